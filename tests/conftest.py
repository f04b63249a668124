import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    """Build the CPU oracle (and oracle/_ref where the reference sources exist) once per session."""
    import common
    try:
        common.build_oracle()
    except Exception as e:  # the GPU box has no reference; prebuilt files are used
        if not os.path.exists(os.path.join(common.ORACLE_DIR, "liborc.so")):
            raise e
    yield
