"""The adaptor translation units of integration/ (the MicroHH-side binding of INTEGRATION.md: the USECUDA halves of
Advec_*, Diff_*, Pres_2 / Pres_4 and Boundary_cyclic re-implemented as calls into the C ABI) are type-checked against the
reference's own headers where the reference is present. include/pres.h pulls in fftw3.h (and cufft.h under USECUDA), which
this image lacks: for the two pressure adaptors the include path carries tests/stubs_syntax_only/, headers that declare the
plan / handle TYPE NAMES those reference headers mention and nothing else -- a syntax check, it builds and pins nothing."""
import glob
import os
import subprocess

import pytest

import common as cm

REF_INC = "/root/reference/include"


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="reference headers not present on this machine")
@pytest.mark.parametrize("src", sorted(glob.glob(os.path.join(cm.ROOT, "integration", "adaptor_*.cxx"))), ids=os.path.basename)
def test_adaptor_type_checks_against_reference_headers(src):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DUSECUDA", "-DRESTRICTKEYWORD=__restrict__",
           "-I" + REF_INC, "-I" + os.path.join(cm.ROOT, "include"), "-I" + os.path.join(cm.ROOT, "integration"), src]
    if "adaptor_pres_" in src:
        cmd.insert(-1, "-I" + os.path.join(cm.ROOT, "tests", "stubs_syntax_only"))
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_adaptors_exist():
    names = {os.path.basename(p) for p in glob.glob(os.path.join(cm.ROOT, "integration", "*"))}
    assert {"mhh_adaptor.h", "adaptor_advec_2.cxx", "adaptor_advec_2i5.cxx", "adaptor_advec_4.cxx", "adaptor_diff_2.cxx", "adaptor_diff_4.cxx",
            "adaptor_diff_smag2.cxx", "adaptor_boundary_cyclic.cxx", "adaptor_pres_2.cxx", "adaptor_pres_4.cxx"} <= names
