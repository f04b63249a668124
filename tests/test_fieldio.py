"""Field / grid files in the reference's on-disk layout (src/field3d_io.cxx:54-230, reader in
cases/taylorgreen/taylorgreen_test.py:53-68): Python module, the C++ Field3d_io mirror, and slab-wise writes."""
import os
import struct
import subprocess

import numpy as np
import pytest

import common as cm
from microhh_amd import fieldio


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_roundtrip_and_layout(tmp_path, dtype):
    g = cm.grid_2nd(12, 10, 8, gc=(3, 3, 1), dtype=dtype)
    a = np.random.RandomState(3).random_sample(g.shape3).astype(dtype)
    fn = fieldio.field_filename(str(tmp_path), "th", 3600)
    assert fn.endswith("th.0003600")
    fieldio.save_field3d(fn, a, g, offset=300.)
    # the layout the reference's scripts read: struct.unpack of nx*ny*nz values reshaped (nz, ny, nx)
    raw = open(fn, "rb").read()
    vals = np.array(struct.unpack("<%d%s" % (g.itot*g.jtot*g.ktot, "d" if dtype == np.float64 else "f"), raw)).reshape(g.ktot, g.jtot, g.itot)
    assert np.array_equal(vals.astype(dtype), a[g.interior] + dtype(300.))
    b = fieldio.load_field3d(fn, g, offset=300.)
    assert np.abs(b[g.interior] - a[g.interior]).max() <= 2 * 300 * np.finfo(dtype).eps      # (x + 300) - 300
    fn0 = fieldio.field_filename(str(tmp_path), "u", 0)
    fieldio.save_field3d(fn0, a, g)
    assert np.array_equal(fieldio.load_field3d(fn0, g)[g.interior], a[g.interior])               # offset 0: exact
    ghost = np.ones(g.shape3, bool); ghost[g.interior] = False
    assert np.all(b[ghost] == 0)
    with pytest.raises(ValueError):
        fieldio.load_field3d(fn, cm.grid_2nd(12, 10, 9, gc=(3, 3, 1), dtype=dtype))


def test_slab_ranks_write_one_global_file(tmp_path):
    gg = cm.grid_2nd(12, 8, 6, gc=(3, 3, 1))
    a = np.random.RandomState(4).random_sample(gg.shape3)
    whole, parts = str(tmp_path / "u.0000000"), str(tmp_path / "u.0000001")
    fieldio.save_field3d(whole, a, gg)
    npy = 2
    with pytest.raises(FileNotFoundError):            # no rank creates or re-sizes the global file while writing its rows
        g1 = cm.grid_2nd(12, 8, 6, gc=(3, 3, 1), npy=npy, mpicoordy=1)
        fieldio.save_field3d(parts, np.zeros(g1.shape3), g1, rank=1, npy=npy)
    fieldio.prepare_global_file(parts, cm.grid_2nd(12, 8, 6, gc=(3, 3, 1), npy=npy, mpicoordy=0), npy)
    for r in (1, 0):                                   # any order
        gs = cm.grid_2nd(12, 8, 6, gc=(3, 3, 1), npy=npy, mpicoordy=r)
        loc = np.zeros(gs.shape3)
        loc[:, gs.jstart:gs.jend, :] = a[:, gg.jstart + r*gs.jmax: gg.jstart + (r+1)*gs.jmax, :]
        fieldio.save_field3d(parts, loc, gs, rank=r, npy=npy)
    assert open(whole, "rb").read() == open(parts, "rb").read()
    fieldio.prepare_global_file(parts, cm.grid_2nd(12, 8, 6, gc=(3, 3, 1), npy=npy, mpicoordy=0), npy)   # right size already: contents stay
    assert open(whole, "rb").read() == open(parts, "rb").read()
    gs = cm.grid_2nd(12, 8, 6, gc=(3, 3, 1), npy=npy, mpicoordy=1)
    b = fieldio.load_field3d(whole, gs, rank=1, npy=npy)
    assert np.array_equal(b[gs.interior], a[:, gg.jstart + gs.jmax: gg.jstart + 2*gs.jmax, :][gg.kstart:gg.kend, :, gg.istart:gg.iend])


def test_grid_file(tmp_path):
    g = cm.grid_2nd(12, 10, 8, gc=(3, 3, 1))
    fn = fieldio.save_grid(str(tmp_path), g)
    assert os.path.basename(fn) == "grid.0000000" and os.path.getsize(fn) == 8*2*(12 + 10 + 8)
    d = fieldio.load_grid(str(tmp_path), 12, 10, 8)
    assert np.allclose(d["x"], (np.arange(12) + 0.5)*g.dx) and np.allclose(d["yh"], np.arange(10)*g.dy)
    assert np.array_equal(d["z"], g.z[g.kstart:g.kend]) and np.array_equal(d["zh"], g.zh[g.kstart:g.kend])


@pytest.mark.parametrize("npy", [1, 2])
def test_cpp_field3d_io_reads_and_writes_the_same_files(tmp_path, npy):
    exe = str(tmp_path / "fieldio_roundtrip")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(cm.ROOT, "tests", "cpp", "fieldio_roundtrip.cpp")], check=True)
    g = cm.grid_2nd(12, 8, 6, gc=(3, 3, 1))
    a = np.random.RandomState(5).random_sample(g.shape3)
    fin, fout = str(tmp_path / "th.0000000"), str(tmp_path / "th.0000001")
    fieldio.save_field3d(fin, a, g, offset=300.)
    r = subprocess.run([exe, fin, fout, "12", "8", "6", "3", "3", "1", str(npy)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(fin, "rb").read() == open(fout, "rb").read()
