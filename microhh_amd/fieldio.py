"""Restart / field files in the reference's on-disk layout, so that fields produced here load into MicroHH's own
tooling and vice versa.

  name.NNNNNNN   raw TF stream of the INTERIOR cells, C order (ktot, jtot, itot), no header, native endianness,
                 `data + offset` on save and `file - offset` on load (Field3d_io::save_field3d / load_field3d,
                 src/field3d_io.cxx:54-140 and :143-230; the MPI build writes the same global file through a
                 subarray view, which is what `rank`/`npy` reproduce here with positioned writes of each y-slab)
  grid.0000000   x, xh [itot], y, yh [jtot], z, zh [ktot] as raw TF (reader: cases/taylorgreen/taylorgreen_test.py:53-68)

Host-side only: arrays are numpy, ghosted (kcells, jcells, icells) like everything else in this package.
"""
import os

import numpy as np


def field_filename(path, name, iteration):
    return os.path.join(path, "%s.%07d" % (name, iteration))


def save_field3d(filename, data, grid, offset=0., rank=0, npy=1, kstart=None, kend=None):
    """Write the interior of a ghosted field. With npy > 1 every rank calls this with its own slab (grid.jmax rows)
    and the same filename; rank r's rows land at j offset r*jmax of the global (k, jtot, itot) file."""
    g = grid
    kstart = g.kstart if kstart is None else kstart
    kend = g.kend if kend is None else kend
    kmax = kend - kstart
    a = np.asarray(data).reshape(g.shape3)
    inner = a[kstart:kend, g.jstart:g.jend, g.istart:g.iend].astype(g.np_dtype, copy=False)
    if offset:
        inner = inner + g.np_dtype.type(offset)
    jtot = g.jmax * npy
    if npy == 1:
        np.ascontiguousarray(inner).tofile(filename)
        return
    # The global file must already exist at its full size (prepare_global_file, called by ONE rank ahead of a barrier:
    # HotPath.save). Nothing here creates, truncates or resizes it -- a rank that did would race the others' writes.
    nbytes = kmax * jtot * g.imax * inner.itemsize
    if not os.path.exists(filename) or os.path.getsize(filename) != nbytes:
        raise FileNotFoundError("%s must be created at %d bytes (fieldio.prepare_global_file) before the slab ranks write into it" % (filename, nbytes))
    mm = np.memmap(filename, dtype=g.np_dtype, mode="r+", shape=(kmax, jtot, g.imax))
    mm[:, rank*g.jmax:(rank+1)*g.jmax, :] = inner
    mm.flush()
    del mm


def prepare_global_file(filename, grid, npy, kstart=None, kend=None):
    """Create (or re-size) the one global file of a slab-decomposed field at its final size, without touching its contents
    when the size is already right. ONE rank calls this, then all ranks meet at a barrier, then every rank writes its rows:
    the collective MPI_File_open + subarray view of the reference (src/field3d_io.cxx) spelled out."""
    g = grid
    kmax = (g.kend if kend is None else kend) - (g.kstart if kstart is None else kstart)
    nbytes = kmax * g.jmax * npy * g.imax * np.dtype(g.np_dtype).itemsize
    with open(filename, "r+b" if os.path.exists(filename) else "w+b") as f:     # never "wb": that would zero rows already written
        f.truncate(nbytes)
    return nbytes


def load_field3d(filename, grid, offset=0., rank=0, npy=1, kstart=None, kend=None, out=None):
    """Read a field file into the interior of a ghosted array (ghost cells are left as they are / zero, as in the
    reference, where the boundary conditions fill them afterwards)."""
    g = grid
    kstart = g.kstart if kstart is None else kstart
    kend = g.kend if kend is None else kend
    kmax = kend - kstart
    jtot = g.jmax * npy
    expect = kmax * jtot * g.imax * np.dtype(g.np_dtype).itemsize
    if os.path.getsize(filename) != expect:
        raise ValueError("%s: %d bytes, expected %d for a (%d, %d, %d) %s field" %
                         (filename, os.path.getsize(filename), expect, kmax, jtot, g.imax, np.dtype(g.np_dtype).name))
    mm = np.memmap(filename, dtype=g.np_dtype, mode="r", shape=(kmax, jtot, g.imax))
    if out is None:
        out = np.zeros(g.shape3, dtype=g.np_dtype)
    view = out.reshape(g.shape3)
    view[kstart:kend, g.jstart:g.jend, g.istart:g.iend] = mm[:, rank*g.jmax:(rank+1)*g.jmax, :]
    if offset:
        view[kstart:kend, g.jstart:g.jend, g.istart:g.iend] -= g.np_dtype.type(offset)
    del mm
    return out


def save_grid(path, grid, itot=None, jtot=None):
    """grid.0000000: cell-centre and face coordinates of the global interior (Grid::save_grid, src/grid.cxx:379)."""
    g = grid
    itot = g.itot if itot is None else itot
    jtot = g.jtot if jtot is None else jtot
    dt = g.np_dtype
    x = ((np.arange(itot) + 0.5) * g.dx).astype(dt); xh = (np.arange(itot) * g.dx).astype(dt)
    y = ((np.arange(jtot) + 0.5) * g.dy).astype(dt); yh = (np.arange(jtot) * g.dy).astype(dt)
    z = np.asarray(g.z[g.kstart:g.kend], dtype=dt); zh = np.asarray(g.zh[g.kstart:g.kend], dtype=dt)
    fn = os.path.join(path, "grid.%07d" % 0)
    with open(fn, "wb") as f:
        for a in (x, xh, y, yh, z, zh):
            np.ascontiguousarray(a).tofile(f)
    return fn


def load_grid(path, itot, jtot, ktot, dtype=np.float64):
    fn = os.path.join(path, "grid.%07d" % 0)
    raw = np.fromfile(fn, dtype=dtype)
    if raw.size != 2*(itot + jtot + ktot):
        raise ValueError("%s: %d values, expected %d" % (fn, raw.size, 2*(itot + jtot + ktot)))
    out, o = {}, 0
    for name, n in (("x", itot), ("xh", itot), ("y", jtot), ("yh", jtot), ("z", ktot), ("zh", ktot)):
        out[name] = raw[o:o+n].copy(); o += n
    return out
