"""Pin the oracle's pressure solver (pres_2 / pres_4). The reference TUs for this stage need fftw3.h and
cannot be built here, so the restatement is pinned by independent mathematics instead:
  * FFT stage vs numpy.fft (pocketfft) through the FFTW half-complex mapping (src/fft.cxx:143-150),
  * modified wave numbers / tridiagonal coefficients vs their closed forms (src/pres_2.cxx:125-153),
  * the spectral solve vs a dense numpy solve of the same banded system,
  * the whole operator through its defining property: after Pres::exec the discrete divergence of
    rho*(u/dt + ut) is zero to rounding, i.e. a second pres_input yields ~0 (projection idempotence)."""
import numpy as np
import pytest

import common as cm
from common import ptr, dbl


def packed(g, rs):
    return rs.random_sample((g.ktot, g.jtot, g.itot)).astype(g.np_dtype)


def hc_from_rfft(x):
    """FFTW R2HC layout of the DFT of the last axis."""
    n = x.shape[-1]
    X = np.fft.rfft(x, axis=-1)
    out = np.empty_like(x)
    out[..., :n//2+1] = X.real
    for k in range(1, (n+1)//2):
        out[..., n-k] = X[..., k].imag
    return out


@pytest.mark.parametrize("shape", [(16, 8, 4), (12, 10, 3), (9, 7, 2), (32, 1, 4)])
def test_fft_matches_numpy(shape):
    itot, jtot, ktot = shape
    g = cm.Grid(itot, jtot, ktot, 1., 1., 1., order=2)
    rs = np.random.RandomState(1)
    a = packed(g, rs)
    want = hc_from_rfft(a)                                             # x transform
    want = np.swapaxes(hc_from_rfft(np.swapaxes(want, 1, 2)), 1, 2)    # then y
    got = a.copy()
    cm.oracle().orc_fft_forward(g.host_struct(), ptr(got))
    assert np.allclose(got, want, rtol=0, atol=1e-13*itot*jtot)
    cm.oracle().orc_fft_backward(g.host_struct(), ptr(got))
    assert np.allclose(got, a, rtol=0, atol=1e-14*np.log2(itot*jtot+1)*4)


def test_pres2_coefficients():
    g = cm.grid_2nd(16, 12, 10)
    c = cm.Case(g)
    bi = np.zeros(g.itot); bj = np.zeros(g.jtot); a = np.zeros(g.kmax); cc = np.zeros(g.kmax)
    cm.oracle().orc_pres2_coeffs(g.host_struct(), ptr(c.rhorefh), ptr(bi), ptr(bj), ptr(a), ptr(cc))
    i = np.arange(g.itot); i = np.minimum(i, g.itot - i)
    j = np.arange(g.jtot); j = np.minimum(j, g.jtot - j)
    assert np.allclose(bi, 2.*(np.cos(2*np.pi*i/g.itot)-1.)/g.dx**2, rtol=1e-14, atol=0)
    assert np.allclose(bj, 2.*(np.cos(2*np.pi*j/g.jtot)-1.)/g.dy**2, rtol=1e-14, atol=0)
    k = np.arange(g.kmax) + g.kgc
    assert np.allclose(a, g.dz[k]*c.rhorefh[k]*g.dzhi[k], rtol=1e-15)
    assert np.allclose(cc, g.dz[k]*c.rhorefh[k+1]*g.dzhi[k+1], rtol=1e-15)


def test_pres2_spectral_solve_is_tridiagonal_solve():
    g = cm.grid_2nd(8, 6, 12)
    c = cm.Case(g)
    rs = np.random.RandomState(3)
    rhs = packed(g, rs)
    p = rhs.copy()
    cm.oracle().orc_pres_spectral_solve(g.host_struct(), 2, ptr(p), ptr(c.rhoref), ptr(c.rhorefh))
    bi = np.zeros(g.itot); bj = np.zeros(g.jtot); a = np.zeros(g.kmax); cc = np.zeros(g.kmax)
    cm.oracle().orc_pres2_coeffs(g.host_struct(), ptr(c.rhorefh), ptr(bi), ptr(bj), ptr(a), ptr(cc))
    kk = np.arange(g.kmax) + g.kgc
    for (i, j) in [(0, 0), (1, 0), (3, 2), (7, 5)]:
        b = g.dz[kk]**2 * c.rhoref[kk]*(bi[i]+bj[j]) - (a+cc)
        b[0] += a[0]
        b[-1] += (-cc[-1] if (i == 0 and j == 0) else cc[-1])
        A = np.diag(b) + np.diag(a[1:], -1) + np.diag(cc[:-1], 1)
        want = np.linalg.solve(A, g.dz[kk]**2 * rhs[:, j, i])
        assert np.allclose(p[:, j, i], want, rtol=1e-9, atol=1e-12*np.abs(want).max())


def _project(g, c, order, dt=0.7):
    O = cm.oracle(); G = g.host_struct()
    pk = np.zeros((g.ktot, g.jtot, g.itot), dtype=g.np_dtype)
    O.orc_pres_exec(G, order, ptr(c.p), ptr(pk), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.ut), ptr(c.vt), ptr(c.wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(dt))
    pk2 = np.zeros_like(pk)
    O.orc_pres_input(G, order, ptr(pk2), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.ut), ptr(c.vt), ptr(c.wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(dt))
    return pk, pk2


@pytest.mark.parametrize("shape", [(16, 12, 10), (8, 8, 6), (12, 1, 8)])
def test_pres2_projects_to_divergence_free(shape):
    g = cm.grid_2nd(*shape, gc=(1, 1, 1))
    c = cm.Case(g, periodic=True)
    div0 = np.zeros((g.ktot, g.jtot, g.itot))
    cm.oracle().orc_pres_input(g.host_struct(), 2, ptr(div0), ptr(c.u), ptr(c.v), ptr(c.w), ptr(c.ut), ptr(c.vt), ptr(c.wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(0.7))
    _, div1 = _project(g, c, 2)
    assert np.abs(div0).max() > 1e-4
    assert np.abs(div1).max() < 1e-11 * np.abs(div0).max() * g.ktot
    # bottom Neumann ghost + periodic ghosts of p
    assert np.array_equal(c.p[g.kstart-1][g.jstart:g.jend, g.istart:g.iend], c.p[g.kstart][g.jstart:g.jend, g.istart:g.iend])
    assert np.array_equal(c.p[:, :, 0], c.p[:, :, g.iend-1])


@pytest.mark.parametrize("shape", [(16, 12, 12), (12, 1, 8)])
def test_pres4_projects_to_divergence_free(shape):
    g = cm.grid_4th(*shape)
    c = cm.Case(g, rho="one", periodic=True)
    # Boundary::set_ghost_cells_w(Conservation_type), src/boundary.cxx:838-871, runs before pres->exec (src/model.cxx:410)
    for m in (1, 2):
        c.w[g.kstart-m] = -c.w[g.kstart+m]
        c.w[g.kend+m] = -c.w[g.kend-m]
    div0 = np.zeros((g.ktot, g.jtot, g.itot))
    ut, vt, wt = c.ut.copy(), c.vt.copy(), c.wt.copy()
    cm.oracle().orc_pres_input(g.host_struct(), 4, ptr(div0), ptr(c.u), ptr(c.v), ptr(c.w), ptr(ut), ptr(vt), ptr(wt), ptr(c.rhoref), ptr(c.rhorefh), dbl(0.7))
    _, div1 = _project(g, c, 4)
    assert np.abs(div0).max() > 1e-3
    assert np.abs(div1).max() < 1e-9 * np.abs(div0).max()
