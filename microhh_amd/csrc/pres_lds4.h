// pres_lds4.h -- Pres_4::exec (src/pres_4.cxx:62-95) as three kernels with the transforms done in LDS: the 4th-order operator in the
// shape of pres_lds.h (same spectral array S[k][kx][j], same transforms, same block -> strip mapping).
//
//   pres4_in_fftx_kernel    Pres_4::input (src/pres_4.cxx:256-317) -> LDS -> real-to-complex transform of 8 rows along x -> S
//   pres4_ysolve_kernel     per kx: transform along y of 8 levels in LDS, forward substitution of the factored 7-band system
//                           (src/pres_4.cxx:358-470, hdma :574-730) down through all levels; back up: back substitution, inverse
//                           transform along y
//   pres4_ifftx_out_kernel  complex-to-real transform of 11 rows along x (the strip's 8 + 2 south + 1 north for the 4-point
//                           gradient in y), normalisation, p with its periodic halo and its four mirrored ghost levels
//                           (src/pres_4.cxx:481-528), the corrections of ut and vt (:533-571)
//   + Pres4WtOp (k_pres.hip), one thread per cell: the correction of wt, which needs p of the level ABOVE -- a level the marching
//     kernel has not transformed yet. It reads p (just written, four levels per cell, three of them cache hits) and passes over wt
//     once: one array pass more than a fused form, no register file of carried levels in the transform kernel.
//
// The staged form of pres_4 is input | x r2c | y c2c | substitutions | y c2c | x c2r | unpack + output: 27.5 array passes + the 7
// factor arrays; this form: 20.5 + the factors. The pressure is the toleranced part of the path (DESIGN.md "Parity"): the
// expressions below are the reference's, their evaluation order inside a cell is not always.
#pragma once
#include "pres_lds.h"

namespace mhh { namespace lds_fft {

// ======================================================================================================================
// (1) Pres_4::input + the transform along x. Block = 8 rows j0..j0+7, marching up through kc levels; thread = column i.
// ======================================================================================================================
template<class TF>
struct Pres4LdsIn
{
    GridDev<TF> g;
    const TF* u; const TF* v; const TF* w; const TF* ut; const TF* vt; const TF* wt;
    TF dti;
    C2<TF>* S; const C2<TF>* Tx;      // Tx[m] = exp(-2 pi i m / itot), m < itot
    int nx;                           // log2(itot/2)
    int kc;                           // levels per block
};
template<class TF, int BT, int NX>
// (second launch bound = waves per SIMD the register budget is set for: a block of 512 threads is two waves per SIMD, so two blocks per CU need four)
__global__ void __launch_bounds__(BT, (BT >= 128 ? 4 : 1)) pres4_in_fftx_kernel(const Pres4LdsIn<TF> a)
{
    HIP_DYNAMIC_SHARED(LdsUnit, lds_raw);
    const GridDev<TF>& g = a.g;
    const int itot = g.itot, jtot = g.jtot, nh = itot >> 1, rp = nh + 2;
    C2<TF>* D = reinterpret_cast<C2<TF>*>(lds_raw);
    C2<TF>* T = D + 8*rp;
    const int tid = threadIdx.x;                   // blockDim.x == itot
    T[tid] = a.Tx[tid];
    int strip, chunk; lds_strip_of_block(jtot >> 3, strip, chunk);
    const int j0 = strip*8, k0 = chunk*a.kc, k1 = (k0 + a.kc < g.kmax) ? k0 + a.kc : g.kmax;
    const int jj = g.icells, kk = g.ijcells;
    const int c0 = (tid + g.igc) + (j0 + g.jgc)*jj;
    const int team = nh >> 3, slot = tid / team, l = tid - slot*team;     // the transform this thread works on in the passes
    const bool active = slot < 8;
    TF* Dr = reinterpret_cast<TF*>(D);
    const int rs = 2*rp;                           // reals per LDS row: itot + 4
    // ut + u/dt (and likewise v, w) of a cell: the summands of the reference's integrand
    auto U = [&](int c) { return a.ut[c] + a.u[c] * a.dti; };
    auto V = [&](int c) { return a.vt[c] + a.v[c] * a.dti; };
    auto W = [&](int c) { return a.wt[c] + a.w[c] * a.dti; };
    // the vertical stencil of level k+1 shares three of its four levels with level k: carried, two in registers and the third in
    // LDS (eight rows of this thread's column; in registers too, the itot = 512 kernel needs scratch at the 128 registers that
    // let two blocks share a CU)
    TF wm[8], wc[8];
    TF* const above0 = reinterpret_cast<TF*>(T + itot);              // row r of this column at above0[r*itot + tid]
    {
        const int c = c0 + (k0 + g.kgc)*kk;
#pragma unroll
        for (int r=0; r<8; ++r) { wm[r] = W(c + r*jj - kk); wc[r] = W(c + r*jj); above0[r*itot + tid] = W(c + r*jj + kk); }
    }
    for (int k=k0; k<k1; ++k)
    {
        // opaque per-level copies of the thread's indices: their address arithmetic stays inside the level instead of being hoisted
        // out of the loop into registers that are then spilled (see pres_ifftx_out_kernel)
        unsigned tl = (unsigned)tid, ll = (unsigned)l, sl = (unsigned)(active ? slot : 0), cl = (unsigned)c0;
        keep_vgpr(tl); keep_vgpr(ll); keep_vgpr(sl); keep_vgpr(cl);
        const int kd = k + g.kgc, c = (int)cl + kd*kk;
        TF* const above = above0 + tl;
        const TF dzi4 = uniform_load(g.dzi4, kd);
        // this column's U of the eight rows -> LDS (element x of a row at x + 1, x = -1 .. itot + 1: the three ghost columns the
        // stencil reaches are read from the ghost cells, like the reference does, by the first two threads)
#pragma unroll
        for (int r=0; r<8; ++r) Dr[r*rs + tid + 1] = U(c + r*jj);
        if (tid < 2)
        {
#pragma unroll
            for (int r=0; r<8; ++r) Dr[r*rs + itot + tid + 1] = U(c + r*jj + itot);
        }
        if (tid == 0)
        {
#pragma unroll
            for (int r=0; r<8; ++r) Dr[r*rs] = U(c + r*jj - 1);
        }
        sched_fence();
        // the y and z terms of pres4_in (cell_ops.h; src/pres_4.cxx:305-316) from this thread's own loads, a few rows at a time
        // (the loads of a group in flight together; few values alive across the barrier: the kernel must fit 128 registers at
        // itot = 1024)
        TF d[8];
        constexpr int RG = (sizeof(TF) == 8) ? 2 : 4;      // rows per group (fp64: the itot = 512 kernel spills with four)
#pragma unroll
        for (int h=0; h<8; h+=RG)
        {
            TF vv[RG+3], wn[RG];
#pragma unroll
            for (int r=0; r<RG+3; ++r) vv[r] = V(c + (h + r - 1)*jj);          // (jtot >= 8 in this form: the y terms always exist)
#pragma unroll
            for (int r=0; r<RG; ++r) wn[r] = W(c + (h + r)*jj + 2*kk);
#pragma unroll
            for (int r=0; r<RG; ++r)
            {
                const TF wp = above[(h+r)*itot];
                TF s = cg4(wm[h+r], wc[h+r], wp, wn[r]) * dzi4;
                s = cg4(vv[r], vv[r+1], vv[r+2], vv[r+3]) * g.dyi_d + s;
                d[h+r] = s;
                wm[h+r] = wc[h+r]; wc[h+r] = wp; above[(h+r)*itot] = wn[r];
            }
            sched_fence();
        }
        lds_barrier();
#pragma unroll
        for (int r=0; r<8; ++r)
        {
            const TF um = Dr[r*rs + tid], uc = Dr[r*rs + tid + 1], up = Dr[r*rs + tid + 2], up2 = Dr[r*rs + tid + 3];
            d[r] = cg4(um, uc, up, up2) * g.dxi_d + d[r];
        }
        lds_barrier();
#pragma unroll
        for (int r=0; r<8; ++r) Dr[2*(r*rp + lds_slot<TF>(tid >> 1)) + (tid & 1)] = d[r];
        lds_barrier();
        { const C2<TF> none[fft_np(NX)][7] = {}; fft_batch_ct<-1, true, NX, false>(D + sl*rp, T, 1, (int)ll, a.nx, active, none); }
        lds_barrier();
        // real-to-complex as in pres_in_fftx_kernel: rows fastest, (X_0, X_nyq) share column 0
        for (int e=(int)tl; e<8*nh; e+=itot)
        {
            const int kx = e >> 3, r = e & 7;
            const C2<TF> za = D[r*rp + lds_slot<TF>(kx)], zb = D[r*rp + lds_slot<TF>((nh - kx) & (nh-1))];
            const C2<TF> ev{TF(0.5)*(za.x + zb.x), TF(0.5)*(za.y - zb.y)};
            const C2<TF> od{TF(0.5)*(za.y + zb.y), TF(0.5)*(zb.x - za.x)};
            C2<TF> x = ev + mul_tw<-1>(od, T[kx]);
            if (kx == 0) x = C2<TF>{za.x + za.y, za.x - za.y};
            a.S[((size_t)k*nh + kx)*jtot + j0 + r] = x;
        }
        lds_barrier();
    }
}

// ======================================================================================================================
// (2) Transforms along y around the substitution sweeps of the factored 7-band system. Block = one kx; thread = one ky;
// eight levels per round. Column 0 carries the modes kx = 0 and kx = itot/2 as in pres_ysolve_kernel ("packed", "two").
// Unknown row r of the reference's kmax+4 system is spectral level r-2; the two boundary rows on either side have a zero
// right-hand side and live in registers only (hdma_solve_kernel, k_pres.hip).
// ======================================================================================================================
template<class TF>
struct Pres4LdsSolve
{
    C2<TF>* S;
    const TF* F;                      // F[(band*(kmax+4) + row)*wlev + col], wlev = (ncol+1)*jtot, col = kx*jtot + ky: the LU factors of
                                      // the mode thread ky of block kx solves (block row ncol: the second modes of "two"); band 3 = 1 / m4
    const TF* M7;                     // the outermost upper band is the matrix's own on every row the back substitution multiplies by a
                                      // non-zero (LU without pivoting leaves it alone): per level, not per mode -- read through the scalar cache
    const C2<TF>* Ty;                 // exp(-2 pi i m / jtot)
    int ncol, jtot, ny, kmax;         // ncol = itot/2 columns; ny = log2(jtot)
};
template<class TF, int BT, int NY>
__global__ void __launch_bounds__(BT) pres4_ysolve_kernel(const Pres4LdsSolve<TF> a)
{
    HIP_DYNAMIC_SHARED(LdsUnit, lds_raw);
    const int N = a.jtot, kmax = a.kmax, rp = N;
    C2<TF>* D = reinterpret_cast<C2<TF>*>(lds_raw);
    C2<TF>* T = D + 8*rp;
    const int ky = threadIdx.x, kx = blockIdx.x;   // blockDim.x == jtot
    T[ky] = a.Ty[ky];
    const int team = N >> 3, slot = ky / team, l = ky - slot*team;        // slot < 8 always
    constexpr bool TWC = (NY > 0 && BT <= 256);                           // twiddles in registers where the register file has the room
    C2<TF> tw[fft_np(NY)][7];
    if constexpr (TWC) { lds_barrier(); fft_twiddles_ct<NY>(T, 0, l, tw); }
    const size_t lev = (size_t)a.ncol*N, wlev = (size_t)(a.ncol + 1)*N, band = (size_t)(kmax + 4)*wlev;
    C2<TF>* Sc = a.S + (size_t)kx*N + ky;
    const TF* Fc = a.F + (size_t)kx*N + ky;           // row r of band b at Fc[b*band + r*wlev]
    const TF* Fc2 = a.F + (size_t)a.ncol*N + ky;
    const bool packed = (kx == 0);
    const bool two = packed && (ky == 0 || ky == (N >> 1));
    const bool upper = ky > (N >> 1);
    const int mir = (N - ky) & (N - 1);
    const int nround = (kmax + 7) >> 3;

    // ---- down: rows of eight levels -> LDS -> transform along y -> L y = q -> S (in place)
    C2<TF> a1{TF(0), TF(0)}, a2{TF(0), TF(0)}, a3{TF(0), TF(0)};     // y of the three rows above (rows 0, 1 of the system: zero)
    C2<TF> q[8];
#pragma unroll
    for (int m=0; m<8; ++m) q[m] = (m < kmax) ? Sc[m*lev] : C2<TF>{TF(0), TF(0)};
    for (int rd=0; rd<nround; ++rd)
    {
        const int k0 = rd << 3;
#pragma unroll
        for (int m=0; m<8; ++m) D[m*rp + lds_slot<TF>(ky)] = q[m];
        // this round's factors (bands 0..2 of rows k+2) and the next round's rows: requested before the transform, used after it
        TF f1[8], f2[8], f3[8];
#pragma unroll
        for (int m=0; m<8; ++m)
        {
            const size_t r = (size_t)((k0 + m < kmax) ? k0 + m + 2 : kmax + 1)*wlev;
            f1[m] = Fc[r]; f2[m] = Fc[band + r]; f3[m] = Fc[2*band + r];
        }
        if (rd + 1 < nround)
        {
#pragma unroll
            for (int m=0; m<8; ++m) if (k0 + 8 + m < kmax) q[m] = Sc[(size_t)(k0 + 8 + m)*lev];
        }
        lds_barrier();
        { unsigned ll = (unsigned)l, sl = (unsigned)slot; keep_vgpr(ll); keep_vgpr(sl);
          fft_batch_ct<-1, (BT <= 512), NY, TWC>(D + sl*rp, T, 0, (int)ll, a.ny, true, tw); }
        if (BT <= 512) lds_barrier();
        C2<TF> r8[8];
#pragma unroll
        for (int m=0; m<8; ++m)
        {
            C2<TF> r = D[m*rp + lds_slot<TF>(ky)];
            if (packed && !two)
            {
                const C2<TF> zm = D[m*rp + lds_slot<TF>(mir)];
                r = upper ? C2<TF>{TF(0.5)*(zm.y + r.y), TF(0.5)*(r.x - zm.x)}       // Y_nyq[N-ky] = (Z[N-ky] - conj Z[ky]) / 2i
                          : C2<TF>{TF(0.5)*(r.x + zm.x), TF(0.5)*(r.y - zm.y)};      // Y_0[ky]     = (Z[ky] + conj Z[N-ky]) / 2
            }
            r8[m] = r;
        }
        lds_barrier();
        if (two)
        {   // the imaginary part is a mode of its own (kx = itot/2): its factors, requested together
            TF g1[8], g2[8], g3[8];
#pragma unroll
            for (int m=0; m<8; ++m)
            {
                const size_t r = (size_t)((k0 + m < kmax) ? k0 + m + 2 : kmax + 1)*wlev;
                g1[m] = Fc2[r]; g2[m] = Fc2[band + r]; g3[m] = Fc2[2*band + r];
            }
#pragma unroll
            for (int m=0; m<8; ++m)
            {
                const int k = k0 + m;
                if (k < kmax)
                {
                    C2<TF> v;
                    v.x = r8[m].x - a1.x*f3[m] - a2.x*f2[m] - a3.x*f1[m];
                    v.y = r8[m].y - a1.y*g3[m] - a2.y*g2[m] - a3.y*g1[m];
                    a3 = a2; a2 = a1; a1 = v;
                    Sc[(size_t)k*lev] = v;
                }
            }
        }
        else
        {
#pragma unroll
            for (int m=0; m<8; ++m)
            {
                const int k = k0 + m;
                if (k < kmax)
                {
                    C2<TF> v;
                    v.x = r8[m].x - a1.x*f3[m] - a2.x*f2[m] - a3.x*f1[m];
                    v.y = r8[m].y - a1.y*f3[m] - a2.y*f2[m] - a3.y*f1[m];
                    a3 = a2; a2 = a1; a1 = v;
                    Sc[(size_t)k*lev] = v;
                }
            }
        }
    }
    // ---- the two boundary rows on top (zero right-hand side), then U x = y from the top down
    C2<TF> b1, b2, b3{TF(0), TF(0)};        // x of the three rows below the one worked (b3: nothing beyond row kmax+3)
    {
        const size_t t0 = (size_t)(kmax + 2)*wlev, t1 = (size_t)(kmax + 3)*wlev;
        auto top = [&](const TF* Fm, TF y1, TF y2, TF y3, TF& x2, TF& x3)
        {
            const TF yt0 = TF(0) - y1*Fm[2*band + t0] - y2*Fm[band + t0] - y3*Fm[t0];
            const TF yt1 = TF(0) - yt0*Fm[2*band + t1] - y1*Fm[band + t1] - y2*Fm[t1];
            x3 = yt1 * Fm[3*band + t1];
            x2 = (yt0 - x3*Fm[4*band + t0]) * Fm[3*band + t0];
        };
        top(Fc, a1.x, a2.x, a3.x, b1.x, b2.x);
        top(two ? Fc2 : Fc, a1.y, a2.y, a3.y, b1.y, b2.y);
    }
    // ---- up: back substitution over eight levels -> LDS -> inverse transform along y -> S
    // (the values this thread reads back are the ones it wrote itself)
    TF h4[8], h5[8], h6[8];
    auto request = [&](int k0)
    {
#pragma unroll
        for (int m=0; m<8; ++m)
        {
            const int k = (k0 + m < kmax) ? k0 + m : kmax - 1;
            const size_t r = (size_t)(k + 2)*wlev;
            q[m] = Sc[(size_t)k*lev];
            h4[m] = Fc[3*band + r]; h5[m] = Fc[4*band + r]; h6[m] = Fc[5*band + r];
        }
    };
    request((nround-1) << 3);
    for (int rd=nround-1; rd>=0; --rd)
    {
        const int k0 = rd << 3;
        C2<TF> o[8];
        TF h7[8];                          // (row kmax+1, level kmax-1, multiplies x of row kmax+4: zero, whatever the band holds)
#pragma unroll
        for (int m=0; m<8; ++m) h7[m] = uniform_load(a.M7, (k0 + m < kmax) ? k0 + m : kmax - 1);
        if (two)
        {
            TF e4[8], e5[8], e6[8];
#pragma unroll
            for (int m=0; m<8; ++m)
            {
                const size_t r = (size_t)(((k0 + m < kmax) ? k0 + m : kmax - 1) + 2)*wlev;
                e4[m] = Fc2[3*band + r]; e5[m] = Fc2[4*band + r]; e6[m] = Fc2[5*band + r];
            }
#pragma unroll
            for (int m=7; m>=0; --m)
            {
                C2<TF> r{TF(0), TF(0)};
                if (k0 + m < kmax)
                {
                    r.x = (q[m].x - b1.x*h5[m] - b2.x*h6[m] - b3.x*h7[m]) * h4[m];
                    r.y = (q[m].y - b1.y*e5[m] - b2.y*e6[m] - b3.y*h7[m]) * e4[m];
                    b3 = b2; b2 = b1; b1 = r;
                }
                o[m] = r;
            }
        }
        else
        {
#pragma unroll
            for (int m=7; m>=0; --m)
            {
                C2<TF> r{TF(0), TF(0)};
                if (k0 + m < kmax)
                {
                    r.x = (q[m].x - b1.x*h5[m] - b2.x*h6[m] - b3.x*h7[m]) * h4[m];
                    r.y = (q[m].y - b1.y*h5[m] - b2.y*h6[m] - b3.y*h7[m]) * h4[m];
                    b3 = b2; b2 = b1; b1 = r;
                }
                o[m] = r;
            }
        }
#pragma unroll
        for (int m=0; m<8; ++m) D[m*rp + lds_slot<TF>(ky)] = o[m];
        if (rd > 0) request(k0 - 8);
        lds_barrier();
        if (packed)                        // Z[ky] = Y_0[ky] + i Y_nyq[ky] again, the upper half from the Hermitian symmetry of both
        {
            C2<TF> z[8];
#pragma unroll
            for (int m=0; m<8; ++m)
            {
                const C2<TF> own = D[m*rp + lds_slot<TF>(ky)], mv = D[m*rp + lds_slot<TF>(mir)];
                z[m] = two ? own : (upper ? C2<TF>{mv.x + own.y, own.x - mv.y} : C2<TF>{own.x - mv.y, own.y + mv.x});
            }
            lds_barrier();
#pragma unroll
            for (int m=0; m<8; ++m) D[m*rp + lds_slot<TF>(ky)] = z[m];
            lds_barrier();
        }
        { unsigned ll = (unsigned)l, sl = (unsigned)slot; keep_vgpr(ll); keep_vgpr(sl);
          fft_batch_ct<+1, (BT <= 512), NY, TWC>(D + sl*rp, T, 0, (int)ll, a.ny, true, tw); }
        if (BT <= 512) lds_barrier();
#pragma unroll
        for (int m=0; m<8; ++m) if (k0 + m < kmax) Sc[(size_t)(k0 + m)*lev] = D[m*rp + lds_slot<TF>(ky)];
        lds_barrier();
    }
}

// ======================================================================================================================
// (3) The transform back along x + p + the corrections of ut, vt. Block = rows j0-2 .. j0+8 (the strip's own are LDS rows
// 2 .. 9), marching up through kc levels; thread = column i.
// ======================================================================================================================
template<class TF>
struct Pres4LdsOut
{
    GridDev<TF> g;
    const C2<TF>* S; const C2<TF>* Tx;
    TF* p; TF* ut; TF* vt;
    int nx, kc;
};
template<class TF, int BT, int NX>
__global__ void __launch_bounds__(BT, (BT >= 128 ? 4 : 1)) pres4_ifftx_out_kernel(const Pres4LdsOut<TF> a)
{
    HIP_DYNAMIC_SHARED(LdsUnit, lds_raw);
    constexpr int NR = 11;
    const GridDev<TF>& g = a.g;
    const int itot = g.itot, jtot = g.jtot, nh = itot >> 1, rp = nh + 2, kmax = g.kmax;
    C2<TF>* D = reinterpret_cast<C2<TF>*>(lds_raw);       // row r of the strip's window at D + r*rp; row 0 = j0-2
    C2<TF>* T = D + NR*rp;
    const int tid = threadIdx.x;                   // blockDim.x == itot
    T[tid] = a.Tx[tid];
    int strip, chunk; lds_strip_of_block(jtot >> 3, strip, chunk);
    const int j0 = strip*8, k0 = chunk*a.kc, k1 = (k0 + a.kc < kmax) ? k0 + a.kc : kmax;
    const int jj = g.icells, kk = g.ijcells;
    const int team = nh >> 3, slot = tid / team, l = tid - slot*team;
    const bool active = slot < NR;
    const TF* Dr = reinterpret_cast<const TF*>(D);
    const TF nrm = (TF(1) / TF(jtot)) * (TF(1) / TF(itot));           // both powers of two: exact
    const int i1 = (tid + itot - 1) & (itot - 1), i2 = (tid + itot - 2) & (itot - 1), ie = (tid + 1) & (itot - 1);
    for (int k=k0; k<k1; ++k)
    {
        const int c = (tid + g.igc) + (j0 + g.jgc)*jj + (k + g.kgc)*kk;
        unsigned tl = (unsigned)tid; keep_vgpr(tl);                      // (see pres_ifftx_out_kernel: index arithmetic stays inside the level)
        // spectral rows -> LDS (columns 0 .. nh-1, column 0 = (X_0, X_nyq); rows fastest in memory)
        for (int e=(int)tl; e<NR*nh; e+=itot)
        {
            const int kx = e / NR, r = e - NR*kx;
            const int j = (j0 - 2 + r + jtot) & (jtot - 1);
            D[r*rp + lds_slot<TF>(kx)] = a.S[((size_t)k*nh + kx)*jtot + j];
        }
        lds_barrier();
        // complex-to-real: Z[kx] = (Xa + conj Xb) + i (Xa - conj Xb) exp(+2 pi i kx / itot), Xb = X[nh - kx]; pairs (kx, nh - kx) in place
        for (int e=(int)tl; e<NR*(nh/2 + 1); e+=itot)
        {
            const int r = e / (nh/2 + 1), kx = e - r*(nh/2 + 1), kb = nh - kx;
            const C2<TF> xa = D[r*rp + lds_slot<TF>(kx)], xb = D[r*rp + lds_slot<TF>(kb & (nh-1))];
            if (kx == 0) D[r*rp] = C2<TF>{xa.x + xa.y, xa.x - xa.y};
            else
            {
                const C2<TF> ev{xa.x + xb.x, xa.y - xb.y}, df{xa.x - xb.x, xa.y + xb.y};
                const C2<TF> od = mul_tw<+1>(df, T[kx]);
                D[r*rp + lds_slot<TF>(kx)] = C2<TF>{ev.x - od.y, ev.y + od.x};
            }
            if (kx != 0 && kb != kx)
            {
                const C2<TF> ev{xb.x + xa.x, xb.y - xa.y}, df{xb.x - xa.x, xb.y + xa.y};
                const C2<TF> od = mul_tw<+1>(df, T[kb]);
                D[r*rp + lds_slot<TF>(kb)] = C2<TF>{ev.x - od.y, ev.y + od.x};
            }
        }
        lds_barrier();
        { unsigned ll = (unsigned)l, sl = (unsigned)(active ? slot : 0); keep_vgpr(ll); keep_vgpr(sl);
          const C2<TF> none[fft_np(NX)][7] = {}; fft_batch_ct<+1, true, NX, false>(D + sl*rp, T, 1, (int)ll, a.nx, active, none); }
        lds_barrier();
        // rows of p: element i of row r at real index 2*lds_slot<TF>(i/2) + (i&1)
        const int oc = 2*lds_slot<TF>(tid >> 1) + (tid & 1), o1 = 2*lds_slot<TF>(i1 >> 1) + (i1 & 1);
        const int o2 = 2*lds_slot<TF>(i2 >> 1) + (i2 & 1), oe = 2*lds_slot<TF>(ie >> 1) + (ie & 1);
        // this column's p of the window's rows: a sliding four for the gradient in y
        TF pm2 = Dr[2*(0*rp) + oc] * nrm, pm1 = Dr[2*(1*rp) + oc] * nrm, pc = Dr[2*(2*rp) + oc] * nrm;
        // the levels this level's p is also stored to: the mirrored ghost levels (src/pres_4.cxx:508-525)
        int nl = 1, lvo[3]; lvo[0] = 0;
        if (k == 0) lvo[nl++] = -kk;                     // p[kstart-1] = p[kstart]
        if (k == 1) lvo[nl++] = -3*kk;                   // p[kstart-2] = p[kstart+1]
        if (k == kmax-1) lvo[nl++] = kk;                 // p[kend]     = p[kend-1]
        if (k == kmax-2) lvo[nl++] = 3*kk;               // p[kend+1]   = p[kend-2]  (at most two of the four hold for one k)
#pragma unroll
        for (int qd=0; qd<8; ++qd)
        {
            const int r = qd + 2, cr = c + qd*jj;
            const TF pn = Dr[2*((r+1)*rp) + oc] * nrm;
            const TF pw1 = Dr[2*(r*rp) + o1] * nrm, pw2 = Dr[2*(r*rp) + o2] * nrm, pe = Dr[2*(r*rp) + oe] * nrm;
            a.ut[cr] -= cg4(pw2, pw1, pc, pe) * g.dxi_d;
            a.vt[cr] -= cg4(pm2, pm1, pc, pn) * g.dyi_d;                    // (jtot >= 8 in this form)
            // p: the cell, its images in the periodic halo, on its own level and on the ghost levels that mirror it
            const int js = j0 + qd;
            for (int lv=0; lv<nl; ++lv)
            {
                const int cl = cr + lvo[lv];
                for (int rowsel=0; rowsel<3; ++rowsel)
                {
                    int off;
                    if (rowsel == 0) off = 0;
                    else if (rowsel == 1) { if (js < jtot - g.jgc) continue; off = -jtot*jj; }     // row js - jtot: the south halo
                    else                  { if (js >= g.jgc) continue;       off =  jtot*jj; }     // row js + jtot: the north halo
                    a.p[cl + off] = pc;
                    if (tid >= itot - g.igc) a.p[cl + off - itot] = pc;
                    if (tid < g.igc)         a.p[cl + off + itot] = pc;
                }
            }
            pm2 = pm1; pm1 = pc; pc = pn;
        }
        lds_barrier();
    }
}

}} // namespace mhh::lds_fft
