// pres_lds.h -- Pres_2::exec (src/pres_2.cxx:66-94) as THREE kernels with the transforms done in LDS, for power-of-two itot, jtot.
//
// The staged form (k_pres.hip) is seven passes over memory: input | x r2c | y c2c | Thomas | y c2c | x c2r | unpack + output,
// each a read and a write of a 512^3 array or more. The transforms are short (a row fits a fraction of a CU's 160 KB LDS), so
// they can ride inside the kernels on either side of them:
//
//   pres_in_fftx_kernel   Pres_2::input (src/pres_2.cxx:156-196) -> LDS -> real-to-complex transform of 8 rows along x
//                         (fft_forward's first half, src/fft.cxx:451-497) -> spectral array S[k][kx][j]
//   pres_ysolve_kernel    per kx: forward transform along y of 8 levels in LDS, the Thomas forward sweep over them
//                         (src/pres_2.cxx:213-249), down through all levels; then back up: back substitution (:251-263),
//                         inverse transform along y, S[k][kx][j] again (now physical in y)
//   pres_ifftx_out_kernel complex-to-real transform of 9 rows along x in LDS (fft_backward's second half, src/fft.cxx:540-583),
//                         normalisation, p with its ghost cells (src/pres_2.cxx:333-363) and Pres_2::output (:365-387)
//
// 19.5 array passes instead of 27.5. The transforms are Stockham radix-8 / 4 / 2 passes on 16-byte (fp64) or 8-byte (fp32)
// complex numbers, eight elements per thread and pass, twiddles from a table made on the host in double precision. The
// reference's transforms are FFTW's; like rocFFT's these agree with them to rounding (DESIGN.md "Parity": the pressure is the
// toleranced part of the path).
#pragma once
#include <type_traits>
#include <gfx950_prims.h>   // angle form: the CPU emulation build (tests/emul) overrides it by include path

namespace mhh { namespace lds_fft {

struct alignas(16) LdsUnit { double x, y; };           // the dynamic LDS block starts 16-byte aligned

template<class TF> __device__ __forceinline__ C2<TF> operator+(C2<TF> a, C2<TF> b) { return C2<TF>{a.x + b.x, a.y + b.y}; }
template<class TF> __device__ __forceinline__ C2<TF> operator-(C2<TF> a, C2<TF> b) { return C2<TF>{a.x - b.x, a.y - b.y}; }

// Where element i of a transform sits in LDS: the low bits of the index are XOR-ed with the bits above them, so that BOTH access
// patterns of a pass are free of bank conflicts (MI355X_MICROARCH.md, LDS): consecutive elements read with ds_read_b128 / b64
// (lane groups of 16 / 32 need distinct 16-byte slots of a 256-byte bank row: a permutation inside aligned blocks keeps that), and
// the strided stores of a radix-8 pass (8 / 16 contiguous lanes, element stride 8: all on one slot without the swizzle).
// fp64 (16-byte elements): 8-element blocks; fp32 (8-byte elements): 16-element blocks. No padding inside a row.
// (Round-2 first form, i + i/8: conflict-free stores, but it shifted the slots of consecutive reads against their lane groups --
// SQ_LDS_BANK_CONFLICT on 50-60 % of the LDS cycles of all three kernels.)
template<class TF> __host__ __device__ __forceinline__ int lds_slot(int i)
{
    return sizeof(TF) == 8 ? (i ^ ((i >> 3) & 7)) : (i ^ ((i >> 4) & 15));
}

// S < 0: forward (e^{-i..}), S > 0: inverse. t = table entry exp(-i phi).
template<int S, class TF> __device__ __forceinline__ C2<TF> mul_tw(C2<TF> a, C2<TF> t)
{
    return S < 0 ? C2<TF>{a.x*t.x - a.y*t.y, a.x*t.y + a.y*t.x} : C2<TF>{a.x*t.x + a.y*t.y, a.y*t.x - a.x*t.y};
}
template<int S, class TF> __device__ __forceinline__ C2<TF> rot(C2<TF> a) { return S < 0 ? C2<TF>{a.y, -a.x} : C2<TF>{-a.y, a.x}; }   // a * (S i)
template<int S, class TF> __device__ __forceinline__ void dft4(C2<TF>& a0, C2<TF>& a1, C2<TF>& a2, C2<TF>& a3)
{
    const C2<TF> s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = rot<S>(a1 - a3);
    a0 = s02 + s13; a1 = d02 + d13; a2 = s02 - s13; a3 = d02 - d13;
}
template<int S, class TF> __device__ __forceinline__ void dft8(C2<TF> (&v)[8])
{
    const TF h = TF(0.70710678118654752440);
    C2<TF> t0 = v[0] + v[4], u0 = v[0] - v[4], t1 = v[1] + v[5], u1 = v[1] - v[5];
    C2<TF> t2 = v[2] + v[6], u2 = v[2] - v[6], t3 = v[3] + v[7], u3 = v[3] - v[7];
    // u_k *= exp(S i pi k / 4)
    u1 = S < 0 ? C2<TF>{(u1.x + u1.y)*h, (u1.y - u1.x)*h} : C2<TF>{(u1.x - u1.y)*h, (u1.x + u1.y)*h};
    u2 = rot<S>(u2);
    u3 = S < 0 ? C2<TF>{(u3.y - u3.x)*h, -(u3.x + u3.y)*h} : C2<TF>{-(u3.x + u3.y)*h, (u3.x - u3.y)*h};
    dft4<S>(t0, t1, t2, t3); dft4<S>(u0, u1, u2, u3);
    v[0] = t0; v[2] = t1; v[4] = t2; v[6] = t3; v[1] = u0; v[3] = u1; v[5] = u2; v[7] = u3;
}

// One Stockham pass over a transform of N = 2^n points that lives in LDS at D[lds_slot<TF>(0..N-1)], done by N/8 threads (l = 0..N/8-1),
// every thread owning eight elements. gather: all reads of the pass; scatter: twiddles, butterflies of radix 2^lr (8 / lr of them
// per thread), writes. The caller puts a barrier between the two and after. ls = log2 of the product of the radices already done.
// T[m << tshift] = exp(-2 pi i m / N).
template<class TF> __device__ __forceinline__ void fft_gather(const C2<TF>* D, int l, int n, C2<TF> (&v)[8])
{
    const int n8 = 1 << (n-3);
#pragma unroll
    for (int m=0; m<8; ++m) v[m] = D[lds_slot<TF>(l + m*n8)];
}
template<int S, class TF> __device__ __forceinline__ void fft_scatter(C2<TF>* D, const C2<TF>* T, int tshift, int l, int n, int lr, int ls, C2<TF> (&v)[8])
{
    const int n8 = 1 << (n-3), mask = (1 << ls) - 1, q = n - ls - lr + tshift;
    if (lr == 3)
    {
        const int k = l & mask;
        if (ls > 0)
        {
#pragma unroll
            for (int r=1; r<8; ++r) v[r] = mul_tw<S>(v[r], T[(r*k) << q]);
        }
        dft8<S>(v);
        const int o = ((l - k) << 3) + k;
#pragma unroll
        for (int r=0; r<8; ++r) D[lds_slot<TF>(o + (r << ls))] = v[r];
    }
    else if (lr == 2)
    {
#pragma unroll
        for (int b=0; b<2; ++b)
        {
            const int j = l + b*n8, k = j & mask;
            C2<TF> a0 = v[b], a1 = v[b+2], a2 = v[b+4], a3 = v[b+6];
            if (ls > 0) { a1 = mul_tw<S>(a1, T[k << q]); a2 = mul_tw<S>(a2, T[(2*k) << q]); a3 = mul_tw<S>(a3, T[(3*k) << q]); }
            dft4<S>(a0, a1, a2, a3);
            const int o = ((j - k) << 2) + k;
            D[lds_slot<TF>(o)] = a0; D[lds_slot<TF>(o + (1 << ls))] = a1; D[lds_slot<TF>(o + (2 << ls))] = a2; D[lds_slot<TF>(o + (3 << ls))] = a3;
        }
    }
    else
    {
#pragma unroll
        for (int b=0; b<4; ++b)
        {
            const int j = l + b*n8, k = j & mask;
            C2<TF> a0 = v[b], a1 = v[b+4];
            if (ls > 0) a1 = mul_tw<S>(a1, T[k << q]);
            const int o = ((j - k) << 1) + k;
            D[lds_slot<TF>(o)] = a0 + a1; D[lds_slot<TF>(o + (1 << ls))] = a0 - a1;
        }
    }
}
// the passes of an N = 2^n transform: the odd radix (n mod 3 bits) first, radix 8 after it
__device__ __forceinline__ int first_radix_log2(int n) { const int r = n % 3; return r ? r : 3; }

// A batch of transforms, one per `slot`, all threads of the block passing through the same synchronisation points. `active`: this
// thread works on transform D (of N/8 threads, as number l); idle threads only keep the count. WL: the N/8 threads of a transform
// sit in one wave (N <= 512), so the passes need no block barrier at all -- the caller's barrier before (data in LDS) and after
// (before other waves read the result) are the only ones.
template<bool WL> __device__ __forceinline__ void fft_sync() { if (WL) wave_sync(); else lds_barrier(); }
template<int S, bool WL, class TF> __device__ __forceinline__ void fft_batch(C2<TF>* D, const C2<TF>* T, int tshift, int l, int n, bool active)
{
    C2<TF> v[8];
    int ls = 0;
    for (int lr = first_radix_log2(n); ls < n; ls += lr, lr = 3)
    {
        if (active) fft_gather(D, l, n, v);
        fft_sync<WL>();
        if (active) fft_scatter<S>(D, T, tshift, l, n, lr, ls, v);
        fft_sync<WL>();
    }
}

// The same with the size known at compile time (NLOG > 0; NLOG == 0 forwards to the run-time form): the passes unroll, radices
// and strides are constants and the LDS addresses of a pass one base plus immediates. TWC: the twiddles of a thread are the same
// in every call, so a kernel that transforms in a loop keeps them in registers (fft_twiddles_ct fills tw once).
constexpr int fft_np(int nlog) { return nlog ? (nlog + 2)/3 : 1; }      // passes (array extent of the register-held twiddles)
template<int NLOG> struct FftCT
{
    static constexpr int first = (NLOG % 3) ? (NLOG % 3) : 3, npass = (NLOG + 2) / 3;
    static constexpr int lr(int p) { return p ? 3 : first; }
    static constexpr int ls(int p) { return p ? first + 3*(p-1) : 0; }
};
template<int NLOG, class TF> __device__ __forceinline__ void fft_twiddles_ct(const C2<TF>* T, int tshift, int l, C2<TF> (&tw)[fft_np(NLOG)][7])
{
    typedef FftCT<NLOG> F;
#pragma unroll
    for (int p=1; p<F::npass; ++p)
    {
        const int k = l & ((1 << F::ls(p)) - 1), q = NLOG - F::ls(p) - 3 + tshift;
#pragma unroll
        for (int r=1; r<8; ++r) tw[p][r-1] = T[(r*k) << q];
    }
}
template<int S, bool WL, int NLOG, bool TWC, class TF>
__device__ __forceinline__ void fft_batch_ct(C2<TF>* D, const C2<TF>* T, int tshift, int l, int n, bool active, const C2<TF> (&tw)[fft_np(NLOG)][7])
{
    if constexpr (NLOG == 0) fft_batch<S, WL>(D, T, tshift, l, n, active);
    else
    {
        typedef FftCT<NLOG> F;
        C2<TF> v[8];
#pragma unroll
        for (int p=0; p<F::npass; ++p)
        {
            if (active) fft_gather(D, l, NLOG, v);
            fft_sync<WL>();
            if (active)
            {
                if constexpr (TWC)
                {
                    if (p > 0)          // radix 8 with this thread's own twiddles
                    {
#pragma unroll
                        for (int r=1; r<8; ++r) v[r] = mul_tw<S>(v[r], tw[p][r-1]);
                        dft8<S>(v);
                        const int k = l & ((1 << F::ls(p)) - 1), o = ((l - k) << 3) + k;
#pragma unroll
                        for (int r=0; r<8; ++r) D[lds_slot<TF>(o + (r << F::ls(p)))] = v[r];
                    }
                    else fft_scatter<S>(D, T, tshift, l, NLOG, F::lr(0), 0, v);
                }
                else fft_scatter<S>(D, T, tshift, l, NLOG, F::lr(p), F::ls(p), v);
            }
            fft_sync<WL>();
        }
    }
}

// Blocks of the x-stage kernels: a strip of 8 rows x a chunk of kc levels, launched as ONE dimension so that the strips an XCD
// works on are neighbours (block id % 8 = XCD): the row a strip shares with the next one (v, vt at j0+8; the spectral row j0-1)
// is then fetched into ONE L2 instead of two.
__device__ __forceinline__ void lds_strip_of_block(int nstrips, int& strip, int& chunk)
{
    const unsigned id = blockIdx.x;
    if ((nstrips & 7) == 0) { const unsigned x = id & 7u, t = id >> 3, per = (unsigned)nstrips >> 3; chunk = (int)(t / per); strip = (int)(x*per + (t - (unsigned)chunk*per)); }
    else { chunk = (int)(id / (unsigned)nstrips); strip = (int)(id - (unsigned)chunk*(unsigned)nstrips); }
}

// Slab-decomposed grids (npx = 1, npy = N; k_slab.hip): the x-stage kernels work on the rank's own rows and talk to the
// all-to-all buffers of the x <-> y transpose directly (src/transpose.cxx:170-219): layout [slice][peer q][k in slice][kxl][row],
// kx = q*nxb + kxl, the itot/2 + 1 modes unpacked (the Nyquist mode in a column of its own: every column is an ordinary complex
// column for the y stage) and padded with zeros to npy*nxb columns. The y stage of this form is slab_yfft_kernel below + the Thomas
// sweeps of k_slab.hip.
struct LdsSlab { int nxb, npy, ks; };                    // modes per peer, peers, levels per slice
// [slice][peer q][k in slice][kxl][row]: rows fastest, like S[k][kx][j] of the single-GPU form -- the x stages store / load eight
// rows of a mode as one 128-byte piece, and after the transpose a column's rows from one source rank are one run of jmax numbers
template<class TF> __device__ __forceinline__ size_t lds_xbuf_index(const LdsSlab& sl, int k, int kx, int row, int nrows)
{
    const int c = k / sl.ks, kk = k - c*sl.ks, q = kx / sl.nxb, kxl = kx - q*sl.nxb;
    return ((((size_t)c*sl.npy + q)*sl.ks + kk)*sl.nxb + kxl)*nrows + row;
}

// ======================================================================================================================
// (1) Pres_2::input + the transform along x. Block = 8 rows j0..j0+7, marching up through KC levels; thread = column i.
// ======================================================================================================================
template<class TF>
struct PresLdsIn
{
    GridDev<TF> g;
    const TF* u; const TF* v; const TF* w; const TF* ut; const TF* vt; const TF* wt; const TF* rhoref; const TF* rhorefh;
    TF dti;
    C2<TF>* S; const C2<TF>* Tx;      // Tx[m] = exp(-2 pi i m / itot), m < itot
    int nx;                           // log2(itot/2)
    int kc;                           // levels per block
    int kbeg, kend;                   // the levels of this launch (a k-slice of the sliced transposes; 0, kmax otherwise)
    int nrows;                        // rows worked: jtot, or the rank's jmax
    LdsSlab sl;                       // SLAB only
};
template<class TF, int RG, int BT, int NX, bool SLAB = false>
__global__ void __launch_bounds__(BT, (BT >= 128 ? 4 : 1)) pres_in_fftx_kernel(const PresLdsIn<TF> a)
{
    HIP_DYNAMIC_SHARED(LdsUnit, lds_raw);
    const GridDev<TF>& g = a.g;
    const int itot = g.itot, jtot = g.jtot, nh = itot >> 1, rp = nh + 2;
    C2<TF>* D = reinterpret_cast<C2<TF>*>(lds_raw);
    C2<TF>* T = D + 8*rp;
    const int tid = threadIdx.x;                   // blockDim.x == itot
    T[tid] = a.Tx[tid];
    int strip, chunk; lds_strip_of_block(a.nrows >> 3, strip, chunk);
    const int j0 = strip*8, k0 = a.kbeg + chunk*a.kc, k1 = (k0 + a.kc < a.kend) ? k0 + a.kc : a.kend;
    const int jj = g.icells, kk = g.ijcells;
    const int c0 = (tid + g.igc) + (j0 + g.jgc)*jj;
    const int team = nh >> 3, slot = tid / team, l = tid - slot*team;     // the transform this thread works on in the passes
    const bool active = slot < 8;
    TF* Dr = reinterpret_cast<TF*>(D);
    // the lower-face term of level k is the upper-face term of level k-1: carried
    TF low[8];
    {
        const int c = c0 + (k0 + g.kgc)*kk; const TF rh = uniform_load(a.rhorefh, k0 + g.kgc);
#pragma unroll
        for (int r=0; r<8; ++r) low[r] = rh * (a.wt[c + r*jj] + a.w[c + r*jj] * a.dti);
    }
    for (int k=k0; k<k1; ++k)
    {
        const int kd = k + g.kgc, c = c0 + kd*kk;
        const TF rk = uniform_load(a.rhoref, kd), rhp = uniform_load(a.rhorefh, kd+1), dzi = uniform_load(g.dzi, kd);
#pragma unroll
        for (int h=0; h<8; h+=RG)                     // RG rows at a time: the loads of a group in flight together
        {
            TF uc[RG], ue[RG], utc[RG], ute[RG], vv[RG+1], vvt[RG+1], wu[RG], wtu[RG];
#pragma unroll
            for (int r=0; r<RG; ++r)
            {
                const int cr = c + (h + r)*jj;
                uc[r] = a.u[cr]; ue[r] = a.u[cr+1]; utc[r] = a.ut[cr]; ute[r] = a.ut[cr+1];
                wu[r] = a.w[cr+kk]; wtu[r] = a.wt[cr+kk];
            }
#pragma unroll
            for (int r=0; r<RG+1; ++r) { const int cr = c + (h + r)*jj; vv[r] = a.v[cr]; vvt[r] = a.vt[cr]; }
#pragma unroll
            for (int r=0; r<RG; ++r)
            {
                const TF up = rhp * (wtu[r] + wu[r] * a.dti);
                // pres2_in (cell_ops.h), with the two vertical face terms named
                const TF d = rk * ( (ute[r] + ue[r] * a.dti) - (utc[r] + uc[r] * a.dti) ) * g.dxi_t
                           + rk * ( (vvt[r+1] + vv[r+1] * a.dti) - (vvt[r] + vv[r] * a.dti) ) * g.dyi_t
                           + ( up - low[h + r] ) * dzi;
                low[h + r] = up;
                Dr[2*((h + r)*rp + lds_slot<TF>(tid >> 1)) + (tid & 1)] = d;
            }
            sched_fence();
        }
        lds_barrier();
        // opaque per-level copies of the thread's indices: their address arithmetic stays inside the level instead of being hoisted
        // out of the loop into registers the transform then spills (see pres_ifftx_out_kernel)
        unsigned tl = (unsigned)tid, ll = (unsigned)l, sl = (unsigned)(active ? slot : 0); keep_vgpr(tl); keep_vgpr(ll); keep_vgpr(sl);
        { const C2<TF> none[fft_np(NX)][7] = {}; fft_batch_ct<-1, true, NX, false>(D + sl*rp, T, 1, (int)ll, a.nx, active, none); }
        lds_barrier();
        // real-to-complex: X[kx] = E + exp(-2 pi i kx / itot) O from Z[kx] and Z[nh - kx]; one (kx, row) element per thread and turn,
        // rows fastest: eight neighbouring threads write one 128-byte (fp64) piece of S[k][kx][j0..j0+7]
        if constexpr (!SLAB)
        {
            for (int e=(int)tl; e<8*nh; e+=itot)
            {
                const int kx = e >> 3, r = e & 7;
                const C2<TF> za = D[r*rp + lds_slot<TF>(kx)], zb = D[r*rp + lds_slot<TF>((nh - kx) & (nh-1))];
                const C2<TF> ev{TF(0.5)*(za.x + zb.x), TF(0.5)*(za.y - zb.y)};         // (Za + conj Zb) / 2
                const C2<TF> od{TF(0.5)*(za.y + zb.y), TF(0.5)*(zb.x - za.x)};         // (Za - conj Zb) / (2 i)
                C2<TF> x = ev + mul_tw<-1>(od, T[kx]);
                if (kx == 0) x = C2<TF>{za.x + za.y, za.x - za.y};                      // (X_0, X_nyq): both real, one column
                a.S[((size_t)k*nh + kx)*jtot + j0 + r] = x;
            }
        }
        else
        {
            // rows fastest, modes kx = 0 .. nh; kx = 0 and kx = nh (Nyquist) are real
            for (int e=(int)tl; e<8*(nh + 1); e+=itot)
            {
                const int kx = e >> 3, r = e & 7;
                const int ka = kx & (nh-1);
                const C2<TF> za = D[r*rp + lds_slot<TF>(ka)], zb = D[r*rp + lds_slot<TF>((nh - ka) & (nh-1))];
                const C2<TF> ev{TF(0.5)*(za.x + zb.x), TF(0.5)*(za.y - zb.y)};
                const C2<TF> od{TF(0.5)*(za.y + zb.y), TF(0.5)*(zb.x - za.x)};
                C2<TF> x = ev + mul_tw<-1>(od, T[ka]);
                if (kx == 0) x = C2<TF>{za.x + za.y, TF(0)};
                if (kx == nh) x = C2<TF>{za.x - za.y, TF(0)};
                a.S[lds_xbuf_index<TF>(a.sl, k, kx, j0 + r, a.nrows)] = x;
            }
        }
        lds_barrier();
    }
}

// ======================================================================================================================
// (2) Transforms along y around the Thomas sweeps. Block = one kx; thread = one ky; eight levels per round.
//
// The modes kx = 0 and kx = itot/2 are real along x, so stage 1 stores them as ONE complex column, S[k][0][j] = (X_0, X_nyq):
// itot/2 columns, a block each -- 256 blocks for 256 CUs at itot = 512 instead of 257. After the transform along y the two
// Hermitian spectra are separated again (Y_0[ky] = (Z[ky] + conj Z[-ky]) / 2, Y_nyq[ky] = (Z[ky] - conj Z[-ky]) / 2i); thread
// ky < N/2 of block 0 solves mode (0, ky), thread N - ky solves mode (itot/2, ky), and threads 0 and N/2 carry two REAL
// modes, (0, ky) in the real and (itot/2, ky) in the imaginary part, each with its own pivots ("two").
// ======================================================================================================================
template<class TF>
struct PresLdsSolve
{
    C2<TF>* S; const TF* W3;          // W3[k][kx][ky] = c[k-1] / w2[k-1] of the mode thread ky of block kx solves; row ncol: the second modes of "two"
    const TF* bmati; const TF* bmatj; const TF* a; const TF* c; const TF* dz; const TF* rho;
    const C2<TF>* Ty;                 // exp(-2 pi i m / jtot)
    int ncol, jtot, ny, kmax;         // ncol = itot/2 columns; ny = log2(jtot)
};
// 1 / x: hardware seed + two Newton steps (the pivots are O(1): no scaling needed)
template<class TF> __device__ __forceinline__ TF recip(TF x)
{
    TF y = recip_seed(x);
    TF e = tfma(-x, y, TF(1)); y = tfma(y, e, y);
    e = tfma(-x, y, TF(1));    y = tfma(y, e, y);
    return y;
}
// the diagonal of level k from its table entries (src/pres_2.cxx:289-330)
template<class TF>
__device__ __forceinline__ TF tdma_diag_vals(TF dz2, TF rho, TF ak, TF ck, TF bm, bool mean, int k, int kmax)
{
    TF b = dz2 * rho*bm - (ak+ck);
    if (k == 0) b += ak;
    if (k == kmax-1) { if (mean) b -= ck; else b += ck; }
    return b;
}
template<class TF>
__device__ __forceinline__ TF tdma_diag_lds(const PresLdsSolve<TF>& a, TF bm, bool mean, int k)
{
    // per-level tables through the scalar cache (uniform_load): as plain loads they become VECTOR loads of a uniform address, each
    // followed by s_waitcnt vmcnt(0) -- which also waits for the rows requested for the next round and for the stores of the last
    // level: eight full memory latencies per round (the y stage ran at 1.05 ms with them)
    const TF dzk = uniform_load(a.dz, k), ak = uniform_load(a.a, k), ck = uniform_load(a.c, k);
    const TF dz2 = dzk*dzk;
    TF b = dz2 * uniform_load(a.rho, k)*bm - (ak+ck);
    if (k == 0) b += ak;
    if (k == a.kmax-1) { if (mean) b -= ck; else b += ck; }
    return b;
}
// which mode thread ky of block kx solves: its bmati index
__device__ __forceinline__ int lds_mode_kx(int kx, int ky, int N, int ncol) { return (kx == 0 && ky > (N >> 1)) ? ncol : kx; }
// the Thomas pivots (src/pres_2.cxx:213-249) as reciprocals: w3[k] = c[k-1] * (1 / w2[k-1]), the same recurrence in the factor
// kernel (plan creation) and in the forward sweep, so that the two sweeps use one and the same factorisation
template<class TF>
__global__ void __launch_bounds__(64) pres_lds_factor_kernel(TF* __restrict__ W3, const PresLdsSolve<TF> a)
{
    const int ky = blockIdx.x*64 + threadIdx.x, row = blockIdx.y;          // rows 0 .. ncol
    if (ky >= a.jtot) return;
    const int kxa = (row == a.ncol) ? a.ncol : lds_mode_kx(row, ky, a.jtot, a.ncol);
    const TF bm = a.bmati[kxa] + a.bmatj[ky];
    const bool mean = (row == 0 && ky == 0);
    const size_t col = (size_t)row*a.jtot + ky, lev = (size_t)(a.ncol + 1)*a.jtot;
    TF inv = recip(tdma_diag_lds(a, bm, mean, 0));
    W3[col] = TF(0);
    for (int k=1; k<a.kmax; ++k)
    {
        const TF w3 = uniform_load(a.c, k-1) * inv;
        W3[col + k*lev] = w3;
        inv = recip(tdma_diag_lds(a, bm, mean, k) - uniform_load(a.a, k)*w3);
    }
}
template<class TF, int BT, int NY>
__global__ void __launch_bounds__(BT) pres_ysolve_kernel(const PresLdsSolve<TF> a)
{
    HIP_DYNAMIC_SHARED(LdsUnit, lds_raw);
    const int N = a.jtot, kmax = a.kmax, rp = N;
    C2<TF>* D = reinterpret_cast<C2<TF>*>(lds_raw);
    C2<TF>* T = D + 8*rp;
    const int ky = threadIdx.x, kx = blockIdx.x;   // blockDim.x == jtot
    T[ky] = a.Ty[ky];
    const int team = N >> 3, slot = ky / team, l = ky - slot*team;        // slot < 8 always
    constexpr bool TWC = (NY > 0 && BT <= 512);                           // twiddles in registers where the register file has the room
    C2<TF> tw[fft_np(NY)][7];
    if constexpr (TWC) { lds_barrier(); fft_twiddles_ct<NY>(T, 0, l, tw); }
    const size_t lev = (size_t)a.ncol*N, wlev = (size_t)(a.ncol + 1)*N;
    C2<TF>* Sc = a.S + (size_t)kx*N + ky;
    const TF* Wc = a.W3 + (size_t)kx*N + ky;
    const TF* Wc2 = a.W3 + (size_t)a.ncol*N + ky;
    const bool packed = (kx == 0);
    const bool two = packed && (ky == 0 || ky == (N >> 1));
    const bool upper = ky > (N >> 1);
    const int mir = (N - ky) & (N - 1);
    const TF bm = a.bmati[lds_mode_kx(kx, ky, N, a.ncol)] + a.bmatj[ky];
    const TF bm2 = a.bmati[a.ncol] + a.bmatj[ky];
    const bool mean = (kx == 0 && ky == 0);
    const int nround = (kmax + 7) >> 3;

    // ---- down: rows of eight levels -> LDS -> transform along y -> forward sweep -> S (in place)
    TF inv = TF(1), inv2 = TF(1); C2<TF> pp{TF(0), TF(0)};
    C2<TF> q[8];
#pragma unroll
    for (int m=0; m<8; ++m) q[m] = (m < kmax) ? Sc[m*lev] : C2<TF>{TF(0), TF(0)};
    for (int rd=0; rd<nround; ++rd)
    {
        const int k0 = rd << 3;
#pragma unroll
        for (int m=0; m<8; ++m) D[m*rp + lds_slot<TF>(ky)] = q[m];
        if (rd + 1 < nround)
        {
#pragma unroll
            for (int m=0; m<8; ++m) if (k0 + 8 + m < kmax) q[m] = Sc[(size_t)(k0 + 8 + m)*lev];
        }
        lds_barrier();
        { unsigned ll = (unsigned)l, sl = (unsigned)slot; keep_vgpr(ll); keep_vgpr(sl);       // per-round copies: see pres_ifftx_out_kernel
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
        // four levels at a time: their table entries are requested together (scalar loads, clamped indices: no branch between
        // them), then the dependent chain runs
#pragma unroll
        for (int h=0; h<8; h+=4)
        {
            TF dzv[4], av[4], cv[4], rv[4];
#pragma unroll
            for (int n=0; n<4; ++n)
            {
                const int kc = (k0 + h + n < kmax) ? k0 + h + n : kmax - 1;
                dzv[n] = uniform_load(a.dz, kc); av[n] = uniform_load(a.a, kc); cv[n] = uniform_load(a.c, kc); rv[n] = uniform_load(a.rho, kc);
            }
            TF cprev = uniform_load(a.c, (k0 + h > 0) ? k0 + h - 1 : 0);
#pragma unroll
            for (int n=0; n<4; ++n)
            {
                const int m = h + n, k = k0 + m;
                if (k < kmax)
                {
                    const TF dz2 = dzv[n]*dzv[n], ak = av[n];
                    TF w2 = tdma_diag_vals(dz2, rv[n], ak, cv[n], bm, mean, k, kmax);
                    C2<TF> r = r8[m];
                    r.x = dz2 * r.x; r.y = dz2 * r.y;
                    if (k > 0)
                    {
                        w2 -= ak * (cprev * inv);
                        r.x -= ak*pp.x; r.y -= ak*pp.y;
                    }
                    inv = recip(w2);
                    r.x *= inv;
                    if (two)
                    {
                        TF w2b = tdma_diag_vals(dz2, rv[n], ak, cv[n], bm2, false, k, kmax);
                        if (k > 0) w2b -= ak * (cprev * inv2);
                        inv2 = recip(w2b);
                        r.y *= inv2;
                    }
                    else r.y *= inv;
                    pp = r;
                    Sc[(size_t)k*lev] = r;
                }
                cprev = cv[n];
            }
        }
    }
    // ---- up: back substitution over eight levels -> LDS -> inverse transform along y -> S
    // (the values this thread reads back are the ones it wrote itself)
    TF w3[8], w3b[8];
    {
        const int k0 = (nround-1) << 3;
#pragma unroll
        for (int m=0; m<8; ++m)
        {
            w3[m] = TF(0); w3b[m] = TF(0);
            if (k0 + m < kmax) q[m] = Sc[(size_t)(k0 + m)*lev];
            if (k0 + m + 1 < kmax) { w3[m] = Wc[(size_t)(k0 + m + 1)*wlev]; if (two) w3b[m] = Wc2[(size_t)(k0 + m + 1)*wlev]; }
        }
    }
    pp = C2<TF>{TF(0), TF(0)};            // the solution of the level above (nothing above the top level: w3 = 0 there)
    for (int rd=nround-1; rd>=0; --rd)
    {
        const int k0 = rd << 3;
#pragma unroll
        for (int m=7; m>=0; --m)
        {
            const int k = k0 + m;
            C2<TF> r{TF(0), TF(0)};
            if (k < kmax)
            {
                r = q[m];
                if (k < kmax-1) { r.x -= w3[m]*pp.x; r.y -= (two ? w3b[m] : w3[m])*pp.y; }
                pp = r;
            }
            D[m*rp + lds_slot<TF>(ky)] = r;
        }
        if (rd > 0)
        {
#pragma unroll
            for (int m=0; m<8; ++m)
            {
                q[m] = Sc[(size_t)(k0 - 8 + m)*lev]; w3[m] = Wc[(size_t)(k0 - 8 + m + 1)*wlev];
                if (two) w3b[m] = Wc2[(size_t)(k0 - 8 + m + 1)*wlev];
            }
        }
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

// ----------------------------------------------------------------------------------------------------------------------
// (2t) The same stage with TWO blocks per kx (twisted factorisation): the levels [0, ksplit) are eliminated from the bottom up by
// block 2 kx -- the forward sweep above -- and the levels [ksplit, kmax) from the top down by block 2 kx + 1 (the mirror image: c
// and a change places, x_k = d''_k - a'_k x_{k-1}); a second launch solves the 2 x 2 system where the two meet
//     x_{m-1} = d'_{m-1} - c'_{m-1} x_m,   x_m = d''_m - a'_m x_{m-1}      (m = ksplit)
// and substitutes outward from there, each block through its own levels and back through the transform along y. Same operation
// count and array passes as the one-block form, half the dependent chain per block, twice the blocks: for grids with fewer columns
// than the chip has CUs (itot <= 256) and for one block per CU (itot = 512), where a round of eight levels is a chain of barriers
// and dependent fp64 operations. A different elimination order than the reference's tdma: inside the pressure tolerance
// (DESIGN.md "Parity"), not the same bits as the one-block form.
// ----------------------------------------------------------------------------------------------------------------------
template<class TF>
struct PresLdsSolveTw
{
    PresLdsSolve<TF> s;
    const TF* A3;                     // A3[k][kx][ky] = a[k] / w2'[k] of the top-down elimination (layout of W3; levels >= ksplit)
    C2<TF>* I;                        // I[kx][0 | 1][ky]: d' of level ksplit-1 and d'' of level ksplit, as the eliminations leave them -- the second
                                      // launch reads them here: in S the other block of the column may already have stored its solution over them
    int ksplit;                       // a multiple of 8, 8 <= ksplit < kmax
};
// the pivots of the top-down elimination as reciprocals, the same recurrence as in the sweep (see pres_lds_factor_kernel)
template<class TF>
__global__ void __launch_bounds__(64) pres_lds_factor_tw_kernel(TF* __restrict__ A3, const PresLdsSolveTw<TF> t)
{
    const PresLdsSolve<TF>& a = t.s;
    const int ky = blockIdx.x*64 + threadIdx.x, row = blockIdx.y;          // rows 0 .. ncol
    if (ky >= a.jtot) return;
    const int kxa = (row == a.ncol) ? a.ncol : lds_mode_kx(row, ky, a.jtot, a.ncol);
    const TF bm = a.bmati[kxa] + a.bmatj[ky];
    const bool mean = (row == 0 && ky == 0);
    const size_t col = (size_t)row*a.jtot + ky, lev = (size_t)(a.ncol + 1)*a.jtot;
    TF inv = TF(1);
    for (int k=a.kmax-1; k>=t.ksplit; --k)
    {
        TF w2 = tdma_diag_lds(a, bm, mean, k);
        if (k < a.kmax-1) w2 -= uniform_load(a.c, k) * (uniform_load(a.a, k+1) * inv);
        inv = recip(w2);
        A3[col + k*lev] = uniform_load(a.a, k) * inv;
    }
}
template<class TF, int BT, int NY, int PHASE>
__global__ void __launch_bounds__(BT) pres_ysolve_tw_kernel(const PresLdsSolveTw<TF> t)
{
    HIP_DYNAMIC_SHARED(LdsUnit, lds_raw);
    const PresLdsSolve<TF>& a = t.s;
    const int N = a.jtot, kmax = a.kmax, rp = N, ks = t.ksplit;
    C2<TF>* D = reinterpret_cast<C2<TF>*>(lds_raw);
    C2<TF>* T = D + 8*rp;
    const int ky = threadIdx.x, kx = blockIdx.x >> 1;                          // blockDim.x == jtot
    T[ky] = a.Ty[ky];
    const int team = N >> 3, slot = ky / team, l = ky - slot*team;
    constexpr bool TWC = (NY > 0 && BT <= 512);
    C2<TF> tw[fft_np(NY)][7];
    if constexpr (TWC) { lds_barrier(); fft_twiddles_ct<NY>(T, 0, l, tw); }
    const size_t lev = (size_t)a.ncol*N, wlev = (size_t)(a.ncol + 1)*N;
    C2<TF>* Sc = a.S + (size_t)kx*N + ky;
    const TF* Wc = a.W3 + (size_t)kx*N + ky;  const TF* Wc2 = a.W3 + (size_t)a.ncol*N + ky;
    const TF* Ac = t.A3 + (size_t)kx*N + ky;  const TF* Ac2 = t.A3 + (size_t)a.ncol*N + ky;
    const bool packed = (kx == 0);
    const bool two = packed && (ky == 0 || ky == (N >> 1));
    const bool upper = ky > (N >> 1);
    const int mir = (N - ky) & (N - 1);
    const TF bm = a.bmati[lds_mode_kx(kx, ky, N, a.ncol)] + a.bmatj[ky];
    const TF bm2 = a.bmati[a.ncol] + a.bmatj[ky];
    const bool mean = (kx == 0 && ky == 0);
    // the direction as a compile-time constant of the body (the register arrays of a round are then indexed by constants; indexed by
    // a run-time direction they live in scratch)
    auto body = [&](auto half_tag) __attribute__((always_inline))
    {
    constexpr int half = decltype(half_tag)::value;
    // this block's levels [kA, kB) in rounds of eight counted from kA
    const int kA = half ? ks : 0, kB = half ? kmax : ks;
    const int nround = (kB - kA + 7) >> 3;
    C2<TF> q[8];

    if constexpr (PHASE == 1)
    {
        // ---- elimination: rows of eight levels -> LDS -> transform along y -> sweep towards the split -> S (in place)
        TF inv = TF(1), inv2 = TF(1); C2<TF> pp{TF(0), TF(0)};
        const int rd0 = half ? nround-1 : 0, rdstep = half ? -1 : 1;
#pragma unroll
        for (int m=0; m<8; ++m) { const int k = kA + (rd0 << 3) + m; q[m] = (k < kB) ? Sc[(size_t)k*lev] : C2<TF>{TF(0), TF(0)}; }
        for (int n=0, rd=rd0; n<nround; ++n, rd+=rdstep)
        {
            const int k0 = kA + (rd << 3);
#pragma unroll
            for (int m=0; m<8; ++m) D[m*rp + lds_slot<TF>(ky)] = q[m];
            if (n + 1 < nround)
            {
                const int kn = k0 + rdstep*8;
#pragma unroll
                for (int m=0; m<8; ++m) if (kn + m < kB) q[m] = Sc[(size_t)(kn + m)*lev];
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
                    r = upper ? C2<TF>{TF(0.5)*(zm.y + r.y), TF(0.5)*(r.x - zm.x)}
                              : C2<TF>{TF(0.5)*(r.x + zm.x), TF(0.5)*(r.y - zm.y)};
                }
                r8[m] = r;
            }
            lds_barrier();
#pragma unroll
            for (int hh=0; hh<2; ++hh)
            {
                const int h = half ? 4 - 4*hh : 4*hh;                 // the group of four nearer to where the sweep comes from first
                TF dzv[4], av[4], cv[4], rv[4];
#pragma unroll
                for (int nn=0; nn<4; ++nn)
                {
                    const int kc = (k0 + h + nn < kmax) ? k0 + h + nn : kmax - 1;
                    dzv[nn] = uniform_load(a.dz, kc); av[nn] = uniform_load(a.a, kc); cv[nn] = uniform_load(a.c, kc); rv[nn] = uniform_load(a.rho, kc);
                }
                // the off-diagonal entry of the level the sweep comes from: c[k-1] going up, a[k+1] going down
                TF oprev = half ? uniform_load(a.a, (k0 + h + 4 < kmax) ? k0 + h + 4 : kmax - 1) : uniform_load(a.c, (k0 + h > 0) ? k0 + h - 1 : 0);
#pragma unroll
                for (int n4=0; n4<4; ++n4)
                {
                    const int nn = half ? 3 - n4 : n4, m = h + nn, k = k0 + m;
                    if (k < kB)
                    {
                        const TF dz2 = dzv[nn]*dzv[nn];
                        const TF off = half ? cv[nn] : av[nn];            // multiplies the neighbour already eliminated
                        const bool first = half ? (k == kmax-1) : (k == 0);
                        TF w2 = tdma_diag_vals(dz2, rv[nn], av[nn], cv[nn], bm, mean, k, kmax);
                        C2<TF> r = r8[m];
                        r.x = dz2 * r.x; r.y = dz2 * r.y;
                        if (!first)
                        {
                            w2 -= off * (oprev * inv);
                            r.x -= off*pp.x; r.y -= off*pp.y;
                        }
                        inv = recip(w2);
                        r.x *= inv;
                        if (two)
                        {
                            TF w2b = tdma_diag_vals(dz2, rv[nn], av[nn], cv[nn], bm2, false, k, kmax);
                            if (!first) w2b -= off * (oprev * inv2);
                            inv2 = recip(w2b);
                            r.y *= inv2;
                        }
                        else r.y *= inv;
                        pp = r;
                        Sc[(size_t)k*lev] = r;
                        if (k == (half ? ks : ks-1)) t.I[((size_t)kx*2 + half)*N + ky] = r;
                    }
                    oprev = half ? av[nn] : cv[nn];
                }
            }
        }
    }
    else
    {
        // ---- where the two sweeps meet: x_m and x_{m-1} from d'_{m-1}, d''_m and the two factors there
        C2<TF> pp;
        {
            const C2<TF> dl = t.I[((size_t)kx*2)*N + ky], du = t.I[((size_t)kx*2 + 1)*N + ky];
            const TF cp = Wc[(size_t)ks*wlev], ap = Ac[(size_t)ks*wlev];
            const TF cp2 = two ? Wc2[(size_t)ks*wlev] : cp, ap2 = two ? Ac2[(size_t)ks*wlev] : ap;
            C2<TF> xm;
            xm.x = (du.x - ap *dl.x) / (TF(1) - ap *cp);
            xm.y = (du.y - ap2*dl.y) / (TF(1) - ap2*cp2);
            pp = half ? C2<TF>{dl.x - cp*xm.x, dl.y - cp2*xm.y} : xm;      // the neighbour's solution this block starts from
        }
        // ---- substitution away from the split over eight levels -> LDS -> inverse transform along y -> S
        TF f3[8], f3b[8];
        const int rd0 = half ? 0 : nround-1, rdstep = half ? 1 : -1;
        auto request = [&](int k0)
        {
#pragma unroll
            for (int m=0; m<8; ++m)
            {
                const int k = (k0 + m < kB) ? k0 + m : kB - 1;
                q[m] = Sc[(size_t)k*lev];
                // going down: x_k = d'_k - c'_k x_{k+1}, c'_k = W3[k+1]; going up: x_k = d''_k - a'_k x_{k-1}
                f3[m] = half ? Ac[(size_t)k*wlev] : Wc[(size_t)(k + 1)*wlev];
                f3b[m] = two ? (half ? Ac2[(size_t)k*wlev] : Wc2[(size_t)(k + 1)*wlev]) : f3[m];
            }
        };
        request(kA + (rd0 << 3));
        for (int n=0, rd=rd0; n<nround; ++n, rd+=rdstep)
        {
            const int k0 = kA + (rd << 3);
#pragma unroll
            for (int n8=0; n8<8; ++n8)
            {
                const int m = half ? n8 : 7 - n8, k = k0 + m;
                C2<TF> r{TF(0), TF(0)};
                if (k < kB)
                {
                    r.x = q[m].x - f3[m]*pp.x; r.y = q[m].y - f3b[m]*pp.y;
                    pp = r;
                }
                D[m*rp + lds_slot<TF>(ky)] = r;
            }
            if (n + 1 < nround) request(k0 + rdstep*8);
            lds_barrier();
            if (packed)
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
            for (int m=0; m<8; ++m) if (k0 + m < kB) Sc[(size_t)(k0 + m)*lev] = D[m*rp + lds_slot<TF>(ky)];
            lds_barrier();
        }
    }
    };
    if (blockIdx.x & 1) body(std::integral_constant<int, 1>{}); else body(std::integral_constant<int, 0>{});
}

// ======================================================================================================================
// (3) The transform back along x + p + Pres_2::output. Block = rows j0-1 .. j0+7 (the first one only for the gradient in y),
// marching up through KC levels (one level below them first, for the gradient in z); thread = column i.
// ======================================================================================================================
template<class TF>
struct PresLdsOut
{
    GridDev<TF> g;
    const C2<TF>* S; const C2<TF>* Tx;
    TF* p; TF* ut; TF* vt; TF* wt;
    int nx, kc;
    int kbeg, kend;                   // the levels of this launch
    int nrows;                        // rows worked: jtot, or the rank's jmax
    LdsSlab sl;                       // SLAB only
    TF* u; TF* v; TF* w; TF cA, cB, rdt;   // RK only: the sub-step of the time integration for u, v, w (src/timeloop.cxx:250-334)
};
// RK: the corrected tendency of u, v, w is final when it is stored here, so the Runge-Kutta sub-step that follows Pres::exec in
// Model::exec (src/model.cxx:411,484; a += cB*dt*at, at *= cA) is applied to it in registers: two array passes per field instead
// of the four of a separate kernel. Same expressions in the same order as mhh_rk_substep: the same bits.
// SLAB: rows are the rank's own; the row south of the first strip belongs to the south neighbour, so vt of the rank's southernmost
// row is left to mhh_pres_output_south_row (after the one-row halo of p), and p gets its x halo only (the y halo is the exchange's)
template<class TF, int RG, int BT, int NX, bool SLAB = false, bool RK = false>
__global__ void __launch_bounds__(BT, (BT >= 128 ? 4 : 1)) pres_ifftx_out_kernel(const PresLdsOut<TF> a)
{
    HIP_DYNAMIC_SHARED(LdsUnit, lds_raw);
    const GridDev<TF>& g = a.g;
    const int itot = g.itot, jtot = g.jtot, nh = itot >> 1, rp = nh + 2;
    C2<TF>* D = reinterpret_cast<C2<TF>*>(lds_raw);       // row r+1 of the strip at D + (r+1)*rp; row 0 = j0-1
    C2<TF>* T = D + 9*rp;
    const int tid = threadIdx.x;                   // blockDim.x == itot
    T[tid] = a.Tx[tid];
    int strip, chunk; lds_strip_of_block(a.nrows >> 3, strip, chunk);
    const int j0 = strip*8, k0 = a.kbeg + chunk*a.kc, k1 = (k0 + a.kc < a.kend) ? k0 + a.kc : a.kend;
    const int jj = g.icells, kk = g.ijcells;
    const int team = nh >> 3, slot = tid / team, l = tid - slot*team;
    const bool active = slot < 9;
    const bool has_south = !SLAB || j0 > 0;                            // row j0-1 is at hand
    const TF* Dr = reinterpret_cast<const TF*>(D);
    const TF nrm = (TF(1) / TF(jtot)) * (TF(1) / TF(itot));           // both powers of two: exact
    const int jsouth = SLAB ? j0 - 1 : ((j0 + jtot - 1) & (jtot - 1));
    const int iw = (tid + itot - 1) & (itot - 1);
    // p of the level below, eight rows of this thread's column: in LDS rather than in sixteen registers (see the note on spills below)
    TF* below = reinterpret_cast<TF*>(T + itot) + tid;               // below[r*itot]
    for (int k = (k0 > 0 ? k0-1 : 0); k<k1; ++k)
    {
        const bool emit = (k >= k0);
        const int c = (tid + g.igc) + (j0 + g.jgc)*jj + (k + g.kgc)*kk;
        const TF dzhi_k = uniform_load(g.dzhi, k + g.kgc);
        // The index arithmetic below depends on the thread only; hoisted out of the level loop it occupies registers across the
        // transform, which then spills -- and every reload of a spilled register is a scratch load followed by s_waitcnt vmcnt(0),
        // i.e. a wait for every tendency store still in flight (cycle stamps: the transform of a level took 31 600 of its 62 600
        // cycles). An opaque copy of the thread index per level keeps that arithmetic inside the level.
        unsigned tl = (unsigned)tid; keep_vgpr(tl);
        // spectral rows -> LDS (columns 0 .. nh-1, column 0 = (X_0, X_nyq); rows fastest in memory)
        if constexpr (!SLAB)
        {
            for (int e=(int)tl; e<9*nh; e+=itot)
            {
                const int kx = e / 9, r = e - 9*kx;
                const int j = (r == 0) ? jsouth : j0 + r - 1;
                D[r*rp + lds_slot<TF>(kx)] = a.S[((size_t)k*nh + kx)*jtot + j];
            }
        }
        else
        {
            TF* Dw = reinterpret_cast<TF*>(D);
            for (int e=(int)tl; e<9*(nh + 1); e+=itot)
            {
                const int kx = e / 9, r = e - 9*kx;
                const int j = j0 + r - 1;
                if (r == 0 && !has_south) continue;                      // (that row of LDS stays undefined; its p is never used)
                const C2<TF> x = a.S[lds_xbuf_index<TF>(a.sl, k, kx, j, a.nrows)];
                if (kx == 0)       Dw[2*(r*rp + lds_slot<TF>(0))] = x.x;          // column 0 = (X_0, X_nyq)
                else if (kx == nh) Dw[2*(r*rp + lds_slot<TF>(0)) + 1] = x.x;
                else               D[r*rp + lds_slot<TF>(kx)] = x;
            }
        }
        lds_barrier();
        // complex-to-real: Z[kx] = (Xa + conj Xb) + i (Xa - conj Xb) exp(+2 pi i kx / itot), Xb = X[nh - kx]; pairs (kx, nh - kx) in place
        for (int e=(int)tl; e<9*(nh/2 + 1); e+=itot)
        {
            const int r = e / (nh/2 + 1), kx = e - r*(nh/2 + 1), kb = nh - kx;
            const C2<TF> xa = D[r*rp + lds_slot<TF>(kx)], xb = D[r*rp + lds_slot<TF>(kb & (nh-1))];
            if (kx == 0) D[r*rp] = C2<TF>{xa.x + xa.y, xa.x - xa.y};                // X_0 and X_nyq are real: Z[0] = (X_0 + X_nyq) + i (X_0 - X_nyq)
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
        { unsigned ll = (unsigned)l, sl = (unsigned)(active ? slot : 0); keep_vgpr(ll); keep_vgpr(sl);      // likewise: the transform's addresses
          const C2<TF> none[fft_np(NX)][7] = {}; fft_batch_ct<+1, true, NX, false>(D + sl*rp, T, 1, (int)ll, a.nx, active, none); }
        lds_barrier();
        // rows of p: element i of row r at real index 2*lds_slot<TF>(i/2) + (i&1)
        const int oc = 2*lds_slot<TF>(tid >> 1) + (tid & 1), ow = 2*lds_slot<TF>(iw >> 1) + (iw & 1);
        TF ps = Dr[2*(0*rp) + oc] * nrm;
#pragma unroll
        for (int h=0; h<8; h+=RG)
        {
            TF tu[RG], tv[RG], tw[RG];
            TF fu[RG], fv[RG], fw[RG];                                 // RK: the fields themselves
            if (emit)
            {
#pragma unroll
                for (int r=0; r<RG; ++r) { tu[r] = a.ut[c + (h+r)*jj]; tv[r] = a.vt[c + (h+r)*jj]; tw[r] = a.wt[c + (h+r)*jj]; }
                if constexpr (RK)
                {
#pragma unroll
                    for (int r=0; r<RG; ++r) { fu[r] = a.u[c + (h+r)*jj]; fv[r] = a.v[c + (h+r)*jj]; fw[r] = a.w[c + (h+r)*jj]; }
                }
            }
#pragma unroll
            for (int q=0; q<RG; ++q)
            {
                const int r = h + q;
                const TF pc = Dr[2*((r+1)*rp) + oc] * nrm, pw = Dr[2*((r+1)*rp) + ow] * nrm;
                if (emit)
                {
                    const int cr = c + r*jj;
                    const TF pb = (k == 0) ? pc : below[r*itot];                          // p[kstart-1] = p[kstart]
                    const TF nut = tu[q] - (pc - pw) * g.dxi_t, nvt = tv[q] - (pc - ps) * g.dyi_t, nwt = tw[q] - (pc - pb) * dzhi_k;
                    if constexpr (RK)
                    {
                        a.u[cr] = fu[q] + a.cB*a.rdt*nut; a.ut[cr] = a.cA*nut;
                        a.v[cr] = fv[q] + a.cB*a.rdt*nvt; a.vt[cr] = a.cA*nvt;
                        a.w[cr] = fw[q] + a.cB*a.rdt*nwt; a.wt[cr] = a.cA*nwt;
                    }
                    else
                    {
                        a.ut[cr] = nut;
                        if (has_south || r > 0) a.vt[cr] = nvt;
                        a.wt[cr] = nwt;
                    }
                    // p: the cell, its images in the periodic halo, and the ghost level below the first one
                    const int js = j0 + r;
                    for (int lv=0; lv<2; ++lv)
                    {
                        if (lv == 1 && k != 0) break;
                        const int cl = cr - lv*kk;
                        for (int rowsel=0; rowsel<3; ++rowsel)
                        {
                            int off;
                            if (rowsel == 0) off = 0;
                            else if (SLAB) continue;                                                        // the y halo is the neighbours'
                            else if (rowsel == 1) { if (js < jtot - g.jgc) continue; off = -jtot*jj; }     // row js - jtot: the south halo
                            else                  { if (js >= g.jgc) continue;       off =  jtot*jj; }     // row js + jtot: the north halo
                            a.p[cl + off] = pc;
                            if (tid >= itot - g.igc) a.p[cl + off - itot] = pc;
                            if (tid < g.igc)         a.p[cl + off + itot] = pc;
                        }
                    }
                }
                below[r*itot] = pc; ps = pc;
            }
            sched_fence();
        }
        lds_barrier();
    }
}


// ======================================================================================================================
// (2s) The transforms along y of the slab form: block = one local mode kxl and eight levels, thread = one row j / mode ky.
// FWD: the receive buffer of the x -> y transpose ([slice][source rank][k][kxl][jl]: a column's rows from one rank are one run of
// jmax numbers) -> LDS -> transform -> specy[k][kxl][ky] (unit stride in ky for the Thomas sweeps of k_slab.hip).
// !FWD: specy -> LDS -> inverse transform -> the send buffer of the y -> x transpose, same layout.
// One read and one write of the rank's spectral slab per direction; the staged y stage (reorder through LDS tiles + rocFFT in place)
// makes two of each.
// ======================================================================================================================
template<class TF>
struct SlabYfft
{
    C2<TF>* xbuf; C2<TF>* specy; const C2<TF>* Ty;
    int jtot, jmax, ny, nxb, kbeg, kend;
    LdsSlab sl;
};
template<class TF, int BT, int NY, bool FWD>
__global__ void __launch_bounds__(BT) slab_yfft_kernel(const SlabYfft<TF> a)
{
    HIP_DYNAMIC_SHARED(LdsUnit, lds_raw);
    const int N = a.jtot, rp = N;
    C2<TF>* D = reinterpret_cast<C2<TF>*>(lds_raw);
    C2<TF>* T = D + 8*rp;
    const int ky = threadIdx.x, kxl = blockIdx.x;            // blockDim.x == jtot
    T[ky] = a.Ty[ky];
    const int team = N >> 3, slot = ky / team, l = ky - slot*team;
    const int k0 = a.kbeg + 8*(int)blockIdx.y;
    const int r = ky / a.jmax, jl = ky - r*a.jmax;           // source / destination rank and row of this thread's y
    const int kx = r*a.sl.nxb + kxl;                         // (lds_xbuf_index takes the global mode: q = kx / nxb = r)
    C2<TF> q[8];
#pragma unroll
    for (int m=0; m<8; ++m)
    {
        const int k = k0 + m;
        q[m] = C2<TF>{TF(0), TF(0)};
        if (k < a.kend) q[m] = FWD ? a.xbuf[lds_xbuf_index<TF>(a.sl, k, kx, jl, a.jmax)] : a.specy[((size_t)k*a.nxb + kxl)*N + ky];
    }
#pragma unroll
    for (int m=0; m<8; ++m) D[m*rp + lds_slot<TF>(ky)] = q[m];
    lds_barrier();
    { const C2<TF> none[fft_np(NY)][7] = {}; fft_batch_ct<(FWD ? -1 : +1), (BT <= 512), NY, false>(D + slot*rp, T, 0, l, a.ny, true, none); }
    if (BT <= 512) lds_barrier();
#pragma unroll
    for (int m=0; m<8; ++m)
    {
        const int k = k0 + m;
        if (k < a.kend)
        {
            const C2<TF> v = D[m*rp + lds_slot<TF>(ky)];
            if (FWD) a.specy[((size_t)k*a.nxb + kxl)*N + ky] = v;
            else     a.xbuf[lds_xbuf_index<TF>(a.sl, k, kx, jl, a.jmax)] = v;
        }
    }
}

}} // namespace mhh::lds_fft
