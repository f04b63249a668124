// cell_ops.h -- per-cell arithmetic of the RHS operators (advec_2 / advec_2i5 / advec_4, diff_2 / diff_4 /
// diff_smag2, pres in/out) as __host__ __device__ inline functions.
//
// Every function returns the increment(s) the reference adds to ONE tendency cell, with the reference's
// expression association (file:line cited per function, paths relative to the reference root), so that
// with -ffp-contract=off the HIP kernels are bit-identical to the reference CPU path in fp64 and fp32.
// The functions take a pointer + flat index + strides, so the same body serves one-cell-per-thread
// kernels and the fused multi-tendency kernels. Compiling this header with a host compiler
// (MHH_HD empty) is how tests/emul checks the arithmetic on the CPU before any GPU time is spent.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#  define MHH_HD __host__ __device__ __forceinline__
#else
#  define MHH_HD inline
#endif

namespace mhh
{
// Kernel-side grid descriptor (subset of Grid_data, include/grid.h:49-135, narrowed to TF on the host).
template<class TF>
struct GridDev
{
    int itot, jtot, ktot, imax, jmax, kmax, igc, jgc, kgc;
    int icells, jcells, ijcells, kcells;
    int istart, iend, jstart, jend, kstart, kend;
    int dim3;                 // jtot != 1 (Advec_4::exec, src/advec_4.cxx:596)
    TF dx, dy, zsize;
    TF dxi_t, dyi_t;          // TF(1.)/dx : spelling of advec_*, pres_2  (TF quotient)
    TF dxi_d, dyi_d;          // TF(1./dx) : spelling of diff_smag2, pres_4, calc_cfl of advec_2/2i5 (double quotient, narrowed)
    TF dxidxi_d, dyidyi_d;    // TF(1./(dx*dx)): diff_smag2 diff_c / dnmul, diff_4 diff_c
    TF dxidxi_t, dyidyi_t;    // TF(1/(dx*dx)) : diff_4 diff_w
    double dxidxi_2, dyidyi_2;// double(TF(1)/(dx*dx)): diff_2 (src/diff_2.cxx:44-45, a double even in SP builds)
    const TF* z; const TF* dz; const TF* dzi; const TF* dzhi; const TF* dzi4; const TF* dzhi4;
};

MHH_HD double tabs(double a) { return __builtin_fabs(a); }
MHH_HD float  tabs(float a)  { return __builtin_fabsf(a); }

// ---- F2: TWO single-precision cells per lane (cells i, i+1 of a row). Every operation is the element-wise fp32 operation, so
// a kernel body written once against its "lane value" type computes, with F2, the same bits per cell as with float -- on
// gfx950 as packed instructions (v_pk_add_f32, v_pk_mul_f32: two cells per issue slot, no contraction involved). Used by the
// k-marching kernel for the fp32 configuration (k_march.hip); scalars (metrics, coefficients) stay plain float and broadcast.
#if defined(__clang__)
typedef float mhh_f2v __attribute__((ext_vector_type(2)));
struct F2
{
    mhh_f2v v;
    MHH_HD F2() {}
    MHH_HD F2(float s) : v{s, s} {}
    MHH_HD explicit F2(double s) : v{(float)s, (float)s} {}
    MHH_HD explicit F2(int s) : v{(float)s, (float)s} {}
    MHH_HD F2(mhh_f2v x) : v(x) {}
    MHH_HD F2(float a, float b) : v{a, b} {}
    MHH_HD float lo() const { return v.x; }
    MHH_HD float hi() const { return v.y; }
};
MHH_HD F2 operator+(F2 a, F2 b) { return F2(a.v + b.v); }
MHH_HD F2 operator-(F2 a, F2 b) { return F2(a.v - b.v); }
MHH_HD F2 operator*(F2 a, F2 b) { return F2(a.v * b.v); }
MHH_HD F2 operator/(F2 a, F2 b) { return F2(a.v / b.v); }
MHH_HD F2 operator-(F2 a) { return F2(-a.v); }
MHH_HD F2 tabs(F2 a) { return F2(__builtin_elementwise_abs(a.v)); }
MHH_HD F2 tfma(F2 a, F2 b, F2 c) { return F2(__builtin_fmaf(a.v.x, b.v.x, c.v.x), __builtin_fmaf(a.v.y, b.v.y, c.v.y)); }
#else   // host compilers (tests/emul): the same type, element by element
struct F2
{
    float v[2];
    F2() {}
    F2(float s) : v{s, s} {}
    explicit F2(double s) : v{(float)s, (float)s} {}
    explicit F2(int s) : v{(float)s, (float)s} {}
    F2(float a, float b) : v{a, b} {}
    float lo() const { return v[0]; }
    float hi() const { return v[1]; }
};
inline F2 operator+(F2 a, F2 b) { return F2(a.v[0] + b.v[0], a.v[1] + b.v[1]); }
inline F2 operator-(F2 a, F2 b) { return F2(a.v[0] - b.v[0], a.v[1] - b.v[1]); }
inline F2 operator*(F2 a, F2 b) { return F2(a.v[0] * b.v[0], a.v[1] * b.v[1]); }
inline F2 operator/(F2 a, F2 b) { return F2(a.v[0] / b.v[0], a.v[1] / b.v[1]); }
inline F2 operator-(F2 a) { return F2(-a.v[0], -a.v[1]); }
inline F2 tabs(F2 a) { return F2(__builtin_fabsf(a.v[0]), __builtin_fabsf(a.v[1])); }
inline F2 tfma(F2 a, F2 b, F2 c) { return F2(__builtin_fmaf(a.v[0], b.v[0], c.v[0]), __builtin_fmaf(a.v[1], b.v[1], c.v[1])); }
#endif
MHH_HD F2 operator+(F2 a) { return a; }
MHH_HD F2 operator+(F2 a, float b) { return a + F2(b); }
MHH_HD F2 operator+(float a, F2 b) { return F2(a) + b; }
MHH_HD F2 operator-(F2 a, float b) { return a - F2(b); }
MHH_HD F2 operator-(float a, F2 b) { return F2(a) - b; }
MHH_HD F2 operator*(F2 a, float b) { return a * F2(b); }
MHH_HD F2 operator*(float a, F2 b) { return F2(a) * b; }
MHH_HD F2 operator/(F2 a, float b) { return a / F2(b); }
MHH_HD F2& operator+=(F2& a, F2 b) { a = a + b; return a; }
MHH_HD F2& operator-=(F2& a, F2 b) { a = a - b; return a; }
// lane-value traits: the scalar type of the arithmetic and the cells a lane carries
template<class VT> struct lane_of { typedef VT scalar; static constexpr int cells = 1; };
template<> struct lane_of<F2> { typedef float scalar; static constexpr int cells = 2; };
template<class TF> MHH_HD TF tmin(TF a, TF b) { return (b < a) ? b : a; }   // std::min semantics
template<class TF> MHH_HD TF tmax(TF a, TF b) { return (a < b) ? b : a; }   // std::max semantics
template<class TF> MHH_HD TF sq(TF a) { return a*a; }

// ---- interpolation weights (include/finite_difference.h:36-155) -------------------------------------
template<class TF> MHH_HD TF i2(TF a, TF b) { return TF(0.5)*(a+b); }
template<class TF> MHH_HD TF i6(TF a, TF b, TF c, TF d, TF e, TF f)
{ return TF(37./60.)*(c+d) - TF(8./60.)*(b+e) + TF(1./60.)*(a+f); }
template<class TF> MHH_HD TF i5(TF a, TF b, TF c, TF d, TF e, TF f)
{ return TF(10./60.)*(d-c) - TF(5./60.)*(e-b) + TF(1./60.)*(f-a); }
template<class TF> MHH_HD TF i4ws(TF a, TF b, TF c, TF d) { return TF(7./12.)*(b+c) - TF(1./12.)*(a+d); }
template<class TF> MHH_HD TF i3ws(TF a, TF b, TF c, TF d) { return TF(3./12.)*(c-b) - TF(1./12.)*(d-a); }

// Four-point weights, summed left to right as the reference writes them. A product by a power of two is exact (no rounding
// unless it underflows: |x| < 2^-1018 in fp64), so RN(w*x + t) with such a w is ONE fma with the bits of the multiplication
// followed by the addition -- an instruction less per such weight (the -1/16 at both ends of the 4th-order interpolation: two of
// its seven operations, 108 of the ~900 vector instructions per cell of advec_4 + diff_4). The other weights keep their two
// roundings. F2 (two fp32 cells per lane) keeps the product-sum: its operations are packed instructions.
constexpr bool w_is_pow2(double w) { double a = w < 0 ? -w : w; if (a == 0) return false; while (a < 1) a *= 2; while (a > 1) a /= 2; return a == 1; }
template<class TF> MHH_HD TF pw2_fma(double w, TF x, TF t) { return TF(w)*x + t; }
MHH_HD double pw2_fma(double w, double x, double t) { return __builtin_fma(w, x, t); }
MHH_HD float  pw2_fma(double w, float x, float t)   { return __builtin_fmaf((float)w, x, t); }
#ifndef MHH_W4_POW2_FMA
#define MHH_W4_POW2_FMA 1
#endif
#define MHH_W4(name, w0, w1, w2, w3) \
    template<class TF> MHH_HD TF name(TF a, TF b, TF c, TF d) { \
        TF t; \
        if constexpr (MHH_W4_POW2_FMA && w_is_pow2(w0))      t = pw2_fma(w0, a, TF(w1)*b); \
        else if constexpr (MHH_W4_POW2_FMA && w_is_pow2(w1)) t = pw2_fma(w1, b, TF(w0)*a); \
        else                                                 t = TF(w0)*a + TF(w1)*b; \
        if constexpr (MHH_W4_POW2_FMA && w_is_pow2(w2)) t = pw2_fma(w2, c, t); else t = t + TF(w2)*c; \
        if constexpr (MHH_W4_POW2_FMA && w_is_pow2(w3)) t = pw2_fma(w3, d, t); else t = t + TF(w3)*d; \
        return t; }
MHH_W4(ci4, -1./16.,  9./16.,  9./16., -1./16.)
MHH_W4(bi4,  5./16., 15./16., -5./16.,  1./16.)
MHH_W4(ti4,  1./16., -5./16., 15./16.,  5./16.)
MHH_W4(cg4,  1./24., -27./24., 27./24., -1./24.)
MHH_W4(bg4, -23./24., 21./24.,  3./24., -1./24.)
MHH_W4(tg4,  1./24., -3./24., -21./24., 23./24.)
#undef MHH_W4
template<class TF> MHH_HD TF i4c(TF a, TF b, TF c, TF d) { return MHH_W4_POW2_FMA ? pw2_fma(-1./16., a+d, TF(9./16.)*(b+c)) : TF(-1./16.)*(a+d) + TF(9./16.)*(b+c); }

// =======================================================================================================
// advec_2 (src/advec_2.cxx:81-202). COMP: 0=u 1=v 2=w (momentum, staggering offset o = -1/-jj/-kk), 3=scalar.
// rt/rb/rc/dz are the face/cell density weights and the metric of the row k (chosen by the caller):
//   u,v,s: rt=rhorefh[k+1] rb=rhorefh[k] rc=rhoref[k] dz=dzi[k];  w: rt=rhoref[k] rb=rhoref[k-1] rc=rhorefh[k] dz=dzhi[k]
// =======================================================================================================
template<class TF>
MHH_HD TF advec2_mom(const TF* __restrict__ f, const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w,
                     int c, int o, int jj, int kk, TF dxi, TF dyi, TF rt, TF rb, TF rc, TF dz)
{
    return
        - ( i2(u[c+1+o], u[c+1]) * i2(f[c], f[c+1])
          - i2(u[c  +o], u[c  ]) * i2(f[c-1], f[c]) ) * dxi
        - ( i2(v[c+jj+o], v[c+jj]) * i2(f[c], f[c+jj])
          - i2(v[c   +o], v[c   ]) * i2(f[c-jj], f[c]) ) * dyi
        - ( rt * i2(w[c+kk+o], w[c+kk]) * i2(f[c], f[c+kk])
          - rb * i2(w[c   +o], w[c   ]) * i2(f[c-kk], f[c]) ) / rc * dz;
}
template<class TF>
MHH_HD TF advec2_s(const TF* __restrict__ s, const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w,
                   int c, int jj, int kk, TF dxi, TF dyi, TF rt, TF rb, TF rc, TF dz)
{
    return
        - ( u[c+1]  * i2(s[c], s[c+1])  - u[c] * i2(s[c-1],  s[c]) ) * dxi
        - ( v[c+jj] * i2(s[c], s[c+jj]) - v[c] * i2(s[c-jj], s[c]) ) * dyi
        - ( rt * w[c+kk] * i2(s[c], s[c+kk]) - rb * w[c] * i2(s[c-kk], s[c]) ) / rc * dz;
}

// =======================================================================================================
// advec_2i5 (src/advec_2i5.cxx:151-728)
// Vertical order table: order of the interpolation on a face. 0 = wall (no flux), 2 = 2nd (no upwind term),
// 4 = 4th centred + 3rd upwind, 6 = 6th centred + 5th upwind. For u,v,scalars `kf` is the bottom face of cell
// kf; for w it is the cell centre (the w equation's "faces").
// =======================================================================================================
MHH_HD int order_face_c(int kf, int kstart, int kend)
{
    if (kf <= kstart || kf >= kend) return 0;
    if (kf == kstart+1 || kf == kend-1) return 2;
    if (kf == kstart+2 || kf == kend-2) return 4;
    return 6;
}
MHH_HD int order_face_w(int kc, int kstart, int kend)
{
    if (kc == kstart || kc == kend-1) return 2;
    if (kc == kstart+1 || kc == kend-2) return 4;
    return 6;
}
// interpolants on the face between f[c-s] and f[c]
template<class TF> MHH_HD TF face_cen(const TF* __restrict__ f, int c, int s, int order)
{
    if (order == 2) return i2(f[c-s], f[c]);
    if (order == 4) return i4ws(f[c-2*s], f[c-s], f[c], f[c+s]);
    return i6(f[c-3*s], f[c-2*s], f[c-s], f[c], f[c+s], f[c+2*s]);
}
template<class TF> MHH_HD TF face_upw(const TF* __restrict__ f, int c, int s, int order)
{
    if (order == 4) return i3ws(f[c-2*s], f[c-s], f[c], f[c+s]);
    return i5(f[c-3*s], f[c-2*s], f[c-s], f[c], f[c+s], f[c+2*s]);
}

// horizontal increment (:181-201, :333-352, :483-503, :612-632); ue/uw/vn/vs = advecting face velocities
template<class TF>
MHH_HD TF advec25_hor(const TF* __restrict__ f, int c, int jj, TF ue, TF uw, TF vn, TF vs, TF dxi, TF dyi)
{
    const TF fm3 = f[c-3], fm2 = f[c-2], fm1 = f[c-1], f0 = f[c], fp1 = f[c+1], fp2 = f[c+2], fp3 = f[c+3];
    const TF gm3 = f[c-3*jj], gm2 = f[c-2*jj], gm1 = f[c-jj], gp1 = f[c+jj], gp2 = f[c+2*jj], gp3 = f[c+3*jj];
    return
        - ( ue * i6(fm2, fm1, f0, fp1, fp2, fp3) - uw * i6(fm3, fm2, fm1, f0, fp1, fp2) ) * dxi
        + ( tabs(ue) * i5(fm2, fm1, f0, fp1, fp2, fp3) - tabs(uw) * i5(fm3, fm2, fm1, f0, fp1, fp2) ) * dxi
        - ( vn * i6(gm2, gm1, f0, gp1, gp2, gp3) - vs * i6(gm3, gm2, gm1, f0, gp1, gp2) ) * dyi
        + ( tabs(vn) * i5(gm2, gm1, f0, gp1, gp2, gp3) - tabs(vs) * i5(gm3, gm2, gm1, f0, gp1, gp2) ) * dyi;
}
// the same with the cell's own value handed over (the marching kernel holds it in a register; f points at the cell in an
// LDS plane of row pitch jj). ue/uw/vn/vs may be SUMS a+b of the two velocities a face averages, with dxi/dyi halved by the
// caller: scaling by 2 commutes with every rounding of the expression (no overflow; exact unless a velocity sum is subnormal),
// so 0.5*(a+b) * I * dxi and (a+b) * I * (0.5*dxi) are the same bits -- one multiplication per face less.
// P = something indexable at the cell (a pointer, or the marching kernel's plane view), TF = the lane value, MT = the metrics' type
template<class P, class TF, class MT>
MHH_HD TF advec25_hor_f0(const P& f, TF f0, int jj, TF ue, TF uw, TF vn, TF vs, MT dxi, MT dyi)
{
    const TF fm3 = f[-3], fm2 = f[-2], fm1 = f[-1], fp1 = f[1], fp2 = f[2], fp3 = f[3];
    const TF gm3 = f[-3*jj], gm2 = f[-2*jj], gm1 = f[-jj], gp1 = f[jj], gp2 = f[2*jj], gp3 = f[3*jj];
    return
        - ( ue * i6(fm2, fm1, f0, fp1, fp2, fp3) - uw * i6(fm3, fm2, fm1, f0, fp1, fp2) ) * dxi
        + ( tabs(ue) * i5(fm2, fm1, f0, fp1, fp2, fp3) - tabs(uw) * i5(fm3, fm2, fm1, f0, fp1, fp2) ) * dxi
        - ( vn * i6(gm2, gm1, f0, gp1, gp2, gp3) - vs * i6(gm3, gm2, gm1, f0, gp1, gp2) ) * dyi
        + ( tabs(vn) * i5(gm2, gm1, f0, gp1, gp2, gp3) - tabs(vs) * i5(gm3, gm2, gm1, f0, gp1, gp2) ) * dyi;
}

// x / d for a divisor known in advance, with r = RN(1/d) from an IEEE division on the host: the correctly rounded quotient in
// five multiply-adds instead of the ~14 issue slots of a full fp64 division (v_rcp_f64, v_div_scale/fmas/fixup).
// q0 = RN(x r) is within 1.5 ulp of x/d; one residual correction makes it a faithful rounding (the residual is exact or
// rounded by < 2^-53 of itself, the correction term carries the 2^-53 relative error of r: together < 2^-51 ulp); from a
// faithful quotient a second correction with the exact residual RN(x - d q) gives RN(x/d) (Markstein's theorem; it needs
// r = RN(1/d) and a significand of d that is not all ones -- the host checks that, known_divisor_ok). Exact for x = 0 and for
// every x whose residual does not underflow (|x| > 2^-960 in fp64): eddy viscosities are never that small.
MHH_HD double tfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
MHH_HD float  tfma(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
template<class VT, class TF>
MHH_HD VT div_known(VT x, TF d, TF r)
{
    const VT md = VT(-d), rr = VT(r);
    VT q = x * rr;
    VT e = tfma(md, q, x);
    q = tfma(e, rr, q);
    e = tfma(md, q, x);
    return tfma(e, rr, q);
}

// vertical increment (:204-299 etc.) for face orders ot (top) / ob (bottom)
template<class TF>
MHH_HD TF advec25_ver(const TF* __restrict__ f, int c, int kk, int ot, int ob, TF wt, TF wb, TF rt, TF rb, TF rc, TF dz)
{
    const TF It = ot ? face_cen(f, c+kk, kk, ot) : TF(0);
    const TF Ib = ob ? face_cen(f, c, kk, ob) : TF(0);
    TF cen;
    if (ob == 0)      cen = - ( rt * wt * It ) / rc * dz;
    else if (ot == 0) cen = - ( -rb * wb * Ib ) / rc * dz;
    else              cen = - ( rt * wt * It - rb * wb * Ib ) / rc * dz;
    const bool ut = (ot >= 4), ub = (ob >= 4);
    if (ut && ub) return cen + ( rt * tabs(wt) * face_upw(f, c+kk, kk, ot) - rb * tabs(wb) * face_upw(f, c, kk, ob) ) / rc * dz;
    if (ut)       return cen + ( rt * tabs(wt) * face_upw(f, c+kk, kk, ot) ) / rc * dz;
    if (ub)       return cen - ( rb * tabs(wb) * face_upw(f, c, kk, ob) ) / rc * dz;
    return cen;
}

// advec_2i4 (src/advec_2i4.cxx:101-640): 2nd-order advecting velocities, 4th-order (-1,9,9,-1)/16 interpolation of the
// advected field, 2nd order on the vertical faces next to a wall; ONE increment per cell (x, y and z terms in one sum).
// ot / ob = order of the top / bottom vertical face: the 2i5 table capped at 4 (0 = wall, 2, 4).
template<class TF>
MHH_HD TF advec24(const TF* __restrict__ f, int c, int jj, int kk, TF ue, TF uw, TF vn, TF vs, TF wt, TF wb,
                  int ot, int ob, TF dxi, TF dyi, TF rt, TF rb, TF rc, TF dz)
{
    const TF It = (ot == 0) ? TF(0) : (ot == 2) ? i2(f[c], f[c+kk]) : i4c(f[c-kk], f[c], f[c+kk], f[c+2*kk]);
    const TF Ib = (ob == 0) ? TF(0) : (ob == 2) ? i2(f[c-kk], f[c]) : i4c(f[c-2*kk], f[c-kk], f[c], f[c+kk]);
    const TF X = ue * i4c(f[c-1 ], f[c], f[c+1 ], f[c+2   ]) - uw * i4c(f[c-2   ], f[c-1 ], f[c], f[c+1 ]);
    const TF Y = vn * i4c(f[c-jj], f[c], f[c+jj], f[c+2*jj]) - vs * i4c(f[c-2*jj], f[c-jj], f[c], f[c+jj]);
    if (ob == 0) return - ( X ) * dxi - ( Y ) * dyi - ( rt * wt * It ) / rc * dz;
    if (ot == 0) return - ( X ) * dxi - ( Y ) * dyi - ( -rb * wb * Ib ) / rc * dz;
    return - ( X ) * dxi - ( Y ) * dyi - ( rt * wt * It - rb * wb * Ib ) / rc * dz;
}

// advec_2i62 (src/advec_2i62.cxx:105-310): 6th-order interpolation of the advected field horizontally, two-point
// interpolation vertically on every level; one increment per cell.
template<class TF>
MHH_HD TF advec262(const TF* __restrict__ f, int c, int jj, int kk, TF ue, TF uw, TF vn, TF vs, TF wt, TF wb,
                   TF dxi, TF dyi, TF rt, TF rb, TF rc, TF dz)
{
    return - ( ue * i6(f[c-2   ], f[c-1 ], f[c], f[c+1 ], f[c+2   ], f[c+3   ]) - uw * i6(f[c-3   ], f[c-2   ], f[c-1 ], f[c], f[c+1 ], f[c+2   ]) ) * dxi
           - ( vn * i6(f[c-2*jj], f[c-jj], f[c], f[c+jj], f[c+2*jj], f[c+3*jj]) - vs * i6(f[c-3*jj], f[c-2*jj], f[c-jj], f[c], f[c+jj], f[c+2*jj]) ) * dyi
           - ( rt * wt * i2(f[c], f[c+kk]) - rb * wb * i2(f[c-kk], f[c]) ) / rc * dz;
}

// Thermo_dry buoyancy tendency of w (src/thermo_dry.cxx:165-197): grav/threfh[k] * (th at the w level - threfh[k])
template<class TF>
MHH_HD TF buoyancy_tend(const TF* __restrict__ th, int c, int kk, int order, TF grav, TF threfh_k)
{
    const TF thh = (order == 4) ? i4c(th[c-2*kk], th[c-kk], th[c], th[c+kk]) : i2(th[c-kk], th[c]);
    return grav/threfh_k * (thh - threfh_k);
}

// =======================================================================================================
// Koren (1993) flux-limited scalar advection (include/advec_monotonic.h:10-180, selected per scalar by
// advec.fluxlimit_list, src/advec_2i5.cxx:921,1030). face = 0: interior face, 1: first face above the bottom wall
// (upwind value for upward flow), 2: last face below the top wall (upwind value for downward flow).
// =======================================================================================================
MHH_HD double tcopysign1(double d) { return __builtin_copysign(1., d); }
MHH_HD float  tcopysign1(float d)  { return __builtin_copysignf(1.f, d); }
template<class TF> MHH_HD TF teps();
template<> MHH_HD double teps<double>() { return 2.220446049250313e-16; }
template<> MHH_HD float  teps<float>()  { return 1.1920928955078125e-07f; }
template<class TF>
MHH_HD TF koren_upwind(TF vel, TF far, TF near_, TF across)          // vel * (near + phi/2 * (near - far)), r from across-near
{
    const TF d = near_ - far;
    const TF denom = tcopysign1(d) * tmax(tabs(d), teps<TF>());
    const TF two_r = TF(2.) * (across - near_) / denom;
    const TF phi = tmax(TF(0.), tmin(two_r, tmin(TF(1./3.)*(TF(1.)+two_r), TF(2.))));
    return vel*(near_ + TF(0.5)*phi*(near_ - far));
}
template<class TF>
MHH_HD TF koren_flux(int face, TF vel, TF sm2, TF sm1, TF sp1, TF sp2)
{
    if (vel >= TF(0.)) return (face == 1) ? vel*sm1 : koren_upwind(vel, sm2, sm1, sp1);
    return (face == 2) ? vel*sp1 : koren_upwind(vel, sp2, sp1, sm1);
}
// increment of st at cell c; lev: 0 interior, 1 kstart, 2 kstart+1, 3 kend-2, 4 kend-1
template<class TF>
MHH_HD TF advec_s_lim_cell(const TF* __restrict__ s, const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w,
                           int c, int jj, int kk, int lev, TF dxi, TF dyi, TF rt, TF rb, TF rc, TF dz)
{
    const TF xm2 = s[c-2], xm1 = s[c-1], s0 = s[c], xp1 = s[c+1], xp2 = s[c+2];
    const TF ym2 = s[c-2*jj], ym1 = s[c-jj], yp1 = s[c+jj], yp2 = s[c+2*jj];
    const TF hor = - ( koren_flux(0, u[c+1],  xm1, s0, xp1, xp2) - koren_flux(0, u[c], xm2, xm1, s0, xp1) ) * dxi
                   - ( koren_flux(0, v[c+jj], ym1, s0, yp1, yp2) - koren_flux(0, v[c], ym2, ym1, s0, yp1) ) * dyi;
    const TF zm1 = s[c-kk], zp1 = s[c+kk];
    if (lev == 1) return hor - ( rt * koren_flux(1, w[c+kk], zm1, s0, zp1, s[c+2*kk]) ) / rc * dz;
    if (lev == 4) return hor - ( - rb * koren_flux(2, w[c], s[c-2*kk], zm1, s0, zp1) ) / rc * dz;
    const TF zm2 = s[c-2*kk], zp2 = s[c+2*kk];
    return hor - ( rt * koren_flux(lev == 3 ? 2 : 0, w[c+kk], zm1, s0, zp1, zp2)
                 - rb * koren_flux(lev == 2 ? 1 : 0, w[c],    zm2, zm1, s0, zp1) ) / rc * dz;
}

// =======================================================================================================
// advec_4 (src/advec_4.cxx:88-486): three separate decrements (x, y if dim3, z); returns them through d[3].
// sd = staggering stride of the advected momentum component (1, jj, kk); is_w selects the w equation, whose
// vertical advecting velocity is the advected field itself incl. the biased wall forms.
// =======================================================================================================
// The arithmetic is written once against a "view" of each field -- view.at<DI,DJ,DK>() = the value DI, DJ, DK cells away
// from the cell being updated -- so that the one-thread-per-cell kernels (GlobalView: flat array) and the k-marching
// kernel (k_march4.hip: LDS planes for horizontal offsets, a register window for the own column) execute the very same
// expressions.
template<class TF> struct GlobalView
{
    const TF* __restrict__ p; int c, jj, kk;
    template<int DI, int DJ, int DK> MHH_HD TF at() const { return p[c + DI + DJ*jj + DK*kk]; }
};
// 4th-order interpolation of `a` to the location of momentum component COMP (along its staggering direction), evaluated
// at the cell (SI, SJ, SK) away: a[-2e], a[-e], a[0], a[+e] with e the unit vector of COMP
template<int COMP, int SI, int SJ, int SK, class TF, class V>
MHH_HD TF stag_ci4(const V& a)
{
    constexpr int EI = (COMP == 0), EJ = (COMP == 1), EK = (COMP == 2);
    return ci4<TF>(a.template at<SI-2*EI, SJ-2*EJ, SK-2*EK>(), a.template at<SI-EI, SJ-EJ, SK-EK>(),
                   a.template at<SI, SJ, SK>(), a.template at<SI+EI, SJ+EJ, SK+EK>());
}
template<int COMP, int M, class TF, class FV, class UV>
MHH_HD TF advec4_px(const FV& f, const UV& u)
{ return stag_ci4<COMP, M-1, 0, 0, TF>(u) * ci4<TF>(f.template at<M-3,0,0>(), f.template at<M-2,0,0>(), f.template at<M-1,0,0>(), f.template at<M,0,0>()); }
template<int COMP, int M, class TF, class FV, class VV>
MHH_HD TF advec4_py(const FV& f, const VV& v)
{ return stag_ci4<COMP, 0, M-1, 0, TF>(v) * ci4<TF>(f.template at<0,M-3,0>(), f.template at<0,M-2,0>(), f.template at<0,M-1,0>(), f.template at<0,M,0>()); }
template<int COMP, int M, class TF, class FV, class WV>
MHH_HD TF advec4_pz(const FV& f, const WV& w, bool bot, bool top)
{
    TF fi;
    if (M == 0 && bot)      fi = bi4<TF>(f.template at<0,0,-2>(), f.template at<0,0,-1>(), f.template at<0,0,0>(), f.template at<0,0,1>());
    else if (M == 3 && top) fi = ti4<TF>(f.template at<0,0,-1>(), f.template at<0,0,0>(), f.template at<0,0,1>(), f.template at<0,0,2>());
    else                    fi = ci4<TF>(f.template at<0,0,M-3>(), f.template at<0,0,M-2>(), f.template at<0,0,M-1>(), f.template at<0,0,M>());
    TF ve;
    if constexpr (COMP == 2) ve = fi;                        // the w equation advects itself vertically
    else                     ve = stag_ci4<COMP, 0, 0, M-1, TF>(w);
    return ve * fi;
}
// COMP = 0, 1, 2: the u, v, w equation; f is the advected component itself (u, v or w)
template<int COMP, class TF, class FV, class UV, class VV, class WV>
MHH_HD void advec4_mom_v(TF d[3], const FV& f, const UV& u, const VV& v, const WV& w, bool bot, bool top, TF dxi, TF dyi, TF dz, bool dim3)
{
    const TF cg0 = TF(1./24.), cg1 = TF(-27./24.), cg2 = TF(27./24.), cg3 = TF(-1./24.);
    d[0] = ( cg0*advec4_px<COMP,0,TF>(f, u) + cg1*advec4_px<COMP,1,TF>(f, u) + cg2*advec4_px<COMP,2,TF>(f, u) + cg3*advec4_px<COMP,3,TF>(f, u) ) * dxi;
    d[1] = TF(0);
    if (dim3)
        d[1] = ( cg0*advec4_py<COMP,0,TF>(f, v) + cg1*advec4_py<COMP,1,TF>(f, v) + cg2*advec4_py<COMP,2,TF>(f, v) + cg3*advec4_py<COMP,3,TF>(f, v) ) * dyi;
    d[2] = ( cg0*advec4_pz<COMP,0,TF>(f, w, bot, top) + cg1*advec4_pz<COMP,1,TF>(f, w, bot, top)
           + cg2*advec4_pz<COMP,2,TF>(f, w, bot, top) + cg3*advec4_pz<COMP,3,TF>(f, w, bot, top) ) * dz;
}
// The same with the VERTICAL face products carried from level to level by a k-marching kernel (k_march4.hip): product M of level
// k+1 is product M+1 of level k -- same operands, same order -- so a level forms only its topmost product (M = 3) and takes the
// other three from the level below (c[0..2]); `fresh` (the first level a thread works, and the levels whose lowest product is the
// biased wall form) forms all four. Fifteen operations and four LDS reads per reused product of the u and v equations, eight
// for w. The sum and its order are advec4_mom_v's: the same bits.
template<int COMP, class TF, class FV, class UV, class VV, class WV>
MHH_HD void advec4_mom_vc(TF d[3], const FV& f, const UV& u, const VV& v, const WV& w, bool bot, bool top, TF dxi, TF dyi, TF dz, bool dim3, TF (&c)[3], bool fresh)
{
    const TF cg0 = TF(1./24.), cg1 = TF(-27./24.), cg2 = TF(27./24.), cg3 = TF(-1./24.);
    d[0] = ( cg0*advec4_px<COMP,0,TF>(f, u) + cg1*advec4_px<COMP,1,TF>(f, u) + cg2*advec4_px<COMP,2,TF>(f, u) + cg3*advec4_px<COMP,3,TF>(f, u) ) * dxi;
    d[1] = TF(0);
    if (dim3)
        d[1] = ( cg0*advec4_py<COMP,0,TF>(f, v) + cg1*advec4_py<COMP,1,TF>(f, v) + cg2*advec4_py<COMP,2,TF>(f, v) + cg3*advec4_py<COMP,3,TF>(f, v) ) * dyi;
    TF p0, p1, p2;
    if (fresh) { p0 = advec4_pz<COMP,0,TF>(f, w, bot, top); p1 = advec4_pz<COMP,1,TF>(f, w, bot, top); p2 = advec4_pz<COMP,2,TF>(f, w, bot, top); }
    else       { p0 = c[0]; p1 = c[1]; p2 = c[2]; }
    const TF p3 = advec4_pz<COMP,3,TF>(f, w, bot, top);
    d[2] = ( cg0*p0 + cg1*p1 + cg2*p2 + cg3*p3 ) * dz;
    c[0] = p1; c[1] = p2; c[2] = p3;
}
// advec_4m (src/advec_4m.cxx:90-478): every term is grad4 over four face products, each the 4th-order interpolated
// advecting velocity times a TWO-point mean of the advected quantity over a widening stencil ((-3,0), (-1,0), (0,1), (0,3)
// cells along the direction); at the walls the outermost vertical product is the mirrored one. One increment per cell.
// COMP 0..2 = u, v, w equation, 3 = scalar (velocities taken at their faces, no interpolation).
template<class TF> MHH_HD TF grad4m(TF a, TF b, TF c, TF d) { return - TF(1./24.)*(d-a) - TF(-27./24.)*(c-b); }
template<int COMP, int DI, int DJ, int DK, class TF, class V>
MHH_HD TF vel4m(const V& a)
{
    if constexpr (COMP == 3) return a.template at<DI, DJ, DK>();
    else
    {   // interp4c, the paired form -1/16 (a+d) + 9/16 (b+c) (advec_4 proper uses the four-term sum)
        constexpr int EI = (COMP == 0), EJ = (COMP == 1), EK = (COMP == 2);
        return i4c<TF>(a.template at<DI-2*EI, DJ-2*EJ, DK-2*EK>(), a.template at<DI-EI, DJ-EJ, DK-EK>(),
                       a.template at<DI, DJ, DK>(), a.template at<DI+EI, DJ+EJ, DK+EK>());
    }
}
template<int COMP, class TF, class FV, class UV, class VV, class WV>
MHH_HD TF advec4m_v(const FV& f, const UV& u, const VV& v, const WV& w, bool bot, bool top, TF dxi, TF dyi, TF dz)
{
    const TF gx = grad4m<TF>(vel4m<COMP,-1,0,0,TF>(u) * i2(f.template at<-3,0,0>(), f.template at<0,0,0>()),
                             vel4m<COMP, 0,0,0,TF>(u) * i2(f.template at<-1,0,0>(), f.template at<0,0,0>()),
                             vel4m<COMP, 1,0,0,TF>(u) * i2(f.template at< 0,0,0>(), f.template at<1,0,0>()),
                             vel4m<COMP, 2,0,0,TF>(u) * i2(f.template at< 0,0,0>(), f.template at<3,0,0>()));
    const TF gy = grad4m<TF>(vel4m<COMP,0,-1,0,TF>(v) * i2(f.template at<0,-3,0>(), f.template at<0,0,0>()),
                             vel4m<COMP,0, 0,0,TF>(v) * i2(f.template at<0,-1,0>(), f.template at<0,0,0>()),
                             vel4m<COMP,0, 1,0,TF>(v) * i2(f.template at<0, 0,0>(), f.template at<0,1,0>()),
                             vel4m<COMP,0, 2,0,TF>(v) * i2(f.template at<0, 0,0>(), f.template at<0,3,0>()));
    const TF p0 = bot ? -vel4m<COMP,0,0, 1,TF>(w) * i2(f.template at<0,0,-1>(), f.template at<0,0,2>())
                      :  vel4m<COMP,0,0,-1,TF>(w) * i2(f.template at<0,0,-3>(), f.template at<0,0,0>());
    const TF p3 = top ? -vel4m<COMP,0,0, 0,TF>(w) * i2(f.template at<0,0,-2>(), f.template at<0,0,1>())
                      :  vel4m<COMP,0,0, 2,TF>(w) * i2(f.template at<0,0, 0>(), f.template at<0,0,3>());
    const TF gz = grad4m<TF>(p0,
                             vel4m<COMP,0,0,0,TF>(w) * i2(f.template at<0,0,-1>(), f.template at<0,0,0>()),
                             vel4m<COMP,0,0,1,TF>(w) * i2(f.template at<0,0, 0>(), f.template at<0,0,1>()),
                             p3);
    return - gx * dxi - gy * dyi - gz * dz;
}
template<class TF>
MHH_HD TF advec4m(int comp, const TF* __restrict__ f, const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w,
                  int c, int jj, int kk, bool bot, bool top, TF dxi, TF dyi, TF dz)
{
    const GlobalView<TF> fv{f, c, jj, kk}, uv{u, c, jj, kk}, vv{v, c, jj, kk}, wv{w, c, jj, kk};
    if (comp == 0) return advec4m_v<0, TF>(fv, uv, vv, wv, bot, top, dxi, dyi, dz);
    if (comp == 1) return advec4m_v<1, TF>(fv, uv, vv, wv, bot, top, dxi, dyi, dz);
    if (comp == 2) return advec4m_v<2, TF>(fv, uv, vv, wv, false, false, dxi, dyi, dz);
    return advec4m_v<3, TF>(fv, uv, vv, wv, bot, top, dxi, dyi, dz);
}
template<class TF>
MHH_HD void advec4_mom(TF d[3], const TF* __restrict__ f, const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w,
                       int c, int sd, bool is_w, int jj, int kk, bool bot, bool top, TF dxi, TF dyi, TF dz, bool dim3)
{
    const GlobalView<TF> fv{f, c, jj, kk}, uv{u, c, jj, kk}, vv{v, c, jj, kk}, wv{w, c, jj, kk};
    if (is_w)          advec4_mom_v<2>(d, fv, uv, vv, wv, bot, top, dxi, dyi, dz, dim3);
    else if (sd == 1)  advec4_mom_v<0>(d, fv, uv, vv, wv, bot, top, dxi, dyi, dz, dim3);
    else               advec4_mom_v<1>(d, fv, uv, vv, wv, bot, top, dxi, dyi, dz, dim3);
}
template<class TF>
MHH_HD void advec4_s(TF d[3], const TF* __restrict__ s, const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w,
                     int c, int jj, int kk, bool bot, bool top, TF dxi, TF dyi, TF dz, bool dim3)
{
    const TF cg0 = TF(1./24.), cg1 = TF(-27./24.), cg2 = TF(27./24.), cg3 = TF(-1./24.);
    d[0] = ( cg0*(u[c-1] * ci4(s[c-3], s[c-2], s[c-1], s[c  ]))
           + cg1*(u[c  ] * ci4(s[c-2], s[c-1], s[c  ], s[c+1]))
           + cg2*(u[c+1] * ci4(s[c-1], s[c  ], s[c+1], s[c+2]))
           + cg3*(u[c+2] * ci4(s[c  ], s[c+1], s[c+2], s[c+3])) ) * dxi;
    d[1] = TF(0);
    if (dim3)
        d[1] = ( cg0*(v[c-jj  ] * ci4(s[c-3*jj], s[c-2*jj], s[c-jj], s[c]))
               + cg1*(v[c     ] * ci4(s[c-2*jj], s[c-jj], s[c], s[c+jj]))
               + cg2*(v[c+jj  ] * ci4(s[c-jj], s[c], s[c+jj], s[c+2*jj]))
               + cg3*(v[c+2*jj] * ci4(s[c], s[c+jj], s[c+2*jj], s[c+3*jj])) ) * dyi;
    const TF f0 = bot ? bi4(s[c-2*kk], s[c-kk], s[c], s[c+kk]) : ci4(s[c-3*kk], s[c-2*kk], s[c-kk], s[c]);
    const TF f3 = top ? ti4(s[c-kk], s[c], s[c+kk], s[c+2*kk]) : ci4(s[c], s[c+kk], s[c+2*kk], s[c+3*kk]);
    d[2] = ( cg0*(w[c-kk  ] * f0)
           + cg1*(w[c     ] * ci4(s[c-2*kk], s[c-kk], s[c], s[c+kk]))
           + cg2*(w[c+kk  ] * ci4(s[c-kk], s[c], s[c+kk], s[c+2*kk]))
           + cg3*(w[c+2*kk] * f3) ) * dz;
}

// calc_cfl integrand (advec_2.cxx:51-78, advec_2i5.cxx:60-148, advec_4.cxx:51-86)
template<class TF>
MHH_HD TF cfl_cell(int scheme, const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w,
                   int c, int jj, int kk, int k, int kstart, int kend, TF dxi, TF dyi, TF dzi_k)
{
    if (scheme == 2)
        return tabs(i2(u[c], u[c+1]))*dxi + tabs(i2(v[c], v[c+jj]))*dyi + tabs(i2(w[c], w[c+kk]))*dzi_k;
    if (scheme == 41)       // src/advec_4m.cxx:51-88: the weighted sum spelled out, metrics dxi = 1./dx and dzi
        return tabs(ci4(u[c-1], u[c], u[c+1], u[c+2]))*dxi + tabs(ci4(v[c-jj], v[c], v[c+jj], v[c+2*jj]))*dyi
             + tabs(ci4(w[c-kk], w[c], w[c+kk], w[c+2*kk]))*dzi_k;
    if (scheme == 262)      // src/advec_2i62.cxx:58-105
        return tabs(i6(u[c-2], u[c-1], u[c], u[c+1], u[c+2], u[c+3]))*dxi + tabs(i6(v[c-2*jj], v[c-jj], v[c], v[c+jj], v[c+2*jj], v[c+3*jj]))*dyi
             + tabs(i2(w[c], w[c+kk]))*dzi_k;
    if (scheme == 24)       // src/advec_2i4.cxx:51-99
        return tabs(i4c(u[c-1], u[c], u[c+1], u[c+2]))*dxi + tabs(i4c(v[c-jj], v[c], v[c+jj], v[c+2*jj]))*dyi
             + tabs((k == kstart || k == kend-1) ? i2(w[c], w[c+kk]) : i4c(w[c-kk], w[c], w[c+kk], w[c+2*kk]))*dzi_k;
    if (scheme == 4)
        return tabs(i4c(u[c-1], u[c], u[c+1], u[c+2]))*dxi + tabs(i4c(v[c-jj], v[c], v[c+jj], v[c+2*jj]))*dyi
             + tabs(i4c(w[c-kk], w[c], w[c+kk], w[c+2*kk]))*dzi_k;
    TF wi;
    if (k == kstart || k == kend-1)        wi = i2(w[c], w[c+kk]);
    else if (k == kstart+1 || k == kend-2 || scheme == 253) wi = i4ws(w[c-kk], w[c], w[c+kk], w[c+2*kk]);   // 2i53: 4th order on every inner level
    else                                   wi = i6(w[c-2*kk], w[c-kk], w[c], w[c+kk], w[c+2*kk], w[c+3*kk]);
    return tabs(i6(u[c-2], u[c-1], u[c], u[c+1], u[c+2], u[c+3]))*dxi
         + tabs(i6(v[c-2*jj], v[c-jj], v[c], v[c+jj], v[c+2*jj], v[c+3*jj]))*dyi
         + tabs(wi)*dzi_k;
}

// =======================================================================================================
// diff_2 (src/diff_2.cxx:39-86): at += visc * ( ... ) with DOUBLE dxidxi/dyidyi. Returns the new tendency value
// (the += is evaluated in the promoted type, so the caller must not re-round).
//   centred fields: gt=dzhi[k+1] gb=dzhi[k] gc=dzi[k];  w: gt=dzi[k] gb=dzi[k-1] gc=dzhi[k]
// =======================================================================================================
template<class TF>
MHH_HD TF diff2_apply(TF t, const TF* __restrict__ a, int c, int jj, int kk, TF visc, double dxidxi, double dyidyi, TF gt, TF gb, TF gc)
{
    t += visc * (
            + ( (a[c+1 ] - a[c]) - (a[c] - a[c-1 ]) ) * dxidxi
            + ( (a[c+jj] - a[c]) - (a[c] - a[c-jj]) ) * dyidyi
            + ( (a[c+kk] - a[c]) * gt - (a[c] - a[c-kk]) * gb ) * gc );
    return t;
}

// =======================================================================================================
// diff_4 (src/diff_4.cxx:41-173): three increments. g4[0..3] = inner metric at the four faces, go = outer metric.
// =======================================================================================================
template<class TF, class AV>
MHH_HD void diff4_v(TF d[3], const AV& a, bool bot, bool top, TF visc, TF dxidxi, TF dyidyi, const TF g4[4], TF go, bool dim3)
{
    const TF cdg0 = TF(-1460./576.), cdg1 = TF(783./576.), cdg2 = TF(-54./576.), cdg3 = TF(1./576.);
    const TF cg0 = TF(1./24.), cg1 = TF(-27./24.), cg2 = TF(27./24.), cg3 = TF(-1./24.);
    d[0] = visc * (cdg3*a.template at<-3,0,0>() + cdg2*a.template at<-2,0,0>() + cdg1*a.template at<-1,0,0>() + cdg0*a.template at<0,0,0>()
                 + cdg1*a.template at<1,0,0>() + cdg2*a.template at<2,0,0>() + cdg3*a.template at<3,0,0>())*dxidxi;
    d[1] = TF(0);
    if (dim3)
        d[1] = visc * (cdg3*a.template at<0,-3,0>() + cdg2*a.template at<0,-2,0>() + cdg1*a.template at<0,-1,0>() + cdg0*a.template at<0,0,0>()
                     + cdg1*a.template at<0,1,0>() + cdg2*a.template at<0,2,0>() + cdg3*a.template at<0,3,0>())*dyidyi;
    const TF g0 = bot ? bg4<TF>(a.template at<0,0,-2>(), a.template at<0,0,-1>(), a.template at<0,0,0>(), a.template at<0,0,1>())
                      : cg4<TF>(a.template at<0,0,-3>(), a.template at<0,0,-2>(), a.template at<0,0,-1>(), a.template at<0,0,0>());
    const TF g3 = top ? tg4<TF>(a.template at<0,0,-1>(), a.template at<0,0,0>(), a.template at<0,0,1>(), a.template at<0,0,2>())
                      : cg4<TF>(a.template at<0,0,0>(), a.template at<0,0,1>(), a.template at<0,0,2>(), a.template at<0,0,3>());
    d[2] = visc * ( cg0*g0 * g4[0]
                  + cg1*cg4<TF>(a.template at<0,0,-2>(), a.template at<0,0,-1>(), a.template at<0,0,0>(), a.template at<0,0,1>()) * g4[1]
                  + cg2*cg4<TF>(a.template at<0,0,-1>(), a.template at<0,0,0>(), a.template at<0,0,1>(), a.template at<0,0,2>()) * g4[2]
                  + cg3*g3 * g4[3] ) * go;
}
// diff4_v with the inner vertical gradients carried from level to level (see advec4_mom_vc): gradient M of level k+1 is gradient
// M+1 of level k; seven operations per reused gradient
template<class TF, class AV>
MHH_HD void diff4_vc(TF d[3], const AV& a, bool bot, bool top, TF visc, TF dxidxi, TF dyidyi, const TF g4[4], TF go, bool dim3, TF (&c)[3], bool fresh)
{
    const TF cdg0 = TF(-1460./576.), cdg1 = TF(783./576.), cdg2 = TF(-54./576.), cdg3 = TF(1./576.);
    const TF cg0 = TF(1./24.), cg1 = TF(-27./24.), cg2 = TF(27./24.), cg3 = TF(-1./24.);
    d[0] = visc * (cdg3*a.template at<-3,0,0>() + cdg2*a.template at<-2,0,0>() + cdg1*a.template at<-1,0,0>() + cdg0*a.template at<0,0,0>()
                 + cdg1*a.template at<1,0,0>() + cdg2*a.template at<2,0,0>() + cdg3*a.template at<3,0,0>())*dxidxi;
    d[1] = TF(0);
    if (dim3)
        d[1] = visc * (cdg3*a.template at<0,-3,0>() + cdg2*a.template at<0,-2,0>() + cdg1*a.template at<0,-1,0>() + cdg0*a.template at<0,0,0>()
                     + cdg1*a.template at<0,1,0>() + cdg2*a.template at<0,2,0>() + cdg3*a.template at<0,3,0>())*dyidyi;
    TF g0, g1, g2;
    if (fresh)
    {
        g0 = bot ? bg4<TF>(a.template at<0,0,-2>(), a.template at<0,0,-1>(), a.template at<0,0,0>(), a.template at<0,0,1>())
                 : cg4<TF>(a.template at<0,0,-3>(), a.template at<0,0,-2>(), a.template at<0,0,-1>(), a.template at<0,0,0>());
        g1 = cg4<TF>(a.template at<0,0,-2>(), a.template at<0,0,-1>(), a.template at<0,0,0>(), a.template at<0,0,1>());
        g2 = cg4<TF>(a.template at<0,0,-1>(), a.template at<0,0,0>(), a.template at<0,0,1>(), a.template at<0,0,2>());
    }
    else { g0 = c[0]; g1 = c[1]; g2 = c[2]; }
    const TF g3 = top ? tg4<TF>(a.template at<0,0,-1>(), a.template at<0,0,0>(), a.template at<0,0,1>(), a.template at<0,0,2>())
                      : cg4<TF>(a.template at<0,0,0>(), a.template at<0,0,1>(), a.template at<0,0,2>(), a.template at<0,0,3>());
    d[2] = visc * ( cg0*g0 * g4[0] + cg1*g1 * g4[1] + cg2*g2 * g4[2] + cg3*g3 * g4[3] ) * go;
    c[0] = g1; c[1] = g2; c[2] = g3;
}
template<class TF>
MHH_HD void diff4_cell(TF d[3], const TF* __restrict__ a, int c, int jj, int kk, bool bot, bool top, TF visc,
                       TF dxidxi, TF dyidyi, const TF g4[4], TF go, bool dim3)
{
    const GlobalView<TF> av{a, c, jj, kk};
    diff4_v(d, av, bot, top, visc, dxidxi, dyidyi, g4, go, dim3);
}

// =======================================================================================================
// diff_smag2 (src/diff_smag2.cxx)
// =======================================================================================================
// calc_strain2 (:47-155); mo = surface model active on this (lowest) level
template<class TF>
MHH_HD TF smag_strain2(const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w, int c, int jj, int kk,
                       bool mo, TF ugradbot, TF vgradbot, TF dxi, TF dyi, TF dzi_k, TF dzhi_k, TF dzhi_kp)
{
    TF acc = sq((u[c+1]-u[c])*dxi);
    acc = acc + sq((v[c+jj]-v[c])*dyi);
    acc = acc + sq((w[c+kk]-w[c])*dzi_k);
    acc = acc + TF(0.125)*sq((u[c      ]-u[c  -jj])*dyi + (v[c      ]-v[c-1   ])*dxi);
    acc = acc + TF(0.125)*sq((u[c+1    ]-u[c+1-jj])*dyi + (v[c+1    ]-v[c     ])*dxi);
    acc = acc + TF(0.125)*sq((u[c  +jj ]-u[c      ])*dyi + (v[c  +jj]-v[c-1+jj])*dxi);
    acc = acc + TF(0.125)*sq((u[c+1+jj ]-u[c+1    ])*dyi + (v[c+1+jj]-v[c  +jj])*dxi);
    if (mo)
    {
        acc = acc + TF(0.5)*sq(ugradbot);
        acc = acc + TF(0.125)*sq((w[c      ]-w[c-1   ])*dxi);
        acc = acc + TF(0.125)*sq((w[c+1    ]-w[c     ])*dxi);
        acc = acc + TF(0.125)*sq((w[c  +kk ]-w[c-1+kk])*dxi);
        acc = acc + TF(0.125)*sq((w[c+1+kk ]-w[c  +kk])*dxi);
        acc = acc + TF(0.5)*sq(vgradbot);
        acc = acc + TF(0.125)*sq((w[c      ]-w[c-jj   ])*dyi);
        acc = acc + TF(0.125)*sq((w[c+jj   ]-w[c      ])*dyi);
        acc = acc + TF(0.125)*sq((w[c   +kk]-w[c-jj+kk])*dyi);
        acc = acc + TF(0.125)*sq((w[c+jj+kk]-w[c   +kk])*dyi);
    }
    else
    {
        acc = acc + TF(0.125)*sq((u[c      ]-u[c  -kk])*dzhi_k  + (w[c      ]-w[c-1   ])*dxi);
        acc = acc + TF(0.125)*sq((u[c+1    ]-u[c+1-kk])*dzhi_k  + (w[c+1    ]-w[c     ])*dxi);
        acc = acc + TF(0.125)*sq((u[c  +kk ]-u[c     ])*dzhi_kp + (w[c  +kk ]-w[c-1+kk])*dxi);
        acc = acc + TF(0.125)*sq((u[c+1+kk ]-u[c+1   ])*dzhi_kp + (w[c+1+kk ]-w[c  +kk])*dxi);
        acc = acc + TF(0.125)*sq((v[c      ]-v[c   -kk])*dzhi_k  + (w[c      ]-w[c-jj   ])*dyi);
        acc = acc + TF(0.125)*sq((v[c+jj   ]-v[c+jj-kk])*dzhi_k  + (w[c+jj   ]-w[c      ])*dyi);
        acc = acc + TF(0.125)*sq((v[c   +kk]-v[c      ])*dzhi_kp + (w[c   +kk]-w[c-jj+kk])*dyi);
        acc = acc + TF(0.125)*sq((v[c+jj+kk]-v[c+jj   ])*dzhi_kp + (w[c+jj+kk]-w[c   +kk])*dyi);
    }
    TF s2 = TF(2.)*acc;
    s2 += 1.e-9;          // Constants::dsmall is a double: promoted add, narrowed on store
    return s2;
}

// evisc from strain^2 (calc_evisc / calc_evisc_neutral, src/diff_smag2.cxx:157-367, the surface-model branches and the
// plain Smagorinsky length for resolved walls): pow(x, 2) and pow(y, 1/2) are evaluated as x*x and sqrt(y).
MHH_HD double dsqrt2(double x) { return __builtin_sqrt(x); }
MHH_HD float  dsqrt2(float x)  { return __builtin_sqrtf(x); }
// the squared mixing length: depends on the level and the column's roughness length only (three divisions and a square root
// per cell in fp64) -- for a horizontally uniform z0m it is a per-level table (mhh_smag2_mlen2_host, mhh_diff_params::mlen2)
template<class TF>
MHH_HD TF evisc_mlen2(int sm, int neutral, TF mlen0_k, TF z_k, TF z0m_ij)
{
    if (!sm) return sq(mlen0_k);
    if (neutral) return sq(TF(1.)/(TF(1.)/mlen0_k + TF(1.)/(TF(0.4)*(z_k+z0m_ij))));
    return sq(dsqrt2(TF(1.)/(TF(1.)/sq(mlen0_k) + TF(1.)/sq(TF(0.4)*(z_k+z0m_ij)))));
}
template<class TF>
MHH_HD TF evisc_from_mlen2(TF fac, TF s2, TF n2, int neutral, TF tPr)
{
    if (neutral) return fac * dsqrt2(s2);
    TF rit = n2 / s2 / tPr;
    rit = tmin(rit, TF(1.-1.e-9));
    return fac * dsqrt2(s2) * dsqrt2(TF(1.)-rit);
}
template<class TF>
MHH_HD TF evisc_value(TF s2, TF n2, int sm, int neutral, TF mlen0_k, TF z_k, TF z0m_ij, TF tPr)
{
    return evisc_from_mlen2(evisc_mlen2(sm, neutral, mlen0_k, z_k, z0m_ij), s2, n2, neutral, tPr);
}

// diff_u / diff_v (:369-571). fb/ft: this level takes the surface flux at the bottom / top instead of the resolved gradient.
template<class TF>
MHH_HD TF smag_diff_u(const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w, const TF* __restrict__ ev,
                      int c, int jj, int kk, bool fb, bool ft, TF fluxbot, TF fluxtop, TF visc, TF dxi, TF dyi,
                      TF rhk, TF rhkp, TF rk, TF dzi_k, TF dzhi_k, TF dzhi_kp)
{
    const TF ee = ev[c] + visc;
    const TF ew = ev[c-1] + visc;
    const TF en = TF(0.25)*(ev[c-1   ] + ev[c   ] + ev[c-1+jj] + ev[c+jj]) + visc;
    const TF es = TF(0.25)*(ev[c-1-jj] + ev[c-jj] + ev[c-1   ] + ev[c   ]) + visc;
    const TF hor = + ( ee*(u[c+1]-u[c  ])*dxi - ew*(u[c  ]-u[c-1])*dxi ) * TF(2.)*dxi
                   + ( en*((u[c+jj]-u[c   ])*dyi + (v[c+jj]-v[c-1+jj])*dxi)
                     - es*((u[c   ]-u[c-jj])*dyi + (v[c   ]-v[c-1   ])*dxi) ) * dyi;
    TF ver;
    if (fb)
    {
        const TF et = TF(0.25)*(ev[c-1] + ev[c] + ev[c-1+kk] + ev[c+kk]) + visc;
        ver = ( rhkp * et*((u[c+kk]-u[c])*dzhi_kp + (w[c+kk]-w[c-1+kk])*dxi) + rhk * fluxbot ) / rk * dzi_k;
    }
    else if (ft)
    {
        const TF eb = TF(0.25)*(ev[c-1-kk] + ev[c-kk] + ev[c-1] + ev[c]) + visc;
        ver = ( - rhkp * fluxtop - rhk * eb*((u[c]-u[c-kk])*dzhi_k + (w[c]-w[c-1])*dxi) ) / rk * dzi_k;
    }
    else
    {
        const TF et = TF(0.25)*(ev[c-1   ] + ev[c   ] + ev[c-1+kk] + ev[c+kk]) + visc;
        const TF eb = TF(0.25)*(ev[c-1-kk] + ev[c-kk] + ev[c-1   ] + ev[c   ]) + visc;
        ver = ( rhkp * et*((u[c+kk]-u[c   ])*dzhi_kp + (w[c+kk]-w[c-1+kk])*dxi)
              - rhk  * eb*((u[c   ]-u[c-kk])*dzhi_k  + (w[c   ]-w[c-1   ])*dxi) ) / rk * dzi_k;
    }
    return hor + ver;
}
template<class TF>
MHH_HD TF smag_diff_v(const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w, const TF* __restrict__ ev,
                      int c, int jj, int kk, bool fb, bool ft, TF fluxbot, TF fluxtop, TF visc, TF dxi, TF dyi,
                      TF rhk, TF rhkp, TF rk, TF dzi_k, TF dzhi_k, TF dzhi_kp)
{
    const TF ee = TF(0.25)*(ev[c  -jj] + ev[c  ] + ev[c+1-jj] + ev[c+1]) + visc;
    const TF ew = TF(0.25)*(ev[c-1-jj] + ev[c-1] + ev[c  -jj] + ev[c  ]) + visc;
    const TF en = ev[c] + visc;
    const TF es = ev[c-jj] + visc;
    const TF hor = + ( ee*((v[c+1]-v[c  ])*dxi + (u[c+1]-u[c+1-jj])*dyi)
                     - ew*((v[c  ]-v[c-1])*dxi + (u[c  ]-u[c  -jj])*dyi) ) * dxi
                   + ( en*(v[c+jj]-v[c   ])*dyi - es*(v[c   ]-v[c-jj])*dyi ) * TF(2.)*dyi;
    TF ver;
    if (fb)
    {
        const TF et = TF(0.25)*(ev[c-jj] + ev[c] + ev[c+kk-jj] + ev[c+kk]) + visc;
        ver = ( rhkp * et*((v[c+kk]-v[c])*dzhi_kp + (w[c+kk]-w[c-jj+kk])*dyi) + rhk * fluxbot ) / rk * dzi_k;
    }
    else if (ft)
    {
        const TF eb = TF(0.25)*(ev[c-kk-jj] + ev[c-kk] + ev[c-jj] + ev[c]) + visc;
        ver = ( - rhkp * fluxtop - rhk * eb*((v[c]-v[c-kk])*dzhi_k + (w[c]-w[c-jj])*dyi) ) / rk * dzi_k;
    }
    else
    {
        const TF et = TF(0.25)*(ev[c   -jj] + ev[c   ] + ev[c+kk-jj] + ev[c+kk]) + visc;
        const TF eb = TF(0.25)*(ev[c-kk-jj] + ev[c-kk] + ev[c   -jj] + ev[c   ]) + visc;
        ver = ( rhkp * et*((v[c+kk]-v[c   ])*dzhi_kp + (w[c+kk]-w[c-jj+kk])*dyi)
              - rhk  * eb*((v[c   ]-v[c-kk])*dzhi_k  + (w[c   ]-w[c-jj   ])*dyi) ) / rk * dzi_k;
    }
    return hor + ver;
}
// diff_w (:573-617): rk=rhoref[k] rkm=rhoref[k-1] rhk=rhorefh[k]
template<class TF>
MHH_HD TF smag_diff_w(const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w, const TF* __restrict__ ev,
                      int c, int jj, int kk, TF visc, TF dxi, TF dyi, TF rk, TF rkm, TF rhk, TF dzi_k, TF dzi_km, TF dzhi_k)
{
    const TF ee = TF(0.25)*(ev[c   -kk] + ev[c   ] + ev[c+1 -kk] + ev[c+1 ]) + visc;
    const TF ew = TF(0.25)*(ev[c-1 -kk] + ev[c-1 ] + ev[c   -kk] + ev[c   ]) + visc;
    const TF en = TF(0.25)*(ev[c   -kk] + ev[c   ] + ev[c+jj-kk] + ev[c+jj]) + visc;
    const TF es = TF(0.25)*(ev[c-jj-kk] + ev[c-jj] + ev[c   -kk] + ev[c   ]) + visc;
    const TF et = ev[c] + visc;
    const TF eb = ev[c-kk] + visc;
    return
        + ( ee*((w[c+1]-w[c  ])*dxi + (u[c+1]-u[c+1-kk])*dzhi_k)
          - ew*((w[c  ]-w[c-1])*dxi + (u[c  ]-u[c  -kk])*dzhi_k) ) * dxi
        + ( en*((w[c+jj]-w[c   ])*dyi + (v[c+jj]-v[c+jj-kk])*dzhi_k)
          - es*((w[c   ]-w[c-jj])*dyi + (v[c   ]-v[c   -kk])*dzhi_k) ) * dyi
        + ( rk  * et*(w[c+kk]-w[c   ])*dzi_k
          - rkm * eb*(w[c   ]-w[c-kk])*dzi_km ) / rhk * TF(2.)*dzhi_k;
}
// diff_c (:619-709)
template<class TF>
MHH_HD TF smag_diff_c(const TF* __restrict__ a, const TF* __restrict__ ev, int c, int jj, int kk, bool fb, bool ft,
                      TF fluxbot, TF fluxtop, TF tPr, TF visc, TF dxidxi, TF dyidyi,
                      TF rhk, TF rhkp, TF rk, TF dzi_k, TF dzhi_k, TF dzhi_kp)
{
    const TF ee = TF(0.5)*(ev[c   ]+ev[c+1 ])/tPr + visc;
    const TF ew = TF(0.5)*(ev[c-1 ]+ev[c   ])/tPr + visc;
    const TF en = TF(0.5)*(ev[c   ]+ev[c+jj])/tPr + visc;
    const TF es = TF(0.5)*(ev[c-jj]+ev[c   ])/tPr + visc;
    const TF hor = + ( ee*(a[c+1 ]-a[c]) - ew*(a[c]-a[c-1 ]) ) * dxidxi
                   + ( en*(a[c+jj]-a[c]) - es*(a[c]-a[c-jj]) ) * dyidyi;
    TF ver;
    if (fb)
    {
        const TF et = TF(0.5)*(ev[c]+ev[c+kk])/tPr + visc;
        ver = ( rhkp * et*(a[c+kk]-a[c])*dzhi_kp + rhk * fluxbot ) / rk * dzi_k;
    }
    else if (ft)
    {
        const TF eb = TF(0.5)*(ev[c-kk]+ev[c])/tPr + visc;
        ver = ( -rhkp * fluxtop - rhk * eb*(a[c]-a[c-kk])*dzhi_k ) / rk * dzi_k;
    }
    else
    {
        const TF et = TF(0.5)*(ev[c   ]+ev[c+kk])/tPr + visc;
        const TF eb = TF(0.5)*(ev[c-kk]+ev[c   ])/tPr + visc;
        ver = ( rhkp * et*(a[c+kk]-a[c   ])*dzhi_kp
              - rhk  * eb*(a[c   ]-a[c-kk])*dzhi_k ) / rk * dzi_k;
    }
    return hor + ver;
}

// =======================================================================================================
// pres_2 / pres_4 pointwise pieces
// =======================================================================================================
// Pres_2::input integrand (src/pres_2.cxx:185-195)
template<class TF>
MHH_HD TF pres2_in(const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w,
                   const TF* __restrict__ ut, const TF* __restrict__ vt, const TF* __restrict__ wt,
                   int c, int jj, int kk, TF dxi, TF dyi, TF dti, TF rk, TF rhk, TF rhkp, TF dzi_k)
{
    return rk * ( (ut[c+1 ] + u[c+1 ] * dti) - (ut[c] + u[c] * dti) ) * dxi
         + rk * ( (vt[c+jj] + v[c+jj] * dti) - (vt[c] + v[c] * dti) ) * dyi
         + ( rhkp * (wt[c+kk] + w[c+kk] * dti)
           - rhk  * (wt[c   ] + w[c   ] * dti) ) * dzi_k;
}
// Pres_4::input integrand (src/pres_4.cxx:305-316); wtm/wtp2 etc. read through the pointers (ghosts set by caller)
template<class TF>
MHH_HD TF pres4_in(const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w,
                   const TF* __restrict__ ut, const TF* __restrict__ vt, const TF* __restrict__ wt,
                   int c, int jj, int kk, TF dxi, TF dyi, TF dti, TF dzi4_k, bool dim3)
{
    TF p = cg4(ut[c-1] + u[c-1]*dti, ut[c] + u[c]*dti, ut[c+1] + u[c+1]*dti, ut[c+2] + u[c+2]*dti) * dxi;
    if (dim3)
        p += cg4(vt[c-jj] + v[c-jj]*dti, vt[c] + v[c]*dti, vt[c+jj] + v[c+jj]*dti, vt[c+2*jj] + v[c+2*jj]*dti) * dyi;
    p += cg4(wt[c-kk] + w[c-kk]*dti, wt[c] + w[c]*dti, wt[c+kk] + w[c+kk]*dti, wt[c+2*kk] + w[c+2*kk]*dti) * dzi4_k;
    return p;
}
// divergence integrands (src/pres_2.cxx:411-415, src/pres_4.cxx:755-759)
template<class TF>
MHH_HD TF div2_cell(const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w, int c, int jj, int kk,
                    TF dxi, TF dyi, TF rk, TF rhk, TF rhkp, TF dzi_k)
{
    return rk*((u[c+1]-u[c])*dxi + (v[c+jj]-v[c])*dyi) + (rhkp*w[c+kk]-rhk*w[c])*dzi_k;
}
template<class TF>
MHH_HD TF div4_cell(const TF* __restrict__ u, const TF* __restrict__ v, const TF* __restrict__ w, int c, int jj, int kk,
                    TF dxi, TF dyi, TF dzi4_k)
{
    return cg4(u[c-1], u[c], u[c+1], u[c+2]) * dxi + cg4(v[c-jj], v[c], v[c+jj], v[c+2*jj]) * dyi
         + cg4(w[c-kk], w[c], w[c+kk], w[c+2*kk]) * dzi4_k;
}

} // namespace mhh
