/*
 * mhh_oracle.cpp -- CPU restatement of MicroHH's RHS + pressure hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle: plain, single-threaded C++ loops that restate the arithmetic of the
 * reference CPU path (adconnolly/microhh, paths below are relative to the reference root) with the
 * same expression association, so that with -ffp-contract=off the results are bit-identical to the
 * reference kernels built with the same flags.  It is pinned by oracle/_ref (the reference's own
 * translation units compiled in place, see oracle/Makefile) for every stencil kernel, and by
 * numpy.fft / discrete identities / the Taylor-Green closed form for the pressure solver, whose
 * reference TU needs fftw3.h and cannot be built in this image (DESIGN.md "Oracle").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Nothing under microhh_amd/ links, imports or calls it.
 *
 * Formulation notes (why this does not look like the reference source): stencils are written with
 * relative-offset accessors, the three momentum equations of each advection scheme share one
 * routine parameterised by the staggering stride, and the 2i5 vertical boundary treatment is
 * expressed as a per-face order table instead of seven copies of the loop.
 */
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <limits>
#include "../include/mhh_hip.h"

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace
{
// ---------------------------------------------------------------------------------------------
// interpolation / gradient weights: include/finite_difference.h:36-155
// ---------------------------------------------------------------------------------------------
template<class TF> inline TF i2(TF a, TF b) { return TF(0.5)*(a+b); }
template<class TF> inline TF i6(TF a, TF b, TF c, TF d, TF e, TF f)
{ return TF(37./60.)*(c+d) - TF(8./60.)*(b+e) + TF(1./60.)*(a+f); }
template<class TF> inline TF i5(TF a, TF b, TF c, TF d, TF e, TF f)
{ return TF(10./60.)*(d-c) - TF(5./60.)*(e-b) + TF(1./60.)*(f-a); }
template<class TF> inline TF i4ws(TF a, TF b, TF c, TF d) { return TF(7./12.)*(b+c) - TF(1./12.)*(a+d); }
template<class TF> inline TF i3ws(TF a, TF b, TF c, TF d) { return TF(3./12.)*(c-b) - TF(1./12.)*(d-a); }

template<class TF> struct W4
{
    static constexpr TF ci0 = -1./16., ci1 = 9./16., ci2 = 9./16., ci3 = -1./16.;
    static constexpr TF bi0 =  5./16., bi1 = 15./16., bi2 = -5./16., bi3 = 1./16.;
    static constexpr TF ti0 =  1./16., ti1 = -5./16., ti2 = 15./16., ti3 = 5./16.;
    static constexpr TF cg0 =  1./24., cg1 = -27./24., cg2 = 27./24., cg3 = -1./24.;
    static constexpr TF bg0 = -23./24., bg1 = 21./24., bg2 = 3./24., bg3 = -1./24.;
    static constexpr TF tg0 =  1./24., tg1 = -3./24., tg2 = -21./24., tg3 = 23./24.;
    static constexpr TF cdg0 = -1460./576., cdg1 = 783./576., cdg2 = -54./576., cdg3 = 1./576.;
};
// 4-point weighted sums, left-associated like the reference's spelled-out expressions
template<class TF> inline TF ci4(TF a, TF b, TF c, TF d) { using W=W4<TF>; return W::ci0*a + W::ci1*b + W::ci2*c + W::ci3*d; }
template<class TF> inline TF bi4(TF a, TF b, TF c, TF d) { using W=W4<TF>; return W::bi0*a + W::bi1*b + W::bi2*c + W::bi3*d; }
template<class TF> inline TF ti4(TF a, TF b, TF c, TF d) { using W=W4<TF>; return W::ti0*a + W::ti1*b + W::ti2*c + W::ti3*d; }
// interp4c of the header (used by advec_4's calc_cfl only): pairs the symmetric points first
template<class TF> inline TF i4c(TF a, TF b, TF c, TF d) { using W=W4<TF>; return W::ci0*(a+d) + W::ci1*(b+c); }
template<class TF> inline TF cg4(TF a, TF b, TF c, TF d) { using W=W4<TF>; return W::cg0*a + W::cg1*b + W::cg2*c + W::cg3*d; }
template<class TF> inline TF bg4(TF a, TF b, TF c, TF d) { using W=W4<TF>; return W::bg0*a + W::bg1*b + W::bg2*c + W::bg3*d; }
template<class TF> inline TF tg4(TF a, TF b, TF c, TF d) { using W=W4<TF>; return W::tg0*a + W::tg1*b + W::tg2*c + W::tg3*d; }

template<class TF> inline const TF* P(const void* p) { return static_cast<const TF*>(p); }
template<class TF> inline TF* P(void* p) { return static_cast<TF*>(p); }

#define FOR_INTERIOR_PLANE(g) \
    for (int j=(g).jstart; j<(g).jend; ++j) \
        for (int i=(g).istart; i<(g).iend; ++i)

// ---------------------------------------------------------------------------------------------
// Boundary_cyclic::exec  (src/boundary_cyclic.cxx:370-443)
// ---------------------------------------------------------------------------------------------
template<class TF>
void cyclic(const mhh_grid& g, TF* a, int edge)
{
    const int jj = g.icells, kk = g.ijcells;
    if (edge == MHH_EDGE_EW || edge == MHH_EDGE_BOTH)
        for (int k=0; k<g.kcells; ++k)
            for (int j=0; j<g.jcells; ++j)
            {
                TF* row = a + j*jj + (size_t)k*kk;
                for (int i=0; i<g.igc; ++i) row[i] = row[g.iend-g.igc+i];
                for (int i=0; i<g.igc; ++i) row[g.iend+i] = row[g.istart+i];
            }
    if (edge == MHH_EDGE_NS || edge == MHH_EDGE_BOTH)
    {
        if (g.jtot > 1)
        {
            for (int k=0; k<g.kcells; ++k)
            {
                TF* pl = a + (size_t)k*kk;
                for (int j=0; j<g.jgc; ++j)
                    for (int i=0; i<g.icells; ++i) pl[i + j*jj] = pl[i + (g.jend-g.jgc+j)*jj];
                for (int j=0; j<g.jgc; ++j)
                    for (int i=0; i<g.icells; ++i) pl[i + (g.jend+j)*jj] = pl[i + (g.jstart+j)*jj];
            }
        }
        else   // 2-D run: replicate the single row (interior k only, as the reference)
        {
            for (int k=g.kstart; k<g.kend; ++k)
            {
                TF* pl = a + (size_t)k*kk;
                for (int j=0; j<g.jgc; ++j)
                    for (int i=0; i<g.icells; ++i)
                    {
                        const TF r = pl[i + g.jstart*jj];
                        pl[i + j*jj] = r;
                        pl[i + (g.jend+j)*jj] = r;
                    }
            }
        }
    }
}

// exec_2d (src/boundary_cyclic.cxx:445-500): one horizontal slice
template<class TF>
void cyclic_2d(const mhh_grid& g, TF* a)
{
    mhh_grid s = g; s.kcells = 1; s.kstart = 0; s.kend = 1;
    // the reference's 2-D variant handles jtot==1 identically on the single slice
    cyclic<TF>(s, a, MHH_EDGE_BOTH);
}

// ---------------------------------------------------------------------------------------------
// advec_2  (src/advec_2.cxx:81-202)
// ---------------------------------------------------------------------------------------------
// Momentum equations share one body: `o` is minus the staggering stride of the advected
// component (-1 for u, -jj for v, -kk for w); the advecting velocity on a face normal to
// direction d is the 2-point average of that velocity component over the staggering direction.
template<class TF>
void advec2_mom(const mhh_grid& g, int comp, TF* t, const TF* u, const TF* v, const TF* w,
                const TF* rhoref, const TF* rhorefh)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzi = P<TF>(g.dzi); const TF* dzhi = P<TF>(g.dzhi);
    const TF* f = (comp==0) ? u : (comp==1) ? v : w;
    const int o = (comp==0) ? -1 : (comp==1) ? -jj : -kk;
    const int k0 = (comp==2) ? g.kstart+1 : g.kstart;
    for (int k=k0; k<g.kend; ++k)
    {
        // density weights of the top / bottom face and of the cell itself
        const TF rt = (comp==2) ? rhoref[k]   : rhorefh[k+1];
        const TF rb = (comp==2) ? rhoref[k-1] : rhorefh[k];
        const TF rc = (comp==2) ? rhorefh[k]  : rhoref[k];
        const TF dz = (comp==2) ? dzhi[k]     : dzi[k];
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            t[c] +=
                - ( i2(u[c+1+o], u[c+1]) * i2(f[c], f[c+1])
                  - i2(u[c  +o], u[c  ]) * i2(f[c-1], f[c]) ) * dxi
                - ( i2(v[c+jj+o], v[c+jj]) * i2(f[c], f[c+jj])
                  - i2(v[c   +o], v[c   ]) * i2(f[c-jj], f[c]) ) * dyi
                - ( rt * i2(w[c+kk+o], w[c+kk]) * i2(f[c], f[c+kk])
                  - rb * i2(w[c   +o], w[c   ]) * i2(f[c-kk], f[c]) ) / rc * dz;
        }
    }
}

template<class TF>
void advec2_s(const mhh_grid& g, TF* t, const TF* s, const TF* u, const TF* v, const TF* w,
              const TF* rhoref, const TF* rhorefh)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzi = P<TF>(g.dzi);
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            t[c] +=
                - ( u[c+1]  * i2(s[c], s[c+1])  - u[c] * i2(s[c-1],  s[c]) ) * dxi
                - ( v[c+jj] * i2(s[c], s[c+jj]) - v[c] * i2(s[c-jj], s[c]) ) * dyi
                - ( rhorefh[k+1] * w[c+kk] * i2(s[c], s[c+kk])
                  - rhorefh[k  ] * w[c   ] * i2(s[c-kk], s[c]) ) / rhoref[k] * dzi[k];
        }
}

// calc_cfl of the three schemes (advec_2.cxx:51-78, advec_2i5.cxx:60-148, advec_4.cxx:51-86); returns cfl*dt
template<class TF>
double advec_cfl(const mhh_grid& g, int scheme, const TF* u, const TF* v, const TF* w, double dt_in)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF* dzi = P<TF>(g.dzi);
    TF dxi, dyi;
    if (scheme == MHH_ADVEC_4) { dxi = TF(1.)/TF(g.dx); dyi = TF(1.)/TF(g.dy); }   // 2, 2i4, 2i5: Grid_data::dxi
    else { dxi = TF(1./TF(g.dx)); dyi = TF(1./TF(g.dy)); }   // "1./dx" evaluated in double, narrowed
    TF cfl = 0;
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            TF a;
            if (scheme == MHH_ADVEC_2)
                a = std::abs(i2(u[c], u[c+1]))*dxi + std::abs(i2(v[c], v[c+jj]))*dyi + std::abs(i2(w[c], w[c+kk]))*dzi[k];
            else if (scheme == MHH_ADVEC_4M)                        // src/advec_4m.cxx:51-88
                a = std::abs(TF(-1./16.)*u[c-1] + TF(9./16.)*u[c] + TF(9./16.)*u[c+1] + TF(-1./16.)*u[c+2])*dxi
                  + std::abs(TF(-1./16.)*v[c-jj] + TF(9./16.)*v[c] + TF(9./16.)*v[c+jj] + TF(-1./16.)*v[c+2*jj])*dyi
                  + std::abs(TF(-1./16.)*w[c-kk] + TF(9./16.)*w[c] + TF(9./16.)*w[c+kk] + TF(-1./16.)*w[c+2*kk])*dzi[k];
            else if (scheme == MHH_ADVEC_2I62)                      // src/advec_2i62.cxx:58-105
                a = std::abs(i6(u[c-2], u[c-1], u[c], u[c+1], u[c+2], u[c+3]))*dxi
                  + std::abs(i6(v[c-2*jj], v[c-jj], v[c], v[c+jj], v[c+2*jj], v[c+3*jj]))*dyi
                  + std::abs(i2(w[c], w[c+kk]))*dzi[k];
            else if (scheme == MHH_ADVEC_2I4)                       // src/advec_2i4.cxx:51-99
                a = std::abs(i4c(u[c-1], u[c], u[c+1], u[c+2]))*dxi
                  + std::abs(i4c(v[c-jj], v[c], v[c+jj], v[c+2*jj]))*dyi
                  + std::abs((k == g.kstart || k == g.kend-1) ? i2(w[c], w[c+kk]) : i4c(w[c-kk], w[c], w[c+kk], w[c+2*kk]))*dzi[k];
            else if (scheme == MHH_ADVEC_4)
                a = std::abs(i4c(u[c-1], u[c], u[c+1], u[c+2]))*dxi
                  + std::abs(i4c(v[c-jj], v[c], v[c+jj], v[c+2*jj]))*dyi
                  + std::abs(i4c(w[c-kk], w[c], w[c+kk], w[c+2*kk]))*dzi[k];
            else
            {
                TF wi;
                if (k == g.kstart || k == g.kend-1)         wi = i2(w[c], w[c+kk]);
                else if (k == g.kstart+1 || k == g.kend-2 || scheme == MHH_ADVEC_2I53)  wi = i4ws(w[c-kk], w[c], w[c+kk], w[c+2*kk]);   // 2i53: 4th order on every inner level
                else                                        wi = i6(w[c-2*kk], w[c-kk], w[c], w[c+kk], w[c+2*kk], w[c+3*kk]);
                a = std::abs(i6(u[c-2], u[c-1], u[c], u[c+1], u[c+2], u[c+3]))*dxi
                  + std::abs(i6(v[c-2*jj], v[c-jj], v[c], v[c+jj], v[c+2*jj], v[c+3*jj]))*dyi
                  + std::abs(wi)*dzi[k];
            }
            cfl = std::max(cfl, a);
        }
    const TF dt = TF(dt_in);
    cfl = cfl*dt;
    return static_cast<double>(cfl);
}

// ---------------------------------------------------------------------------------------------
// advec_2i5  (src/advec_2i5.cxx:151-728)
// ---------------------------------------------------------------------------------------------
// Vertical face orders. For fields at cell centres in z (u, v, scalars) the face `kf` is the
// bottom face of cell kf; for w the "face" is the cell centre kf. order 0 = no flux (wall),
// 2 = 2nd order without upwind term, 4 = 4th/3rd order, 6 = 6th/5th order.
inline int face_order_c(const mhh_grid& g, int kf)   // faces of centred fields
{
    if (kf <= g.kstart || kf >= g.kend) return 0;
    if (kf == g.kstart+1 || kf == g.kend-1) return 2;
    if (kf == g.kstart+2 || kf == g.kend-2) return 4;
    return 6;
}
inline int face_order_w(const mhh_grid& g, int kc)   // centres, used by the w equation
{
    if (kc == g.kstart || kc == g.kend-1) return 2;
    if (kc == g.kstart+1 || kc == g.kend-2) return 4;
    return 6;
}

// centred and upwind interpolants of f on the face whose upper neighbour is f[c] (stride s):
// face lies between f[c-s] and f[c].
template<class TF> inline TF face_c(const TF* f, int c, int s, int order)
{
    if (order == 2) return i2(f[c-s], f[c]);
    if (order == 4) return i4ws(f[c-2*s], f[c-s], f[c], f[c+s]);
    return i6(f[c-3*s], f[c-2*s], f[c-s], f[c], f[c+s], f[c+2*s]);
}
template<class TF> inline TF face_u(const TF* f, int c, int s, int order)
{
    if (order == 4) return i3ws(f[c-2*s], f[c-s], f[c], f[c+s]);
    return i5(f[c-3*s], f[c-2*s], f[c-s], f[c], f[c+s], f[c+2*s]);
}

// one vertical update: returns the increment of the tendency given face data.
// ot/ob: orders of the top/bottom faces; wt/wb: advecting velocities; It,Ib centred; Dt,Db upwind.
template<class TF>
inline TF vert_incr(int ot, int ob, TF rt, TF rb, TF rc, TF dz, TF wt, TF wb, TF It, TF Ib, TF Dt, TF Db)
{
    TF cen;
    if (ob == 0)      cen = - ( rt * wt * It ) / rc * dz;
    else if (ot == 0) cen = - ( -rb * wb * Ib ) / rc * dz;
    else              cen = - ( rt * wt * It - rb * wb * Ib ) / rc * dz;
    const bool ut = (ot >= 4), ub = (ob >= 4);
    if (ut && ub) return cen + ( rt * std::abs(wt) * Dt - rb * std::abs(wb) * Db ) / rc * dz;
    if (ut)       return cen + ( rt * std::abs(wt) * Dt ) / rc * dz;
    if (ub)       return cen - ( rb * std::abs(wb) * Db ) / rc * dz;
    return cen;
}

template<class TF>
void advec25_mom(const mhh_grid& g, int comp, TF* t, const TF* u, const TF* v, const TF* w,
                 const TF* rhoref, const TF* rhorefh, int cap = 6)   // cap = 4: advec_2i53 (src/advec_2i53.cxx), the same scheme with 4th/3rd order vertically
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzi = P<TF>(g.dzi); const TF* dzhi = P<TF>(g.dzhi);
    const TF* f = (comp==0) ? u : (comp==1) ? v : w;
    const int o = (comp==0) ? -1 : (comp==1) ? -jj : -kk;
    const int k0 = (comp==2) ? g.kstart+1 : g.kstart;

    // pass 1: horizontal terms (advec_2i5.cxx:181-201, :333-352, :483-503)
    for (int k=k0; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const TF ue = i2(u[c+1+o], u[c+1]),   uw = i2(u[c+o], u[c]);
            const TF vn = i2(v[c+jj+o], v[c+jj]), vs = i2(v[c+o], v[c]);
            t[c] +=
                - ( ue * i6(f[c-2], f[c-1], f[c], f[c+1], f[c+2], f[c+3])
                  - uw * i6(f[c-3], f[c-2], f[c-1], f[c], f[c+1], f[c+2]) ) * dxi
                + ( std::abs(ue) * i5(f[c-2], f[c-1], f[c], f[c+1], f[c+2], f[c+3])
                  - std::abs(uw) * i5(f[c-3], f[c-2], f[c-1], f[c], f[c+1], f[c+2]) ) * dxi
                - ( vn * i6(f[c-2*jj], f[c-jj], f[c], f[c+jj], f[c+2*jj], f[c+3*jj])
                  - vs * i6(f[c-3*jj], f[c-2*jj], f[c-jj], f[c], f[c+jj], f[c+2*jj]) ) * dyi
                + ( std::abs(vn) * i5(f[c-2*jj], f[c-jj], f[c], f[c+jj], f[c+2*jj], f[c+3*jj])
                  - std::abs(vs) * i5(f[c-3*jj], f[c-2*jj], f[c-jj], f[c], f[c+jj], f[c+2*jj]) ) * dyi;
        }

    // pass 2: vertical terms, second accumulation into the tendency (advec_2i5.cxx:204-299, :355-449, :506-579)
    for (int k=k0; k<g.kend; ++k)
    {
        int ot, ob; TF rt, rb, rc, dz;
        if (comp == 2) { ot = std::min(face_order_w(g, k), cap);   ob = std::min(face_order_w(g, k-1), cap); rt = rhoref[k];    rb = rhoref[k-1]; rc = rhorefh[k]; dz = dzhi[k]; }
        else           { ot = std::min(face_order_c(g, k+1), cap); ob = std::min(face_order_c(g, k), cap);   rt = rhorefh[k+1]; rb = rhorefh[k];  rc = rhoref[k];  dz = dzi[k];  }
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const TF wt = i2(w[c+kk+o], w[c+kk]), wb = i2(w[c+o], w[c]);
            const TF It = ot ? face_c(f, c+kk, kk, ot) : TF(0), Ib = ob ? face_c(f, c, kk, ob) : TF(0);
            const TF Dt = (ot>=4) ? face_u(f, c+kk, kk, ot) : TF(0), Db = (ob>=4) ? face_u(f, c, kk, ob) : TF(0);
            t[c] += vert_incr(ot, ob, rt, rb, rc, dz, wt, wb, It, Ib, Dt, Db);
        }
    }
}

template<class TF>
void advec25_s(const mhh_grid& g, TF* t, const TF* s, const TF* u, const TF* v, const TF* w,
               const TF* rhoref, const TF* rhorefh, int cap = 6)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzi = P<TF>(g.dzi);
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            t[c] +=
                - ( u[c+1] * i6(s[c-2], s[c-1], s[c], s[c+1], s[c+2], s[c+3])
                  - u[c  ] * i6(s[c-3], s[c-2], s[c-1], s[c], s[c+1], s[c+2]) ) * dxi
                + ( std::abs(u[c+1]) * i5(s[c-2], s[c-1], s[c], s[c+1], s[c+2], s[c+3])
                  - std::abs(u[c  ]) * i5(s[c-3], s[c-2], s[c-1], s[c], s[c+1], s[c+2]) ) * dxi
                - ( v[c+jj] * i6(s[c-2*jj], s[c-jj], s[c], s[c+jj], s[c+2*jj], s[c+3*jj])
                  - v[c   ] * i6(s[c-3*jj], s[c-2*jj], s[c-jj], s[c], s[c+jj], s[c+2*jj]) ) * dyi
                + ( std::abs(v[c+jj]) * i5(s[c-2*jj], s[c-jj], s[c], s[c+jj], s[c+2*jj], s[c+3*jj])
                  - std::abs(v[c   ]) * i5(s[c-3*jj], s[c-2*jj], s[c-jj], s[c], s[c+jj], s[c+2*jj]) ) * dyi;
        }
    for (int k=g.kstart; k<g.kend; ++k)
    {
        const int ot = std::min(face_order_c(g, k+1), cap), ob = std::min(face_order_c(g, k), cap);
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const TF It = ot ? face_c(s, c+kk, kk, ot) : TF(0), Ib = ob ? face_c(s, c, kk, ob) : TF(0);
            const TF Dt = (ot>=4) ? face_u(s, c+kk, kk, ot) : TF(0), Db = (ob>=4) ? face_u(s, c, kk, ob) : TF(0);
            t[c] += vert_incr(ot, ob, rhorefh[k+1], rhorefh[k], rhoref[k], dzi[k], w[c+kk], w[c], It, Ib, Dt, Db);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// advec_4  (src/advec_4.cxx:88-486)
// ---------------------------------------------------------------------------------------------
// tend -= sum_m cg_m * ( I4(vel)_m * I4(f)_m ) * dinv  in each direction, three separate updates.
// Faces m = 0..3 sit at offsets (m-1.5) cells from the cell in the sweep direction.
template<class TF>
void advec4_mom(const mhh_grid& g, int comp, TF* t, const TF* u, const TF* v, const TF* w)
{
    using W = W4<TF>;
    const int jj = g.icells, kk = g.ijcells;
    const bool dim3 = (g.jtot != 1);
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzi4 = P<TF>(g.dzi4); const TF* dzhi4 = P<TF>(g.dzhi4);
    const TF* f = (comp==0) ? u : (comp==1) ? v : w;
    const int sd = (comp==0) ? 1 : (comp==1) ? jj : kk;      // staggering stride of the component
    const int k0 = (comp==2) ? g.kstart+1 : g.kstart;
    const TF cg[4] = {W::cg0, W::cg1, W::cg2, W::cg3};
    for (int k=k0; k<g.kend; ++k)
    {
        const bool bot = (k == k0), top = (k == g.kend-1);
        const TF dz = (comp==2) ? dzhi4[k] : dzi4[k];
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            // x
            {
                TF pr[4];
                for (int m=0; m<4; ++m)
                {
                    const int b = c + (m-1);
                    pr[m] = ci4(u[b-2*sd], u[b-sd], u[b], u[b+sd]) * ci4(f[c+m-3], f[c+m-2], f[c+m-1], f[c+m]);
                }
                t[c] -= ( cg[0]*pr[0] + cg[1]*pr[1] + cg[2]*pr[2] + cg[3]*pr[3] ) * dxi;
            }
            if (dim3)
            {
                TF pr[4];
                for (int m=0; m<4; ++m)
                {
                    const int b = c + (m-1)*jj;
                    pr[m] = ci4(v[b-2*sd], v[b-sd], v[b], v[b+sd]) * ci4(f[c+(m-3)*jj], f[c+(m-2)*jj], f[c+(m-1)*jj], f[c+m*jj]);
                }
                t[c] -= ( cg[0]*pr[0] + cg[1]*pr[1] + cg[2]*pr[2] + cg[3]*pr[3] ) * dyi;
            }
            {
                TF pr[4];
                for (int m=0; m<4; ++m)
                {
                    const int b = c + (m-1)*kk;
                    TF fi;
                    if (bot && m==0)      fi = bi4(f[c-2*kk], f[c-kk], f[c], f[c+kk]);
                    else if (top && m==3) fi = ti4(f[c-kk], f[c], f[c+kk], f[c+2*kk]);
                    else                  fi = ci4(f[c+(m-3)*kk], f[c+(m-2)*kk], f[c+(m-1)*kk], f[c+m*kk]);
                    // for w the advecting velocity IS the advected field, incl. the biased forms
                    const TF ve = (comp==2) ? fi : ci4(w[b-2*sd], w[b-sd], w[b], w[b+sd]);
                    pr[m] = ve * fi;
                }
                t[c] -= ( cg[0]*pr[0] + cg[1]*pr[1] + cg[2]*pr[2] + cg[3]*pr[3] ) * dz;
            }
        }
    }
}

template<class TF>
void advec4_s(const mhh_grid& g, TF* t, const TF* s, const TF* u, const TF* v, const TF* w)
{
    using W = W4<TF>;
    const int jj = g.icells, kk = g.ijcells;
    const bool dim3 = (g.jtot != 1);
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzi4 = P<TF>(g.dzi4);
    for (int k=g.kstart; k<g.kend; ++k)
    {
        const bool bot = (k == g.kstart), top = (k == g.kend-1);
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            t[c] -= ( W::cg0*(u[c-1] * ci4(s[c-3], s[c-2], s[c-1], s[c  ]))
                    + W::cg1*(u[c  ] * ci4(s[c-2], s[c-1], s[c  ], s[c+1]))
                    + W::cg2*(u[c+1] * ci4(s[c-1], s[c  ], s[c+1], s[c+2]))
                    + W::cg3*(u[c+2] * ci4(s[c  ], s[c+1], s[c+2], s[c+3])) ) * dxi;
            if (dim3)
                t[c] -= ( W::cg0*(v[c-jj  ] * ci4(s[c-3*jj], s[c-2*jj], s[c-jj], s[c]))
                        + W::cg1*(v[c     ] * ci4(s[c-2*jj], s[c-jj], s[c], s[c+jj]))
                        + W::cg2*(v[c+jj  ] * ci4(s[c-jj], s[c], s[c+jj], s[c+2*jj]))
                        + W::cg3*(v[c+2*jj] * ci4(s[c], s[c+jj], s[c+2*jj], s[c+3*jj])) ) * dyi;
            const TF f0 = bot ? bi4(s[c-2*kk], s[c-kk], s[c], s[c+kk]) : ci4(s[c-3*kk], s[c-2*kk], s[c-kk], s[c]);
            const TF f3 = top ? ti4(s[c-kk], s[c], s[c+kk], s[c+2*kk]) : ci4(s[c], s[c+kk], s[c+2*kk], s[c+3*kk]);
            t[c] -= ( W::cg0*(w[c-kk  ] * f0)
                    + W::cg1*(w[c     ] * ci4(s[c-2*kk], s[c-kk], s[c], s[c+kk]))
                    + W::cg2*(w[c+kk  ] * ci4(s[c-kk], s[c], s[c+kk], s[c+2*kk]))
                    + W::cg3*(w[c+2*kk] * f3) ) * dzi4[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// diff_2 (src/diff_2.cxx:39-86) -- dxidxi/dyidyi are DOUBLE even for TF=float
// ---------------------------------------------------------------------------------------------
template<class TF>
void diff2(const mhh_grid& g, bool is_w, TF* t, const TF* a, TF visc)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dx = TF(g.dx), dy = TF(g.dy);
    const double dxidxi = 1/(dx*dx);
    const double dyidyi = 1/(dy*dy);
    const TF* dzi = P<TF>(g.dzi); const TF* dzhi = P<TF>(g.dzhi);
    for (int k = is_w ? g.kstart+1 : g.kstart; k<g.kend; ++k)
    {
        const TF gt = is_w ? dzi[k]   : dzhi[k+1];
        const TF gb = is_w ? dzi[k-1] : dzhi[k];
        const TF gc = is_w ? dzhi[k]  : dzi[k];
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            t[c] += visc * (
                    + ( (a[c+1 ] - a[c]) - (a[c] - a[c-1 ]) ) * dxidxi
                    + ( (a[c+jj] - a[c]) - (a[c] - a[c-jj]) ) * dyidyi
                    + ( (a[c+kk] - a[c]) * gt - (a[c] - a[c-kk]) * gb ) * gc );
        }
    }
}

// ---------------------------------------------------------------------------------------------
// diff_4 (src/diff_4.cxx:41-173)
// ---------------------------------------------------------------------------------------------
template<class TF>
void diff4(const mhh_grid& g, bool is_w, TF* t, const TF* a, TF visc)
{
    using W = W4<TF>;
    const int jj = g.icells, kk = g.ijcells;
    const bool dim3 = (g.jtot != 1);
    const TF dx = TF(g.dx), dy = TF(g.dy);
    // diff_c spells 1./(dx*dx) (double quotient, narrowed), diff_w spells 1/(dx*dx) (TF quotient)
    const TF dxidxi = is_w ? TF(1/(dx*dx)) : TF(1./(dx*dx));
    const TF dyidyi = is_w ? TF(1/(dy*dy)) : TF(1./(dy*dy));
    const TF* dzi4 = P<TF>(g.dzi4); const TF* dzhi4 = P<TF>(g.dzhi4);
    const int k0 = is_w ? g.kstart+1 : g.kstart;
    for (int k=k0; k<g.kend; ++k)
    {
        const bool bot = (k == k0), top = (k == g.kend-1);
        // inner gradient metric at the four faces, outer metric at the cell
        const TF* gi = is_w ? dzi4 : dzhi4;
        const int s = is_w ? -1 : 0;     // w: faces k-2..k+1, centred: k-1..k+2
        const TF go = is_w ? dzhi4[k] : dzi4[k];
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            t[c] += visc * (W::cdg3*a[c-3] + W::cdg2*a[c-2] + W::cdg1*a[c-1] + W::cdg0*a[c] + W::cdg1*a[c+1] + W::cdg2*a[c+2] + W::cdg3*a[c+3])*dxidxi;
            if (dim3)
                t[c] += visc * (W::cdg3*a[c-3*jj] + W::cdg2*a[c-2*jj] + W::cdg1*a[c-jj] + W::cdg0*a[c] + W::cdg1*a[c+jj] + W::cdg2*a[c+2*jj] + W::cdg3*a[c+3*jj])*dyidyi;
            const TF g0 = bot ? bg4(a[c-2*kk], a[c-kk], a[c], a[c+kk]) : cg4(a[c-3*kk], a[c-2*kk], a[c-kk], a[c]);
            const TF g3 = top ? tg4(a[c-kk], a[c], a[c+kk], a[c+2*kk]) : cg4(a[c], a[c+kk], a[c+2*kk], a[c+3*kk]);
            t[c] += visc * ( W::cg0*g0 * gi[k-1+s]
                           + W::cg1*cg4(a[c-2*kk], a[c-kk], a[c], a[c+kk]) * gi[k+s]
                           + W::cg2*cg4(a[c-kk], a[c], a[c+kk], a[c+2*kk]) * gi[k+1+s]
                           + W::cg3*g3 * gi[k+2+s] ) * go;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// diff_smag2 (src/diff_smag2.cxx)
// ---------------------------------------------------------------------------------------------
template<class TF> inline TF sq(TF a) { return a*a; }
const double dsmall = 1.e-9;   // Constants::dsmall (include/constants.h:97) -- a double
template<class TF> constexpr TF kappa = 0.4;   // Constants::kappa

// calc_strain2 (:47-155)
template<class TF>
void smag_strain2(const mhh_grid& g, bool sm, TF* s2, const TF* u, const TF* v, const TF* w,
                  const TF* ugradbot, const TF* vgradbot)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1./TF(g.dx)), dyi = TF(1./TF(g.dy));
    const TF* dzi = P<TF>(g.dzi); const TF* dzhi = P<TF>(g.dzhi);
    for (int k=g.kstart; k<g.kend; ++k)
    {
        const bool mo = sm && (k == g.kstart);
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const int ij = i + j*jj;
            TF acc = sq((u[c+1]-u[c])*dxi);
            acc = acc + sq((v[c+jj]-v[c])*dyi);
            acc = acc + sq((w[c+kk]-w[c])*dzi[k]);
            acc = acc + TF(0.125)*sq((u[c      ]-u[c  -jj])*dyi + (v[c      ]-v[c-1   ])*dxi);
            acc = acc + TF(0.125)*sq((u[c+1    ]-u[c+1-jj])*dyi + (v[c+1    ]-v[c     ])*dxi);
            acc = acc + TF(0.125)*sq((u[c  +jj ]-u[c      ])*dyi + (v[c  +jj]-v[c-1+jj])*dxi);
            acc = acc + TF(0.125)*sq((u[c+1+jj ]-u[c+1    ])*dyi + (v[c+1+jj]-v[c  +jj])*dxi);
            if (mo)
            {
                acc = acc + TF(0.5)*sq(ugradbot[ij]);
                acc = acc + TF(0.125)*sq((w[c      ]-w[c-1   ])*dxi);
                acc = acc + TF(0.125)*sq((w[c+1    ]-w[c     ])*dxi);
                acc = acc + TF(0.125)*sq((w[c  +kk ]-w[c-1+kk])*dxi);
                acc = acc + TF(0.125)*sq((w[c+1+kk ]-w[c  +kk])*dxi);
                acc = acc + TF(0.5)*sq(vgradbot[ij]);
                acc = acc + TF(0.125)*sq((w[c      ]-w[c-jj   ])*dyi);
                acc = acc + TF(0.125)*sq((w[c+jj   ]-w[c      ])*dyi);
                acc = acc + TF(0.125)*sq((w[c   +kk]-w[c-jj+kk])*dyi);
                acc = acc + TF(0.125)*sq((w[c+jj+kk]-w[c   +kk])*dyi);
            }
            else
            {
                acc = acc + TF(0.125)*sq((u[c      ]-u[c  -kk])*dzhi[k  ] + (w[c      ]-w[c-1   ])*dxi);
                acc = acc + TF(0.125)*sq((u[c+1    ]-u[c+1-kk])*dzhi[k  ] + (w[c+1    ]-w[c     ])*dxi);
                acc = acc + TF(0.125)*sq((u[c  +kk ]-u[c     ])*dzhi[k+1] + (w[c  +kk ]-w[c-1+kk])*dxi);
                acc = acc + TF(0.125)*sq((u[c+1+kk ]-u[c+1   ])*dzhi[k+1] + (w[c+1+kk ]-w[c  +kk])*dxi);
                acc = acc + TF(0.125)*sq((v[c      ]-v[c   -kk])*dzhi[k  ] + (w[c      ]-w[c-jj   ])*dyi);
                acc = acc + TF(0.125)*sq((v[c+jj   ]-v[c+jj-kk])*dzhi[k  ] + (w[c+jj   ]-w[c      ])*dyi);
                acc = acc + TF(0.125)*sq((v[c   +kk]-v[c      ])*dzhi[k+1] + (w[c   +kk]-w[c-jj+kk])*dyi);
                acc = acc + TF(0.125)*sq((v[c+jj+kk]-v[c+jj   ])*dzhi[k+1] + (w[c+jj+kk]-w[c   +kk])*dyi);
            }
            s2[c] = TF(2.)*acc;
            s2[c] += dsmall;       // TF += double: evaluated in double, narrowed on store
        }
    }
}

template<class TF>
void evisc_mirror_walls(const mhh_grid& g, TF* evisc)
{
    const int jj = g.icells, kk = g.ijcells;
    for (int j=0; j<g.jcells; ++j)
        for (int i=0; i<g.icells; ++i)
        {
            const int b = i + j*jj + g.kstart*kk, t = i + j*jj + (g.kend-1)*kk;
            evisc[b-kk] = evisc[b];
            evisc[t+kk] = evisc[t];
        }
}

// calc_evisc (:254-367), stratified; evisc holds strain2 on entry
template<class TF>
void smag_evisc(const mhh_grid& g, bool sm, TF* evisc, const TF* N2, const TF* bgradbot, const TF* z0m, TF cs, TF tPr)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dx = TF(g.dx), dy = TF(g.dy);
    const TF* z = P<TF>(g.z); const TF* dz = P<TF>(g.dz);
    if (!sm)
    {
        for (int k=g.kstart; k<g.kend; ++k)
        {
            const TF mlen = cs*std::pow(dx*dy*dz[k], TF(1./3.));
            const TF fac = sq(mlen);
            FOR_INTERIOR_PLANE(g)
            {
                const int c = i + j*jj + k*kk;
                TF rit = N2[c] / evisc[c] / tPr;
                rit = std::min(rit, TF(1.-dsmall));
                evisc[c] = fac * std::sqrt(evisc[c]) * std::sqrt(TF(1.)-rit);
            }
        }
        evisc_mirror_walls(g, evisc);
    }
    else
    {
        const TF n = 2.;
        for (int k=g.kstart; k<g.kend; ++k)
        {
            const TF mlen0 = cs*std::pow(dx*dy*dz[k], TF(1./3.));
            FOR_INTERIOR_PLANE(g)
            {
                const int c = i + j*jj + k*kk;
                const int ij = i + j*jj;
                TF rit = ((k == g.kstart) ? bgradbot[ij] : N2[c]) / evisc[c] / tPr;
                rit = std::min(rit, TF(1.-dsmall));
                const TF mlen = std::pow(TF(1.)/(TF(1.)/std::pow(mlen0, n) + TF(1.)/(std::pow(kappa<TF>*(z[k]+z0m[ij]), n))), TF(1.)/n);
                evisc[c] = sq(mlen) * std::sqrt(evisc[c]) * std::sqrt(TF(1.)-rit);
            }
        }
    }
    cyclic<TF>(g, evisc, MHH_EDGE_BOTH);
}

// calc_evisc_neutral (:157-252)
template<class TF>
void smag_evisc_neutral(const mhh_grid& g, bool sm, TF* evisc, const TF* u, const TF* v, const TF* z0m, TF cs, TF visc)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dx = TF(g.dx), dy = TF(g.dy), zsize = TF(g.zsize);
    const TF* z = P<TF>(g.z); const TF* dz = P<TF>(g.dz); const TF* dzhi = P<TF>(g.dzhi);
    constexpr TF n_mason = TF(1.);
    constexpr TF A_vandriest = TF(26.);
    if (!sm)
    {
        for (int k=g.kstart; k<g.kend; ++k)
        {
            const TF mlen_smag = cs*std::pow(dx*dy*dz[k], TF(1./3.));
            FOR_INTERIOR_PLANE(g)
            {
                const int cb = i + j*jj + g.kstart*kk, ct = i + j*jj + g.kend*kk;
                const TF utb = std::pow( sq( visc*(u[cb] - u[cb-kk])*dzhi[g.kstart] ) + sq( visc*(v[cb] - v[cb-kk])*dzhi[g.kstart] ), TF(0.25) );
                const TF utt = std::pow( sq( visc*(u[ct] - u[ct-kk])*dzhi[g.kend] ) + sq( visc*(v[ct] - v[ct-kk])*dzhi[g.kend] ), TF(0.25) );
                const TF fb = TF(1.) - std::exp( -(        z[k] *utb) / (A_vandriest*visc) );
                const TF ft = TF(1.) - std::exp( -((zsize-z[k])*utt) / (A_vandriest*visc) );
                const TF fac = std::min(fb, ft);
                const int c = i + j*jj + k*kk;
                evisc[c] = sq(fac * mlen_smag) * std::sqrt(evisc[c]);
            }
        }
        evisc_mirror_walls(g, evisc);
    }
    else
    {
        for (int k=g.kstart; k<g.kend; ++k)
        {
            const TF mlen0 = cs*std::pow(dx*dy*dz[k], TF(1./3.));
            FOR_INTERIOR_PLANE(g)
            {
                const int c = i + j*jj + k*kk;
                const int ij = i + j*jj;
                const TF mlen = std::pow(TF(1.)/(TF(1.)/std::pow(mlen0, n_mason) + TF(1.)/(std::pow(kappa<TF>*(z[k]+z0m[ij]), n_mason))), TF(1.)/n_mason);
                evisc[c] = sq(mlen) * std::sqrt(evisc[c]);
            }
        }
    }
    cyclic<TF>(g, evisc, MHH_EDGE_BOTH);
}

// diff_u / diff_v (:369-571): comp 0 = u (stagger stride 1, other horizontal stride jj), comp 1 = v.
template<class TF>
void smag_diff_uv(const mhh_grid& g, int comp, bool sm, TF* t, const TF* u, const TF* v, const TF* w,
                  const TF* ev, const TF* fluxbot, const TF* fluxtop, const TF* rhoref, const TF* rhorefh, TF visc)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1./TF(g.dx)), dyi = TF(1./TF(g.dy));
    const TF* dzi = P<TF>(g.dzi); const TF* dzhi = P<TF>(g.dzhi);
    for (int k=g.kstart; k<g.kend; ++k)
    {
        const bool fb = sm && (k == g.kstart), ft = sm && (k == g.kend-1);
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const int ij = i + j*jj;
            TF hor, evt = 0, evb = 0;
            if (comp == 0)
            {
                const TF ee = ev[c] + visc;
                const TF ew = ev[c-1] + visc;
                const TF en = TF(0.25)*(ev[c-1   ] + ev[c   ] + ev[c-1+jj] + ev[c+jj]) + visc;
                const TF es = TF(0.25)*(ev[c-1-jj] + ev[c-jj] + ev[c-1   ] + ev[c   ]) + visc;
                if (!ft) evt = TF(0.25)*(ev[c-1   ] + ev[c   ] + ev[c-1+kk] + ev[c+kk]) + visc;
                if (!fb) evb = TF(0.25)*(ev[c-1-kk] + ev[c-kk] + ev[c-1   ] + ev[c   ]) + visc;
                hor = + ( ee*(u[c+1]-u[c  ])*dxi - ew*(u[c  ]-u[c-1])*dxi ) * TF(2.)*dxi
                      + ( en*((u[c+jj]-u[c   ])*dyi + (v[c+jj]-v[c-1+jj])*dxi)
                        - es*((u[c   ]-u[c-jj])*dyi + (v[c   ]-v[c-1   ])*dxi) ) * dyi;
                TF ver;
                if (fb)      ver = ( rhorefh[k+1] * evt*((u[c+kk]-u[c])*dzhi[k+1] + (w[c+kk]-w[c-1+kk])*dxi) + rhorefh[k] * fluxbot[ij] ) / rhoref[k] * dzi[k];
                else if (ft) ver = ( - rhorefh[k+1] * fluxtop[ij] - rhorefh[k] * evb*((u[c]-u[c-kk])*dzhi[k] + (w[c]-w[c-1])*dxi) ) / rhoref[k] * dzi[k];
                else         ver = ( rhorefh[k+1] * evt*((u[c+kk]-u[c   ])*dzhi[k+1] + (w[c+kk]-w[c-1+kk])*dxi)
                                   - rhorefh[k  ] * evb*((u[c   ]-u[c-kk])*dzhi[k  ] + (w[c   ]-w[c-1   ])*dxi) ) / rhoref[k] * dzi[k];
                t[c] += hor + ver;
            }
            else
            {
                const TF ee = TF(0.25)*(ev[c  -jj] + ev[c  ] + ev[c+1-jj] + ev[c+1]) + visc;
                const TF ew = TF(0.25)*(ev[c-1-jj] + ev[c-1] + ev[c  -jj] + ev[c  ]) + visc;
                const TF en = ev[c] + visc;
                const TF es = ev[c-jj] + visc;
                if (!ft) evt = TF(0.25)*(ev[c   -jj] + ev[c   ] + ev[c+kk-jj] + ev[c+kk]) + visc;
                if (!fb) evb = TF(0.25)*(ev[c-kk-jj] + ev[c-kk] + ev[c   -jj] + ev[c   ]) + visc;
                hor = + ( ee*((v[c+1]-v[c  ])*dxi + (u[c+1]-u[c+1-jj])*dyi)
                        - ew*((v[c  ]-v[c-1])*dxi + (u[c  ]-u[c  -jj])*dyi) ) * dxi
                      + ( en*(v[c+jj]-v[c   ])*dyi - es*(v[c   ]-v[c-jj])*dyi ) * TF(2.)*dyi;
                TF ver;
                if (fb)      ver = ( rhorefh[k+1] * evt*((v[c+kk]-v[c])*dzhi[k+1] + (w[c+kk]-w[c-jj+kk])*dyi) + rhorefh[k] * fluxbot[ij] ) / rhoref[k] * dzi[k];
                else if (ft) ver = ( - rhorefh[k+1] * fluxtop[ij] - rhorefh[k] * evb*((v[c]-v[c-kk])*dzhi[k] + (w[c]-w[c-jj])*dyi) ) / rhoref[k] * dzi[k];
                else         ver = ( rhorefh[k+1] * evt*((v[c+kk]-v[c   ])*dzhi[k+1] + (w[c+kk]-w[c-jj+kk])*dyi)
                                   - rhorefh[k  ] * evb*((v[c   ]-v[c-kk])*dzhi[k  ] + (w[c   ]-w[c-jj   ])*dyi) ) / rhoref[k] * dzi[k];
                t[c] += hor + ver;
            }
        }
    }
}

// diff_w (:573-617)
template<class TF>
void smag_diff_w(const mhh_grid& g, TF* t, const TF* u, const TF* v, const TF* w, const TF* ev,
                 const TF* rhoref, const TF* rhorefh, TF visc)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1./TF(g.dx)), dyi = TF(1./TF(g.dy));
    const TF* dzi = P<TF>(g.dzi); const TF* dzhi = P<TF>(g.dzhi);
    for (int k=g.kstart+1; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const TF ee = TF(0.25)*(ev[c   -kk] + ev[c   ] + ev[c+1 -kk] + ev[c+1 ]) + visc;
            const TF ew = TF(0.25)*(ev[c-1 -kk] + ev[c-1 ] + ev[c   -kk] + ev[c   ]) + visc;
            const TF en = TF(0.25)*(ev[c   -kk] + ev[c   ] + ev[c+jj-kk] + ev[c+jj]) + visc;
            const TF es = TF(0.25)*(ev[c-jj-kk] + ev[c-jj] + ev[c   -kk] + ev[c   ]) + visc;
            const TF et = ev[c] + visc;
            const TF eb = ev[c-kk] + visc;
            t[c] +=
                + ( ee*((w[c+1]-w[c  ])*dxi + (u[c+1]-u[c+1-kk])*dzhi[k])
                  - ew*((w[c  ]-w[c-1])*dxi + (u[c  ]-u[c  -kk])*dzhi[k]) ) * dxi
                + ( en*((w[c+jj]-w[c   ])*dyi + (v[c+jj]-v[c+jj-kk])*dzhi[k])
                  - es*((w[c   ]-w[c-jj])*dyi + (v[c   ]-v[c   -kk])*dzhi[k]) ) * dyi
                + ( rhoref[k  ] * et*(w[c+kk]-w[c   ])*dzi[k  ]
                  - rhoref[k-1] * eb*(w[c   ]-w[c-kk])*dzi[k-1] ) / rhorefh[k] * TF(2.)*dzhi[k];
        }
}

// diff_c (:619-709)
template<class TF>
void smag_diff_c(const mhh_grid& g, bool sm, TF* t, const TF* a, const TF* ev, const TF* fluxbot, const TF* fluxtop,
                 const TF* rhoref, const TF* rhorefh, TF tPr, TF visc)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dx = TF(g.dx), dy = TF(g.dy);
    const TF dxidxi = TF(1./(dx*dx)), dyidyi = TF(1./(dy*dy));
    const TF* dzi = P<TF>(g.dzi); const TF* dzhi = P<TF>(g.dzhi);
    for (int k=g.kstart; k<g.kend; ++k)
    {
        const bool fb = sm && (k == g.kstart), ft = sm && (k == g.kend-1);
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const int ij = i + j*jj;
            const TF ee = TF(0.5)*(ev[c   ]+ev[c+1 ])/tPr + visc;
            const TF ew = TF(0.5)*(ev[c-1 ]+ev[c   ])/tPr + visc;
            const TF en = TF(0.5)*(ev[c   ]+ev[c+jj])/tPr + visc;
            const TF es = TF(0.5)*(ev[c-jj]+ev[c   ])/tPr + visc;
            TF et = 0, eb = 0;
            if (!ft) et = TF(0.5)*(ev[c   ]+ev[c+kk])/tPr + visc;
            if (!fb) eb = TF(0.5)*(ev[c-kk]+ev[c   ])/tPr + visc;
            const TF hor = + ( ee*(a[c+1 ]-a[c]) - ew*(a[c]-a[c-1 ]) ) * dxidxi
                           + ( en*(a[c+jj]-a[c]) - es*(a[c]-a[c-jj]) ) * dyidyi;
            TF ver;
            if (fb)      ver = ( rhorefh[k+1] * et*(a[c+kk]-a[c])*dzhi[k+1] + rhorefh[k] * fluxbot[ij] ) / rhoref[k] * dzi[k];
            else if (ft) ver = ( -rhorefh[k+1] * fluxtop[ij] - rhorefh[k] * eb*(a[c]-a[c-kk])*dzhi[k] ) / rhoref[k] * dzi[k];
            else         ver = ( rhorefh[k+1] * et*(a[c+kk]-a[c   ])*dzhi[k+1]
                               - rhorefh[k  ] * eb*(a[c   ]-a[c-kk])*dzhi[k]   ) / rhoref[k] * dzi[k];
            t[c] += hor + ver;
        }
    }
}

// calc_dnmul (:711-736) + get_dn's "1./(dx*dx)" double evaluation
template<class TF>
double smag_dnmul(const mhh_grid& g, const TF* ev, TF tPr)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dx = TF(g.dx), dy = TF(g.dy);
    const TF dxidxi = TF(1./(dx*dx)), dyidyi = TF(1./(dy*dy));
    const TF* dzi = P<TF>(g.dzi);
    const TF tPrfac_i = TF(1)/std::min(TF(1.), tPr);
    TF dnmul = 0;
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            dnmul = std::max(dnmul, std::abs(ev[c]*tPrfac_i*(dxidxi + dyidyi + dzi[k]*dzi[k])));
        }
    return static_cast<double>(dnmul);
}

// Thermo_dry calc_N2 (src/thermo_dry.cxx:66-78)
template<class TF>
void calc_N2(const mhh_grid& g, TF* N2, const TF* th, const TF* thref, TF grav)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF* dzi = P<TF>(g.dzi);
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            N2[c] = grav/thref[k]*TF(0.5)*(th[c+kk] - th[c-kk])*dzi[k];
        }
}

// ---------------------------------------------------------------------------------------------
// FFT: FFTW r2r R2HC / HC2R semantics (src/fft.cxx:143-150, :334-448), written from the DFT
// definition. Half-complex layout: r0, r1, ..., r_{n/2}, i_{(n+1)/2-1}, ..., i_1.
// Power-of-two lengths use an iterative radix-2 complex FFT, other lengths the O(n^2) sum.
// This is NOT FFTW: agreement with the reference is to rounding, not bitwise ("parity unpinned"
// for the spectral stage).
// ---------------------------------------------------------------------------------------------
struct Dft
{
    int n; bool pow2;
    std::vector<double> cs, sn;      // cos/sin(2 pi m / n)
    std::vector<int> rev;
    explicit Dft(int n_) : n(n_), pow2((n_ & (n_-1)) == 0), cs(n_), sn(n_)
    {
        const long double tp = 2.0L*acosl(-1.0L)/n;
        for (int m=0; m<n; ++m) { cs[m] = (double)cosl(tp*m); sn[m] = (double)sinl(tp*m); }
        if (pow2)
        {
            rev.resize(n);
            int lg = 0; while ((1<<lg) < n) ++lg;
            for (int m=0; m<n; ++m) { int r=0; for (int b=0; b<lg; ++b) if (m & (1<<b)) r |= 1<<(lg-1-b); rev[m]=r; }
        }
    }
    // X_k = sum_j x_j exp(sign * 2 pi i j k / n), in place on (re, im)
    void cfft(std::vector<double>& re, std::vector<double>& im, int sign) const
    {
        if (pow2)
        {
            for (int m=0; m<n; ++m) if (rev[m] > m) { std::swap(re[m], re[rev[m]]); std::swap(im[m], im[rev[m]]); }
            for (int len=2; len<=n; len<<=1)
            {
                const int half = len>>1, step = n/len;
                for (int s=0; s<n; s+=len)
                    for (int q=0; q<half; ++q)
                    {
                        const double wr = cs[q*step], wi = sign*sn[q*step];
                        const int a = s+q, b = a+half;
                        const double xr = re[b]*wr - im[b]*wi, xi = re[b]*wi + im[b]*wr;
                        re[b] = re[a]-xr; im[b] = im[a]-xi; re[a] += xr; im[a] += xi;
                    }
            }
        }
        else
        {
            std::vector<double> or_(n), oi(n);
            for (int k=0; k<n; ++k)
            {
                long double ar=0, ai=0;
                for (int j=0; j<n; ++j)
                {
                    const int m = (int)(((long long)j*k) % n);
                    const double wr = cs[m], wi = sign*sn[m];
                    ar += re[j]*wr - im[j]*wi; ai += re[j]*wi + im[j]*wr;
                }
                or_[k] = (double)ar; oi[k] = (double)ai;
            }
            re.swap(or_); im.swap(oi);
        }
    }
    template<class TF> void r2hc(const TF* in, int istride, TF* out, int ostride) const
    {
        std::vector<double> re(n), im(n, 0.0);
        for (int j=0; j<n; ++j) re[j] = in[(size_t)j*istride];
        cfft(re, im, -1);
        for (int k=0; k<=n/2; ++k) out[(size_t)k*ostride] = TF(re[k]);
        for (int k=1; k<(n+1)/2; ++k) out[(size_t)(n-k)*ostride] = TF(im[k]);
    }
    template<class TF> void hc2r(const TF* in, int istride, TF* out, int ostride) const
    {
        std::vector<double> re(n), im(n);
        re[0] = in[0]; im[0] = 0;
        for (int k=1; k<(n+1)/2; ++k)
        {
            re[k] = in[(size_t)k*istride]; im[k] = in[(size_t)(n-k)*istride];
            re[n-k] = re[k]; im[n-k] = -im[k];
        }
        if (n%2 == 0) { re[n/2] = in[(size_t)(n/2)*istride]; im[n/2] = 0; }
        cfft(re, im, +1);
        for (int j=0; j<n; ++j) out[(size_t)j*ostride] = TF(re[j]);
    }
};

// FFT::exec_forward (serial, src/fft.cxx:334-390): x transforms of every row, then y transforms
template<class TF>
void fft_forward(const mhh_grid& g, TF* data)
{
    const int itot = g.itot, jtot = g.jtot;
    Dft dx(itot), dy(jtot);
    std::vector<TF> tmp(std::max(itot, jtot));
    for (int k=0; k<g.ktot; ++k)
    {
        TF* pl = data + (size_t)k*itot*jtot;
        for (int j=0; j<jtot; ++j) { dx.r2hc(pl + (size_t)j*itot, 1, tmp.data(), 1); std::copy(tmp.begin(), tmp.begin()+itot, pl + (size_t)j*itot); }
        for (int i=0; i<itot; ++i) { dy.r2hc(pl + i, itot, tmp.data(), 1); for (int j=0; j<jtot; ++j) pl[i + (size_t)j*itot] = tmp[j]; }
    }
}
// FFT::exec_backward (src/fft.cxx:392-448): y back (/jtot), x back (/itot)
template<class TF>
void fft_backward(const mhh_grid& g, TF* data)
{
    const int itot = g.itot, jtot = g.jtot;
    Dft dx(itot), dy(jtot);
    std::vector<TF> tmp(std::max(itot, jtot));
    for (int k=0; k<g.ktot; ++k)
    {
        TF* pl = data + (size_t)k*itot*jtot;
        for (int i=0; i<itot; ++i) { dy.hc2r(pl + i, itot, tmp.data(), 1); for (int j=0; j<jtot; ++j) pl[i + (size_t)j*itot] = tmp[j] / jtot; }
        for (int j=0; j<jtot; ++j) { dx.hc2r(pl + (size_t)j*itot, 1, tmp.data(), 1); for (int i=0; i<itot; ++i) pl[i + (size_t)j*itot] = tmp[i] / itot; }
    }
}

// ---------------------------------------------------------------------------------------------
// pres_2 (src/pres_2.cxx)
// ---------------------------------------------------------------------------------------------
// Pres_2::set_values (:125-153)
template<class TF>
void pres2_set_values(const mhh_grid& g, const TF* rhorefh, std::vector<TF>& bmati, std::vector<TF>& bmatj, std::vector<TF>& a, std::vector<TF>& c)
{
    const TF dx = TF(g.dx), dy = TF(g.dy);
    const TF dxidxi = 1./(dx*dx), dyidyi = 1./(dy*dy);
    const TF pi = std::acos(-1.);
    bmati.resize(g.itot); bmatj.resize(g.jtot); a.resize(g.kmax); c.resize(g.kmax);
    for (int j=0; j<g.jtot/2+1; ++j) bmatj[j] = 2. * (std::cos(2.*pi*(TF)j/(TF)g.jtot)-1.) * dyidyi;
    for (int j=g.jtot/2+1; j<g.jtot; ++j) bmatj[j] = bmatj[g.jtot-j];
    for (int i=0; i<g.itot/2+1; ++i) bmati[i] = 2. * (std::cos(2.*pi*(TF)i/(TF)g.itot)-1.) * dxidxi;
    for (int i=g.itot/2+1; i<g.itot; ++i) bmati[i] = bmati[g.itot-i];
    const TF* dz = P<TF>(g.dz); const TF* dzhi = P<TF>(g.dzhi);
    for (int k=0; k<g.kmax; ++k)
    {
        a[k] = dz[k+g.kgc] * rhorefh[k+g.kgc  ]*dzhi[k+g.kgc  ];
        c[k] = dz[k+g.kgc] * rhorefh[k+g.kgc+1]*dzhi[k+g.kgc+1];
    }
}

// Pres_2::input (:156-196); p is packed imax*jmax*kmax
template<class TF>
void pres2_input(const mhh_grid& g, TF* p, const TF* u, const TF* v, const TF* w, TF* ut, TF* vt, TF* wt,
                 const TF* rhoref, const TF* rhorefh, TF dt)
{
    const int jj = g.icells, kk = g.ijcells;
    const int jjp = g.imax, kkp = g.imax*g.jmax;
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy), dti = TF(1.)/dt;
    const TF* dzi = P<TF>(g.dzi);
    cyclic<TF>(g, ut, MHH_EDGE_EW);
    cyclic<TF>(g, vt, MHH_EDGE_NS);
    for (int k=0; k<g.kmax; ++k)
        for (int j=0; j<g.jmax; ++j)
            for (int i=0; i<g.imax; ++i)
            {
                const int cp = i + j*jjp + k*kkp;
                const int c  = i+g.igc + (j+g.jgc)*jj + (k+g.kgc)*kk;
                const int kc = k+g.kgc;
                p[cp] = rhoref[kc] * ( (ut[c+1 ] + u[c+1 ] * dti) - (ut[c] + u[c] * dti) ) * dxi
                      + rhoref[kc] * ( (vt[c+jj] + v[c+jj] * dti) - (vt[c] + v[c] * dti) ) * dyi
                      + ( rhorefh[kc+1] * (wt[c+kk] + w[c+kk] * dti)
                        - rhorefh[kc  ] * (wt[c   ] + w[c   ] * dti) ) * dzi[kc];
            }
}

// spectral tridiagonal solve: Pres_2::solve's matrix set-up (:289-326) + tdma (:202-263)
template<class TF>
void pres2_spectral_solve(const mhh_grid& g, TF* p, const TF* rhoref, const std::vector<TF>& bmati, const std::vector<TF>& bmatj,
                          const std::vector<TF>& a, const std::vector<TF>& c)
{
    const int ib = g.itot, jb = g.jtot, kmax = g.kmax, kgc = g.kgc;   // single rank: iblock=itot, jblock=jtot
    const size_t kk = (size_t)ib*jb;
    const TF* dz = P<TF>(g.dz);
    std::vector<TF> b(kk*kmax), work3d(kk*kmax), work2d(kk);
    for (int k=0; k<kmax; ++k)
        for (int j=0; j<jb; ++j)
            for (int i=0; i<ib; ++i)
            {
                const size_t ijk = i + (size_t)j*ib + k*kk;
                b[ijk] = dz[k+kgc]*dz[k+kgc] * rhoref[k+kgc]*(bmati[i]+bmatj[j]) - (a[k]+c[k]);
                p[ijk] = dz[k+kgc]*dz[k+kgc] * p[ijk];
            }
    for (int j=0; j<jb; ++j)
        for (int i=0; i<ib; ++i)
        {
            const size_t ij = i + (size_t)j*ib;
            b[ij] += a[0];
            const size_t top = ij + (kmax-1)*kk;
            if (i == 0 && j == 0) b[top] -= c[kmax-1];
            else                  b[top] += c[kmax-1];
        }
    // Thomas algorithm, column by column (same operation order per column as the reference's plane sweeps)
    for (size_t ij=0; ij<kk; ++ij)
    {
        TF w2 = b[ij];
        p[ij] /= w2;
        for (int k=1; k<kmax; ++k)
        {
            const size_t ijk = ij + k*kk;
            work3d[ijk] = c[k-1] / w2;
            w2 = b[ijk] - a[k]*work3d[ijk];
            p[ijk] -= a[k]*p[ijk-kk];
            p[ijk] /= w2;
        }
        for (int k=kmax-2; k>=0; --k)
        {
            const size_t ijk = ij + k*kk;
            p[ijk] -= work3d[ijk+kk]*p[ijk+kk];
        }
    }
    (void)work2d;
}

// unpack + bottom ghost + cyclic (:333-362)
template<class TF>
void pres2_unpack(const mhh_grid& g, TF* p, const TF* packed)
{
    const int jj = g.icells, kk = g.ijcells;
    for (int k=0; k<g.kmax; ++k)
        for (int j=0; j<g.jmax; ++j)
            for (int i=0; i<g.imax; ++i)
                p[i+g.igc + (j+g.jgc)*jj + (size_t)(k+g.kgc)*kk] = packed[i + j*g.imax + (size_t)k*g.imax*g.jmax];
    FOR_INTERIOR_PLANE(g)
    {
        const int c = i + j*jj + g.kstart*kk;
        p[c-kk] = p[c];
    }
    cyclic<TF>(g, p, MHH_EDGE_BOTH);
}

// Pres_2::output (:365-387)
template<class TF>
void pres2_output(const mhh_grid& g, TF* ut, TF* vt, TF* wt, const TF* p)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzhi = P<TF>(g.dzhi);
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            ut[c] -= (p[c] - p[c-1 ]) * dxi;
            vt[c] -= (p[c] - p[c-jj]) * dyi;
            wt[c] -= (p[c] - p[c-kk]) * dzhi[k];
        }
}

// Pres_2::calc_divergence (:391-422)
template<class TF>
double pres2_divergence(const mhh_grid& g, const TF* u, const TF* v, const TF* w, const TF* rhoref, const TF* rhorefh)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzi = P<TF>(g.dzi);
    TF divmax = 0.;
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const TF div = rhoref[k]*((u[c+1]-u[c])*dxi + (v[c+jj]-v[c])*dyi)
                         + (rhorefh[k+1]*w[c+kk]-rhorefh[k]*w[c])*dzi[k];
            divmax = std::max(divmax, std::abs(div));
        }
    return static_cast<double>(divmax);
}

// ---------------------------------------------------------------------------------------------
// pres_4 (src/pres_4.cxx)
// ---------------------------------------------------------------------------------------------
template<class TF>
struct Pres4Mat { std::vector<TF> bmati, bmatj, m1, m2, m3, m4, m5, m6, m7; };

// Pres_4::set_values (:179-252)
template<class TF>
void pres4_set_values(const mhh_grid& g, Pres4Mat<TF>& M)
{
    const int itot = g.itot, jtot = g.jtot, kmax = g.kmax, kstart = g.kstart;
    const TF dx = TF(g.dx), dy = TF(g.dy);
    const TF dxidxi = 1./(dx*dx), dyidyi = 1./(dy*dy);
    const TF pi = std::acos(-1.);
    M.bmati.resize(itot); M.bmatj.resize(jtot);
    for (auto* m : {&M.m1,&M.m2,&M.m3,&M.m4,&M.m5,&M.m6,&M.m7}) m->assign(kmax, TF(0));
    for (int j=0; j<jtot/2+1; j++)
        M.bmatj[j] = ( 2.* (1./576.)    * std::cos(6.*pi*(double)j/(double)jtot)
                     - 2.* (54./576.)   * std::cos(4.*pi*(double)j/(double)jtot)
                     + 2.* (783./576.)  * std::cos(2.*pi*(double)j/(double)jtot)
                     -     (1460./576.) ) * dyidyi;
    for (int j=jtot/2+1; j<jtot; j++) M.bmatj[j] = M.bmatj[jtot-j];
    for (int i=0; i<itot/2+1; i++)
        M.bmati[i] = ( 2.* (1./576.)    * std::cos(6.*pi*(double)i/(double)itot)
                     - 2.* (54./576.)   * std::cos(4.*pi*(double)i/(double)itot)
                     + 2.* (783./576.)  * std::cos(2.*pi*(double)i/(double)itot)
                     -     (1460./576.) ) * dxidxi;
    for (int i=itot/2+1; i<itot; i++) M.bmati[i] = M.bmati[itot-i];

    const TF* dzi4 = P<TF>(g.dzi4); const TF* h = P<TF>(g.dzhi4);
    int k = 0, kc = kstart;
    M.m1[k] = 0.;
    M.m2[k] = (1./576.) * (               -  27.*h[kc]                               ) * dzi4[kc];
    M.m3[k] = (1./576.) * ( -1.*h[kc+1] + 729.*h[kc] +  27.*h[kc+1]                  ) * dzi4[kc];
    M.m4[k] = (1./576.) * ( 27.*h[kc+1] - 729.*h[kc] - 729.*h[kc+1] -  1.*h[kc+2]    ) * dzi4[kc];
    M.m5[k] = (1./576.) * (-27.*h[kc+1] +  27.*h[kc] + 729.*h[kc+1] + 27.*h[kc+2]    ) * dzi4[kc];
    M.m6[k] = (1./576.) * (  1.*h[kc+1]              -  27.*h[kc+1] - 27.*h[kc+2]    ) * dzi4[kc];
    M.m7[k] = (1./576.) * (                                         +  1.*h[kc+2]    ) * dzi4[kc];
    for (k=1; k<kmax-1; k++)
    {
        kc = kstart+k;
        M.m1[k] = (1./576.) * (   1.*h[kc-1]                                                 ) * dzi4[kc];
        M.m2[k] = (1./576.) * ( -27.*h[kc-1] -  27.*h[kc]                                    ) * dzi4[kc];
        M.m3[k] = (1./576.) * (  27.*h[kc-1] + 729.*h[kc] +  27.*h[kc+1]                     ) * dzi4[kc];
        M.m4[k] = (1./576.) * (  -1.*h[kc-1] - 729.*h[kc] - 729.*h[kc+1] -  1.*h[kc+2]       ) * dzi4[kc];
        M.m5[k] = (1./576.) * (              +  27.*h[kc] + 729.*h[kc+1] + 27.*h[kc+2]       ) * dzi4[kc];
        M.m6[k] = (1./576.) * (                           -  27.*h[kc+1] - 27.*h[kc+2]       ) * dzi4[kc];
        M.m7[k] = (1./576.) * (                                          +  1.*h[kc+2]       ) * dzi4[kc];
    }
    k = kmax-1; kc = kstart+k;
    M.m1[k] = (1./576.) * (   1.*h[kc-1]                                             ) * dzi4[kc];
    M.m2[k] = (1./576.) * ( -27.*h[kc-1] -  27.*h[kc]                +  1.*h[kc]     ) * dzi4[kc];
    M.m3[k] = (1./576.) * (  27.*h[kc-1] + 729.*h[kc] +  27.*h[kc+1] - 27.*h[kc]     ) * dzi4[kc];
    M.m4[k] = (1./576.) * (  -1.*h[kc-1] - 729.*h[kc] - 729.*h[kc+1] + 27.*h[kc]     ) * dzi4[kc];
    M.m5[k] = (1./576.) * (              +  27.*h[kc] + 729.*h[kc+1] -  1.*h[kc]     ) * dzi4[kc];
    M.m6[k] = (1./576.) * (                           -  27.*h[kc+1]                 ) * dzi4[kc];
    M.m7[k] = 0.;
}

// Pres_4::input (:256-317)
template<class TF>
void pres4_input(const mhh_grid& g, TF* p, const TF* u, const TF* v, const TF* w, TF* ut, TF* vt, TF* wt, TF dt)
{
    using W = W4<TF>;
    const int jj = g.icells, kk = g.ijcells;
    const bool dim3 = (g.jtot != 1);
    const TF dxi = 1./TF(g.dx), dyi = 1./TF(g.dy), dti = 1./dt;
    const TF* dzi4 = P<TF>(g.dzi4);
    cyclic<TF>(g, ut, MHH_EDGE_EW);
    if (dim3) cyclic<TF>(g, vt, MHH_EDGE_NS);
    for (int j=0; j<g.jmax; j++)
        for (int i=0; i<g.imax; i++)
        {
            const int b = i+g.igc + (j+g.jgc)*jj + g.kgc*kk;
            wt[b-kk] = -wt[b+kk];
            const int t = i+g.igc + (j+g.jgc)*jj + (g.kmax+g.kgc)*kk;
            wt[t+kk] = -wt[t-kk];
        }
    for (int k=0; k<g.kmax; k++)
        for (int j=0; j<g.jmax; j++)
            for (int i=0; i<g.imax; i++)
            {
                const size_t cp = i + j*g.imax + (size_t)k*g.imax*g.jmax;
                const int c = i+g.igc + (j+g.jgc)*jj + (k+g.kgc)*kk;
                p[cp]  = (W::cg0*(ut[c-1] + u[c-1]*dti) + W::cg1*(ut[c] + u[c]*dti) + W::cg2*(ut[c+1] + u[c+1]*dti) + W::cg3*(ut[c+2] + u[c+2]*dti)) * dxi;
                if (dim3)
                    p[cp] += (W::cg0*(vt[c-jj] + v[c-jj]*dti) + W::cg1*(vt[c] + v[c]*dti) + W::cg2*(vt[c+jj] + v[c+jj]*dti) + W::cg3*(vt[c+2*jj] + v[c+2*jj]*dti)) * dyi;
                p[cp] += (W::cg0*(wt[c-kk] + w[c-kk]*dti) + W::cg1*(wt[c] + w[c]*dti) + W::cg2*(wt[c+kk] + w[c+kk]*dti) + W::cg3*(wt[c+2*kk] + w[c+2*kk]*dti)) * dzi4[k+g.kgc];
            }
}

// Pres_4::solve matrix fill (:358-470) + hdma (:574-730), one column at a time
template<class TF>
void pres4_spectral_solve(const mhh_grid& g, TF* p, const Pres4Mat<TF>& M)
{
    const int ib = g.itot, jb = g.jtot, kmax = g.kmax;
    const size_t kk = (size_t)ib*jb;
    const int n = kmax+4;
    std::vector<TF> m1(n), m2(n), m3(n), m4(n), m5(n), m6(n), m7(n), q(n);
    for (int j=0; j<jb; ++j)
        for (int i=0; i<ib; ++i)
        {
            m1[0]=0; m2[0]=0; m3[0]=0; m4[0]=1; m5[0]=0;  m6[0]=0; m7[0]=-1; q[0]=0;
            m1[1]=0; m2[1]=0; m3[1]=0; m4[1]=1; m5[1]=-1; m6[1]=0; m7[1]=0;  q[1]=0;
            for (int k=0; k<kmax; ++k)
            {
                m1[k+2]=M.m1[k]; m2[k+2]=M.m2[k]; m3[k+2]=M.m3[k];
                m4[k+2]=M.m4[k] + M.bmati[i] + M.bmatj[j];
                m5[k+2]=M.m5[k]; m6[k+2]=M.m6[k]; m7[k+2]=M.m7[k];
                q[k+2] = p[i + (size_t)j*ib + k*kk];
            }
            const int t = kmax+2;
            if (i == 0 && j == 0)
            {
                m1[t]=TF(0.);   m2[t]=TF(-1/3.); m3[t]=TF(2.); m4[t]=TF(1.);
                m1[t+1]=TF(-2.); m2[t+1]=TF(9.);  m3[t+1]=TF(0.); m4[t+1]=TF(1.);
            }
            else
            {
                m1[t]=TF(0.);   m2[t]=TF(0.); m3[t]=TF(-1.); m4[t]=TF(1.);
                m1[t+1]=TF(-1.); m2[t+1]=TF(0.); m3[t+1]=TF(0.);  m4[t+1]=TF(1.);
            }
            m5[t]=0; m6[t]=0; m7[t]=0; q[t]=0; m5[t+1]=0; m6[t+1]=0; m7[t+1]=0; q[t+1]=0;

            // LU factorisation without pivoting
            int k = 0;
            m1[k]=1; m2[k]=1; m3[k]=TF(1.)/m4[k]; m4[k]=1; m5[k]=m5[k]*m3[k]; m6[k]=m6[k]*m3[k]; m7[k]=m7[k]*m3[k];
            k = 1;
            m1[k]=1; m2[k]=1; m3[k]=m3[k]/m4[k-1];
            m4[k]=m4[k]-m3[k]*m5[k-1]; m5[k]=m5[k]-m3[k]*m6[k-1]; m6[k]=m6[k]-m3[k]*m7[k-1];
            k = 2;
            m1[k]=1; m2[k]=m2[k]/m4[k-2];
            m3[k]=( m3[k] - m2[k]*m5[k-2] ) / m4[k-1];
            m4[k]=m4[k] - m3[k]*m5[k-1] - m2[k]*m6[k-2];
            m5[k]=m5[k] - m3[k]*m6[k-1] - m2[k]*m7[k-2];
            m6[k]=m6[k] - m3[k]*m7[k-1];
            for (k=3; k<kmax+4; ++k)
            {
                if (k == kmax+2) { /* the reference first sets m7[kmax+1] = 1 */ m7[kmax+1] = TF(1.); }
                m1[k]=( m1[k] ) / m4[k-3];
                m2[k]=( m2[k] - m1[k]*m5[k-3]) / m4[k-2];
                m3[k]=( m3[k] - m2[k]*m5[k-2] - m1[k]*m6[k-3]) / m4[k-1];
                m4[k]=  m4[k] - m3[k]*m5[k-1] - m2[k]*m6[k-2] - m1[k]*m7[k-3];
                if (k < kmax+3) m5[k]=  m5[k] - m3[k]*m6[k-1] - m2[k]*m7[k-2];
                if (k < kmax+2) m6[k]=  m6[k] - m3[k]*m7[k-1];
                if (k == kmax+2) { m6[k]=TF(1.); m7[k]=TF(1.); }
                if (k == kmax+3) { m5[k]=1.; m6[k]=1.; m7[k]=1.; }
            }
            // forward substitution L y = q
            q[0] = q[0]*m3[0];
            q[1] = q[1] - q[0]*m3[1];
            q[2] = q[2] - q[1]*m3[2] - q[0]*m2[2];
            for (k=3; k<kmax+4; ++k) q[k] = q[k] - q[k-1]*m3[k] - q[k-2]*m2[k] - q[k-3]*m1[k];
            // backward substitution U x = y
            k = kmax+3;
            q[k  ] =   q[k  ] / m4[k  ];
            q[k-1] = ( q[k-1] - q[k  ]*m5[k-1] ) / m4[k-1];
            q[k-2] = ( q[k-2] - q[k-1]*m5[k-2] - q[k]*m6[k-2] ) / m4[k-2];
            for (k=kmax; k>=0; --k) q[k] = ( q[k] - q[k+1]*m5[k] - q[k+2]*m6[k] - q[k+3]*m7[k] ) / m4[k];
            for (k=0; k<kmax; ++k) p[i + (size_t)j*ib + k*kk] = q[k+2];
        }
}

// unpack + mirrored ghosts + cyclic (:481-528)
template<class TF>
void pres4_unpack(const mhh_grid& g, TF* p, const TF* packed)
{
    const int jj = g.icells, kk = g.ijcells;
    for (int k=0; k<g.kmax; ++k)
        for (int j=0; j<g.jmax; ++j)
            for (int i=0; i<g.imax; ++i)
                p[i+g.igc + (j+g.jgc)*jj + (size_t)(k+g.kgc)*kk] = packed[i + j*g.imax + (size_t)k*g.imax*g.jmax];
    FOR_INTERIOR_PLANE(g)
    {
        const int b = i + j*jj + g.kstart*kk;
        p[b-kk] = p[b]; p[b-2*kk] = p[b+kk];
        const int t = i + j*jj + (g.kend-1)*kk;
        p[t+kk] = p[t]; p[t+2*kk] = p[t-kk];
    }
    cyclic<TF>(g, p, MHH_EDGE_BOTH);
}

// Pres_4::output (:533-571)
template<class TF>
void pres4_output(const mhh_grid& g, TF* ut, TF* vt, TF* wt, const TF* p)
{
    const int jj = g.icells, kk = g.ijcells;
    const bool dim3 = (g.jtot != 1);
    const TF dxi = 1./TF(g.dx), dyi = 1./TF(g.dy);
    const TF* dzhi4 = P<TF>(g.dzhi4);
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            ut[c] -= cg4(p[c-2], p[c-1], p[c], p[c+1]) * dxi;
            if (dim3) vt[c] -= cg4(p[c-2*jj], p[c-jj], p[c], p[c+jj]) * dyi;
            if (k > g.kstart) wt[c] -= cg4(p[c-2*kk], p[c-kk], p[c], p[c+kk]) * dzhi4[k];
        }
}

// Pres_4::calc_divergence (:733-767)
template<class TF>
double pres4_divergence(const mhh_grid& g, const TF* u, const TF* v, const TF* w)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = 1./TF(g.dx), dyi = 1./TF(g.dy);
    const TF* dzi4 = P<TF>(g.dzi4);
    TF divmax = 0;
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const TF div = cg4(u[c-1], u[c], u[c+1], u[c+2]) * dxi
                         + cg4(v[c-jj], v[c], v[c+jj], v[c+2*jj]) * dyi
                         + cg4(w[c-kk], w[c], w[c+kk], w[c+2*kk]) * dzi4[k];
            divmax = std::max(divmax, std::abs(div));
        }
    return static_cast<double>(divmax);
}

// ---------------------------------------------------------------------------------------------
// Timeloop rk3 / rk4 (src/timeloop.cxx:250-334): a += cB*dt*at ; at *= cA(next substep)
// ---------------------------------------------------------------------------------------------
template<class TF>
void rk_substep(const mhh_grid& g, int order, int substep, TF dt, TF* a, TF* at)
{
    const int jj = g.icells, kk = g.ijcells;
    TF cA, cB;
    if (order == 3)
    {
        const TF A[] = {0., -5./9., -153./128.};
        const TF B[] = {1./3., 15./16., 8./15.};
        cA = A[(substep+1)%3]; cB = B[substep];
    }
    else
    {
        const TF A[] = { 0., -567301805773./1357537059087., -2404267990393./2016746695238., -3550918686646./2091501179385., -1275806237668./842570457699.};
        const TF B[] = { 1432997174477./9575080441755., 5161836677717./13612068292357., 1720146321549./2090206949498., 3134564353537./4481467310338., 2277821191437./14882151754819.};
        cA = A[(substep+1)%5]; cB = B[substep];
    }
    for (int k=g.kstart; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            a[c] = a[c] + cB*dt*at[c];
        }
    // the step that wraps to substep 0 zeroes the whole array incl. ghost cells, otherwise interior only
    const int nsub = (order == 3) ? 3 : 5;
    if ((substep+1) % nsub == 0)
        for (long long n=0; n<g.ncells; ++n) at[n] = TF(0.);
    else
        for (int k=g.kstart; k<g.kend; ++k)
            FOR_INTERIOR_PLANE(g)
            {
                const int c = i + j*jj + k*kk;
                at[c] = cA*at[c];
            }
}

// ---------------------------------------------------------------------------------------------
// Boundary::set_ghost_cells / set_ghost_cells_w (src/boundary.cxx:686-907, 919-1007). bc: 0 Dirichlet, 1 Neumann/flux
// ---------------------------------------------------------------------------------------------
template<class TF>
void ghost_cells(const mhh_grid& g, int order, TF* a, int bcbot, int bctop, const TF* abot, const TF* agradbot, const TF* atop, const TF* agradtop)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF* dzh = P<TF>(g.dzh); const TF* z = P<TF>(g.z);
    const int ks = g.kstart, ke = g.kend;
    for (int j=0; j<g.jcells; ++j)
        for (int i=0; i<g.icells; ++i)
        {
            const int ij = i + j*jj;
            const int b = ij + ks*kk, t = ij + (ke-1)*kk;
            if (order == 2)
            {
                if (bcbot == 0) a[b-kk] = TF(2.)*abot[ij] - a[b]; else a[b-kk] = -agradbot[ij]*dzh[ks] + a[b];
                if (bctop == 0) a[t+kk] = TF(2.)*atop[ij] - a[t]; else a[t+kk] = agradtop[ij]*dzh[ke] + a[t];
            }
            else
            {
                if (bcbot == 0) { a[b-kk] = TF(8./3.)*abot[ij] - TF(2.)*a[b] + TF(1./3.)*a[b+kk]; a[b-2*kk] = TF(8.)*abot[ij] - TF(9.)*a[b] + TF(2.)*a[b+kk]; }
                else
                {
                    const TF gr = ( - W4<TF>::cg0*(z[ks+1]-z[ks-2]) - W4<TF>::cg1*(z[ks]-z[ks-1]) );      // grad4, finite_difference.h:128
                    a[b-kk] = TF(-1.)*gr*agradbot[ij] + a[b]; a[b-2*kk] = TF(-3.)*gr*agradbot[ij] + a[b+kk];
                }
                if (bctop == 0) { a[t+kk] = TF(8./3.)*atop[ij] - TF(2.)*a[t] + TF(1./3.)*a[t-kk]; a[t+2*kk] = TF(8.)*atop[ij] - TF(9.)*a[t] + TF(2.)*a[t-kk]; }
                else
                {
                    const TF gr = ( - W4<TF>::cg0*(z[ke+1]-z[ke-2]) - W4<TF>::cg1*(z[ke]-z[ke-1]) );
                    a[t+kk] = TF(1.)*gr*agradtop[ij] + a[t]; a[t+2*kk] = TF(3.)*gr*agradtop[ij] + a[t-kk];
                }
            }
        }
}
// 4th-order w ghost cells: type 0 = Normal (extrapolation), 1 = Conservation (mirror)
template<class TF>
void ghost_cells_w(const mhh_grid& g, TF* w, int type)
{
    const int jj = g.icells, kk = g.ijcells;
    for (int j=0; j<g.jcells; ++j)
        for (int i=0; i<g.icells; ++i)
        {
            const int b = i + j*jj + g.kstart*kk, t = i + j*jj + g.kend*kk;
            if (type == 1) { w[b-kk] = -w[b+kk]; w[b-2*kk] = -w[b+2*kk]; w[t+kk] = -w[t-kk]; w[t+2*kk] = -w[t-2*kk]; }
            else { w[b-kk] = TF(-6.)*w[b+kk] + TF(4.)*w[b+2*kk] - w[b+3*kk]; w[t+kk] = TF(-6.)*w[t-kk] + TF(4.)*w[t-2*kk] - w[t-3*kk]; }
        }
}

} // namespace

// =================================================================================================
// C entry points (dtype dispatch)
// =================================================================================================
#define DISPATCH(g, CALL_F64, CALL_F32) do { if ((g)->dtype == MHH_F64) { CALL_F64; } else { CALL_F32; } } while (0)
#define D(x) P<double>(x)
#define F(x) P<float>(x)

ORC_API void orc_boundary_cyclic(const mhh_grid* g, void* a, int edge)
{ DISPATCH(g, cyclic<double>(*g, D(a), edge), cyclic<float>(*g, F(a), edge)); }
ORC_API void orc_boundary_cyclic_2d(const mhh_grid* g, void* a)
{ DISPATCH(g, cyclic_2d<double>(*g, D(a)), cyclic_2d<float>(*g, F(a))); }

namespace {
// -------------------------------------------------------------------------------------------------------
// advec_2i4 (src/advec_2i4.cxx:101-640): second-order fluxes, advected quantity interpolated with (-1, 9, 9, -1)/16,
// two-point interpolation on the vertical faces next to a wall. Restated with the face-order table of the 2i5 scheme
// capped at 4; comp 0..2 = u, v, w (staggering offset o), 3 = scalar. One increment per cell.
// -------------------------------------------------------------------------------------------------------
template<class TF>
void advec24_any(const mhh_grid& g, int comp, TF* t, const TF* f, const TF* u, const TF* v, const TF* w,
                 const TF* rhoref, const TF* rhorefh)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1./TF(g.dx)), dyi = TF(1./TF(g.dy));          // Grid_data::dxi (src/grid.cxx), not TF(1.)/dx
    const TF* dzi = P<TF>(g.dzi); const TF* dzhi = P<TF>(g.dzhi);
    const int o = (comp==0) ? -1 : (comp==1) ? -jj : (comp==2) ? -kk : 0;
    const int k0 = (comp==2) ? g.kstart+1 : g.kstart;
    auto cap = [](int order) { return order > 4 ? 4 : order; };
    for (int k=k0; k<g.kend; ++k)
    {
        int ot, ob; TF rt, rb, rc, dz;
        if (comp == 2) { ot = cap(face_order_w(g, k));   ob = cap(face_order_w(g, k-1)); rt = rhoref[k];    rb = rhoref[k-1]; rc = rhorefh[k]; dz = dzhi[k]; }
        else           { ot = cap(face_order_c(g, k+1)); ob = cap(face_order_c(g, k));   rt = rhorefh[k+1]; rb = rhorefh[k];  rc = rhoref[k];  dz = dzi[k];  }
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            TF ue, uw, vn, vs, wt, wb;
            if (comp == 3) { ue = u[c+1]; uw = u[c]; vn = v[c+jj]; vs = v[c]; wt = w[c+kk]; wb = w[c]; }
            else
            {
                ue = i2(u[c+1+o], u[c+1]);  uw = i2(u[c+o], u[c]);
                vn = i2(v[c+jj+o], v[c+jj]); vs = i2(v[c+o], v[c]);
                wt = i2(w[c+kk+o], w[c+kk]); wb = i2(w[c+o], w[c]);
            }
            const TF fx = ue * i4c(f[c-1], f[c], f[c+1], f[c+2]) - uw * i4c(f[c-2], f[c-1], f[c], f[c+1]);
            const TF fy = vn * i4c(f[c-jj], f[c], f[c+jj], f[c+2*jj]) - vs * i4c(f[c-2*jj], f[c-jj], f[c], f[c+jj]);
            const TF top = (ot == 0) ? TF(0) : rt * wt * ((ot == 2) ? i2(f[c], f[c+kk]) : i4c(f[c-kk], f[c], f[c+kk], f[c+2*kk]));
            TF fz;
            if (ob == 0)      fz = top;
            else
            {
                const TF Ib = (ob == 2) ? i2(f[c-kk], f[c]) : i4c(f[c-2*kk], f[c-kk], f[c], f[c+kk]);
                if (ot == 0)  fz = -rb * wb * Ib;
                else          fz = top - rb * wb * Ib;
            }
            t[c] += - fx * dxi - fy * dyi - fz / rc * dz;
        }
    }
}

}

namespace {
// advec_2i62 (src/advec_2i62.cxx:105-310): 6th-order interpolation of the advected quantity on the x and y faces,
// two-point interpolation on the z faces of every level; comp 0..2 = u, v, w, 3 = scalar.
template<class TF>
void advec262_any(const mhh_grid& g, int comp, TF* t, const TF* f, const TF* u, const TF* v, const TF* w, const TF* rhoref, const TF* rhorefh)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzi = P<TF>(g.dzi); const TF* dzhi = P<TF>(g.dzhi);
    const int o = (comp==0) ? -1 : (comp==1) ? -jj : (comp==2) ? -kk : 0;
    for (int k=(comp==2 ? g.kstart+1 : g.kstart); k<g.kend; ++k)
    {
        const TF rt = (comp==2) ? rhoref[k] : rhorefh[k+1], rb = (comp==2) ? rhoref[k-1] : rhorefh[k];
        const TF rc = (comp==2) ? rhorefh[k] : rhoref[k], dz = (comp==2) ? dzhi[k] : dzi[k];
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            TF ue, uw, vn, vs, wt, wb;
            if (comp == 3) { ue = u[c+1]; uw = u[c]; vn = v[c+jj]; vs = v[c]; wt = w[c+kk]; wb = w[c]; }
            else
            {
                ue = i2(u[c+1+o], u[c+1]);  uw = i2(u[c+o], u[c]);
                vn = i2(v[c+jj+o], v[c+jj]); vs = i2(v[c+o], v[c]);
                wt = i2(w[c+kk+o], w[c+kk]); wb = i2(w[c+o], w[c]);
            }
            t[c] += - ( ue * i6(f[c-2], f[c-1], f[c], f[c+1], f[c+2], f[c+3]) - uw * i6(f[c-3], f[c-2], f[c-1], f[c], f[c+1], f[c+2]) ) * dxi
                    - ( vn * i6(f[c-2*jj], f[c-jj], f[c], f[c+jj], f[c+2*jj], f[c+3*jj]) - vs * i6(f[c-3*jj], f[c-2*jj], f[c-jj], f[c], f[c+jj], f[c+2*jj]) ) * dyi
                    - ( rt * wt * i2(f[c], f[c+kk]) - rb * wb * i2(f[c-kk], f[c]) ) / rc * dz;
        }
    }
}
}

namespace {
// advec_4m (src/advec_4m.cxx:90-478). For each direction: four face products P_m = V_m * mean(f[lo_m], f[hi_m]) with
// (lo, hi) = (-3,0), (-1,0), (0,1), (0,3) cells and V_m the advecting velocity at face m-1 (4th-order interpolated along
// the staggering direction of the equation; taken as is for scalars); the term is -(1/24)(P3-P0) + (27/24)(P2-P1) times
// the metric. Walls (u, v, scalars): the bottom row replaces P0 by -V(+1) * mean(f[-1], f[+2]), the top row P3 by
// -V(0) * mean(f[-2], f[+1]).
template<class TF>
void advec4m_any(const mhh_grid& g, int comp, TF* t, const TF* f, const TF* u, const TF* v, const TF* w)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1./TF(g.dx)), dyi = TF(1./TF(g.dy));
    const TF* dzi4 = P<TF>(g.dzi4); const TF* dzhi4 = P<TF>(g.dzhi4);
    const int se = (comp==0) ? 1 : (comp==1) ? jj : (comp==2) ? kk : 0;     // staggering stride of the equation
    const int lo[4] = {-3, -1, 0, 0}, hi[4] = {0, 0, 1, 3};
    auto vel = [&](const TF* a, int at) -> TF      // advecting velocity component a at cell offset `at`
    {
        if (comp == 3) return a[at];
        return TF(-1./16.)*(a[at-2*se] + a[at+se]) + TF(9./16.)*(a[at-se] + a[at]);      // interp4c: the paired form
    };
    auto grad = [](TF a, TF b, TF c, TF d) -> TF { return - TF(1./24.)*(d-a) - TF(-27./24.)*(c-b); };
    for (int k=(comp==2 ? g.kstart+1 : g.kstart); k<g.kend; ++k)
    {
        const bool bot = (comp != 2) && (k == g.kstart), top = (comp != 2) && (k == g.kend-1);
        const TF dz = (comp==2) ? dzhi4[k] : dzi4[k];
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            TF px[4], py[4], pz[4];
            for (int m=0; m<4; ++m)
            {
                px[m] = vel(u, c + (m-1))    * (TF(0.5)*(f[c+lo[m]]    + f[c+hi[m]]));
                py[m] = vel(v, c + (m-1)*jj) * (TF(0.5)*(f[c+lo[m]*jj] + f[c+hi[m]*jj]));
                pz[m] = vel(w, c + (m-1)*kk) * (TF(0.5)*(f[c+lo[m]*kk] + f[c+hi[m]*kk]));
            }
            if (bot) pz[0] = -vel(w, c + kk) * (TF(0.5)*(f[c-kk]   + f[c+2*kk]));
            if (top) pz[3] = -vel(w, c)      * (TF(0.5)*(f[c-2*kk] + f[c+kk]));
            t[c] += - grad(px[0], px[1], px[2], px[3]) * dxi - grad(py[0], py[1], py[2], py[3]) * dyi - grad(pz[0], pz[1], pz[2], pz[3]) * dz;
        }
    }
}
}

template<class TF>
static void advec_mom_t(const mhh_grid& g, int scheme, int comp, void* t, const void* u, const void* v, const void* w, const void* r, const void* rh)
{
    if (scheme == MHH_ADVEC_2)        advec2_mom<TF>(g, comp, P<TF>(t), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh));
    else if (scheme == MHH_ADVEC_2I5) advec25_mom<TF>(g, comp, P<TF>(t), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh));
    else if (scheme == MHH_ADVEC_2I53) advec25_mom<TF>(g, comp, P<TF>(t), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh), 4);
    else if (scheme == MHH_ADVEC_4M) advec4m_any<TF>(g, comp, P<TF>(t), comp==0 ? P<TF>(u) : comp==1 ? P<TF>(v) : P<TF>(w), P<TF>(u), P<TF>(v), P<TF>(w));
    else if (scheme == MHH_ADVEC_2I62) advec262_any<TF>(g, comp, P<TF>(t), comp==0 ? P<TF>(u) : comp==1 ? P<TF>(v) : P<TF>(w), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh));
    else if (scheme == MHH_ADVEC_2I4) advec24_any<TF>(g, comp, P<TF>(t), comp==0 ? P<TF>(u) : comp==1 ? P<TF>(v) : P<TF>(w), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh));
    else                              advec4_mom<TF>(g, comp, P<TF>(t), P<TF>(u), P<TF>(v), P<TF>(w));
}
ORC_API void orc_advec_u(const mhh_grid* g, int scheme, void* t, const void* u, const void* v, const void* w, const void* r, const void* rh)
{ DISPATCH(g, advec_mom_t<double>(*g, scheme, 0, t,u,v,w,r,rh), advec_mom_t<float>(*g, scheme, 0, t,u,v,w,r,rh)); }
ORC_API void orc_advec_v(const mhh_grid* g, int scheme, void* t, const void* u, const void* v, const void* w, const void* r, const void* rh)
{ DISPATCH(g, advec_mom_t<double>(*g, scheme, 1, t,u,v,w,r,rh), advec_mom_t<float>(*g, scheme, 1, t,u,v,w,r,rh)); }
ORC_API void orc_advec_w(const mhh_grid* g, int scheme, void* t, const void* u, const void* v, const void* w, const void* r, const void* rh)
{ DISPATCH(g, advec_mom_t<double>(*g, scheme, 2, t,u,v,w,r,rh), advec_mom_t<float>(*g, scheme, 2, t,u,v,w,r,rh)); }

template<class TF>
static void advec_s_t(const mhh_grid& g, int scheme, void* t, const void* s, const void* u, const void* v, const void* w, const void* r, const void* rh)
{
    if (scheme == MHH_ADVEC_2)        advec2_s<TF>(g, P<TF>(t), P<TF>(s), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh));
    else if (scheme == MHH_ADVEC_2I5) advec25_s<TF>(g, P<TF>(t), P<TF>(s), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh));
    else if (scheme == MHH_ADVEC_2I53) advec25_s<TF>(g, P<TF>(t), P<TF>(s), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh), 4);
    else if (scheme == MHH_ADVEC_4M) advec4m_any<TF>(g, 3, P<TF>(t), P<TF>(s), P<TF>(u), P<TF>(v), P<TF>(w));
    else if (scheme == MHH_ADVEC_2I62) advec262_any<TF>(g, 3, P<TF>(t), P<TF>(s), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh));
    else if (scheme == MHH_ADVEC_2I4) advec24_any<TF>(g, 3, P<TF>(t), P<TF>(s), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(r), P<TF>(rh));
    else                              advec4_s<TF>(g, P<TF>(t), P<TF>(s), P<TF>(u), P<TF>(v), P<TF>(w));
}
ORC_API void orc_advec_s(const mhh_grid* g, int scheme, void* t, const void* s, const void* u, const void* v, const void* w, const void* r, const void* rh)
{ DISPATCH(g, advec_s_t<double>(*g, scheme, t,s,u,v,w,r,rh), advec_s_t<float>(*g, scheme, t,s,u,v,w,r,rh)); }

// -------------------------------------------------------------------------------------------------------
// Flux-limited scalar advection: Advec_monotonic::advec_s_lim (include/advec_monotonic.h:79-180) with the Koren
// (1993) limiter flux_lim / flux_lim_bot / flux_lim_top (:10-77). Restated as one sweep with a per-level choice of
// the vertical face forms; each face flux is vel * (upwind value + phi/2 * upwind difference).
// -------------------------------------------------------------------------------------------------------
template<class TF>
static TF koren(TF vel, TF a, TF b, TF c, TF d, bool wall_below, bool wall_above)
{
    // stencil a b | c d around the face; flow from b to c when vel >= 0
    const TF eps = std::numeric_limits<TF>::epsilon();
    TF up, upup, down;
    if (vel >= TF(0.)) { if (wall_below) return vel*b; up = b; upup = a; down = c; }
    else               { if (wall_above) return vel*c; up = c; upup = d; down = b; }
    const TF diff = up - upup;
    const TF denom = TF(std::copysign(1., (double)diff)) * std::max(std::abs(diff), eps);
    const TF two_r = TF(2.) * (down - up) / denom;
    const TF phi = std::max(TF(0.), std::min(two_r, std::min(TF(1./3.)*(TF(1.)+two_r), TF(2.))));
    return vel*(up + TF(0.5)*phi*(up - upup));
}
template<class TF>
static void advec_s_lim(const mhh_grid& g, TF* t, const TF* s, const TF* u, const TF* v, const TF* w, const TF* rhoref, const TF* rhorefh)
{
    const int jj = g.icells, kk = g.ijcells;
    const TF dxi = TF(1.)/TF(g.dx), dyi = TF(1.)/TF(g.dy);
    const TF* dzi = P<TF>(g.dzi);
    for (int k=g.kstart; k<g.kend; ++k)
    {
        const bool no_bot = (k == g.kstart), no_top = (k == g.kend-1);          // wall faces carry no flux
        const bool bot1 = (k == g.kstart+1), top1 = (k == g.kend-2);            // faces next to a wall: one-sided
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const TF fe = koren<TF>(u[c+1],  s[c-1],    s[c],    s[c+1],  s[c+2],    false, false);
            const TF fw = koren<TF>(u[c],    s[c-2],    s[c-1],  s[c],    s[c+1],    false, false);
            const TF fn = koren<TF>(v[c+jj], s[c-jj],   s[c],    s[c+jj], s[c+2*jj], false, false);
            const TF fs = koren<TF>(v[c],    s[c-2*jj], s[c-jj], s[c],    s[c+jj],   false, false);
            TF acc = - ( fe - fw ) * dxi - ( fn - fs ) * dyi;
            if (no_bot)
                acc = acc - ( rhorefh[k+1] * koren<TF>(w[c+kk], s[c-kk], s[c], s[c+kk], s[c+2*kk], true, false) ) / rhoref[k] * dzi[k];
            else if (no_top)
                acc = acc - ( - rhorefh[k] * koren<TF>(w[c], s[c-2*kk], s[c-kk], s[c], s[c+kk], false, true) ) / rhoref[k] * dzi[k];
            else
                acc = acc - ( rhorefh[k+1] * koren<TF>(w[c+kk], s[c-kk],   s[c],    s[c+kk], s[c+2*kk], false, top1)
                            - rhorefh[k  ] * koren<TF>(w[c],    s[c-2*kk], s[c-kk], s[c],    s[c+kk],   bot1, false) ) / rhoref[k] * dzi[k];
            t[c] += acc;
        }
    }
}
ORC_API void orc_advec_s_lim(const mhh_grid* g, void* t, const void* s, const void* u, const void* v, const void* w, const void* r, const void* rh)
{ DISPATCH(g, advec_s_lim<double>(*g, D(t), D(s), D(u), D(v), D(w), D(r), D(rh)), advec_s_lim<float>(*g, F(t), F(s), F(u), F(v), F(w), F(r), F(rh))); }

ORC_API double orc_advec_cfl(const mhh_grid* g, int scheme, const void* u, const void* v, const void* w, double dt)
{
    if (g->dtype == MHH_F64) return advec_cfl<double>(*g, scheme, D(u), D(v), D(w), dt);
    return advec_cfl<float>(*g, scheme, F(u), F(v), F(w), dt);
}

ORC_API void orc_diff_c(const mhh_grid* g, int order, void* t, const void* a, double visc)
{
    if (order == 2) DISPATCH(g, diff2<double>(*g, false, D(t), D(a), visc), diff2<float>(*g, false, F(t), F(a), (float)visc));
    else            DISPATCH(g, diff4<double>(*g, false, D(t), D(a), visc), diff4<float>(*g, false, F(t), F(a), (float)visc));
}
ORC_API void orc_diff_w(const mhh_grid* g, int order, void* t, const void* a, double visc)
{
    if (order == 2) DISPATCH(g, diff2<double>(*g, true, D(t), D(a), visc), diff2<float>(*g, true, F(t), F(a), (float)visc));
    else            DISPATCH(g, diff4<double>(*g, true, D(t), D(a), visc), diff4<float>(*g, true, F(t), F(a), (float)visc));
}

ORC_API void orc_smag2_strain2(const mhh_grid* g, int sm, void* s2, const void* u, const void* v, const void* w, const void* dudz, const void* dvdz)
{ DISPATCH(g, smag_strain2<double>(*g, sm, D(s2), D(u), D(v), D(w), D(dudz), D(dvdz)), smag_strain2<float>(*g, sm, F(s2), F(u), F(v), F(w), F(dudz), F(dvdz))); }
ORC_API void orc_smag2_evisc(const mhh_grid* g, int sm, void* ev, const void* N2, const void* bgradbot, const void* z0m, double cs, double tPr)
{ DISPATCH(g, smag_evisc<double>(*g, sm, D(ev), D(N2), D(bgradbot), D(z0m), cs, tPr), smag_evisc<float>(*g, sm, F(ev), F(N2), F(bgradbot), F(z0m), (float)cs, (float)tPr)); }
ORC_API void orc_smag2_evisc_neutral(const mhh_grid* g, int sm, void* ev, const void* u, const void* v, const void* z0m, double cs, double visc)
{ DISPATCH(g, smag_evisc_neutral<double>(*g, sm, D(ev), D(u), D(v), D(z0m), cs, visc), smag_evisc_neutral<float>(*g, sm, F(ev), F(u), F(v), F(z0m), (float)cs, (float)visc)); }
ORC_API void orc_smag2_diff_u(const mhh_grid* g, int sm, void* t, const void* u, const void* v, const void* w, const void* ev, const void* fb, const void* ft, const void* r, const void* rh, double visc)
{ DISPATCH(g, smag_diff_uv<double>(*g, 0, sm, D(t), D(u), D(v), D(w), D(ev), D(fb), D(ft), D(r), D(rh), visc), smag_diff_uv<float>(*g, 0, sm, F(t), F(u), F(v), F(w), F(ev), F(fb), F(ft), F(r), F(rh), (float)visc)); }
ORC_API void orc_smag2_diff_v(const mhh_grid* g, int sm, void* t, const void* u, const void* v, const void* w, const void* ev, const void* fb, const void* ft, const void* r, const void* rh, double visc)
{ DISPATCH(g, smag_diff_uv<double>(*g, 1, sm, D(t), D(u), D(v), D(w), D(ev), D(fb), D(ft), D(r), D(rh), visc), smag_diff_uv<float>(*g, 1, sm, F(t), F(u), F(v), F(w), F(ev), F(fb), F(ft), F(r), F(rh), (float)visc)); }
ORC_API void orc_smag2_diff_w(const mhh_grid* g, void* t, const void* u, const void* v, const void* w, const void* ev, const void* r, const void* rh, double visc)
{ DISPATCH(g, smag_diff_w<double>(*g, D(t), D(u), D(v), D(w), D(ev), D(r), D(rh), visc), smag_diff_w<float>(*g, F(t), F(u), F(v), F(w), F(ev), F(r), F(rh), (float)visc)); }
ORC_API void orc_smag2_diff_c(const mhh_grid* g, int sm, void* t, const void* a, const void* ev, const void* fb, const void* ft, const void* r, const void* rh, double tPr, double visc)
{ DISPATCH(g, smag_diff_c<double>(*g, sm, D(t), D(a), D(ev), D(fb), D(ft), D(r), D(rh), tPr, visc), smag_diff_c<float>(*g, sm, F(t), F(a), F(ev), F(fb), F(ft), F(r), F(rh), (float)tPr, (float)visc)); }
ORC_API double orc_smag2_dnmul(const mhh_grid* g, const void* ev, double tPr)
{
    if (g->dtype == MHH_F64) return smag_dnmul<double>(*g, D(ev), tPr);
    return smag_dnmul<float>(*g, F(ev), (float)tPr);
}
// Thermo_dry buoyancy tendency (src/thermo_dry.cxx:165-197; that TU needs netcdf.h, so this line is restated, not linked)
template<class TF>
static void buoyancy_tend(const mhh_grid& g, int order, TF* wt, const TF* th, const TF* threfh, TF grav)
{
    const int jj = g.icells, kk = g.ijcells;
    for (int k=g.kstart+1; k<g.kend; ++k)
        FOR_INTERIOR_PLANE(g)
        {
            const int c = i + j*jj + k*kk;
            const TF thh = (order == 4) ? TF(-1./16.)*(th[c-2*kk] + th[c+kk]) + TF(9./16.)*(th[c-kk] + th[c])
                                        : TF(0.5)*(th[c-kk] + th[c]);
            wt[c] += grav/threfh[k] * (thh - threfh[k]);
        }
}
ORC_API void orc_buoyancy_tend(const mhh_grid* g, int order, void* wt, const void* th, const void* threfh, double grav)
{ DISPATCH(g, buoyancy_tend<double>(*g, order, D(wt), D(th), D(threfh), grav), buoyancy_tend<float>(*g, order, F(wt), F(th), F(threfh), (float)grav)); }

ORC_API void orc_calc_N2(const mhh_grid* g, void* N2, const void* th, const void* thref, double grav)
{ DISPATCH(g, calc_N2<double>(*g, D(N2), D(th), D(thref), grav), calc_N2<float>(*g, F(N2), F(th), F(thref), (float)grav)); }

// FFT stages on packed data (for pinning against numpy.fft)
ORC_API void orc_fft_forward(const mhh_grid* g, void* data)
{ DISPATCH(g, fft_forward<double>(*g, D(data)), fft_forward<float>(*g, F(data))); }
ORC_API void orc_fft_backward(const mhh_grid* g, void* data)
{ DISPATCH(g, fft_backward<double>(*g, D(data)), fft_backward<float>(*g, F(data))); }

template<class TF>
static void pres_input_t(const mhh_grid& g, int order, void* p, const void* u, const void* v, const void* w, void* ut, void* vt, void* wt, const void* r, const void* rh, double dt)
{
    if (order == 2) pres2_input<TF>(g, P<TF>(p), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(ut), P<TF>(vt), P<TF>(wt), P<TF>(r), P<TF>(rh), TF(dt));
    else            pres4_input<TF>(g, P<TF>(p), P<TF>(u), P<TF>(v), P<TF>(w), P<TF>(ut), P<TF>(vt), P<TF>(wt), TF(dt));
}
ORC_API void orc_pres_input(const mhh_grid* g, int order, void* p_packed, const void* u, const void* v, const void* w, void* ut, void* vt, void* wt, const void* r, const void* rh, double dt)
{ DISPATCH(g, pres_input_t<double>(*g, order, p_packed,u,v,w,ut,vt,wt,r,rh,dt), pres_input_t<float>(*g, order, p_packed,u,v,w,ut,vt,wt,r,rh,dt)); }

// spectral stage only (between forward and backward FFT); used to pin the tridiagonal / heptadiagonal solves
template<class TF>
static void pres_spectral_t(const mhh_grid& g, int order, void* p, const void* r, const void* rh)
{
    if (order == 2)
    {
        std::vector<TF> bi, bj, a, c;
        pres2_set_values<TF>(g, P<TF>(rh), bi, bj, a, c);
        pres2_spectral_solve<TF>(g, P<TF>(p), P<TF>(r), bi, bj, a, c);
    }
    else
    {
        Pres4Mat<TF> M; pres4_set_values<TF>(g, M);
        pres4_spectral_solve<TF>(g, P<TF>(p), M);
    }
}
ORC_API void orc_pres_spectral_solve(const mhh_grid* g, int order, void* p_packed, const void* r, const void* rh)
{ DISPATCH(g, pres_spectral_t<double>(*g, order, p_packed, r, rh), pres_spectral_t<float>(*g, order, p_packed, r, rh)); }

// Pres::solve: forward FFT, spectral solve, backward FFT, unpack into ghosted p
template<class TF>
static void pres_solve_t(const mhh_grid& g, int order, void* p, void* packed, const void* r, const void* rh)
{
    fft_forward<TF>(g, P<TF>(packed));
    pres_spectral_t<TF>(g, order, packed, r, rh);
    fft_backward<TF>(g, P<TF>(packed));
    if (order == 2) pres2_unpack<TF>(g, P<TF>(p), P<TF>(packed));
    else            pres4_unpack<TF>(g, P<TF>(p), P<TF>(packed));
}
ORC_API void orc_pres_solve(const mhh_grid* g, int order, void* p, void* p_packed, const void* r, const void* rh)
{ DISPATCH(g, pres_solve_t<double>(*g, order, p, p_packed, r, rh), pres_solve_t<float>(*g, order, p, p_packed, r, rh)); }

ORC_API void orc_pres_output(const mhh_grid* g, int order, void* ut, void* vt, void* wt, const void* p)
{
    if (order == 2) DISPATCH(g, pres2_output<double>(*g, D(ut), D(vt), D(wt), D(p)), pres2_output<float>(*g, F(ut), F(vt), F(wt), F(p)));
    else            DISPATCH(g, pres4_output<double>(*g, D(ut), D(vt), D(wt), D(p)), pres4_output<float>(*g, F(ut), F(vt), F(wt), F(p)));
}
ORC_API double orc_pres_divergence(const mhh_grid* g, int order, const void* u, const void* v, const void* w, const void* r, const void* rh)
{
    if (order == 2) return (g->dtype == MHH_F64) ? pres2_divergence<double>(*g, D(u), D(v), D(w), D(r), D(rh)) : pres2_divergence<float>(*g, F(u), F(v), F(w), F(r), F(rh));
    return (g->dtype == MHH_F64) ? pres4_divergence<double>(*g, D(u), D(v), D(w)) : pres4_divergence<float>(*g, F(u), F(v), F(w));
}
// Pres::exec
ORC_API void orc_pres_exec(const mhh_grid* g, int order, void* p, void* p_packed, const void* u, const void* v, const void* w,
                           void* ut, void* vt, void* wt, const void* r, const void* rh, double dt)
{
    orc_pres_input(g, order, p_packed, u, v, w, ut, vt, wt, r, rh, dt);
    orc_pres_solve(g, order, p, p_packed, r, rh);
    orc_pres_output(g, order, ut, vt, wt, p);
}
// pressure-solver coefficient tables (for pinning set_values)
ORC_API void orc_pres2_coeffs(const mhh_grid* g, const void* rh, double* bmati, double* bmatj, double* a, double* c)
{
    std::vector<double> bi, bj, aa, cc;
    pres2_set_values<double>(*g, D(rh), bi, bj, aa, cc);
    std::copy(bi.begin(), bi.end(), bmati); std::copy(bj.begin(), bj.end(), bmatj);
    std::copy(aa.begin(), aa.end(), a); std::copy(cc.begin(), cc.end(), c);
}

ORC_API void orc_rk_substep(const mhh_grid* g, int order, int substep, double dt, void* a, void* at)
{ DISPATCH(g, rk_substep<double>(*g, order, substep, dt, D(a), D(at)), rk_substep<float>(*g, order, substep, (float)dt, F(a), F(at))); }

ORC_API void orc_ghost_cells(const mhh_grid* g, int order, void* a, int bcbot, int bctop, const void* abot, const void* agradbot, const void* atop, const void* agradtop)
{ DISPATCH(g, ghost_cells<double>(*g, order, D(a), bcbot, bctop, D(abot), D(agradbot), D(atop), D(agradtop)), ghost_cells<float>(*g, order, F(a), bcbot, bctop, F(abot), F(agradbot), F(atop), F(agradtop))); }
ORC_API void orc_ghost_cells_w(const mhh_grid* g, void* w, int type)
{ DISPATCH(g, ghost_cells_w<double>(*g, D(w), type), ghost_cells_w<float>(*g, F(w), type)); }
