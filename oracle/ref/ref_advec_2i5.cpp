#include "/root/reference/src/advec_2i5.cxx"
#include "ref_common.h"
template<class TF> static void run(const mhh_grid* g, int comp, void* t, const void* s, const void* u, const void* v, const void* w, const void* r, const void* rh)
{
    const TF dx = TF(g->dx), dy = TF(g->dy);
    if (comp == 0) advec_u<TF>(MP<TF>(t), CP<TF>(u), CP<TF>(v), CP<TF>(w), CP<TF>(g->dzi), dx, dy, CP<TF>(r), CP<TF>(rh), GRID_BOUNDS(g), g->icells, g->ijcells);
    if (comp == 1) advec_v<TF>(MP<TF>(t), CP<TF>(u), CP<TF>(v), CP<TF>(w), CP<TF>(g->dzi), dx, dy, CP<TF>(r), CP<TF>(rh), GRID_BOUNDS(g), g->icells, g->ijcells);
    if (comp == 2) advec_w<TF>(MP<TF>(t), CP<TF>(u), CP<TF>(v), const_cast<TF*>(CP<TF>(w)), CP<TF>(g->dzhi), dx, dy, CP<TF>(r), CP<TF>(rh), GRID_BOUNDS(g), g->icells, g->ijcells);
    if (comp == 3) advec_s<TF>(MP<TF>(t), CP<TF>(s), CP<TF>(u), CP<TF>(v), CP<TF>(w), CP<TF>(g->dzi), dx, dy, CP<TF>(r), CP<TF>(rh), GRID_BOUNDS(g), g->icells, g->ijcells);
}
REF_API void ref_advec_2i5(const mhh_grid* g, int comp, void* t, const void* s, const void* u, const void* v, const void* w, const void* r, const void* rh)
{ if (g->dtype == MHH_F64) run<double>(g, comp, t, s, u, v, w, r, rh); else run<float>(g, comp, t, s, u, v, w, r, rh); }
REF_API double ref_advec_2i5_cfl(const mhh_grid* g, const void* u, const void* v, const void* w, double dt)
{
    static Master* master = new Master();
    if (g->dtype == MHH_F64) return calc_cfl<double>(CP<double>(u), CP<double>(v), CP<double>(w), CP<double>(g->dzi), g->dx, g->dy, dt, *master, GRID_BOUNDS(g), g->icells, g->ijcells);
    return calc_cfl<float>(CP<float>(u), CP<float>(v), CP<float>(w), CP<float>(g->dzi), (float)g->dx, (float)g->dy, (float)dt, *master, GRID_BOUNDS(g), g->icells, g->ijcells);
}
// Advec_monotonic::advec_s_lim (include/advec_monotonic.h:79, pulled in by advec_2i5.cxx)
REF_API void ref_advec_s_lim(const mhh_grid* g, void* t, const void* s, const void* u, const void* v, const void* w, const void* r, const void* rh)
{
    if (g->dtype == MHH_F64)
        Advec_monotonic::advec_s_lim<double>(MP<double>(t), CP<double>(s), CP<double>(u), CP<double>(v), CP<double>(w), CP<double>(g->dzi), g->dx, g->dy,
                                             CP<double>(r), CP<double>(rh), GRID_BOUNDS(g), g->icells, g->ijcells);
    else
        Advec_monotonic::advec_s_lim<float>(MP<float>(t), CP<float>(s), CP<float>(u), CP<float>(v), CP<float>(w), CP<float>(g->dzi), (float)g->dx, (float)g->dy,
                                            CP<float>(r), CP<float>(rh), GRID_BOUNDS(g), g->icells, g->ijcells);
}
