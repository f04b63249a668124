#include "/root/reference/src/advec_4m.cxx"
#include "ref_common.h"
template<class TF> static void run(const mhh_grid* g, int comp, void* t, const void* s, const void* u, const void* v, const void* w)
{
    const TF dx = TF(g->dx), dy = TF(g->dy);
    if (comp == 0) advec_u<TF,1>(MP<TF>(t), CP<TF>(u), CP<TF>(v), CP<TF>(w), CP<TF>(g->dzi4), dx, dy, GRID_BOUNDS(g), g->icells, g->ijcells);
    if (comp == 1) advec_v<TF,1>(MP<TF>(t), CP<TF>(u), CP<TF>(v), CP<TF>(w), CP<TF>(g->dzi4), dx, dy, GRID_BOUNDS(g), g->icells, g->ijcells);
    if (comp == 2) advec_w<TF,1>(MP<TF>(t), CP<TF>(u), CP<TF>(v), const_cast<TF*>(CP<TF>(w)), CP<TF>(g->dzhi4), dx, dy, GRID_BOUNDS(g), g->icells, g->ijcells);
    if (comp == 3) advec_s<TF,1>(MP<TF>(t), CP<TF>(s), CP<TF>(u), CP<TF>(v), CP<TF>(w), CP<TF>(g->dzi4), dx, dy, GRID_BOUNDS(g), g->icells, g->ijcells);
}
REF_API void ref_advec_4m(const mhh_grid* g, int comp, void* t, const void* s, const void* u, const void* v, const void* w, const void*, const void*)
{ if (g->dtype == MHH_F64) run<double>(g, comp, t, s, u, v, w); else run<float>(g, comp, t, s, u, v, w); }
REF_API double ref_advec_4m_cfl(const mhh_grid* g, const void* u, const void* v, const void* w, double dt)
{
    static Master* master = new Master();
    if (g->dtype == MHH_F64) return calc_cfl<double>(CP<double>(u), CP<double>(v), CP<double>(w), CP<double>(g->dzi), g->dx, g->dy, dt, *master, GRID_BOUNDS(g), g->icells, g->ijcells);
    return calc_cfl<float>(CP<float>(u), CP<float>(v), CP<float>(w), CP<float>(g->dzi), (float)g->dx, (float)g->dy, (float)dt, *master, GRID_BOUNDS(g), g->icells, g->ijcells);
}
