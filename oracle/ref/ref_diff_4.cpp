#include "/root/reference/src/diff_4.cxx"
#include "ref_common.h"
template<class TF, bool dim3> static void run(const mhh_grid* g, int is_w, void* t, const void* a, double visc)
{
    if (is_w) diff_w<TF,dim3>(MP<TF>(t), CP<TF>(a), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells, TF(g->dx), TF(g->dy), CP<TF>(g->dzi4), CP<TF>(g->dzhi4));
    else      diff_c<TF,dim3>(MP<TF>(t), CP<TF>(a), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells, TF(g->dx), TF(g->dy), CP<TF>(g->dzi4), CP<TF>(g->dzhi4));
}
REF_API void ref_diff_4(const mhh_grid* g, int is_w, void* t, const void* a, double visc)
{
    const bool dim3 = (g->jtot != 1);
    if (g->dtype == MHH_F64) { if (dim3) run<double,true>(g, is_w, t, a, visc); else run<double,false>(g, is_w, t, a, visc); }
    else                     { if (dim3) run<float,true>(g, is_w, t, a, visc);  else run<float,false>(g, is_w, t, a, visc); }
}
