#include "/root/reference/src/diff_2.cxx"
#include "ref_common.h"
template<class TF> static void run(const mhh_grid* g, int is_w, void* t, const void* a, double visc)
{
    if (is_w) diff_w<TF>(MP<TF>(t), CP<TF>(a), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells, TF(g->dx), TF(g->dy), CP<TF>(g->dzi), CP<TF>(g->dzhi));
    else      diff_c<TF>(MP<TF>(t), CP<TF>(a), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells, TF(g->dx), TF(g->dy), CP<TF>(g->dzi), CP<TF>(g->dzhi));
}
REF_API void ref_diff_2(const mhh_grid* g, int is_w, void* t, const void* a, double visc)
{ if (g->dtype == MHH_F64) run<double>(g, is_w, t, a, visc); else run<float>(g, is_w, t, a, visc); }
