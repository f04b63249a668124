#include "/root/reference/src/diff_smag2.cxx"
#include "ref_common.h"
// calc_evisc / calc_evisc_neutral are NOT wrapped: they end in Boundary_cyclic::exec -> Grid::get_grid_data,
// and src/grid.cxx needs netcdf.h, which this image lacks (see DESIGN.md "Oracle").
template<class TF> static void strain2(const mhh_grid* g, int sm, void* s2, const void* u, const void* v, const void* w, const void* dudz, const void* dvdz)
{
    // call-site argument spelling of Diff_smag2::exec_viscosity (src/diff_smag2.cxx:1059-1089): 1./gd.dx narrowed to TF
    const TF dx = TF(g->dx), dy = TF(g->dy);
    if (sm) calc_strain2<TF, Surface_model::Enabled >(MP<TF>(s2), CP<TF>(u), CP<TF>(v), CP<TF>(w), CP<TF>(dudz), CP<TF>(dvdz), CP<TF>(g->z), CP<TF>(g->dzi), CP<TF>(g->dzhi), 1./dx, 1./dy, GRID_BOUNDS(g), g->icells, g->ijcells);
    else    calc_strain2<TF, Surface_model::Disabled>(MP<TF>(s2), CP<TF>(u), CP<TF>(v), CP<TF>(w), nullptr, nullptr, CP<TF>(g->z), CP<TF>(g->dzi), CP<TF>(g->dzhi), 1./dx, 1./dy, GRID_BOUNDS(g), g->icells, g->ijcells);
}
REF_API void ref_smag2_strain2(const mhh_grid* g, int sm, void* s2, const void* u, const void* v, const void* w, const void* dudz, const void* dvdz)
{ if (g->dtype == MHH_F64) strain2<double>(g, sm, s2, u, v, w, dudz, dvdz); else strain2<float>(g, sm, s2, u, v, w, dudz, dvdz); }

template<class TF> static void duvw(const mhh_grid* g, int comp, int sm, void* t, const void* u, const void* v, const void* w, const void* ev,
                                    const void* fb, const void* ft, const void* r, const void* rh, double visc)
{
    const TF dx = TF(g->dx), dy = TF(g->dy);
#define SMAG_ARGS MP<TF>(t), CP<TF>(u), CP<TF>(v), CP<TF>(w), CP<TF>(g->dzi), CP<TF>(g->dzhi), 1./dx, 1./dy, CP<TF>(ev)
    if (comp == 0)
    {
        if (sm) diff_u<TF, Surface_model::Enabled >(SMAG_ARGS, CP<TF>(fb), CP<TF>(ft), CP<TF>(r), CP<TF>(rh), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells);
        else    diff_u<TF, Surface_model::Disabled>(SMAG_ARGS, CP<TF>(fb), CP<TF>(ft), CP<TF>(r), CP<TF>(rh), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells);
    }
    else if (comp == 1)
    {
        if (sm) diff_v<TF, Surface_model::Enabled >(SMAG_ARGS, CP<TF>(fb), CP<TF>(ft), CP<TF>(r), CP<TF>(rh), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells);
        else    diff_v<TF, Surface_model::Disabled>(SMAG_ARGS, CP<TF>(fb), CP<TF>(ft), CP<TF>(r), CP<TF>(rh), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells);
    }
    else
        diff_w<TF>(SMAG_ARGS, CP<TF>(r), CP<TF>(rh), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells);
}
REF_API void ref_smag2_diff_uvw(const mhh_grid* g, int comp, int sm, void* t, const void* u, const void* v, const void* w, const void* ev,
                                const void* fb, const void* ft, const void* r, const void* rh, double visc)
{ if (g->dtype == MHH_F64) duvw<double>(g, comp, sm, t, u, v, w, ev, fb, ft, r, rh, visc); else duvw<float>(g, comp, sm, t, u, v, w, ev, fb, ft, r, rh, visc); }

template<class TF> static void dc(const mhh_grid* g, int sm, void* t, const void* a, const void* ev, const void* fb, const void* ft, const void* r, const void* rh, double tPr, double visc)
{
    const TF dx = TF(g->dx), dy = TF(g->dy);
    if (sm) diff_c<TF, Surface_model::Enabled >(MP<TF>(t), CP<TF>(a), CP<TF>(g->dzi), CP<TF>(g->dzhi), 1./(dx*dx), 1./(dy*dy), CP<TF>(ev), CP<TF>(fb), CP<TF>(ft), CP<TF>(r), CP<TF>(rh), TF(tPr), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells);
    else    diff_c<TF, Surface_model::Disabled>(MP<TF>(t), CP<TF>(a), CP<TF>(g->dzi), CP<TF>(g->dzhi), 1./(dx*dx), 1./(dy*dy), CP<TF>(ev), CP<TF>(fb), CP<TF>(ft), CP<TF>(r), CP<TF>(rh), TF(tPr), TF(visc), GRID_BOUNDS(g), g->icells, g->ijcells);
}
REF_API void ref_smag2_diff_c(const mhh_grid* g, int sm, void* t, const void* a, const void* ev, const void* fb, const void* ft, const void* r, const void* rh, double tPr, double visc)
{ if (g->dtype == MHH_F64) dc<double>(g, sm, t, a, ev, fb, ft, r, rh, tPr, visc); else dc<float>(g, sm, t, a, ev, fb, ft, r, rh, tPr, visc); }

REF_API double ref_smag2_dnmul(const mhh_grid* g, const void* ev, double tPr)
{
    if (g->dtype == MHH_F64) { const double dx = g->dx, dy = g->dy; return calc_dnmul<double>(CP<double>(ev), CP<double>(g->dzi), 1./(dx*dx), 1./(dy*dy), tPr, GRID_BOUNDS(g), g->icells, g->ijcells); }
    const float dx = (float)g->dx, dy = (float)g->dy;
    return calc_dnmul<float>(CP<float>(ev), CP<float>(g->dzi), 1./(dx*dx), 1./(dy*dy), (float)tPr, GRID_BOUNDS(g), g->icells, g->ijcells);
}
