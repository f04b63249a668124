// integration/mhh_adaptor.h -- shared helper of the adaptor translation units that replace MicroHH's .cu files
// (INTEGRATION.md). Written against the reference's own headers; compiled there with -DUSECUDA (the backend macro keeps
// its name) and linked with -lmhh_hip. tests/test_integration_compile.py type-checks these files against
// /root/reference/include whenever the reference is present.
#ifndef MHH_ADAPTOR_H
#define MHH_ADAPTOR_H
#include <stdexcept>
#include <string>
#include "mhh_hip.h"
#include "grid.h"      // include/grid.h:49-135  (Grid_data)
#include "fields.h"    // include/fields.h:132-190
#include "master.h"    // include/master.h:34-53  (MPI_data)

template<typename TF> constexpr int mhh_dtype() { return sizeof(TF) == 8 ? MHH_F64 : MHH_F32; }
inline void mhh_check(int rc) { if (rc) throw std::runtime_error(std::string("mhh: ") + mhh_last_error()); }   // as tools.h:49-56 / pres.cu:185

template<typename TF>
mhh_grid mhh_make_grid(const Grid_data<TF>& gd, const MPI_data& md)
{
    mhh_grid g{};
    g.itot = gd.itot; g.jtot = gd.jtot; g.ktot = gd.ktot; g.imax = gd.imax; g.jmax = gd.jmax; g.kmax = gd.kmax;
    g.igc = gd.igc; g.jgc = gd.jgc; g.kgc = gd.kgc; g.icells = gd.icells; g.jcells = gd.jcells; g.ijcells = gd.ijcells; g.kcells = gd.kcells;
    g.istart = gd.istart; g.jstart = gd.jstart; g.kstart = gd.kstart; g.iend = gd.iend; g.jend = gd.jend; g.kend = gd.kend;
    g.dtype = mhh_dtype<TF>(); g.npx = md.npx; g.npy = md.npy; g.mpicoordx = md.mpicoordx; g.mpicoordy = md.mpicoordy;
    g.ncells = gd.ncells; g.xsize = gd.xsize; g.ysize = gd.ysize; g.zsize = gd.zsize; g.dx = gd.dx; g.dy = gd.dy;
    g.z = gd.z_g; g.zh = gd.zh_g; g.dz = gd.dz_g; g.dzh = gd.dzh_g; g.dzi = gd.dzi_g; g.dzhi = gd.dzhi_g; g.dzi4 = gd.dzi4_g; g.dzhi4 = gd.dzhi4_g;
    return g;
}

template<typename TF>
mhh_fields mhh_make_fields(Fields<TF>& f)
{
    mhh_fields a{};
    a.u = f.mp.at("u")->fld_g; a.v = f.mp.at("v")->fld_g; a.w = f.mp.at("w")->fld_g;
    a.ut = f.mt.at("u")->fld_g; a.vt = f.mt.at("v")->fld_g; a.wt = f.mt.at("w")->fld_g;
    int n = 0;
    for (auto& it : f.sp)
    {
        if (n >= MHH_MAX_SCALARS) throw std::runtime_error("mhh: more than MHH_MAX_SCALARS prognostic scalars");
        a.s[n] = it.second->fld_g; a.st[n] = f.st.at(it.first)->fld_g; a.svisc[n] = it.second->visc;
        a.s_fluxbot[n] = it.second->flux_bot_g; a.s_fluxtop[n] = it.second->flux_top_g;
        ++n;
    }
    a.nscalars = n;
    a.evisc = f.sd.count("evisc") ? f.sd.at("evisc")->fld_g : nullptr;
    a.p = f.sd.count("p") ? f.sd.at("p")->fld_g : nullptr;
    a.rhoref = f.rhoref_g; a.rhorefh = f.rhorefh_g; a.visc = f.visc;
    a.u_fluxbot = f.mp.at("u")->flux_bot_g; a.u_fluxtop = f.mp.at("u")->flux_top_g;
    a.v_fluxbot = f.mp.at("v")->flux_bot_g; a.v_fluxtop = f.mp.at("v")->flux_top_g;
    return a;
}
template<typename TF>
int mhh_scalar_index(Fields<TF>& f, const std::string& name)
{
    int n = 0;
    for (auto& it : f.sp) { if (it.first == name) return n; ++n; }
    return -1;
}
#endif
