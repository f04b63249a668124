// integration/adaptor_diff_smag2.cxx -- replaces the USECUDA half of the reference's Diff_smag2 (src/diff_smag2.cu:519-870).
#include <algorithm>
#include <memory>
#include <vector>
#include "grid.h"
#include "fields.h"
#include "master.h"
#include "diff_smag2.h"
#include "boundary.h"
#include "defines.h"
#include "constants.h"
#include "thermo.h"
#include "stats.h"
#include "mhh_adaptor.h"

#ifdef USECUDA
// The library takes device pointers only; allocation and copies stay with the host code. These two are the HIP runtime
// entry points the maintainer's build already links (hipMalloc / hipMemcpy); declared here so that this file needs no HIP header.
extern "C" int hipMalloc(void** ptr, size_t size);
extern "C" int hipFree(void* ptr);
extern "C" int hipMemcpy(void* dst, const void* src, size_t size, int kind);   // kind 1 = host to device

namespace
{
    template<typename TF>
    mhh_diff_params make_params(Diff_smag2<TF>&, double cs, double tPr, TF* mlen_g, Boundary<TF>& boundary, Thermo<TF>* thermo, Fields<TF>& fields,
                                const void* n2_g)
    {
        mhh_diff_params p{};
        p.cs = cs; p.tPr = tPr; p.mlen0 = mlen_g;
        p.surface_model = (boundary.get_switch() != "default");
        p.neutral = thermo ? (thermo->get_switch() == "0") : 0;
        p.N2 = n2_g; p.th_for_N2 = -1; p.grav = Constants::grav<TF>;
        (void)fields;
        return p;
    }
}

template<typename TF>
void Diff_smag2<TF>::prepare_device(Boundary<TF>&)
{
    auto& gd = grid.get_grid_data();
    std::vector<TF> ml(gd.kcells);
    mhh_grid gh = mhh_make_grid(gd, master.get_MPI_data());
    gh.dz = gd.dz.data();                                             // HOST metrics for the host helper
    mhh_check(mhh_smag2_mlen0_host(&gh, this->cs, ml.data()));       // cs*pow(dx*dy*dz, 1/3) with the CPU path's libm
    if (hipMalloc(reinterpret_cast<void**>(&mlen_g), gd.kcells*sizeof(TF)) != 0) throw std::runtime_error("hipMalloc");
    if (hipMemcpy(mlen_g, ml.data(), gd.kcells*sizeof(TF), 1) != 0) throw std::runtime_error("hipMemcpy");
}

template<typename TF>
void Diff_smag2<TF>::clear_device()
{
    hipFree(mlen_g);
    mlen_g = nullptr;
}

template<typename TF>
void Diff_smag2<TF>::exec_viscosity(Thermo<TF>& thermo)
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    mhh_fields f = mhh_make_fields(fields);
    f.dudz = boundary.get_dudz_g(); f.dvdz = boundary.get_dvdz_g(); f.dbdz = boundary.get_dbdz_g(); f.z0m = boundary.get_z0m_g();
    // thermo hands over its N2 field (src/diff_smag2.cxx:1143); the library can also evaluate it inline from th (th_for_N2, thref)
    auto tmp = fields.get_tmp_g();
    const bool neutral = (thermo.get_switch() == "0");
    if (!neutral)
        thermo.get_thermo_field_g(*tmp, "N2", false);
    mhh_diff_params p = make_params(*this, this->cs, this->tPr, mlen_g, boundary, &thermo, fields, neutral ? nullptr : tmp->fld_g);
    mhh_check(mhh_diff_exec_viscosity(&g, MHH_DIFF_SMAG2, &f, &p, nullptr));    // ends with the cyclic fill of evisc
    fields.release_tmp_g(tmp);
}

template<typename TF>
void Diff_smag2<TF>::exec(Stats<TF>& stats)
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    mhh_fields f = mhh_make_fields(fields);
    mhh_diff_params p = make_params<TF>(*this, this->cs, this->tPr, mlen_g, boundary, nullptr, fields, nullptr);
    mhh_check(mhh_diff_exec(&g, MHH_DIFF_SMAG2, &f, &p, nullptr));
    mhh_check(mhh_synchronize(nullptr));                      // as cudaDeviceSynchronize() ahead of the statistics, src/advec_2.cu:219
    stats.calc_tend(*fields.mt.at("u"), tend_name);
    stats.calc_tend(*fields.mt.at("v"), tend_name);
    stats.calc_tend(*fields.mt.at("w"), tend_name);
    for (auto& it : fields.st)
        stats.calc_tend(*it.second, tend_name);
}

template<typename TF>
double Diff_smag2<TF>::get_dn(const double dt)
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    double dnmul_l = 0;
    auto tmp = fields.get_tmp_g();
    mhh_check(mhh_smag2_dnmul(&g, fields.sd.at("evisc")->fld_g, this->tPr, tmp->fld_g, &dnmul_l, nullptr));
    fields.release_tmp_g(tmp);
    master.max(&dnmul_l, 1);
    return dnmul_l*dt;
}

template<typename TF>
unsigned long Diff_smag2<TF>::get_time_limit(const unsigned long idt, const double dt)
{
    const double dn = std::max(get_dn(dt), 1.e-20);                  // src/diff_smag2.cxx:884-900: avoid a division by zero
    return idt * dnmax / dn;
}

template void Diff_smag2<double>::prepare_device(Boundary<double>&);
template void Diff_smag2<double>::clear_device();
template void Diff_smag2<double>::exec_viscosity(Thermo<double>&);
template void Diff_smag2<double>::exec(Stats<double>&);
template double Diff_smag2<double>::get_dn(double);
template unsigned long Diff_smag2<double>::get_time_limit(unsigned long, double);
#endif
