// integration/adaptor_pres_2.cxx -- replaces the USECUDA half of the reference's Pres_2 (src/pres_2.cu) and, with it, the cuFFT
// code of src/pres.cu:170-495: the plan object of the library owns the rocFFT plans, the modified wave numbers, the
// tridiagonal / heptadiagonal coefficients and all work space (no fields.get_tmp_g() inside the solve).
#include <map>
#include <vector>
#include "master.h"
#include "grid.h"
#include "fields.h"
#include "stats.h"
#include "pres_2.h"
#include "mhh_adaptor.h"

#ifdef USECUDA
namespace
{
    // The plan would be a new member (mhh_pres_plan* plan) next to the cufftHandle members it replaces (include/pres.h:75-78);
    // kept beside the class here so that this file compiles against the unmodified header.
    std::map<const void*, mhh_pres_plan*> plans;
    const std::string tend_name = "pres";       // src/pres_2.cxx: the name under which Stats books the pressure tendency
}

template<typename TF>
void Pres_2<TF>::prepare_device()
{
    auto& gd = grid.get_grid_data();
    mhh_grid g = mhh_make_grid(gd, master.get_MPI_data());
    mhh_pres_plan* plan = nullptr;
    // set_values runs on the host in the reference too (src/pres_2.cxx): the metrics and the base state go in as HOST pointers
    mhh_check(mhh_pres_plan_create(&g, 2, gd.dz.data(), gd.dzhi.data(), gd.dzi4.data(), gd.dzhi4.data(),
                                   fields.rhoref.data(), fields.rhorefh.data(), &plan));
    plans[this] = plan;
}

template<typename TF>
void Pres_2<TF>::clear_device()
{
    auto it = plans.find(this);
    if (it != plans.end()) { mhh_pres_plan_destroy(it->second); plans.erase(it); }
}

template<typename TF>
void Pres_2<TF>::exec(double dt, Stats<TF>& stats)
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    mhh_fields f = mhh_make_fields(fields);
    mhh_check(mhh_pres_exec(plans.at(this), &g, &f, dt, /*stream*/ nullptr));   // cyclic fill of the tendencies, input, transforms, solve, output
    mhh_check(mhh_synchronize(nullptr));                                        // as cudaDeviceSynchronize() ahead of the statistics
    stats.calc_tend(*fields.mt.at("u"), tend_name);
    stats.calc_tend(*fields.mt.at("v"), tend_name);
    stats.calc_tend(*fields.mt.at("w"), tend_name);
}

template<typename TF>
TF Pres_2<TF>::check_divergence()
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    mhh_fields f = mhh_make_fields(fields);
    auto tmp = fields.get_tmp_g();
    double div = 0;
    mhh_check(mhh_pres_check_divergence(&g, 2, &f, tmp->fld_g, &div, nullptr));   // synchronises; local maximum
    fields.release_tmp_g(tmp);
    master.max(&div, 1);
    return static_cast<TF>(div);
}

template void Pres_2<double>::prepare_device();
template void Pres_2<double>::clear_device();
template void Pres_2<double>::exec(double, Stats<double>&);
template double Pres_2<double>::check_divergence();
template void Pres_2<float>::prepare_device();
template void Pres_2<float>::clear_device();
template void Pres_2<float>::exec(double, Stats<float>&);
template float Pres_2<float>::check_divergence();
#endif
