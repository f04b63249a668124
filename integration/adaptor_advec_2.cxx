// integration/adaptor_advec_2.cxx -- replaces the USECUDA half of the reference's Advec_2 (src/advec_2.cu).
#include <algorithm>
#include "advec_2.h"
#include "grid.h"
#include "fields.h"
#include "stats.h"
#include "mhh_adaptor.h"

#ifdef USECUDA
template<typename TF>
void Advec_2<TF>::exec(Stats<TF>& stats)
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    mhh_fields f = mhh_make_fields(fields);
    mhh_check(mhh_advec_exec(&g, MHH_ADVEC_2, &f, /*stream*/ nullptr));
    mhh_check(mhh_synchronize(nullptr));                      // as cudaDeviceSynchronize() ahead of the statistics, src/advec_2.cu:219
    stats.calc_tend(*fields.mt.at("u"), tend_name);
    stats.calc_tend(*fields.mt.at("v"), tend_name);
    stats.calc_tend(*fields.mt.at("w"), tend_name);
    for (auto& it : fields.st)
        stats.calc_tend(*it.second, tend_name);
}

template<typename TF>
double Advec_2<TF>::get_cfl(const double dt)
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    double cfl = 0;
    auto tmp = fields.get_tmp_g();
    mhh_check(mhh_advec_cfl(&g, MHH_ADVEC_2, fields.mp.at("u")->fld_g, fields.mp.at("v")->fld_g, fields.mp.at("w")->fld_g,
                            dt, tmp->fld_g, &cfl, nullptr));
    fields.release_tmp_g(tmp);
    master.max(&cfl, 1);
    return cfl;
}

template<typename TF>
unsigned long Advec_2<TF>::get_time_limit(unsigned long idt, double dt)
{
    double cfl = get_cfl(dt);
    cfl = std::max(cflmin, cfl);
    return idt * cflmax / cfl;
}

template void Advec_2<double>::exec(Stats<double>&);
template double Advec_2<double>::get_cfl(double);
template unsigned long Advec_2<double>::get_time_limit(unsigned long, double);
template void Advec_2<float>::exec(Stats<float>&);
template double Advec_2<float>::get_cfl(double);
template unsigned long Advec_2<float>::get_time_limit(unsigned long, double);
#endif
