// integration/adaptor_diff_4.cxx -- replaces the USECUDA half of the reference's Diff_4 (src/diff_4.cu): exec only;
// the constant-viscosity time limit (get_dn, get_time_limit) is host arithmetic and stays in src/diff_4.cxx.
#include <memory>
#include "grid.h"
#include "fields.h"
#include "master.h"
#include "diff_4.h"
#include "defines.h"
#include "stats.h"
#include "mhh_adaptor.h"

#ifdef USECUDA
template<typename TF>
void Diff_4<TF>::exec(Stats<TF>& stats)
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    mhh_fields f = mhh_make_fields(fields);
    mhh_check(mhh_diff_exec(&g, MHH_DIFF_4, &f, nullptr, nullptr));
    mhh_check(mhh_synchronize(nullptr));                      // as cudaDeviceSynchronize() ahead of the statistics, src/advec_2.cu:219
    stats.calc_tend(*fields.mt.at("u"), tend_name);
    stats.calc_tend(*fields.mt.at("v"), tend_name);
    stats.calc_tend(*fields.mt.at("w"), tend_name);
    for (auto& it : fields.st)
        stats.calc_tend(*it.second, tend_name);
}
template void Diff_4<double>::exec(Stats<double>&);
template void Diff_4<float>::exec(Stats<float>&);
#endif
