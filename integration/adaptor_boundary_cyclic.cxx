// integration/adaptor_boundary_cyclic.cxx -- replaces src/boundary_cyclic.cu:91-127.
#include "grid.h"
#include "boundary_cyclic.h"
#include "mhh_adaptor.h"

#ifdef USECUDA
template<typename TF>
void Boundary_cyclic<TF>::exec_g(TF* data)
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    mhh_check(mhh_boundary_cyclic(&g, data, MHH_EDGE_BOTH, nullptr));
}
template<typename TF>
void Boundary_cyclic<TF>::exec_2d_g(TF* data)
{
    mhh_grid g = mhh_make_grid(grid.get_grid_data(), master.get_MPI_data());
    mhh_check(mhh_boundary_cyclic_2d(&g, data, nullptr));
}
template void Boundary_cyclic<double>::exec_g(double*);
template void Boundary_cyclic<double>::exec_2d_g(double*);
template void Boundary_cyclic<float>::exec_g(float*);
template void Boundary_cyclic<float>::exec_2d_g(float*);
#endif
