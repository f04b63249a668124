// pres_lds_slab.h -- the x stages of Pres_2::exec with the transforms in LDS (pres_lds.h) for a slab rank: defined in
// k_pres.hip (where the kernels are instantiated), used by k_slab.hip. Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mhh_hip.h"
namespace mhh
{
int lds_slab_usable(const mhh_grid* g);
int lds_slab_twiddles(const mhh_grid* g, void** tx);
int lds_slab_twiddles_y(const mhh_grid* g, void** ty);
int lds_slab_yfft(const mhh_grid* g, bool fwd, void* xbuf, void* specy, const void* ty, int nxb, int npy, int ks, int kbeg, int kend, hipStream_t st);
int lds_slab_stage_in(const mhh_grid* g, const mhh_fields* f, double dt, void* xbuf, const void* tx, int nxb, int npy, int ks, int kbeg, int kend, hipStream_t st);
int lds_slab_stage_out(const mhh_grid* g, const mhh_fields* f, const void* xbuf, const void* tx, int nxb, int npy, int ks, int kbeg, int kend, hipStream_t st);
}
