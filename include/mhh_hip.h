/*
 * mhh_hip.h -- C ABI of the MI355X-native RHS + pressure hot path (libmhh_hip.so).
 *
 * Drop-in boundary for the reference's Advec / Diff / Pres / Boundary_cyclic operators
 * (adconnolly/microhh). The reference has no FFI; its GPU seam is the set of member
 * functions that the .cu files re-define under USECUDA. Every entry point below names the
 * reference member function / kernel it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the adaptor TU a MicroHH maintainer would add.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void*
 *     (NULL = default stream). Nothing here synchronises the device unless stated.
 *   - all field pointers are DEVICE pointers, layout identical to the reference's host layout:
 *     ijk = i + j*icells + k*ijcells, i fastest, ghost cells included (include/grid.h:70-80).
 *   - element type is selected by mhh_grid.dtype (MHH_F64 = reference default build,
 *     MHH_F32 = reference -DUSESP build). Scalars cross the ABI as double and are narrowed
 *     exactly where the reference narrows them.
 *   - return value: 0 = ok, otherwise an MHH_E* code; mhh_last_error() gives the text. The C++
 *     adaptor turns non-zero into std::runtime_error like the reference host code does
 *     (include/tools.h:49-56, src/pres.cu:185-186).
 *   - buffers are owned by the caller (Field3d::init_device, src/field3d.cu:32-47); kernels
 *     borrow and never retain pointers. Not re-entrant per plan object.
 */
#ifndef MHH_HIP_H
#define MHH_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MHH_F64 0
#define MHH_F32 1

#define MHH_OK        0
#define MHH_EINVAL    1   /* bad argument / unsupported combination */
#define MHH_EHIP      2   /* HIP runtime error */
#define MHH_EFFT      3   /* rocFFT error (reference: runtime_error("FFT error")) */
#define MHH_ENOMEM    4

/* Boundary_cyclic Edge (include/boundary_cyclic.h:33) */
#define MHH_EDGE_EW   0
#define MHH_EDGE_NS   1
#define MHH_EDGE_BOTH 2

/* advection schemes (src/advec.cxx:55-83 swadvec) */
#define MHH_ADVEC_2    2
#define MHH_ADVEC_2I5  25
#define MHH_ADVEC_4M   41   /* 4th-order kinetic-energy-conserving form (src/advec_4m.cxx); per-field entry points only */
#define MHH_ADVEC_2I53 253  /* advec_2i5 with 4th/3rd instead of 6th/5th order vertically (src/advec_2i53.cxx); per-field entry points only */
#define MHH_ADVEC_2I62 262  /* 6th-order interpolation horizontally, 2nd order vertically (src/advec_2i62.cxx); per-field entry points only */
#define MHH_ADVEC_2I4  24   /* 2nd-order fluxes with 4th-order interpolation (src/advec_2i4.cxx); per-field entry points only */
#define MHH_ADVEC_4    4
/* diffusion schemes (src/diff.cxx:57-85 swdiff) */
#define MHH_DIFF_2     2
#define MHH_DIFF_4     4
#define MHH_DIFF_SMAG2 22

/* Mirror of Grid_data<TF> (include/grid.h:49-135) reduced to what the hot path reads.
 * Metric arrays are device pointers of kcells elements of the grid's dtype. */
typedef struct mhh_grid
{
    int itot, jtot, ktot;
    int imax, jmax, kmax;          /* per-rank interior; == itot,jtot,ktot on one GPU   */
    int igc, jgc, kgc;
    int icells, jcells, ijcells, kcells;
    int istart, jstart, kstart;
    int iend, jend, kend;
    int dtype;                     /* MHH_F64 | MHH_F32                                  */
    int npx, npy;                  /* slab decomposition: npx must be 1                   */
    int mpicoordx, mpicoordy;      /* rank coordinates (include/master.h:34-53)           */
    long long ncells;
    double xsize, ysize, zsize;
    double dx, dy;
    const void* z;
    const void* zh;
    const void* dz;
    const void* dzh;
    const void* dzi;
    const void* dzhi;
    const void* dzi4;
    const void* dzhi4;
} mhh_grid;

#define MHH_MAX_SCALARS 8

/* Mirror of the slice of Fields<TF> the operators touch (include/fields.h:132-161):
 * mp (u,v,w), mt (ut,vt,wt), sp/st (scalars and their tendencies), sd["evisc"], sd["p"],
 * rhoref/rhorefh, and the 2-D surface arrays consumed by diff_smag2. NULL where unused. */
typedef struct mhh_fields
{
    void* u;  void* v;  void* w;
    void* ut; void* vt; void* wt;
    int   nscalars;
    void* s [MHH_MAX_SCALARS];
    void* st[MHH_MAX_SCALARS];
    double svisc[MHH_MAX_SCALARS];   /* Field3d::visc of each scalar                      */
    void* evisc;
    void* p;
    const void* rhoref;              /* [kcells] */
    const void* rhorefh;             /* [kcells] */
    double visc;                     /* Fields::visc                                       */
    /* surface model inputs (Boundary_surface outputs; NULL => resolved walls, "default") */
    const void* u_fluxbot; const void* u_fluxtop;
    const void* v_fluxbot; const void* v_fluxtop;
    const void* s_fluxbot[MHH_MAX_SCALARS]; const void* s_fluxtop[MHH_MAX_SCALARS];
    const void* dudz; const void* dvdz; const void* dbdz;   /* boundary.get_dudz() etc.    */
    const void* z0m;
    int   s_fluxlimit[MHH_MAX_SCALARS];   /* scalar is in advec.fluxlimit_list (src/advec_2i5.cxx:39,921) */
} mhh_fields;

const char* mhh_last_error(void);
int mhh_version(void);
/* Block the host until everything queued on `stream` has finished (NULL = default stream): what the reference's
 * cudaDeviceSynchronize() at the end of every exec does before Stats reads the tendencies (src/advec_2.cu:219). */
int mhh_synchronize(void* stream);

/* ---- Boundary_cyclic -------------------------------------------------------------------
 * replaces Boundary_cyclic<TF>::exec_g / exec (src/boundary_cyclic.cu:91-127,
 * src/boundary_cyclic.cxx:370-443) and exec_2d(_g) (:445-500). jtot==1 replicates row. */
int mhh_boundary_cyclic   (const mhh_grid* g, void* data, int edge, void* stream);
int mhh_boundary_cyclic_2d(const mhh_grid* g, void* data, void* stream);
/* Boundary_cyclic::exec(unsigned int*, Edge) / exec_2d(unsigned int*) (src/boundary_cyclic.cxx:510-660): the same fills for
 * 32-bit integer fields (the immersed-boundary / land-surface index masks), whatever the grid's dtype                    */
int mhh_boundary_cyclic_u32(const mhh_grid* g, void* data /* unsigned int [ncells] */, int edge, void* stream);
int mhh_boundary_cyclic_2d_u32(const mhh_grid* g, void* data /* unsigned int [ijcells] */, void* stream);
/* several fields in one launch (Boundary::set_prognostic_cyclic_bcs, src/boundary.cxx:447-458) */
int mhh_boundary_cyclic_n (const mhh_grid* g, void* const* data, int nfields, int edge, void* stream);

/* ---- Advection (kernel granularity of the reference) -----------------------------------
 * scheme 2  : src/advec_2.cxx:81-202   (advec_u/v/w/s),  GPU src/advec_2.cu:38-117
 * scheme 25 : src/advec_2i5.cxx:151-728,                 GPU src/advec_2i5.cu:41-510
 * scheme 4  : src/advec_4.cxx:88-486,                    GPU src/advec_4.cu:37-442
 * scheme 24 : src/advec_2i4.cxx:101-640 (advec_u/v/w/s), calc_cfl :51-99
 * scheme 262: src/advec_2i62.cxx:105-310,                calc_cfl :58-105
 * scheme 253: src/advec_2i53.cxx:120-700,                calc_cfl :55-118
 * scheme 41 : src/advec_4m.cxx:90-478,                   calc_cfl :51-88
 * Each call adds the advective tendency of one field (read-modify-write of the tendency). */
int mhh_advec_u(const mhh_grid* g, int scheme, void* ut, const void* u, const void* v, const void* w,
                const void* rhoref, const void* rhorefh, void* stream);
int mhh_advec_v(const mhh_grid* g, int scheme, void* vt, const void* u, const void* v, const void* w,
                const void* rhoref, const void* rhorefh, void* stream);
int mhh_advec_w(const mhh_grid* g, int scheme, void* wt, const void* u, const void* v, const void* w,
                const void* rhoref, const void* rhorefh, void* stream);
int mhh_advec_s(const mhh_grid* g, int scheme, void* st, const void* s, const void* u, const void* v, const void* w,
                const void* rhoref, const void* rhorefh, void* stream);
/* Koren-limited scalar advection, Advec_monotonic::advec_s_lim (include/advec_monotonic.h:79-180),
 * used by Advec_2i5::exec for the scalars in fluxlimit_list (src/advec_2i5.cxx:1030). */
int mhh_advec_s_lim(const mhh_grid* g, void* st, const void* s, const void* u, const void* v, const void* w,
                    const void* rhoref, const void* rhorefh, void* stream);
/* Advec::exec (src/advec_2.cxx:297-335, advec_2i5.cxx:977-1044, advec_4.cxx:592-650) */
int mhh_advec_exec(const mhh_grid* g, int scheme, const mhh_fields* f, void* stream);
/* calc_cfl + Master::max + *dt  (src/advec_2.cxx:51-78, advec_2i5.cxx:60-148, advec_4.cxx:51-86);
 * *cfl_out is a host double; the call synchronises `stream`. `work` = device scratch of
 * mhh_reduce_work_bytes() bytes. */
int mhh_advec_cfl(const mhh_grid* g, int scheme, const void* u, const void* v, const void* w,
                  double dt, void* work, double* cfl_out, void* stream);
unsigned long long mhh_reduce_work_bytes(void);

/* ---- Diffusion ---------------------------------------------------------------------------
 * diff_2: src/diff_2.cxx:39-86 (diff_c, diff_w)   GPU src/diff_2.cu:34-90
 * diff_4: src/diff_4.cxx:41-173                   GPU src/diff_4.cu:37-160          */
int mhh_diff_c(const mhh_grid* g, int order, void* at, const void* a, double visc, void* stream);
int mhh_diff_w(const mhh_grid* g, int order, void* wt, const void* w, double visc, void* stream);

/* diff_smag2 (src/diff_smag2.cxx; parity target is the CPU path, NOT src/diff_smag2.cu, see
 * DESIGN.md). surface_model = 1 <=> boundary.get_switch() != "default".                  */
int mhh_smag2_strain2(const mhh_grid* g, int surface_model, void* strain2,
                      const void* u, const void* v, const void* w,
                      const void* dudz, const void* dvdz, void* stream);                 /* :47-155  */
/* mlen0[kcells] = cs*pow(dx*dy*dz[k],1/3) is a per-level table computed on the HOST with the C library's pow
 * (mhh_smag2_mlen0_host; `g` then carries HOST metric pointers) and uploaded by the caller -- what
 * Diff_smag2::prepare_device does in the reference's GPU path (src/diff_smag2.cu:521-542).               */
int mhh_smag2_mlen0_host(const mhh_grid* g_host, double cs, void* mlen0_host_out);
/* mlen2[kcells]: the squared mixing length of calc_evisc / calc_evisc_neutral (src/diff_smag2.cxx:273-276,325-328:
 * wall-damped with kappa*(z+z0m) under a surface model, mlen0^2 otherwise) for a HORIZONTALLY UNIFORM roughness length
 * z0m, evaluated on the host with the same IEEE operations as the kernels (same bits). Optional: passed as
 * mhh_diff_params::mlen2 it replaces three divisions and a square root per cell of exec_viscosity by a table look-up.
 * `g_host` carries HOST metric pointers (z); valid while z0m stays uniform and unchanged. */
int mhh_smag2_mlen2_host(const mhh_grid* g_host, int surface_model, int neutral, const void* mlen0_host, double z0m, void* mlen2_host_out);
int mhh_smag2_evisc(const mhh_grid* g, int surface_model, void* evisc, const void* N2,
                    const void* bgradbot, const void* z0m, const void* mlen0, double tPr, void* stream); /* :254-367 incl. cyclic fill */
int mhh_smag2_evisc_neutral(const mhh_grid* g, int surface_model, void* evisc,
                    const void* u, const void* v, const void* z0m,
                    const void* mlen0, double visc, void* stream);                       /* :157-252 incl. cyclic fill */
int mhh_smag2_diff_u(const mhh_grid* g, int surface_model, void* ut, const void* u, const void* v, const void* w,
                     const void* evisc, const void* fluxbot, const void* fluxtop,
                     const void* rhoref, const void* rhorefh, double visc, void* stream); /* :369-468 */
int mhh_smag2_diff_v(const mhh_grid* g, int surface_model, void* vt, const void* u, const void* v, const void* w,
                     const void* evisc, const void* fluxbot, const void* fluxtop,
                     const void* rhoref, const void* rhorefh, double visc, void* stream); /* :470-571 */
int mhh_smag2_diff_w(const mhh_grid* g, void* wt, const void* u, const void* v, const void* w,
                     const void* evisc, const void* rhoref, const void* rhorefh, double visc, void* stream); /* :573-617 */
int mhh_smag2_diff_c(const mhh_grid* g, int surface_model, void* at, const void* a,
                     const void* evisc, const void* fluxbot, const void* fluxtop,
                     const void* rhoref, const void* rhorefh, double tPr, double visc, void* stream); /* :619-709 */
int mhh_smag2_dnmul(const mhh_grid* g, const void* evisc, double tPr, void* work,
                    double* dnmul_out, void* stream);                                      /* :711-736 + master.max */
/* Thermo_dry calc_N2 hook (src/thermo_dry.cxx:66-78): input of calc_evisc */
int mhh_calc_N2(const mhh_grid* g, void* N2, const void* th, const void* thref, double grav, void* stream);

/* Diff::exec_viscosity / Diff::exec (src/diff_smag2.cxx:1046-1188, :939-1043; diff_2.cxx:150-180;
 * diff_4.cxx:262-300). For smag2, N2 (3-D) must be supplied unless `th_for_N2` >= 0, in which case
 * N2 is evaluated inline from scalar th_for_N2 with thref/grav (fusion of thermo.get_thermo_field). */
typedef struct mhh_diff_params
{
    double cs, tPr;          /* [diff] cs, tPr  (src/diff_smag2.cxx:853-855)              */
    int    surface_model;    /* boundary switch != "default"                               */
    int    neutral;          /* thermo switch == "0"                                       */
    const void* N2;          /* 3-D buoyancy frequency, or NULL with th_for_N2 >= 0        */
    int    th_for_N2;
    const void* thref;       /* [kcells]                                                   */
    double grav;
    const void* mlen0;       /* [kcells] device table from mhh_smag2_mlen0_host            */
    /* dry buoyancy folded into mhh_rhs_exec (Thermo_dry::exec precedes Advec::exec in Model::exec,
     * src/model.cxx:365,388): wt += grav/threfh[k]*(th_h - threfh[k]) for scalar th_for_N2, first. */
    int    buoyancy;         /* 0 = off, 2 / 4 = interpolation order (swspatialorder)      */
    const void* threfh;      /* [kcells]                                                   */
    /* slab decomposition (npy > 1): also evaluate evisc on the ghost rows jstart-1 and jend from the
     * u, v, w (, th) halos instead of exchanging it -- the same operands as on the neighbour, hence
     * the same bits; the diffusion kernels read evisc one row beyond the slab only. Needs jgc >= 2 and
     * the 2-D surface inputs valid on those rows.                                                      */
    int    evisc_ghost_rows;
    const void* mlen2;       /* [kcells] optional device table from mhh_smag2_mlen2_host (uniform z0m), or NULL */
} mhh_diff_params;
int mhh_diff_exec_viscosity(const mhh_grid* g, int scheme, const mhh_fields* f, const mhh_diff_params* p, void* stream);
/* exec_viscosity over the rows [j0, j1) of [jstart-1, jend+1) only (the wall mirror and the east-west wrap still cover
 * all rows): lets the slab driver evaluate the rows that need no north-south halo while the halos travel. */
int mhh_diff_exec_viscosity_rows(const mhh_grid* g, int scheme, const mhh_fields* f, const mhh_diff_params* p,
                                 int j0, int j1, void* stream);
int mhh_diff_exec_viscosity_rows2(const mhh_grid* g, int scheme, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, int j2, int j3, void* stream);
/* diagnostic: launches of the k-marching form of exec_viscosity so far (it needs 16-byte aligned rows; other layouts
 * take the one-thread-per-cell kernel, same bits) */
unsigned long long mhh_stat_visc_march_launches(void);
unsigned long long mhh_stat_rhs44_march_launches(void);   /* same for the k-marching form of (advec_4, diff_4) in mhh_rhs_exec */
/* self test of a device primitive of exec_viscosity: the square root for arguments known to be >= 2^-767 (csrc/gfx950_prims.h,
 * sqrt_in_range) against the compiler's sqrt on n arguments drawn from `seed`; *mismatches (host memory) must come back 0 */
int mhh_selftest_sqrt_in_range(unsigned long long n, unsigned long long seed, unsigned long long* mismatches, void* stream);
int mhh_diff_exec(const mhh_grid* g, int scheme, const mhh_fields* f, const mhh_diff_params* p, void* stream);

/* Thermo_dry::exec buoyancy tendency, calc_buoyancy_tend_2nd / _4th (src/thermo_dry.cxx:165-197,
 * GPU src/thermo_dry.cu): wt[k] += grav/threfh[k]*(interp(th) - threfh[k]) for k in (kstart, kend). */
int mhh_thermo_dry_buoyancy_tend(const mhh_grid* g, int order, void* wt, const void* th, const void* threfh,
                                 double grav, void* stream);

/* ---- Fused RHS: advec.exec + diff.exec in one pass over the tendencies ------------------
 * Same arithmetic, same order of accumulation into each tendency as calling
 * mhh_advec_exec then mhh_diff_exec (bit-identical results), one read of every input
 * and one read-modify-write of every tendency for the pairs (2,2) (25,22) (4,4); any other
 * pair of valid schemes runs as the two operator calls.                                   */
int mhh_rhs_exec(const mhh_grid* g, int advec_scheme, int diff_scheme, const mhh_fields* f,
                 const mhh_diff_params* p, void* stream);

/* the (advec_2i5, diff_smag2) pass over the rows [j0, j1) of the interior only; u, v, w and at most one (unlimited)
 * scalar; same bits as the whole-slab call on those rows */
int mhh_rhs_exec_rows(const mhh_grid* g, int advec_scheme, int diff_scheme, const mhh_fields* f,
                      const mhh_diff_params* p, int j0, int j1, void* stream);
/* the same over TWO disjoint, ordered row ranges in one launch (a slab's two edge strips once its north-south halos are in) */
int mhh_rhs_exec_rows2(const mhh_grid* g, int advec_scheme, int diff_scheme, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, int j2, int j3, void* stream);

/* ---- Pressure ----------------------------------------------------------------------------
 * Plan object = Pres_2 / Pres_4 private state: bmati/bmatj, a/c (m1..m7), rocFFT plans,
 * work buffers (Pres_2::init/set_values/prepare_device: src/pres_2.cxx:107-153,
 * src/pres_2.cu:217-240; pres_4.cxx:179-252).  rhoref/rhorefh/dz/dzhi(/dzi4/dzhi4) are read at
 * creation through HOST pointers (set_values runs on the host in the reference too).       */
typedef struct mhh_pres_plan mhh_pres_plan;
int  mhh_pres_plan_create(const mhh_grid* g, int order /*2|4*/,
                          const void* host_dz, const void* host_dzhi,
                          const void* host_dzi4, const void* host_dzhi4,
                          const void* host_rhoref, const void* host_rhorefh,
                          mhh_pres_plan** out);
void mhh_pres_plan_destroy(mhh_pres_plan* plan);
/* Pres::exec(dt): input -> solve -> output (src/pres_2.cxx:66-94, pres_4.cxx:64-140) */
int mhh_pres_exec(mhh_pres_plan* plan, const mhh_grid* g, const mhh_fields* f, double dt, void* stream);
/* Power-of-two itot, jtot: three kernels that do the transforms in LDS (pres_2: csrc/pres_lds.h, pres_4: csrc/pres_lds4.h) instead
 * of the seven passes of the staged form. mhh_pres_exec takes this form from 2^24 cells on (where it is faster on MI355X:
 * profiles/r3_pres_forms.md); MHH_PRES_LDS=1 / 0 selects it wherever the plan has it / never. The stages one by one, for tests:
 * 1 = Pres::input + transform along x (src/pres_2.cxx:156-196, src/pres_4.cxx:256-317, src/fft.cxx:451-497), 2 = transforms along y
 * around the k sweeps (Thomas, src/pres_2.cxx:202-263; the factored 7-band system, src/pres_4.cxx:358-470, 574-730), 3 = transform
 * back along x + p with its ghost cells + Pres::output (src/pres_2.cxx:333-387; src/pres_4.cxx:481-571). pres_2 with itot <= 256:
 * stage 2 runs two blocks per column (a twisted factorisation: bottom-up and top-down eliminations that meet half way; MHH_PRES_Y_TWISTED=0 / 1). */
int   mhh_pres_lds_stage(mhh_pres_plan* plan, const mhh_grid* g, const mhh_fields* f, double dt, int stage, void* stream);
int   mhh_pres_plan_has_lds_form(const mhh_pres_plan* plan);
int   mhh_pres_exec_form(const mhh_pres_plan* plan);   /* what mhh_pres_exec will run: 0 = staged (rocFFT), 1 = transforms in LDS */
void* mhh_pres_plan_spectral(mhh_pres_plan* plan);   /* device array between the stages: S[k][kx][j], complex */
/* stages, exposed for the slab-decomposed driver and for parity tests */
int mhh_pres_input (mhh_pres_plan* plan, const mhh_grid* g, const mhh_fields* f, double dt, void* p_packed, void* stream); /* pres_2.cxx:156-196, pres_4.cxx:256-317 */
int mhh_pres_solve (mhh_pres_plan* plan, const mhh_grid* g, const mhh_fields* f, void* p_packed, void* stream);            /* pres_2.cxx:267-362, pres_4.cxx:320-529 */
int mhh_pres_output(mhh_pres_plan* plan, const mhh_grid* g, const mhh_fields* f, void* stream);                            /* pres_2.cxx:365-387, pres_4.cxx:533-571 */
/* Pres::check_divergence (src/pres_2.cxx:391-422, pres_4.cxx:733-767); synchronises stream */
int mhh_pres_check_divergence(const mhh_grid* g, int order, const mhh_fields* f, void* work, double* div_out, void* stream);

/* plan-free forms of the pointwise stages (used by the slab driver; with npy > 1 the north-south halo of vt
 * is the caller's exchange, the east-west wrap of ut stays local) */
int mhh_pres_input_packed(const mhh_grid* g, int order, const mhh_fields* f, double dt, void* p_packed, void* stream);
int mhh_pres_output_order(const mhh_grid* g, int order, const mhh_fields* f, void* stream);

/* ---- Slab decomposition in y over the GPUs of one node (npx = 1, npy = N; the reference has no multi-GPU
 * mode, its CPU-MPI path is the model: src/boundary_cyclic.cxx:116-176, src/transpose.cxx:170-219,
 * src/fft.cxx:451-583, src/pres_2.cxx:297-299). The library packs/unpacks; the HOST issues the exchanges
 * (ring send/recv for halos, all-to-all for the x<->y transpose) with RCCL through torch.distributed.
 * With npy > 1, mhh_boundary_cyclic* only accept MHH_EDGE_EW and the operators that end in a cyclic fill
 * (exec_viscosity, evisc) do the east-west wrap only.                                                     */
/* halo buffers: [field][k][jgc][icells]; send_south = southernmost interior rows (-> south neighbour's north
 * ghosts), send_north = northernmost interior rows (-> north neighbour's south ghosts)                     */
unsigned long long mhh_halo_buffer_elems(const mhh_grid* g, int nfields);
int mhh_halo_pack_ns  (const mhh_grid* g, void* const* fields, int nfields, void* send_south, void* send_north, void* stream);
int mhh_halo_unpack_ns(const mhh_grid* g, void* const* fields, int nfields, const void* recv_from_south, const void* recv_from_north, void* stream);
/* partial exchange: rows_south rows travel south, rows_north rows travel north (0..jgc each), buffers
 * [field][k][rows][icells]. pres_2 input reads vt[j+1] only (src/pres_2.cxx:181,193): rows_south = 1, rows_north = 0;
 * pres_2 output reads p[j-1] only (:383-385): rows_south = 0, rows_north = 1. On unpack, recv_from_south holds the
 * rows_north rows the south neighbour sent north, recv_from_north the rows_south rows the north neighbour sent south. */
int mhh_halo_pack_rows  (const mhh_grid* g, void* const* fields, int nfields, int rows_south, int rows_north,
                         void* send_south, void* send_north, void* stream);
int mhh_halo_unpack_rows(const mhh_grid* g, void* const* fields, int nfields, int rows_south, int rows_north,
                         const void* recv_from_south, const void* recv_from_north, void* stream);
/* pres_2 split at the transposes. All-to-all buffers hold mhh_pres_slab_xbuf_elems() COMPLEX elements,
 * laid out [peer][k][jl][kxl] so that one equal-split all_to_all moves them.                               */
typedef struct mhh_pres_slab_plan mhh_pres_slab_plan;
int  mhh_pres_slab_plan_create(const mhh_grid* g, const void* host_dz, const void* host_dzhi,
                               const void* host_rhoref, const void* host_rhorefh, mhh_pres_slab_plan** out);
void mhh_pres_slab_plan_destroy(mhh_pres_slab_plan* plan);
unsigned long long mhh_pres_slab_xbuf_elems(const mhh_pres_slab_plan* plan);
void* mhh_pres_slab_packed(mhh_pres_slab_plan* plan);   /* plan-owned packed-divergence buffer (imax*jmax*kmax) */
int mhh_pres_fwd_x_pack       (mhh_pres_slab_plan* plan, const mhh_grid* g, void* p_packed, void* sendbuf, void* stream);
int mhh_pres_fwd_y_solve_bwd_y(mhh_pres_slab_plan* plan, const mhh_grid* g, void* recvbuf, void* sendbuf, void* stream);
int mhh_pres_bwd_x_unpack     (mhh_pres_slab_plan* plan, const mhh_grid* g, void* recvbuf, const mhh_fields* f, void* stream);
/* the same with Pres_2::output (src/pres_2.cxx:365-387) in the unpack kernel, for everything but vt on the southernmost row
 * (its p[j-1] lives on the south neighbour): exchange the one-row halo of p, then mhh_pres_output_south_row. Same bits as
 * mhh_pres_bwd_x_unpack + halo + mhh_pres_output_order, one pass over p, ut, vt, wt less. */
int mhh_pres_bwd_x_unpack_output(mhh_pres_slab_plan* plan, const mhh_grid* g, void* recvbuf, const mhh_fields* f, void* stream);
int mhh_pres_output_south_row (const mhh_grid* g, const mhh_fields* f, void* stream);
/* The same solve in nchunks slices of k (ktot % nchunks == 0), so that the host can overlap the all-to-all of slice c with the
 * transforms of slice c+1 (the reference's FFT::exec_forward transposes and transforms plane batches in turn as well,
 * src/fft.cxx:451-583). The all-to-all buffers are then laid out [slice][peer][k in slice][jl][kxl]: slice c is the equal-split
 * all-to-all of elements [c*n, (c+1)*n), n = mhh_pres_slab_xbuf_elems() / nchunks. Call order per solve:
 *   for c: fwd_x_pack_chunk(c) -> all-to-all(c);  for c: fwd_y_chunk(c);  solve_y;
 *   for c: bwd_y_chunk(c) -> all-to-all(c);        for c: bwd_x_chunk(c);  unpack_output_slab; p halo; output_south_row.      */
int mhh_pres_slab_set_chunks(mhh_pres_slab_plan* plan, int nchunks);          /* 1 = the unsliced calls above */
int mhh_pres_slab_chunks(const mhh_pres_slab_plan* plan);
int mhh_pres_fwd_x_pack_chunk(mhh_pres_slab_plan* plan, const mhh_grid* g, void* p_packed, void* sendbuf, int c, void* stream);
int mhh_pres_fwd_y_chunk     (mhh_pres_slab_plan* plan, const mhh_grid* g, void* recvbuf, int c, void* stream);
int mhh_pres_solve_y         (mhh_pres_slab_plan* plan, const mhh_grid* g, void* stream);
int mhh_pres_bwd_y_chunk     (mhh_pres_slab_plan* plan, const mhh_grid* g, void* sendbuf, int c, void* stream);
int mhh_pres_bwd_x_chunk     (mhh_pres_slab_plan* plan, const mhh_grid* g, void* recvbuf, int c, void* stream);
int mhh_pres_unpack_output_slab(mhh_pres_slab_plan* plan, const mhh_grid* g, const mhh_fields* f, void* stream);
/* The x stages of the slab solve with the transforms in LDS (csrc/pres_lds.h; power-of-two itot, jmax a multiple of 8;
 * mhh_pres_slab_has_lds tells): Pres_2::input + the transform along x (src/pres_2.cxx:156-196, src/fft.cxx:451-497) of k-slice c
 * written straight into segment c of the send buffer of Transpose::exec_xy (src/transpose.cxx:170-193), and the transform back
 * along x + p with its x halo + Pres_2::output (src/fft.cxx:540-583, src/pres_2.cxx:333-387) read straight from segment c of the
 * receive buffer of Transpose::exec_yx. They replace mhh_pres_input_packed + fwd_x_pack[_chunk] and bwd_x[_chunk] +
 * unpack_output_slab; the y stage, the one-row halo of p and mhh_pres_output_south_row stay. c = 0 with unsliced transposes.
 * Same tolerance as the staged form (different transforms, not the same bits). */
int mhh_pres_slab_has_lds(const mhh_pres_slab_plan* plan);
int mhh_pres_slab_lds_fwd(mhh_pres_slab_plan* plan, const mhh_grid* g, const mhh_fields* f, double dt, void* sendbuf, int c, void* stream);
int mhh_pres_slab_lds_bwd(mhh_pres_slab_plan* plan, const mhh_grid* g, const void* recvbuf, const mhh_fields* f, int c, void* stream);
/* the y stage of that form: the all-to-all buffers are laid out [slice][peer][k][kxl][row] (rows fastest: a column's rows from one rank
 * are one run), so the transforms along y read / write them directly (src/fft.cxx:499-538) around the Thomas sweeps (mhh_pres_solve_y).
 * Call order per solve: for c: lds_fwd(c) -> all-to-all(c);  for c: lds_fwd_y(c);  solve_y;  for c: lds_bwd_y(c) -> all-to-all(c);
 * for c: lds_bwd(c);  p halo (one row);  output_south_row. */
int mhh_pres_slab_lds_fwd_y(mhh_pres_slab_plan* plan, const mhh_grid* g, void* recvbuf, int c, void* stream);
int mhh_pres_slab_lds_bwd_y(mhh_pres_slab_plan* plan, const mhh_grid* g, void* sendbuf, int c, void* stream);

/* ---- Vertical ghost cells --------------------------------------------------------------
 * Boundary::set_ghost_cells: calc_ghost_cells_{bot,top}_{2nd,4th} (src/boundary.cxx:686-836); bc 0 = Dirichlet
 * (abot/atop), 1 = Neumann or flux (agradbot/agradtop); 2-D arrays are [ijcells]. set_ghost_cells_w (4th order
 * only): type 0 = Normal (:874-907), 1 = Conservation (:838-871).                                              */
int mhh_boundary_ghost_cells(const mhh_grid* g, int order, void* a, int bcbot, int bctop,
                             const void* abot, const void* agradbot, const void* atop, const void* agradtop, void* stream);
int mhh_boundary_ghost_cells_w(const mhh_grid* g, void* w, int type, void* stream);

/* ---- Timeloop RK3/RK4 substep (src/timeloop.cxx:250-334, src/timeloop.cu:35-122) -------- */
int mhh_rk_substep(const mhh_grid* g, int rkorder, int substep, double dt, void* a, void* at, void* stream);
/* pres->exec(sub_dt) followed by timeloop.exec() for u, v, w (src/model.cxx:411,484): the sub-step rides in the pres_2 kernel that
 * stores the corrected tendencies (two array passes per field less than a separate kernel); the bits of mhh_pres_exec followed by
 * mhh_rk_substep(u), (v), (w), which is also what it falls back to (pres_4, the last sub-step of a step). f->u, v, w are WRITTEN. */
int mhh_pres_exec_rk(mhh_pres_plan* plan, const mhh_grid* g, const mhh_fields* f, double sub_dt, int rkorder, int substep, double dt, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MHH_HIP_H */
