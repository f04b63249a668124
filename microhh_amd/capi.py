"""ctypes binding of the C ABI in include/mhh_hip.h (libmhh_hip.so, hand-written HIP for gfx950).

This is the ONLY compute path of the package: if the shared library is missing the import of a symbol
fails loudly -- there is no CPU or PyTorch fallback. (tests/ can bind the same signatures onto other
libraries -- the CPU emulation build of the kernels -- through ``bind``.)
"""
import ctypes as C
import os

from .grid import MhhGrid, MAX_SCALARS

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmhh_hip.so")

vp, ci, cd = C.c_void_p, C.c_int, C.c_double
GP = C.POINTER(MhhGrid)


class MhhFields(C.Structure):
    _fields_ = [("u", vp), ("v", vp), ("w", vp), ("ut", vp), ("vt", vp), ("wt", vp),
                ("nscalars", ci),
                ("s", vp * MAX_SCALARS), ("st", vp * MAX_SCALARS), ("svisc", cd * MAX_SCALARS),
                ("evisc", vp), ("p", vp), ("rhoref", vp), ("rhorefh", vp), ("visc", cd),
                ("u_fluxbot", vp), ("u_fluxtop", vp), ("v_fluxbot", vp), ("v_fluxtop", vp),
                ("s_fluxbot", vp * MAX_SCALARS), ("s_fluxtop", vp * MAX_SCALARS),
                ("dudz", vp), ("dvdz", vp), ("dbdz", vp), ("z0m", vp),
                ("s_fluxlimit", ci * MAX_SCALARS)]


class MhhDiffParams(C.Structure):
    _fields_ = [("cs", cd), ("tPr", cd), ("surface_model", ci), ("neutral", ci), ("N2", vp),
                ("th_for_N2", ci), ("thref", vp), ("grav", cd), ("mlen0", vp),
                ("buoyancy", ci), ("threfh", vp), ("evisc_ghost_rows", ci), ("mlen2", vp)]


FP = C.POINTER(MhhFields)
DP = C.POINTER(MhhDiffParams)
PLAN = vp

SIGNATURES = {
    "mhh_synchronize": (ci, [vp]),
    "mhh_version": (ci, []),
    "mhh_last_error": (C.c_char_p, []),
    "mhh_reduce_work_bytes": (C.c_ulonglong, []),
    "mhh_boundary_cyclic": (ci, [GP, vp, ci, vp]),
    "mhh_boundary_cyclic_2d": (ci, [GP, vp, vp]),
    "mhh_boundary_cyclic_u32": (ci, [GP, vp, ci, vp]),
    "mhh_boundary_cyclic_2d_u32": (ci, [GP, vp, vp]),
    "mhh_boundary_cyclic_n": (ci, [GP, C.POINTER(vp), ci, ci, vp]),
    "mhh_advec_u": (ci, [GP, ci, vp, vp, vp, vp, vp, vp, vp]),
    "mhh_advec_v": (ci, [GP, ci, vp, vp, vp, vp, vp, vp, vp]),
    "mhh_advec_w": (ci, [GP, ci, vp, vp, vp, vp, vp, vp, vp]),
    "mhh_advec_s": (ci, [GP, ci, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mhh_advec_s_lim": (ci, [GP, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mhh_advec_exec": (ci, [GP, ci, FP, vp]),
    "mhh_stat_visc_march_launches": (C.c_ulonglong, []),
    "mhh_selftest_sqrt_in_range": (ci, [C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_ulonglong), vp]),
    "mhh_diff_exec_viscosity_rows": (ci, [GP, ci, FP, DP, ci, ci, vp]),
    "mhh_diff_exec_viscosity_rows2": (ci, [GP, ci, FP, DP, ci, ci, ci, ci, vp]),
    "mhh_rhs_exec_rows": (ci, [GP, ci, ci, FP, DP, ci, ci, vp]),
    "mhh_rhs_exec_rows2": (ci, [GP, ci, ci, FP, DP, ci, ci, ci, ci, vp]),
    "mhh_stat_rhs44_march_launches": (C.c_ulonglong, []),
    "mhh_thermo_dry_buoyancy_tend": (ci, [GP, ci, vp, vp, vp, cd, vp]),
    "mhh_advec_cfl": (ci, [GP, ci, vp, vp, vp, cd, vp, C.POINTER(cd), vp]),
    "mhh_diff_c": (ci, [GP, ci, vp, vp, cd, vp]),
    "mhh_diff_w": (ci, [GP, ci, vp, vp, cd, vp]),
    "mhh_smag2_strain2": (ci, [GP, ci, vp, vp, vp, vp, vp, vp, vp]),
    "mhh_smag2_mlen0_host": (ci, [GP, cd, vp]),
    "mhh_smag2_mlen2_host": (ci, [GP, ci, ci, vp, cd, vp]),
    "mhh_smag2_evisc": (ci, [GP, ci, vp, vp, vp, vp, vp, cd, vp]),
    "mhh_smag2_evisc_neutral": (ci, [GP, ci, vp, vp, vp, vp, vp, cd, vp]),
    "mhh_smag2_diff_u": (ci, [GP, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, cd, vp]),
    "mhh_smag2_diff_v": (ci, [GP, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, cd, vp]),
    "mhh_smag2_diff_w": (ci, [GP, vp, vp, vp, vp, vp, vp, vp, cd, vp]),
    "mhh_smag2_diff_c": (ci, [GP, ci, vp, vp, vp, vp, vp, vp, vp, cd, cd, vp]),
    "mhh_smag2_dnmul": (ci, [GP, vp, cd, vp, C.POINTER(cd), vp]),
    "mhh_calc_N2": (ci, [GP, vp, vp, vp, cd, vp]),
    "mhh_diff_exec_viscosity": (ci, [GP, ci, FP, DP, vp]),
    "mhh_diff_exec": (ci, [GP, ci, FP, DP, vp]),
    "mhh_rhs_exec": (ci, [GP, ci, ci, FP, DP, vp]),
    "mhh_pres_plan_create": (ci, [GP, ci, vp, vp, vp, vp, vp, vp, C.POINTER(PLAN)]),
    "mhh_pres_plan_destroy": (None, [PLAN]),
    "mhh_pres_exec": (ci, [PLAN, GP, FP, cd, vp]),
    "mhh_pres_lds_stage": (ci, [PLAN, GP, FP, cd, ci, vp]),
    "mhh_pres_plan_has_lds_form": (ci, [PLAN]),
    "mhh_pres_exec_form": (ci, [PLAN]),
    "mhh_pres_plan_spectral": (vp, [PLAN]),
    "mhh_pres_input": (ci, [PLAN, GP, FP, cd, vp, vp]),
    "mhh_pres_solve": (ci, [PLAN, GP, FP, vp, vp]),
    "mhh_pres_output": (ci, [PLAN, GP, FP, vp]),
    "mhh_pres_check_divergence": (ci, [GP, ci, FP, vp, C.POINTER(cd), vp]),
    "mhh_rk_substep": (ci, [GP, ci, ci, cd, vp, vp, vp]),
    "mhh_pres_exec_rk": (ci, [PLAN, GP, FP, C.c_double, ci, ci, C.c_double, vp]),
    "mhh_boundary_ghost_cells": (ci, [GP, ci, vp, ci, ci, vp, vp, vp, vp, vp]),
    "mhh_boundary_ghost_cells_w": (ci, [GP, vp, ci, vp]),
    "mhh_pres_input_packed": (ci, [GP, ci, FP, cd, vp, vp]),
    "mhh_pres_output_order": (ci, [GP, ci, FP, vp]),
    "mhh_halo_buffer_elems": (C.c_ulonglong, [GP, ci]),
    "mhh_halo_pack_ns": (ci, [GP, C.POINTER(vp), ci, vp, vp, vp]),
    "mhh_halo_unpack_ns": (ci, [GP, C.POINTER(vp), ci, vp, vp, vp]),
    "mhh_halo_pack_rows": (ci, [GP, C.POINTER(vp), ci, ci, ci, vp, vp, vp]),
    "mhh_halo_unpack_rows": (ci, [GP, C.POINTER(vp), ci, ci, ci, vp, vp, vp]),
    "mhh_pres_slab_plan_create": (ci, [GP, vp, vp, vp, vp, C.POINTER(PLAN)]),
    "mhh_pres_slab_plan_destroy": (None, [PLAN]),
    "mhh_pres_slab_xbuf_elems": (C.c_ulonglong, [PLAN]),
    "mhh_pres_slab_packed": (vp, [PLAN]),
    "mhh_pres_fwd_x_pack": (ci, [PLAN, GP, vp, vp, vp]),
    "mhh_pres_fwd_y_solve_bwd_y": (ci, [PLAN, GP, vp, vp, vp]),
    "mhh_pres_bwd_x_unpack": (ci, [PLAN, GP, vp, FP, vp]),
    "mhh_pres_bwd_x_unpack_output": (ci, [PLAN, GP, vp, FP, vp]),
    "mhh_pres_output_south_row": (ci, [GP, FP, vp]),
    "mhh_pres_slab_set_chunks": (ci, [PLAN, ci]),
    "mhh_pres_slab_chunks": (ci, [PLAN]),
    "mhh_pres_fwd_x_pack_chunk": (ci, [PLAN, GP, vp, vp, ci, vp]),
    "mhh_pres_fwd_y_chunk": (ci, [PLAN, GP, vp, ci, vp]),
    "mhh_pres_solve_y": (ci, [PLAN, GP, vp]),
    "mhh_pres_bwd_y_chunk": (ci, [PLAN, GP, vp, ci, vp]),
    "mhh_pres_bwd_x_chunk": (ci, [PLAN, GP, vp, ci, vp]),
    "mhh_pres_unpack_output_slab": (ci, [PLAN, GP, FP, vp]),
    "mhh_pres_slab_has_lds": (ci, [PLAN]),
    "mhh_pres_slab_lds_fwd": (ci, [PLAN, GP, FP, C.c_double, vp, ci, vp]),
    "mhh_pres_slab_lds_bwd": (ci, [PLAN, GP, vp, FP, ci, vp]),
    "mhh_pres_slab_lds_fwd_y": (ci, [PLAN, GP, vp, ci, vp]),
    "mhh_pres_slab_lds_bwd_y": (ci, [PLAN, GP, vp, ci, vp]),
}


class MhhError(RuntimeError):
    """Non-zero status from the C ABI (the C++ adaptor throws std::runtime_error in the same place)."""


def bind(lib, required=True):
    """Attach argtypes/restype for every declared entry point; missing symbols raise."""
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if required:
                raise
            continue
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    """The HIP library. Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        path = os.environ.get("MHH_LIB") or LIB_PATH      # tuning sweeps point this at a variant build of the SAME sources
        if path != LIB_PATH:
            _lib = bind(C.CDLL(path))
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `python -m microhh_amd.build` (needs hipcc); "
                              "microhh_amd has no CPU fallback" % LIB_PATH)
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def check(rc, library=None):
    if rc != 0:
        msg = (library or lib()).mhh_last_error()
        raise MhhError("mhh status %d: %s" % (rc, msg.decode() if msg else "?"))
