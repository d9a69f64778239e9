"""ctypes binding of include/ffm.h (libffm.so, HIP kernels for gfx950).

Host-side mirror of the OpenFOAM interface on the reference's hot path:
``lduMatrix`` (Amul/Tmul/sumA/residual, solver/pEqn.H:5,39) and
``lduMatrix::solver::New(...)->solve(psi, source)`` with the fvSolution keywords
(solver, preconditioner/smoother, tolerance, relTol, minIter, maxIter, nSweeps;
cases/steckler/system/fvSolution:21-61).  Device fields are torch CUDA tensors
(fp64); torch is plumbing only (memory + process launch).
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.environ.get("FFM_LIB") or os.path.join(_HERE, "lib", "libffm.so")      # FFM_LIB: kernel experiments only
_HDR = os.path.join(_ROOT, "include", "ffm.h")

SOLVERS = {"PCG": 0, "PBiCGStab": 1, "PBiCG": 2, "diagonal": 3, "smoothSolver": 4}
PRECONDS = {"none": 0, "DIC": 1, "DILU": 2, "GaussSeidel": 3, "symGaussSeidel": 4, "diagonal": 5}


class FfmError(RuntimeError):
    pass


class Perf(C.Structure):
    _fields_ = [("initialResidual", C.c_double), ("finalResidual", C.c_double),
                ("nIterations", C.c_int), ("converged", C.c_int), ("singular", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


HOST_ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)
HOST_EXCHANGE2_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.POINTER(C.c_double), C.POINTER(C.c_double))
HOST_EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double))


def libpath():
    return _SO


def build(force=False):
    """Compile csrc/*.hip for gfx950 into lib/libffm.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return _SO


def declared_symbols():
    """Function names declared in include/ffm.h."""
    txt = open(_HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ffm_[a-z0-9_]+)\s*\(", txt)) - {"ffm_host_allreduce_fn", "ffm_host_exchange_fn"})


def exported_symbols():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _SO], text=True)
    return sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("ffm_"))


_lib = None


def lib():
    """Load libffm.so.  torch is imported first so that the HIP/RCCL runtime libraries
    already in the process (same SONAMEs) are the ones libffm binds to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise FfmError("libffm.so is not built (%s); run __graft_entry__.build() -- there is no CPU fallback" % _SO)
    import torch  # noqa: F401
    L = C.CDLL(_SO, mode=C.RTLD_GLOBAL)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.POINTER(C.c_int)   # device pointers travel as void*
    hp = C.POINTER(C.c_double)
    sig = {
        "ffm_ctx_create": ([C.c_int, vp, C.POINTER(vp)], C.c_int),
        "ffm_ctx_destroy": ([vp], C.c_int),
        "ffm_ctx_sync": ([vp], C.c_int),
        "ffm_ctx_stream": ([vp], vp),
        "ffm_last_error": ([], C.c_char_p),
        "ffm_version": ([], C.c_char_p),
        "ffm_ldu_create": ([vp, C.c_int, C.c_int, ip, ip, C.POINTER(vp)], C.c_int),
        "ffm_ldu_destroy": ([vp], C.c_int),
        "ffm_ldu_ncells": ([vp], C.c_int),
        "ffm_ldu_nfaces": ([vp], C.c_int),
        "ffm_ldu_nlevels": ([vp], C.c_int),
        "ffm_tile_hint_from_centres": ([C.c_int, hp, C.c_int, ip], C.c_int),
        "ffm_ldu_sweep_mode": ([vp], C.c_int),
        "ffm_ldu_is_native_order": ([vp], C.c_int),
        "ffm_ldu_get_cell_order": ([vp, ip], C.c_int),
        "ffm_renumber_levels": ([C.c_int, C.c_int, ip, ip, ip, ip], C.c_int),
        "ffm_ldu_set_coeffs": ([vp, hp, hp, hp], C.c_int),
        "ffm_ldu_set_coeffs_d": ([vp, dp, dp, dp], C.c_int),
        "ffm_ldu_n_native_faces": ([vp], C.c_int),
        "ffm_ldu_get_face_map": ([vp, ip], C.c_int),
        "ffm_ldu_set_coeffs_native_d": ([vp, dp, dp, dp], C.c_int),
        "ffm_ldu_bind_coeffs_native_d": ([vp, dp, dp, dp, C.c_int], C.c_int),
        "ffm_ldu_set_interfaces": ([vp, C.c_int, ip, C.POINTER(ip), C.POINTER(hp), C.POINTER(hp), ip], C.c_int),
        "ffm_ldu_set_global_cells": ([vp, C.c_long], C.c_int),
        "ffm_spmv": ([vp, dp, dp], C.c_int),
        "ffm_tmul": ([vp, dp, dp], C.c_int),
        "ffm_sumA": ([vp, dp], C.c_int),
        "ffm_residual": ([vp, dp, dp, dp], C.c_int),
        "ffm_precond_setup": ([vp, C.c_int, dp], C.c_int),
        "ffm_precond_apply": ([vp, C.c_int, C.c_int, dp, dp], C.c_int),
        "ffm_gs_smooth": ([vp, C.c_int, C.c_int, dp, dp], C.c_int),
        "ffm_solve_d": ([vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, dp, dp,
                         C.POINTER(Perf)], C.c_int),
        "ffm_solve": ([vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, hp, hp,
                       C.POINTER(Perf)], C.c_int),
        "ffm_solve_multi_d": ([vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_void_p), dp, dp,
                               C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(Perf)], C.c_int),
        "ffm_ldu_unbind_coeffs": ([vp], C.c_int),
        "ffm_bench_spmv": ([vp, dp, dp, C.c_int, hp], C.c_int),
        "ffm_bench_precond": ([vp, C.c_int, dp, dp, C.c_int, hp], C.c_int),
        "ffm_debug_tile_trace": ([vp, C.c_void_p, C.c_int], C.c_int),
        "ffm_debug_set_sweep_ticket": ([vp, C.c_uint], C.c_int),
        "ffm_ldu_set_exchange_tags": ([vp, C.c_int, C.c_int, ip], C.c_int),
        "ffm_field_binary": ([vp, C.c_int, C.c_long, dp, dp, dp], C.c_int),
        "ffm_field_scalar": ([vp, C.c_int, C.c_long, dp, C.c_double, C.c_int, dp], C.c_int),
        "ffm_field_unary": ([vp, C.c_int, C.c_long, dp, dp], C.c_int),
        "ffm_field_eval": ([vp, C.c_long, C.c_int, C.POINTER(C.c_void_p), C.c_int, hp, C.c_int, C.POINTER(C.c_ushort), dp], C.c_int),
        "ffm_reduce_sum": ([vp, dp, C.c_long, hp], C.c_int),
        "ffm_reduce_min": ([vp, dp, C.c_long, hp], C.c_int),
        "ffm_reduce_max": ([vp, dp, C.c_long, hp], C.c_int),
        "ffm_reduce_dot": ([vp, dp, dp, C.c_long, hp], C.c_int),
        "ffm_reduce_summag": ([vp, dp, C.c_long, hp], C.c_int),
        "ffm_mesh_create": ([vp, hp, hp, hp, hp, hp, hp, C.c_int, ip, C.POINTER(ip), C.POINTER(hp), C.POINTER(hp), C.POINTER(vp)], C.c_int),
        "ffm_mesh_destroy": ([vp], C.c_int),
        "ffm_mesh_set_face_centres": ([vp, hp], C.c_int),
        "ffm_mesh_set_nonorth_correction": ([vp, hp], C.c_int),
        "ffm_pyro_create": ([vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(vp)], C.c_int),
        "ffm_pyro_set_solids": ([vp, hp, hp], C.c_int),
        "ffm_pyro_set_reaction": ([vp, C.c_double, C.c_double, C.c_double, C.c_double], C.c_int),
        "ffm_pyro_step": ([vp, C.c_double, dp, C.c_int, C.c_double], C.c_int),
        "ffm_pyro_get": ([vp, C.c_char_p, hp], C.c_int),
        "ffm_pyro_surface_T_d": ([vp], C.c_void_p),
        "ffm_pyro_phiGas_d": ([vp], C.c_void_p),
        "ffm_pyro_couple_d": ([vp, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp], C.c_int),
        "ffm_pyro_qSurf_d": ([vp], C.c_void_p),
        "ffm_pyro_diff_no": ([vp, C.c_double, hp], C.c_int),
        "ffm_pyro_set_model": ([vp, C.c_int, C.c_int, C.c_int], C.c_int),
        "ffm_pyro_set_back": ([vp, C.c_int, C.c_double, C.c_double], C.c_int),
        "ffm_pyro_set_surface_radiation": ([vp] + [C.c_double] * 4, C.c_int),
        "ffm_pyro_evolve_d": ([vp, C.c_double, dp, dp, dp, dp, C.c_double, C.c_double], C.c_int),
        "ffm_pyro_gas_side_d": ([vp, dp, dp, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, dp], C.c_int),
        "ffm_pyro_destroy": ([vp], C.c_int),
        "ffm_thermo_create": ([vp, C.c_int] + [hp] * 8 + [C.c_double, C.POINTER(vp)], C.c_int),
        "ffm_thermo_correct_d": ([vp, C.c_long, C.POINTER(vp), dp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_thermo_he_d": ([vp, C.c_long, C.POINTER(vp), dp, dp], C.c_int),
        "ffm_thermo_properties_d": ([vp, C.c_long, C.POINTER(vp), dp, dp, dp, dp], C.c_int),
        "ffm_thermo_Cp_d": ([vp, C.c_long, C.POINTER(vp), dp, dp], C.c_int),
        "ffm_thermo_destroy": ([vp], C.c_int),
        "ffm_edc_correct_d": ([vp, C.c_long] + [dp] * 6 + [C.c_double] * 7 + [dp, dp], C.c_int),
        "ffm_les_keqn_nut_d": ([vp, C.c_long, C.c_double, C.c_double, dp, dp, dp, dp, dp], C.c_int),
        "ffm_gamg_face_area_pair_weights": ([C.c_int, hp, hp], C.c_int),
        "ffm_ctx_set_gamg_forward": ([vp, C.c_int], C.c_int),
        "ffm_ctx_gamg_forward": ([vp], C.c_int),
        "ffm_gamg_create": ([vp, vp, C.c_int, C.c_int, ip, ip, hp, C.c_int, C.c_int, C.POINTER(vp)], C.c_int),
        "ffm_gamg_set_sweeps": ([vp, C.c_int, C.c_int, C.c_int], C.c_int),
        "ffm_gamg_set_matrix_d": ([vp, dp, dp, dp], C.c_int),
        "ffm_gamg_set_matrix_native_d": ([vp, dp, dp, dp], C.c_int),
        "ffm_gamg_solve_d": ([vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, C.POINTER(Perf)], C.c_int),
        "ffm_gamg_nlevels": ([vp], C.c_int),
        "ffm_gamg_level_size": ([vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "ffm_gamg_get_level_addressing": ([vp, C.c_int, ip, ip], C.c_int),
        "ffm_gamg_get_level_coeffs": ([vp, C.c_int, hp, hp, hp], C.c_int),
        "ffm_gamg_coarsest_solves": ([vp, C.c_int, C.POINTER(Perf)], C.c_int),
        "ffm_gamg_destroy": ([vp], C.c_int),
        "ffm_partition_rcb": ([C.c_int, hp, C.c_int, ip], C.c_int),
        "ffm_partition_graph": ([C.c_int, C.c_int, ip, ip, C.c_int, ip], C.c_int),
        "ffm_subdomain_create": ([C.c_int, C.c_int, ip, ip, ip, C.c_int, C.c_int, C.POINTER(vp)], C.c_int),
        "ffm_subdomain_destroy": ([vp], C.c_int),
        "ffm_subdomain_sizes": ([vp] + [C.POINTER(C.c_int)] * 6, C.c_int),
        "ffm_subdomain_cells": ([vp, ip], C.c_int),
        "ffm_subdomain_faces": ([vp, ip, ip, ip, ip], C.c_int),
        "ffm_subdomain_exchange": ([vp, ip, ip, ip, ip, ip], C.c_int),
        "ffm_subdomain_cut_faces": ([vp, ip, ip, ip, ip], C.c_int),
        "ffm_polymesh_read": ([C.c_char_p, C.POINTER(vp)], C.c_int),
        "ffm_polymesh_destroy": ([vp], C.c_int),
        "ffm_polymesh_sizes": ([vp] + [C.POINTER(C.c_int)] * 5, C.c_int),
        "ffm_polymesh_addressing": ([vp, ip, ip], C.c_int),
        "ffm_polymesh_geometry": ([vp] + [C.POINTER(C.c_double)] * 8, C.c_int),
        "ffm_polymesh_patch": ([vp, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "ffm_polymesh_patch_processor": ([vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "ffm_polymesh_patch_geometry": ([vp, C.c_int, ip] + [C.POINTER(C.c_double)] * 3, C.c_int),
        "ffm_fv_lust_correction": ([vp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fvc_snGrad_correction": ([vp, dp, dp, dp, dp], C.c_int),
        "ffm_fv_linear_upwind_correction": ([vp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fvm_relax": ([vp, C.c_double, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fv_filtered_linear2V_weights": ([vp, C.c_double, C.c_double, dp] + [C.POINTER(C.c_void_p)] * 4 + [dp], C.c_int),
        "ffm_fvc_grad_multi": ([vp, C.c_int] + [C.POINTER(C.c_void_p)] * 5, C.c_int),
        "ffm_fvm_scalar_transport_multi": ([vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp]
                                           + [C.POINTER(C.c_void_p)] * 16, C.c_int),
        "ffm_fvm_scalar_transport_multi_w": ([vp, C.c_int, dp, C.c_double, dp, dp, dp, dp, dp, dp] + [C.POINTER(C.c_void_p)] * 12, C.c_int),
        "ffm_fv_multivariate_weights": ([vp, C.c_int, C.POINTER(C.c_int), C.c_double, C.c_double, C.c_double, dp] + [C.POINTER(C.c_void_p)] * 4 + [dp], C.c_int),
        "ffm_fvm_lust_source3": ([vp, C.c_double, dp, dp] + [C.POINTER(C.c_void_p)] * 5, C.c_int),
        "ffm_pc_phig": ([vp, dp, dp, dp, dp], C.c_int),
        "ffm_ue_buoyancy_flux": ([vp, dp, dp, dp, dp], C.c_int),
        "ffm_pc_phiHbyA": ([vp, dp, dp, dp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_pc_flux": ([vp, dp, dp, dp, dp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fvc_div_dev2T_gradU": ([vp, C.POINTER(vp), dp, dp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)], C.c_int),
        "ffm_les_keqn_G": ([vp, C.POINTER(vp), dp, dp], C.c_int),
        "ffm_fvm_HbyA3": ([vp, dp, dp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), dp, C.POINTER(vp)], C.c_int),
        "ffm_fvc_flux_rho": ([vp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fvm_pressure_eqn": ([vp, C.c_double] + [dp] * 6 + [C.c_double] + [dp] * 9, C.c_int),
        "ffm_mesh_nboundary": ([vp], C.c_int),
        "ffm_mesh_nnative": ([vp], C.c_int),
        "ffm_faces_to_native": ([vp, hp, dp], C.c_int),
        "ffm_faces_from_native": ([vp, dp, hp], C.c_int),
        "ffm_fvc_interpolate": ([vp, dp, dp, dp], C.c_int),
        "ffm_fvc_snGrad": ([vp, dp, dp], C.c_int),
        "ffm_fvc_snGrad_b": ([vp, dp, dp, dp], C.c_int),
        "ffm_fvc_flux": ([vp, dp, dp, dp, dp], C.c_int),
        "ffm_fvc_surface_integrate": ([vp, dp, dp, dp], C.c_int),
        "ffm_fvc_surface_sum": ([vp, dp, dp, dp], C.c_int),
        "ffm_fvc_grad": ([vp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fvc_reconstruct": ([vp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fv_limited_weights": ([vp, C.c_int, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fv_limited_limiter": ([vp, C.c_int, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp, C.c_int], C.c_int),
        "ffm_fv_weights_from_limiter": ([vp, dp, dp, dp], C.c_int),
        "ffm_fvm_transport": ([vp, C.c_double, dp, dp, dp, dp, C.c_int, dp, dp, dp], C.c_int),
        "ffm_fvm_boundary_coeffs": ([vp, dp, dp, C.c_int, dp, dp, dp, dp, dp], C.c_int),
        "ffm_bc_values": ([vp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fvm_add_boundary": ([vp, dp, dp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fvm_A": ([vp, C.c_int, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fvm_H": ([vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_fvm_flux": ([vp, dp, dp, dp, dp, dp, dp, dp], C.c_int),
        "ffm_plume_create": ([vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(vp)], C.c_int),
        "ffm_plume_destroy": ([vp], C.c_int),
        "ffm_plume_step": ([vp], C.c_int),
        "ffm_plume_set_tight": ([vp, C.c_int], C.c_int),
        "ffm_plume_set_solvers": ([vp, C.c_int], C.c_int),
        "ffm_plume_set_radiation": ([vp, C.c_int, C.c_int, C.c_int, vp, vp], C.c_int),
        "ffm_plume_set_radiation_model": ([vp, C.c_double, C.c_double, C.c_double], C.c_int),
        "ffm_plume_set_initial_state": ([vp, C.POINTER(C.c_void_p), hp, hp, hp, C.c_double], C.c_int),
        "ffm_plume_override_mv_weights": ([vp, hp], C.c_int),
        "ffm_plume_ncells": ([vp], C.c_int),
        "ffm_plume_nfaces": ([vp], C.c_int),
        "ffm_plume_get_field": ([vp, C.c_char_p, hp], C.c_int),
        "ffm_plume_nsolves": ([vp], C.c_int),
        "ffm_plume_get_solve": ([vp, C.c_int, C.c_char_p, C.POINTER(Perf)], C.c_int),
        "ffm_plume_get_raw": ([vp, C.c_char_p, hp, C.c_long], C.c_long),
        "ffm_plume_ldu": ([vp], vp),
        "ffm_plume_mesh": ([vp], vp),
        "ffm_fv_multivariate_weights_tiled": ([vp, C.c_int, C.POINTER(C.c_int), C.c_double, C.c_double, C.c_double, dp] + [C.POINTER(C.c_void_p)] * 2 + [dp], C.c_int),
        "ffm_comm_unique_id": ([vp], C.c_int),
        "ffm_comm_init": ([vp, C.c_int, C.c_int, vp], C.c_int),
        "ffm_comm_init_host": ([vp, C.c_int, C.c_int, vp, HOST_ALLREDUCE_FN, HOST_EXCHANGE_FN], C.c_int),
        "ffm_comm_set_host_exchange2": ([vp, HOST_EXCHANGE2_FN], C.c_int),
        "ffm_plume_create_block": ([vp, C.c_int, C.c_int, C.c_int, ip, ip, ip, C.c_double, C.c_double, C.POINTER(vp)], C.c_int),
        "ffm_ldu_create_hint": ([vp, C.c_int, C.c_int, C.c_int, ip, ip, ip, C.POINTER(vp)], C.c_int),
        "ffm_renumber_hint": ([C.c_int, C.c_int, C.c_int, ip, ip, ip, ip, ip], C.c_int),
        "ffm_ldu_create_ext": ([vp, C.c_int, C.c_int, C.c_int, ip, ip, C.POINTER(vp)], C.c_int),
        "ffm_renumber_levels_ext": ([C.c_int, C.c_int, C.c_int, ip, ip, ip, ip], C.c_int),
        "ffm_ldu_set_ghost_exchange": ([vp, C.c_int, ip, ip, ip, ip], C.c_int),
        "ffm_halo_refresh_d": ([vp, dp], C.c_int),
        "ffm_ldu_nowned": ([vp], C.c_int),
        "ffm_comm_rank": ([vp], C.c_int),
        "ffm_comm_size": ([vp], C.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes, fn.restype = args, res
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise FfmError("%s failed (%d): %s" % (what, rc, lib().ffm_last_error().decode()))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _hp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def tile_hint_from_centres(C_, tileCells=0):
    """group hint (int32[nCells]) for ffm.renumber_levels / lduMatrix from cell centres C_[3][nCells] (ffm_tile_hint_from_centres)"""
    C_ = np.ascontiguousarray(C_, np.float64)
    n = C_.shape[1]
    hint = np.empty(n, np.int32)
    _check(lib().ffm_tile_hint_from_centres(n, _hp(C_), int(tileCells), _ip(hint)), "ffm_tile_hint_from_centres")
    return hint


def renumber_levels(nCells, lowerAddr, upperAddr, groupHint=None, nGhost=0):
    """The library's preferred cell order (pure host code in libffm): returns (newToOldCell, newToOldFace).
    nGhost > 0: nCells owned cells followed by nGhost ghost cells (ffm_renumber_levels_ext; the ghosts keep their places)."""
    l = np.ascontiguousarray(lowerAddr, np.int32)
    u = np.ascontiguousarray(upperAddr, np.int32)
    c2 = np.empty(nCells + nGhost, np.int32)
    f2 = np.empty(len(l), np.int32)
    if nGhost and groupHint is None:
        _check(lib().ffm_renumber_levels_ext(nCells, nGhost, len(l), _ip(l), _ip(u), _ip(c2), _ip(f2)), "ffm_renumber_levels_ext")
    elif groupHint is None:
        _check(lib().ffm_renumber_levels(nCells, len(l), _ip(l), _ip(u), _ip(c2), _ip(f2)), "ffm_renumber_levels")
    else:
        gh = np.ascontiguousarray(groupHint, np.int32)
        _check(lib().ffm_renumber_hint(nCells, nGhost, len(l), _ip(l), _ip(u), _ip(gh), _ip(c2), _ip(f2)), "ffm_renumber_hint")
    return c2, f2


class Context:
    """One GPU, one stream (ffm_ctx)."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise FfmError("no GPU visible: the HIP path has no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(device)
        h = C.c_void_p()
        _check(lib().ffm_ctx_create(device, None, C.byref(h)), "ffm_ctx_create")
        self.h = h
        self._cb = None

    def close(self):
        if getattr(self, "h", None):
            lib().ffm_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        _check(lib().ffm_ctx_sync(self.h), "ffm_ctx_sync")

    def to_device(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a, np.float64), device=self.device)

    def empty(self, n):
        return self.torch.empty(int(n), dtype=self.torch.float64, device=self.device)

    # element-wise field algebra (the Foam layer's operators): op = FFM_OP_* / FFM_UN_* of include/ffm.h
    def field_binary(self, op, a, b):
        out = self.empty(a.numel()); self._ready()
        _check(lib().ffm_field_binary(self.h, op, a.numel(), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr())), "ffm_field_binary")
        return out

    def field_scalar(self, op, a, s, scalarFirst=False):
        out = self.empty(a.numel()); self._ready()
        _check(lib().ffm_field_scalar(self.h, op, a.numel(), C.c_void_p(a.data_ptr()), float(s), int(scalarFirst), C.c_void_p(out.data_ptr())), "ffm_field_scalar")
        return out

    def field_unary(self, op, a):
        out = self.empty(a.numel()); self._ready()
        _check(lib().ffm_field_unary(self.h, op, a.numel(), C.c_void_p(a.data_ptr()), C.c_void_p(out.data_ptr())), "ffm_field_unary")
        return out

    def field_eval(self, arrays, imm, program):
        """program: [("load", k) | ("imm", k) | ("binary", op) | ("unary", op)] in postfix order (ffm_field_eval)"""
        kinds = {"load": 1, "imm": 2, "binary": 3, "unary": 4}
        code = (C.c_ushort * len(program))(*[kinds[k] << 12 | int(a) for k, a in program])
        ptrs = (C.c_void_p * max(len(arrays), 1))(*[t.data_ptr() for t in arrays])
        im = np.ascontiguousarray(imm, np.float64) if len(imm) else np.zeros(1)
        n = arrays[0].numel()
        out = self.empty(n); self._ready()
        _check(lib().ffm_field_eval(self.h, n, len(arrays), ptrs, len(imm), im.ctypes.data_as(C.POINTER(C.c_double)), len(program), code, C.c_void_p(out.data_ptr())), "ffm_field_eval")
        return out

    def zeros(self, n):
        z = self.torch.zeros(int(n), dtype=self.torch.float64, device=self.device)
        self._ready()               # the fill runs on torch's stream: complete before libffm's stream writes into the tensor
        return z

    def _ready(self):
        # tensors produced on torch's stream must be complete before libffm's stream reads them
        self.torch.cuda.current_stream().synchronize()

    def comm_init_rccl(self, rank, nRanks, unique_id_bytes):
        buf = C.create_string_buffer(bytes(unique_id_bytes), 128)
        _check(lib().ffm_comm_init(self.h, rank, nRanks, buf), "ffm_comm_init")

    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().ffm_comm_unique_id(buf), "ffm_comm_unique_id")
        return buf.raw

    def comm_init_host(self, rank, nRanks, allreduce, exchange, exchange_var=None):
        """allreduce(np_array_inplace, op) / exchange(sizes, ranks, offsets, send_np, recv_np) / exchange_var(ranks, sends, recvs):
        the last one moves a different number of values each way per neighbour (general partitions: firefoam-dev_amd/decompose.py);
        without it the ghost exchange needs equal counts (structured blocks)."""
        def _ar(user, vals, n, op):
            allreduce(np.ctypeslib.as_array(vals, shape=(n,)), op)

        def _ex(user, nP, size, rank_, off, send, recv):
            sizes = [size[i] for i in range(nP)]
            ranks = [rank_[i] for i in range(nP)]
            offs = [off[i] for i in range(nP)]
            tot = sum(sizes)
            exchange(sizes, ranks, offs, np.ctypeslib.as_array(send, shape=(tot,)),
                     np.ctypeslib.as_array(recv, shape=(tot,)))
        def _ex2(user, nN, rank_, soff, roff, send, recv):
            ranks = [rank_[i] for i in range(nN)]
            so = [soff[i] for i in range(nN + 1)]
            ro = [roff[i] for i in range(nN + 1)]
            sizes_s = [so[i + 1] - so[i] for i in range(nN)]
            sb = np.ctypeslib.as_array(send, shape=(max(so[-1], 1),))
            rb = np.ctypeslib.as_array(recv, shape=(max(ro[-1], 1),))
            sends = [np.ascontiguousarray(sb[so[q]:so[q + 1]]) for q in range(nN)]
            recvs = [np.empty(ro[q + 1] - ro[q]) for q in range(nN)]
            if exchange_var is not None:
                exchange_var(ranks, sends, recvs)              # all neighbours at once
            else:
                # the equal-count callback, one neighbour at a time
                for q in range(nN):
                    if len(sends[q]) != len(recvs[q]):
                        raise FfmError("ghost exchange with %d values out and %d in for rank %d needs comm_init_host(..., exchange_var=...)"
                                       % (len(sends[q]), len(recvs[q]), ranks[q]))
                    exchange([len(sends[q])], [ranks[q]], [0], sends[q], recvs[q])
            for q in range(nN):
                rb[ro[q]:ro[q + 1]] = recvs[q]
        self._cb = (HOST_ALLREDUCE_FN(_ar), HOST_EXCHANGE_FN(_ex), HOST_EXCHANGE2_FN(_ex2))
        _check(lib().ffm_comm_init_host(self.h, rank, nRanks, None, self._cb[0], self._cb[1]), "ffm_comm_init_host")
        _check(lib().ffm_comm_set_host_exchange2(self.h, self._cb[2]), "ffm_comm_set_host_exchange2")

    def _reduce(self, fn, *tensors):
        self._ready()
        out = C.c_double()
        args = [self.h] + [C.c_void_p(t.data_ptr()) for t in tensors] + [tensors[0].numel(), C.byref(out)]
        _check(fn(*args), "ffm_reduce")
        return out.value

    def gSum(self, x):
        return self._reduce(lib().ffm_reduce_sum, x)

    def gMin(self, x):
        return self._reduce(lib().ffm_reduce_min, x)

    def gMax(self, x):
        return self._reduce(lib().ffm_reduce_max, x)

    def gSumProd(self, x, y):
        return self._reduce(lib().ffm_reduce_dot, x, y)

    def gSumMag(self, x):
        return self._reduce(lib().ffm_reduce_summag, x)


class lduMatrix:
    """Device lduMatrix over lduAddressing (lowerAddr, upperAddr)."""

    def __init__(self, ctx, nCells, lowerAddr, upperAddr, groupHint=None, nGhost=0):
        """nGhost > 0: one rank of a decomposed case (ffm_ldu_create_ext): nCells owned cells followed by nGhost ghost cells;
        cell fields then have nCells + nGhost entries"""
        self.ctx = ctx
        l = np.ascontiguousarray(lowerAddr, np.int32)
        u = np.ascontiguousarray(upperAddr, np.int32)
        h = C.c_void_p()
        if groupHint is None and not nGhost:
            _check(lib().ffm_ldu_create(ctx.h, int(nCells), len(l), _ip(l), _ip(u), C.byref(h)), "ffm_ldu_create")
        else:
            gh = None if groupHint is None else np.ascontiguousarray(groupHint, np.int32)
            _check(lib().ffm_ldu_create_hint(ctx.h, int(nCells), int(nGhost), len(l), _ip(l), _ip(u), None if gh is None else _ip(gh), C.byref(h)),
                   "ffm_ldu_create_hint")
        self.h = h
        self.nOwned = int(nCells)
        self.nCells, self.nFaces = int(nCells) + int(nGhost), len(l)

    def set_ghost_exchange(self, nbrRank, sendCount, sendCells, recvCount, tags=None, globalCells=None):
        """ffm_ldu_set_ghost_exchange (+ pair tags, + the global cell count for gAverage): see firefoam-dev_amd/decompose.py"""
        a = [np.ascontiguousarray(x, np.int32) for x in (nbrRank, sendCount, sendCells, recvCount)]
        _check(lib().ffm_ldu_set_ghost_exchange(self.h, len(a[0]), _ip(a[0]), _ip(a[1]), _ip(a[2]), _ip(a[3])), "ffm_ldu_set_ghost_exchange")
        if tags is not None:
            t = np.ascontiguousarray(tags, np.int32)
            _check(lib().ffm_ldu_set_exchange_tags(self.h, 1, len(t), _ip(t)), "ffm_ldu_set_exchange_tags")
        if globalCells is not None:
            _check(lib().ffm_ldu_set_global_cells(self.h, int(globalCells)), "ffm_ldu_set_global_cells")
        return self

    def close(self):
        if getattr(self, "h", None):
            lib().ffm_ldu_destroy(self.h)
            self.h = None

    @property
    def nLevels(self):
        return lib().ffm_ldu_nlevels(self.h)

    @property
    def sweep_mode(self):
        """0 level-scheduled, 2 tiled wavefront"""
        return lib().ffm_ldu_sweep_mode(self.h)

    @property
    def nNative(self):
        """entries of a face array in the library's native layout (sliced owner-ELL incl. padding)"""
        return lib().ffm_ldu_n_native_faces(self.h)

    def face_map(self):
        """callerToNative[f]: the native index of the caller's face f (ffm_ldu_get_face_map; native cell order only)"""
        m = np.empty(self.nFaces, np.int32)
        _check(lib().ffm_ldu_get_face_map(self.h, _ip(m)), "ffm_ldu_get_face_map")
        return m

    def bind_coeffs_native(self, diag, upper, lower=None):
        """zero-copy: the matrix reads device tensors in the library's native layout (diag [nCells], upper / lower [nNative])
        until the next set / bind call; the caller keeps them alive (ffm_ldu_bind_coeffs_native_d)"""
        self._bound = (diag, upper, lower)
        _check(lib().ffm_ldu_bind_coeffs_native_d(self.h, C.c_void_p(diag.data_ptr()), C.c_void_p(upper.data_ptr()),
                                                  None if lower is None else C.c_void_p(lower.data_ptr()), 0), "ffm_ldu_bind_coeffs_native_d")
        return self

    def debug_set_sweep_ticket(self, value):
        """tests: preset the group ticket counter of the tiled sweeps (each sweep launch zeroes it again)"""
        _check(lib().ffm_debug_set_sweep_ticket(self.h, int(value) & 0xFFFFFFFF), "ffm_debug_set_sweep_ticket")

    @property
    def native_order(self):
        return bool(lib().ffm_ldu_is_native_order(self.h))

    def cell_order(self):
        o = np.empty(self.nCells, np.int32)
        _check(lib().ffm_ldu_get_cell_order(self.h, _ip(o)), "ffm_ldu_get_cell_order")
        return o

    def set_coeffs(self, diag, upper, lower=None):
        """diag()/upper()/lower(); numpy (host) or torch CUDA tensors (device)."""
        T = self.ctx.torch
        if T.is_tensor(diag):
            self.ctx._ready()
            lo = None if lower is None else C.c_void_p(lower.data_ptr())
            _check(lib().ffm_ldu_set_coeffs_d(self.h, C.c_void_p(diag.data_ptr()), C.c_void_p(upper.data_ptr()), lo),
                   "ffm_ldu_set_coeffs_d")
            self.ctx.sync()
        else:
            d = np.ascontiguousarray(diag, np.float64)
            u = np.ascontiguousarray(upper, np.float64)
            lo = None if lower is None else np.ascontiguousarray(lower, np.float64)
            _check(lib().ffm_ldu_set_coeffs(self.h, _hp(d), _hp(u), None if lo is None else _hp(lo)), "ffm_ldu_set_coeffs")
        return self

    def set_interfaces(self, faceCells, bouCoeffs, neighbRank, intCoeffs=None, globalCells=None):
        n = len(faceCells)
        fc = [np.ascontiguousarray(a, np.int32) for a in faceCells]
        bc = [np.ascontiguousarray(a, np.float64) for a in bouCoeffs]
        ic = bc if intCoeffs is None else [np.ascontiguousarray(a, np.float64) for a in intCoeffs]
        sizes = (C.c_int * n)(*[len(a) for a in fc])
        ranks = (C.c_int * n)(*[int(r) for r in neighbRank])
        fcp = (C.POINTER(C.c_int) * n)(*[_ip(a) for a in fc])
        bcp = (C.POINTER(C.c_double) * n)(*[_hp(a) for a in bc])
        icp = (C.POINTER(C.c_double) * n)(*[_hp(a) for a in ic])
        _check(lib().ffm_ldu_set_interfaces(self.h, n, sizes, fcp, bcp, icp, ranks), "ffm_ldu_set_interfaces")
        if globalCells is not None:
            _check(lib().ffm_ldu_set_global_cells(self.h, int(globalCells)), "ffm_ldu_set_global_cells")
        return self

    def _unary(self, fn, x, what):
        self.ctx._ready()
        y = self.ctx.empty(self.nCells)
        _check(fn(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())), what)
        self.ctx.sync()
        return y

    def Amul(self, psi):
        return self._unary(lib().ffm_spmv, psi, "ffm_spmv")

    def Tmul(self, psi):
        return self._unary(lib().ffm_tmul, psi, "ffm_tmul")

    def sumA(self):
        s = self.ctx.empty(self.nCells)
        _check(lib().ffm_sumA(self.h, C.c_void_p(s.data_ptr())), "ffm_sumA")
        self.ctx.sync()
        return s

    def residual(self, psi, source):
        self.ctx._ready()
        r = self.ctx.empty(self.nCells)
        _check(lib().ffm_residual(self.h, C.c_void_p(psi.data_ptr()), C.c_void_p(source.data_ptr()),
                                  C.c_void_p(r.data_ptr())), "ffm_residual")
        self.ctx.sync()
        return r

    def reciprocalD(self, preconditioner):
        rD = self.ctx.empty(self.nCells)
        _check(lib().ffm_precond_setup(self.h, PRECONDS[preconditioner], C.c_void_p(rD.data_ptr())), "ffm_precond_setup")
        self.ctx.sync()
        return rD

    def precondition(self, preconditioner, rA, transpose=False):
        self.ctx._ready()
        w = self.ctx.empty(self.nCells)
        _check(lib().ffm_precond_apply(self.h, PRECONDS[preconditioner], 1 if transpose else 0,
                                       C.c_void_p(rA.data_ptr()), C.c_void_p(w.data_ptr())), "ffm_precond_apply")
        self.ctx.sync()
        return w

    def smooth(self, psi, source, nSweeps=1, smoother="symGaussSeidel"):
        self.ctx._ready()
        psi = psi.clone()
        self.ctx._ready()
        _check(lib().ffm_gs_smooth(self.h, 1 if smoother == "symGaussSeidel" else 0, nSweeps,
                                   C.c_void_p(psi.data_ptr()), C.c_void_p(source.data_ptr())), "ffm_gs_smooth")
        self.ctx.sync()
        return psi

    def solve(self, psi, source, solver="PCG", preconditioner="DIC", smoother=None, tolerance=1e-6,
              relTol=0.0, minIter=0, maxIter=1000, nSweeps=1):
        """lduMatrix::solver::New(dict)->solve(psi, source): psi (torch CUDA fp64) is updated in place.
        Returns the SolverPerformance as a dict."""
        self.ctx._ready()
        p = PRECONDS[smoother if (solver == "smoothSolver" and smoother) else preconditioner]
        perf = Perf()
        _check(lib().ffm_solve_d(self.h, SOLVERS[solver], p, tolerance, relTol, minIter, maxIter, nSweeps,
                                 C.c_void_p(psi.data_ptr()), C.c_void_p(source.data_ptr()), C.byref(perf)), "ffm_solve_d")
        return perf.as_dict()

    def solve_multi(self, diags, upper, lower, psis, sources, solver="PBiCGStab", preconditioner="DILU", tolerance=1e-6, relTol=0.0,
                    minIter=0, maxIter=1000):
        """systems with common off-diagonals (native layout, ffm_solve_multi_d): psis updated in place, a perf dict per system"""
        self.ctx._ready()
        n = len(diags)
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        perf = (Perf * n)()
        self._bound = (diags, upper, lower)
        _check(lib().ffm_solve_multi_d(self.h, n, SOLVERS[solver], PRECONDS[preconditioner], tolerance, relTol, minIter, maxIter, arr(diags),
                                       C.c_void_p(upper.data_ptr()), None if lower is None else C.c_void_p(lower.data_ptr()), arr(psis), arr(sources), perf),
               "ffm_solve_multi_d")
        return [perf[i].as_dict() for i in range(n)]

    def solve_host(self, psi, source, solver="PCG", preconditioner="DIC", smoother=None, tolerance=1e-6,
                   relTol=0.0, minIter=0, maxIter=1000, nSweeps=1):
        """Same through host pointers (what an OpenFOAM-side shim calls): returns (psi, perf)."""
        psi = np.ascontiguousarray(psi, np.float64).copy()
        b = np.ascontiguousarray(source, np.float64)
        p = PRECONDS[smoother if (solver == "smoothSolver" and smoother) else preconditioner]
        perf = Perf()
        _check(lib().ffm_solve(self.h, SOLVERS[solver], p, tolerance, relTol, minIter, maxIter, nSweeps,
                               _hp(psi), _hp(b), C.byref(perf)), "ffm_solve")
        return psi, perf.as_dict()

    def bench_Amul(self, x, reps=20):
        self.ctx._ready()
        y = self.ctx.empty(self.nCells)
        ms = C.c_double()
        _check(lib().ffm_bench_spmv(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), reps, C.byref(ms)),
               "ffm_bench_spmv")
        return ms.value


class Plume:
    """The synthetic buoyant-plume case (ffm_plume_*): one fireFoam time step per step()."""

    def __init__(self, ctx, n, h=0.05, deltaT=1e-3, lo=None, hi=None, nbrRank=None):
        """Whole box (default) or one rank's block [lo,hi) with neighbour ranks (-x,+x,-y,+y,-z,+z; -1 = boundary)."""
        self.ctx = ctx
        hnd = C.c_void_p()
        if lo is None:
            _check(lib().ffm_plume_create(ctx.h, int(n[0]), int(n[1]), int(n[2]), float(h), float(deltaT), C.byref(hnd)), "ffm_plume_create")
        else:
            lo_ = (C.c_int * 3)(*[int(v) for v in lo]); hi_ = (C.c_int * 3)(*[int(v) for v in hi])
            nb_ = (C.c_int * 6)(*[int(v) for v in nbrRank])
            _check(lib().ffm_plume_create_block(ctx.h, int(n[0]), int(n[1]), int(n[2]), lo_, hi_, nb_, float(h), float(deltaT),
                                                C.byref(hnd)), "ffm_plume_create_block")
        self.h = hnd
        self.nCells = lib().ffm_plume_ncells(hnd)
        self.nFaces = lib().ffm_plume_nfaces(hnd)

    def close(self):
        if getattr(self, "h", None):
            lib().ffm_plume_destroy(self.h)
            self.h = None

    def set_solvers(self, steckler=True):
        """transport equations with smoothSolver + symGaussSeidel (maxIter 10) as cases/steckler/system/fvSolution:49-62"""
        _check(lib().ffm_plume_set_solvers(self.h, 1 if steckler else 0), "ffm_plume_set_solvers")

    def set_radiation(self, solverFreq=100, nPhi=2, nTheta=4, rays=None):
        """fvDOM stand-in: 4*nPhi*nTheta upwind ray solves every solverFreq steps; rays = [(dAve[3], omega)] or None"""
        if rays is None:
            _check(lib().ffm_plume_set_radiation(self.h, solverFreq, nPhi, nTheta, None, None), "ffm_plume_set_radiation")
            return
        import numpy as np
        d = np.ascontiguousarray(np.concatenate([np.asarray(r[0], dtype=np.float64) for r in rays]))
        o = np.ascontiguousarray(np.array([r[1] for r in rays], dtype=np.float64))
        if len(o) != 4 * nPhi * nTheta:
            raise ValueError("rays must hold 4*nPhi*nTheta entries")
        _check(lib().ffm_plume_set_radiation(self.h, solverFreq, nPhi, nTheta, d.ctypes.data, o.ctypes.data), "ffm_plume_set_radiation")

    def set_radiation_model(self, absorption, Ehrr1, Ehrr2):
        """the reference's absorption / emission model and radiation->Sh in the enthalpy equation (ffm_plume_set_radiation_model)"""
        _check(lib().ffm_plume_set_radiation_model(self.h, float(absorption), float(Ehrr1), float(Ehrr2)), "ffm_plume_set_radiation_model")

    def set_initial_state(self, Y, h, Yamb, Yin, hAmb):
        """start state (Y[5][N], h[N] in natural cell order) and the species' / enthalpy's ambient and inflow values; redoes the
        hydrostatic initialisation (ffm_plume_set_initial_state)"""
        Y = [np.ascontiguousarray(y, np.float64) for y in Y]; h = np.ascontiguousarray(h, np.float64)
        if len(Y) != 5 or any(y.shape != (self.nCells,) for y in Y) or h.shape != (self.nCells,):
            raise ValueError("set_initial_state: five species fields and h of nCells values each")
        yp = (C.c_void_p * 5)(*[y.ctypes.data for y in Y])
        a = np.ascontiguousarray(Yamb, np.float64); b = np.ascontiguousarray(Yin, np.float64)
        _check(lib().ffm_plume_set_initial_state(self.h, yp, _hp(h), _hp(a), _hp(b), float(hAmb)), "ffm_plume_set_initial_state")

    def override_mv_weights(self, w):
        """the next step's species / h convection weights, natural face order (ffm_plume_override_mv_weights; tests)"""
        w = np.ascontiguousarray(w, np.float64)
        if w.shape != (self.nFaces,):
            raise ValueError("override_mv_weights: one weight per internal face")
        _check(lib().ffm_plume_override_mv_weights(self.h, _hp(w)), "ffm_plume_override_mv_weights")

    def set_tight(self, on=True):
        _check(lib().ffm_plume_set_tight(self.h, 1 if on else 0), "ffm_plume_set_tight")

    def step(self):
        _check(lib().ffm_plume_step(self.h), "ffm_plume_step")

    def field(self, name):
        out = np.empty(self.nCells)
        _check(lib().ffm_plume_get_field(self.h, name.encode(), _hp(out)), "ffm_plume_get_field")
        return out

    def solves(self):
        res = []
        for i in range(lib().ffm_plume_nsolves(self.h)):
            nm = C.create_string_buffer(16)
            pf = Perf()
            _check(lib().ffm_plume_get_solve(self.h, i, nm, C.byref(pf)), "ffm_plume_get_solve")
            res.append((nm.value.decode(), pf.as_dict()))
        return res

    def ldu_handle(self):
        return lib().ffm_plume_ldu(self.h)

    def raw(self, name):
        """an array of the case in the library's own orders (ffm_plume_get_raw): cells [N], faces [F] or boundary faces [B]"""
        cap = max(self.nCells, self.nFaces)
        out = np.empty(cap)
        n = lib().ffm_plume_get_raw(self.h, name.encode(), _hp(out), cap)
        if n < 0:
            raise FfmError("ffm_plume_get_raw(%s) failed (%d): %s" % (name, n, lib().ffm_last_error().decode()))
        return out[:n].copy()

    def mesh(self):
        """the case's device mesh as an fvMesh wrapper (not owned: do not close it); tests call operators on it"""
        m = fvMesh.__new__(fvMesh)
        m.ctx, m.h, m.ldu = self.ctx, C.c_void_p(lib().ffm_plume_mesh(self.h)), None
        m.nNative, m.nBoundary = lib().ffm_mesh_nnative(m.h), lib().ffm_mesh_nboundary(m.h)
        return m


class Thermo:
    """hePsiThermo of a janaf / sutherland / perfectGas mixture on the device (ffm_thermo_*).  table: {specie: dict(W, Tlow, Thigh,
    Tcommon, high[7], low[7], As, Ts)} with the thermo file's (molar) coefficients; species: their order"""

    def __init__(self, ctx, species, table, RR):
        self.ctx, self.n = ctx, len(species)
        col = lambda k: np.ascontiguousarray([float(table[s][k]) for s in species], np.float64)
        hi = np.ascontiguousarray([np.asarray(table[s]["high"], float) for s in species]); lo = np.ascontiguousarray([np.asarray(table[s]["low"], float) for s in species])
        h = C.c_void_p()
        _check(lib().ffm_thermo_create(ctx.h, self.n, _hp(col("W")), _hp(col("Tlow")), _hp(col("Thigh")), _hp(col("Tcommon")), _hp(hi), _hp(lo),
                                       _hp(col("As")), _hp(col("Ts")), float(RR), C.byref(h)), "ffm_thermo_create")
        self.h = h

    def _Y(self, Y):
        self._keep = Y
        return (C.c_void_p * self.n)(*[y.data_ptr() for y in Y])

    def correct(self, Y, he, p, T, psi=None, mu=None, alpha=None):
        """in place: T (start value in, T(he) out), psi, mu, alpha; torch CUDA fp64 tensors, Y a list of nSpecies"""
        self.ctx._ready()
        P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _check(lib().ffm_thermo_correct_d(self.h, he.numel(), self._Y(Y), P(he), P(p), P(T), P(psi), P(mu), P(alpha)), "ffm_thermo_correct_d")
        self.ctx.sync()             # the outputs are the caller's tensors: complete before torch's stream reads them

    def he(self, Y, T, out):
        self.ctx._ready()
        _check(lib().ffm_thermo_he_d(self.h, T.numel(), self._Y(Y), C.c_void_p(T.data_ptr()), C.c_void_p(out.data_ptr())), "ffm_thermo_he_d")
        self.ctx.sync()

    def Cp(self, Y, T, out):
        self.ctx._ready()
        _check(lib().ffm_thermo_Cp_d(self.h, T.numel(), self._Y(Y), C.c_void_p(T.data_ptr()), C.c_void_p(out.data_ptr())), "ffm_thermo_Cp_d")
        self.ctx.sync()

    def close(self):
        if getattr(self, "h", None):
            lib().ffm_thermo_destroy(self.h)
            self.h = None


class GAMG:
    """GAMGSolver for an lduMatrix (ffm_gamg_*): the agglomeration is built once from the addressing and the face weights
    (faceAreaPair: weights=None and Sf given), set_matrix() agglomerates the coefficients, solve() runs V-cycles."""

    def __init__(self, ctx, A, l, u, Sf=None, weights=None, nCellsInCoarsestLevel=10, mergeLevels=1, forward=True):
        """forward: pairGAMGAgglomeration::forward_ at the start of this agglomeration (True: the first of a run); None: continue with the
        direction the context's previous agglomeration ended with, as a second mesh / region of one OpenFOAM run does"""
        self.ctx, self.A = ctx, A
        if forward is not None:
            _check(lib().ffm_ctx_set_gamg_forward(ctx.h, 1 if forward else 0), "ffm_ctx_set_gamg_forward")
        l = np.ascontiguousarray(l, np.int32); u = np.ascontiguousarray(u, np.int32)
        if weights is None:
            Sf = np.ascontiguousarray(Sf, np.float64).reshape(-1, 3)
            weights = np.empty(len(l))
            _check(lib().ffm_gamg_face_area_pair_weights(len(l), _hp(Sf), _hp(weights)), "ffm_gamg_face_area_pair_weights")
        weights = np.ascontiguousarray(weights, np.float64)
        h = C.c_void_p()
        _check(lib().ffm_gamg_create(ctx.h, A.h, A.nCells, len(l), _ip(l), _ip(u), _hp(weights), int(nCellsInCoarsestLevel), int(mergeLevels),
                                     C.byref(h)), "ffm_gamg_create")
        self.h = h
        self.weights = weights

    @property
    def nLevels(self):
        return lib().ffm_gamg_nlevels(self.h)

    def level_size(self, level):
        a, b = C.c_int(), C.c_int()
        _check(lib().ffm_gamg_level_size(self.h, level, C.byref(a), C.byref(b)), "ffm_gamg_level_size")
        return a.value, b.value

    def level_addressing(self, level):
        n, f = self.level_size(level)
        l = np.empty(f, np.int32); u = np.empty(f, np.int32)
        _check(lib().ffm_gamg_get_level_addressing(self.h, level, _ip(l), _ip(u)), "ffm_gamg_get_level_addressing")
        return l, u

    def level_coeffs(self, level):
        n, f = self.level_size(level)
        d = np.empty(n); up = np.empty(f); lo = np.empty(f)
        _check(lib().ffm_gamg_get_level_coeffs(self.h, level, _hp(d), _hp(up), _hp(lo)), "ffm_gamg_get_level_coeffs")
        return d, up, lo

    def set_sweeps(self, nPreSweeps=0, nPostSweeps=2, nFinestSweeps=2):
        _check(lib().ffm_gamg_set_sweeps(self.h, nPreSweeps, nPostSweeps, nFinestSweeps), "ffm_gamg_set_sweeps")

    def set_matrix(self, diag, upper, lower=None):
        """torch CUDA fp64 tensors in the caller's cell / face order (kept alive here)"""
        self.ctx._ready()
        self._coef = (diag, upper, lower)
        _check(lib().ffm_gamg_set_matrix_d(self.h, C.c_void_p(diag.data_ptr()), C.c_void_p(upper.data_ptr()),
                                           None if lower is None else C.c_void_p(lower.data_ptr())), "ffm_gamg_set_matrix_d")
        return self

    def solve(self, psi, source, smoother="GaussSeidel", tolerance=1e-6, relTol=0.0, minIter=0, maxIter=1000):
        self.ctx._ready()
        pf = Perf()
        _check(lib().ffm_gamg_solve_d(self.h, PRECONDS[smoother], tolerance, relTol, minIter, maxIter, C.c_void_p(psi.data_ptr()),
                                      C.c_void_p(source.data_ptr()), C.byref(pf)), "ffm_gamg_solve_d")
        return pf.as_dict()

    def coarsest_solves(self):
        n = lib().ffm_gamg_coarsest_solves(self.h, 0, None)
        buf = (Perf * max(n, 1))()
        lib().ffm_gamg_coarsest_solves(self.h, n, buf)
        return [buf[i].as_dict() for i in range(n)]

    def close(self):
        if getattr(self, "h", None):
            lib().ffm_gamg_destroy(self.h)
            self.h = None


class PyrolysisPanel:
    """reactingOneDim for a panel of independent columns (ffm_pyro_*): step(dt, qSurf) advances rho, Y, h, T of every column"""

    def __init__(self, ctx, nCol, nLay=8, thickness=0.0127, area=1.0, T0=298.15, Yw0=1.0):
        self.ctx, self.nCol, self.nLay = ctx, int(nCol), int(nLay)
        h = C.c_void_p()
        _check(lib().ffm_pyro_create(ctx.h, self.nCol, self.nLay, thickness, area, T0, Yw0, C.byref(h)), "ffm_pyro_create")
        self.h = h

    def step(self, dt, qSurf, Tback=None):
        """qSurf: torch CUDA fp64 [nCol]"""
        self.ctx._ready()
        _check(lib().ffm_pyro_step(self.h, float(dt), C.c_void_p(qSurf.data_ptr()), 0 if Tback is None else 1, 0.0 if Tback is None else float(Tback)), "ffm_pyro_step")

    def field(self, name):
        col = name in ("Tsurf", "phiGas", "qSurf", "Twall")
        out = np.empty(self.nCol if col else self.nCol * self.nLay)
        _check(lib().ffm_pyro_get(self.h, name.encode(), _hp(out)), "ffm_pyro_get")
        return out if col else out.reshape(self.nCol, self.nLay)

    def couple(self, Tgas_cell, kappaDelta, qin, emissivity, absorptivity, rho_b, magSf, nf, hocSolid, qFuel, refT, U, map=None):
        """the mapped patch conditions (ffm_pyro_couple_d); all torch CUDA tensors (nf, U: lists of three); map: int32 tensor or None"""
        self.ctx._ready()
        P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _check(lib().ffm_pyro_couple_d(self.h, P(map), P(Tgas_cell), P(kappaDelta), P(qin), float(emissivity), float(absorptivity), P(rho_b), P(magSf),
                                       P(nf[0]), P(nf[1]), P(nf[2]), float(hocSolid), float(qFuel), P(refT), P(U[0]), P(U[1]), P(U[2])), "ffm_pyro_couple_d")
        self.ctx.sync()             # refT and U are the caller's tensors: complete before torch's stream reads them

    def set_model(self, model="reactingOneDim", alphaScheme="linear", kappaScheme="linear", back=None, radiation=None):
        """the region's dictionaries: pyrolysisModel (reactingOneDim | reactingOneDim21), the two laplacian schemes (linear | harmonic),
        back face None | ("fixed", T) | ("constH", h, Tinf), radiation None | dict(v=(absorptivity, emissivity), char=(...))"""
        if model not in ("reactingOneDim", "reactingOneDim21"):
            raise ValueError("pyrolysisModel %r is not built" % model)
        _check(lib().ffm_pyro_set_model(self.h, 1 if model == "reactingOneDim21" else 0, 1 if alphaScheme == "harmonic" else 0,
                                        1 if kappaScheme == "harmonic" else 0), "ffm_pyro_set_model")
        if back is None:
            _check(lib().ffm_pyro_set_back(self.h, 0, 0.0, 298.15), "ffm_pyro_set_back")
        elif back[0] == "fixed":
            _check(lib().ffm_pyro_set_back(self.h, 1, 0.0, float(back[1])), "ffm_pyro_set_back")
        else:
            _check(lib().ffm_pyro_set_back(self.h, 2, float(back[1]), float(back[2])), "ffm_pyro_set_back")
        if radiation is not None:
            (aV, eV), (aC, eC) = radiation["v"], radiation["char"]
            _check(lib().ffm_pyro_set_surface_radiation(self.h, float(aV), float(eV), float(aC), float(eC)), "ffm_pyro_set_surface_radiation")

    def evolve(self, dt, Tgas_cell, kappaDelta, qin, emissivity=1.0, absorptivity=1.0, map=None):
        """evolveRegion with the coupled wall condition evaluated inside the step (ffm_pyro_evolve_d); torch CUDA tensors"""
        self.ctx._ready()
        P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _check(lib().ffm_pyro_evolve_d(self.h, float(dt), P(map), P(Tgas_cell), P(kappaDelta), P(qin), float(emissivity), float(absorptivity)), "ffm_pyro_evolve_d")

    def gas_side(self, rho_b, magSf, nf, hocSolid, qFuel, refT, U, emissivity=None, map=None):
        """the gas-side patch values from the panel's new state (ffm_pyro_gas_side_d)"""
        self.ctx._ready()
        P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _check(lib().ffm_pyro_gas_side_d(self.h, P(map), P(rho_b), P(magSf), P(nf[0]), P(nf[1]), P(nf[2]), float(hocSolid), float(qFuel), P(refT),
                                         P(U[0]), P(U[1]), P(U[2]), P(emissivity)), "ffm_pyro_gas_side_d")
        self.ctx.sync()

    def diff_no(self, dt):
        """solidRegionDiffNo() (ffm_pyro_diff_no)"""
        self.ctx._ready()
        out = np.zeros(1)
        _check(lib().ffm_pyro_diff_no(self.h, float(dt), _hp(out)), "ffm_pyro_diff_no")
        return float(out[0])

    def step_coupled(self, dt, Tback=None):
        """one step with the heat flux of the last couple()"""
        self.ctx._ready()
        _check(lib().ffm_pyro_step(self.h, float(dt), C.c_void_p(lib().ffm_pyro_qSurf_d(self.h)), 0 if Tback is None else 1, 0.0 if Tback is None else float(Tback)), "ffm_pyro_step")

    def close(self):
        if getattr(self, "h", None):
            lib().ffm_pyro_destroy(self.h)
            self.h = None


class fvMesh:
    """Device fvMesh over an lduMatrix in the library's cell order (ffm_mesh_*): geometry in LDU face order on input,
    face fields converted with to_native/from_native.  Thin wrapper for the operator-level parity tests."""

    def __init__(self, ldu, V, C_, Sf, magSf, weights, deltaCoeffs, patches):
        """patches: list of (faceCells[int32], Sf[3][n], deltaCoeffs[n])"""
        self.ldu, self.ctx = ldu, ldu.ctx
        f64 = lambda a: np.ascontiguousarray(a, np.float64)
        V, C_, Sf, magSf, weights, deltaCoeffs = map(f64, (V, C_, Sf, magSf, weights, deltaCoeffs))
        n = len(patches)
        fc = [np.ascontiguousarray(p[0], np.int32) for p in patches]
        ps = [f64(p[1]) for p in patches]
        pd = [f64(p[2]) for p in patches]
        sizes = (C.c_int * n)(*[len(a) for a in fc])
        fcp = (C.POINTER(C.c_int) * n)(*[_ip(a) for a in fc])
        psp = (C.POINTER(C.c_double) * n)(*[_hp(a) for a in ps])
        pdp = (C.POINTER(C.c_double) * n)(*[_hp(a) for a in pd])
        h = C.c_void_p()
        _check(lib().ffm_mesh_create(ldu.h, _hp(V), _hp(C_), _hp(Sf), _hp(magSf), _hp(weights), _hp(deltaCoeffs), n, sizes, fcp, psp, pdp,
                                     C.byref(h)), "ffm_mesh_create")
        self.h = h
        self.nNative = lib().ffm_mesh_nnative(h)
        self.nBoundary = lib().ffm_mesh_nboundary(h)

    def close(self):
        if getattr(self, "h", None):
            lib().ffm_mesh_destroy(self.h)
            self.h = None

    def set_nonorth_correction(self, corrVec):
        """nonOrthCorrectionVectors[3][F] in LDU face order (the `corrected` snGrad / laplacian schemes)."""
        c = np.ascontiguousarray(corrVec, np.float64)
        _check(lib().ffm_mesh_set_nonorth_correction(self.h, _hp(c)), "ffm_mesh_set_nonorth_correction")

    def set_face_centres(self, Cf):
        """Cf[3][F] in LDU face order (mesh.Cf(), needed by the LUST correction)."""
        a = np.ascontiguousarray(Cf, np.float64)
        _check(lib().ffm_mesh_set_face_centres(self.h, _hp(a)), "ffm_mesh_set_face_centres")

    def to_native(self, faceField):
        out = self.ctx.zeros(max(self.nNative, 1))
        a = np.ascontiguousarray(faceField, np.float64)
        _check(lib().ffm_faces_to_native(self.h, _hp(a), C.c_void_p(out.data_ptr())), "ffm_faces_to_native")
        self.ctx.sync()
        return out

    def from_native(self, t):
        self.ctx._ready()
        out = np.empty(self.ldu.nFaces)
        _check(lib().ffm_faces_from_native(self.h, C.c_void_p(t.data_ptr()), _hp(out)), "ffm_faces_from_native")
        return out

    def call(self, name, *args):
        """ffm_<name>(mesh, *args): torch tensors -> device pointers, None -> NULL, numbers as is."""
        self.ctx._ready()
        conv = []
        for a in args:
            if a is None:
                conv.append(None)
            elif isinstance(a, (list, tuple)):          # host array of device pointers (None -> NULL)
                conv.append((C.c_void_p * len(a))(*[None if t is None else t.data_ptr() for t in a]))
            elif hasattr(a, "data_ptr"):
                conv.append(C.c_void_p(a.data_ptr()))
            else:
                conv.append(a)
        _check(getattr(lib(), "ffm_" + name)(self.h, *conv), "ffm_" + name)
        self.ctx.sync()
