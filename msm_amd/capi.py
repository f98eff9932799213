"""ctypes binding of libmsm_amd.so (the C ABI declared in include/msm_amd.h).

The library is the product: there is no Python or CPU implementation behind
these calls.  Loading fails loudly when the shared object has not been built.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MSM_AMD_LIB: another build of the same library (A/B timing of compile-time choices)
LIB_PATH = os.environ.get("MSM_AMD_LIB") or os.path.join(HERE, "libmsm_amd.so")

LMAX = 10
U_NAMES = ("ushort_real", "ushort_self", "ushort_csr", "ulong_real", "ulong_four", "ulong_self", "utotal")
U_COUNT = 7
ACC_GRADIENT, ACC_LITERAL, ACC_FORCE = 0, 1, 2
EWALD_FULL, EWALD_LITERAL = 0, 1
MODE_AUTO, MODE_LITERAL, MODE_SCALABLE = 0, 1, 2
STENCILS_HOST, STENCILS_DEVICE = 0, 1
GRID_Q, GRID_E, GRID_EFOUR = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class PlanDesc(C.Structure):
    """struct msm_amd_plan_desc"""
    _fields_ = [("N", C.c_int), ("p", C.c_int), ("mu", C.c_int), ("L", C.c_int), ("kmax", C.c_int),
                ("acc_convention", C.c_int), ("M", C.c_int * 3), ("zplane0", C.c_int), ("nzplanes", C.c_int),
                ("n_owned", C.c_int), ("A", C.c_double * 3), ("a", C.c_double), ("beta", C.c_double),
                ("open_boundary", C.c_int), ("origin", C.c_double * 3), ("extent", C.c_double * 3)]


class MsmStats(C.Structure):
    """struct msm_stats (960 bytes), field for field as include/msm_amd.h declares it."""
    _fields_ = [("N", C.c_int), ("L", C.c_int), ("mu", C.c_int),
                ("h", C.c_double), ("hx", C.c_double), ("hy", C.c_double), ("hz", C.c_double),
                ("a", C.c_double), ("aL", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("p", C.c_int), ("s", C.c_int), ("kmax", C.c_int), ("csr", C.c_double),
                ("chinx", C.c_int), ("chiny", C.c_int), ("chinz", C.c_int), ("chisize", C.c_int),
                ("cvecx", _dp), ("cvecy", _dp), ("cvecz", _dp),
                ("clenx", C.c_int), ("cleny", C.c_int), ("clenz", C.c_int),
                ("chisum", C.c_double), ("chi", _dp),
                ("sLsize", C.c_int), ("sLnx", C.c_int), ("sLny", C.c_int), ("sLnz", C.c_int),
                ("M0", C.c_int * 3), ("Mx", C.c_int), ("My", C.c_int), ("Mz", C.c_int),
                ("MLx", C.c_int), ("MLy", C.c_int), ("MLz", C.c_int),
                ("ML2x", C.c_int), ("ML2y", C.c_int), ("ML2z", C.c_int),
                ("Mlx", C.c_int * LMAX), ("Mly", C.c_int * LMAX), ("Mlz", C.c_int * LMAX),
                ("slsizex", C.c_int * LMAX), ("slsizey", C.c_int * LMAX), ("slsizez", C.c_int * LMAX),
                ("Ax", C.c_int), ("Ay", C.c_int), ("Az", C.c_int),
                ("size", (C.c_int * 3) * LMAX),
                ("rminx", C.c_double), ("rminy", C.c_double), ("rminz", C.c_double),
                ("rmaxx", C.c_double), ("rmaxy", C.c_double), ("rmaxz", C.c_double),
                ("volume", C.c_double), ("density", C.c_double),
                ("short_range_interactions", C.c_long), ("storage", C.c_int),
                ("netcharge", C.c_double), ("netgridcharge", C.c_double * LMAX),
                ("cputime_initialization", C.c_double), ("cputime_anterpolation", C.c_double),
                ("cputime_restriction", C.c_double), ("cputime_stencill", C.c_double),
                ("cputime_stencilL", C.c_double), ("cputime_grid_to_grid_mapping", C.c_double),
                ("cputime_top_level", C.c_double), ("cputime_prolongation", C.c_double),
                ("cputime_interpolation", C.c_double), ("cputime_potential_long", C.c_double),
                ("cputime_short_range", C.c_double), ("cputime_self", C.c_double), ("cputime_total", C.c_double),
                ("potential_total", C.c_double), ("potential_short_total", C.c_double),
                ("potential_short_real", C.c_double), ("potential_short_csr", C.c_double),
                ("potential_short_self", C.c_double), ("potential_long_total", C.c_double),
                ("potential_long_real", C.c_double), ("potential_long_self", C.c_double),
                ("potential_long_fourier", C.c_double)]


# struct msm_amd_collectives: the exchange steps of a slab rank as a table of functions (stream = hipStream_t)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
HALO_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                               C.c_int, C.c_void_p)


class Collectives(C.Structure):
    """struct msm_amd_collectives"""
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("ctx", C.c_void_p), ("all_gather", ALL_GATHER_FN),
                ("all_reduce_sum", ALL_REDUCE_FN), ("halo_exchange", HALO_EXCHANGE_FN)]


SLAB_AUTO, SLAB_ALLGATHER, SLAB_HALO = 0, 1, 2

# every symbol include/msm_amd.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
SIGNATURES = {
    # drop-in layer
    "msmperiodic": (MsmStats, [C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_int, _dp, _dp] + [C.c_double] * 3),
    "msmPeriodicRunner": (C.c_double, [_dp, _dp, _dp, C.c_int, C.c_int, C.c_int] + [C.c_double] * 4),
    "mylog2": (C.c_double, [C.c_double]),
    "choosebetaNew": (C.c_double, [C.c_double] * 3),
    "choosekmax": (C.c_double, [C.c_double] * 3),
    "chooseM": (C.c_int, [C.c_double, C.c_double]),
    "chooseL": (C.c_int, [C.c_int] * 4),
    "minOf3Integers": (C.c_int, [C.c_int] * 3),
    "data_read": (C.c_int, [C.c_char_p, C.POINTER(_dp), C.POINTER(_dp), _ip, _dp, _dp, _dp]),
    "short_range_periodic": (C.c_double, [C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_double, _dp] + [C.c_double] * 3),
    "print_stats": (None, [MsmStats]),
    "msm4g_linkedlist_new": (_vp, []),
    "msm4g_linkedlist_add": (None, [_vp, C.c_int]),
    "msm4g_linkedlist_addvalue": (None, [_vp, C.c_int, C.c_double]),
    "msm4g_linkedlist_search": (C.c_int, [_vp, C.c_int]),
    "msm4g_linkedlist_getvalue": (C.c_double, [_vp, C.c_int]),
    "msm4g_linkedlist_destroy": (None, [_vp]),
    # handle layer
    "msm_amd_set_mode": (None, [C.c_int]),
    "msm_amd_get_mode": (C.c_int, []),
    "msm_amd_last_call_mode": (C.c_int, []),
    "msm_amd_release_cached_plan": (None, []),
    "msm_amd_plan_builds": (C.c_long, []),
    "msm_amd_set_stencil_backend": (None, [C.c_int]),
    "msm_amd_get_stencil_backend": (C.c_int, []),
    "msm_amd_set_quiet": (None, [C.c_int]),
    "msm_amd_plan_auto": (C.c_int, [C.c_int] + [C.c_double] * 3 + [C.c_int, C.c_int, C.c_double, C.c_int,
                                                                  C.POINTER(PlanDesc)]),
    "msm_amd_create": (C.c_int, [C.POINTER(PlanDesc), C.c_int, _vp, _vpp]),
    "msm_amd_create_literal": (C.c_int, [C.POINTER(PlanDesc), C.c_int, _vp, _vpp]),
    "msm_amd_destroy": (None, [_vp]),
    "msm_amd_eval": (C.c_int, [_vp, _vp, _vp, _vp, _dp]),
    "msm_amd_eval_host": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "msm_amd_eval_enqueue": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "msm_amd_eval_reuse": (C.c_int, [_vp, _vp, _vp, _vp, _dp, C.c_double]),
    "msm_amd_reuse_stats": (C.c_int, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "msm_amd_stage_bin": (C.c_int, [_vp, _vp, _vp]),
    "msm_amd_stage_pairs": (C.c_int, [_vp, _vp]),
    "msm_amd_stage_anterpolate": (C.c_int, [_vp]),
    "msm_amd_stage_grid": (C.c_int, [_vp]),
    "msm_amd_stage_interpolate": (C.c_int, [_vp, _vp]),
    "msm_amd_stage_energies": (C.c_int, [_vp, _dp]),
    "msm_amd_set_counts": (C.c_int, [_vp, C.c_int, C.c_int]),
    "msm_amd_set_planes": (C.c_int, [_vp, C.c_int, C.c_int]),
    "msm_amd_plan_open_holds": (C.c_int, [C.POINTER(PlanDesc), _dp, _dp]),
    "msm_amd_stage_raw_sums": (C.c_int, [_vp, _dp]),
    "msm_amd_raw_sums_device_ptr": (_vp, [_vp]),
    "msm_amd_combine_energies": (C.c_int, [_vp, _dp, _dp]),
    "msm_amd_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "msm_amd_rccl_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.POINTER(Collectives))]),
    "msm_amd_rccl_destroy": (None, [C.POINTER(Collectives)]),
    "msm_amd_rccl_available": (C.c_int, []),
    "msm_amd_rccl_comm_info": (C.c_int, [C.POINTER(Collectives), _ip, _ip]),
    "msm_amd_slab_exchange_mode": (C.c_int, [_vp, C.c_int, C.c_int]),
    "msm_amd_slab_step": (C.c_int, [_vp, C.POINTER(Collectives), _vp, _vp, _vp, C.c_int]),
    "msm_amd_slab_energies": (C.c_int, [_vp, C.POINTER(Collectives), _dp]),
    "msm_amd_grid_device_ptr": (_vp, [_vp, C.c_int, C.c_int]),
    "msm_amd_grid_dims": (C.c_int, [_vp, C.c_int, _ip]),
    "msm_amd_copy_grid": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "msm_amd_set_grid": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "msm_amd_stencil_info": (C.c_int, [_vp, C.c_int, _ip, _ip]),
    "msm_amd_copy_stencil": (C.c_int, [_vp, C.c_int, _dp]),
    "msm_amd_copy_wprime": (C.c_int, [_vp, _dp, _ip]),
    "msm_amd_op_restrict": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "msm_amd_op_prolong_add": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "msm_amd_op_convolve": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "direct": (None, [C.c_int, _dp, _dp, _dp, _dp]),
    "msm3d": (MsmStats, [C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp]),
    "msm_amd_plan_open": (C.c_int, [C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(PlanDesc)]),
    "ewaldClassical": (C.c_double, [_dp, _dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "ewaldClassicalRunner": (None, [C.c_char_p, C.c_double, C.c_double, C.c_double]),
    "msm_amd_ewald_work_doubles": (C.c_longlong, [C.c_int, C.c_double, C.c_int]),
    "msm_amd_ewald_device": (C.c_int, [C.c_int, _vp, _vp, _dp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                       _vp, _vp, _vp, _vp, _dp]),
    "msm_amd_direct_work_doubles": (C.c_int, [C.c_int]),
    "msm_amd_direct_device": (C.c_int, [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _dp]),
    "msm_amd_direct_noalloc_max_slice": (C.c_int, []),
    "msm_amd_direct_device_noalloc": (C.c_int, [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "msm_amd_last_timings": (C.c_int, [_vp, _dp]),
    "msm_amd_last_kernel_ms": (C.c_int, [_vp, C.c_char_p, _dp]),
    "msm_amd_set_profiling": (None, [_vp, C.c_int]),
    "msm_amd_set_pair_trace": (C.c_int, [_vp, _vp]),
    "msm_amd_pair_trace_slots": (C.c_long, [_vp]),
    "msm_amd_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "msm_amd_hostplan_create": (C.c_int, [C.POINTER(PlanDesc), C.c_int, _vpp]),
    "msm_amd_hostplan_write_msmsimple_cache": (C.c_int, [_vp, C.c_char_p]),
    "msm_amd_write_msmsimple_cache": (C.c_int, [_vp, C.c_char_p]),
    "msm_amd_hostplan_destroy": (None, [_vp]),
    "msm_amd_hostplan_stencil_info": (C.c_int, [_vp, C.c_int, _ip, _ip]),
    "msm_amd_hostplan_copy_stencil": (C.c_int, [_vp, C.c_int, _dp]),
    "msm_amd_hostplan_copy_wprime": (C.c_int, [_vp, _dp, _ip]),
    "msm_amd_hostplan_scalars": (C.c_int, [_vp, _dp]),
    "msm_amd_last_error": (C.c_char_p, []),
    "msm_amd_device_count": (C.c_int, []),
    "msm_amd_version": (C.c_char_p, []),
}

_lib = None


class MsmAmdError(RuntimeError):
    pass


def load():
    """Load libmsm_amd.so and set every prototype.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MsmAmdError("msm_amd/libmsm_amd.so is not built: run `make -C msm_amd/csrc` "
                          "(or __graft_entry__.build()); there is no fallback implementation")
    # A process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64 under the same soname as the
    # system one this library links against; whichever is loaded first serves both.  The Engine uses torch for
    # device memory, so torch's copy has to win: import it before the library pulls in the system runtime
    # (loading this library first and torch afterwards leaves torch on a runtime it was not built for, and the
    # device then shows up as absent).  Plain C callers are not affected: they never load torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().msm_amd_last_error()
        raise MsmAmdError("%s failed (%d): %s" % (what or "msm_amd call", rc, msg.decode() if msg else ""))
