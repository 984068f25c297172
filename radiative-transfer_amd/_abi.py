"""ctypes mirror of include/rt_amd.h (struct layouts, enums, library loader).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc,
--offload-arch=gfx950) as ``radiative-transfer_amd/csrc/librt_amd.so``.  There
is no CPU implementation behind this module: if the library is missing or no
MI355X is visible the calls fail loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "librt_amd.so")

RT_ABI_VERSION = 8

BOUNDARY = {"reflect": 0, "absorb": 1, "reemit": 2}
SCATTER = {"hg": 0, "iso": 1, "draine": 2, "draine_exact": 3}
GEOMETRY = {"slab": 0, "shell": 1}
ORDER = {"mc": 0, "gray": 1}
RNG = {"philox52": 0, "philox32x4": 1, "philox40": 2}
FLAG_RUNTIME_KERNEL = 1

FATE_ESCAPED, FATE_ESCAPED_UNCOUNTED, FATE_ABSORBED, FATE_BOUNDARY, FATE_GUARD = 1, 2, 3, 4, 5

(SUM_INITIAL, SUM_ESCAPED, SUM_BOUNDARY, SUM_ESCAPED_UNCOUNTED, SUM_TRANSFER_SQ, SUM_TRANSFER_P0,
 SUM_P0_SQ, SUM_ABSORBED_ENERGY, SUM_DUST_MOMENTUM, SUM_REEMITTED) = range(10)
NSUM = 10
(CNT_SCATTER, CNT_ESCAPED, CNT_ESCAPED_UNCOUNTED, CNT_ABSORBED, CNT_BOUNDARY_ABSORBED, CNT_REEMITTED,
 CNT_GUARD, CNT_PHOTONS, CNT_FLIGHTS, CNT_BOUNDARY_HITS) = range(10)
NCNT = 10

c_double_p = C.POINTER(C.c_double)
c_u64_p = C.POINTER(C.c_uint64)


class RtParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_atm", C.c_int32), ("boundary", C.c_int32), ("scatter", C.c_int32), ("loop_order", C.c_int32),
        ("rand_mu", C.c_int32), ("abs_only", C.c_int32), ("n_mu_bins", C.c_int32), ("n_depth_bins", C.c_int32),
        ("max_events_per_photon", C.c_int32), ("n_slabs", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64), ("photon_id_base", C.c_uint64),
        ("monochromatic_wl", C.c_double), ("alpha", C.c_double), ("h", C.c_double),
        ("tau_atm", c_double_p), ("n_photons", c_u64_p),
        ("rng_mode", C.c_int32), ("geometry", C.c_int32), ("table_of_atm", C.POINTER(C.c_int32)),
        ("shell_inner", C.c_double), ("escaped_mu", c_double_p),
    ]


class RtTallies(C.Structure):
    _fields_ = [
        ("sums", c_double_p), ("counts", c_u64_p), ("mu_hist", c_u64_p),
        ("depth_momentum", c_double_p), ("depth_energy", c_double_p), ("slab_sums", c_double_p),
        ("slab_depth", c_double_p),
    ]


class RtGrainSet(C.Structure):
    _fields_ = [("n_sizes", C.c_int32), ("a", c_double_p), ("offset", C.POINTER(C.c_int32)), ("wl", c_double_p),
                ("q_abs", c_double_p), ("q_sca", c_double_p), ("g", c_double_p)]


class RtGrainTables(C.Structure):
    _fields_ = [("kappa_rp", c_double_p), ("kappa_f", c_double_p), ("sigma_abs", c_double_p), ("sigma_sca", c_double_p),
                ("g_av", c_double_p), ("q_ext", c_double_p), ("kappa_av_rp", C.c_double), ("kappa_av_f", C.c_double),
                ("lum", C.c_double)]


class RtShellGalaxy(C.Structure):
    _fields_ = [("n_gal", C.c_int32), ("n_views", C.c_int32), ("cm_pc", C.c_double), ("ratio_rp_lambda_rp", C.c_double)] \
             + [(n, c_double_p) for n in ("lha_ratio_f", "dist_to_ref", "mass_tot")] \
             + [("view_of_region", C.POINTER(C.c_int32)), ("self_row", C.POINTER(C.c_int32))]


class RtShellModel(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("aging", "static_momentum", "n_tau", "n_age", "n_lage")] + [("struct_size", C.c_uint32)] \
             + [(n, C.c_double) for n in ("G", "c", "sigma_sb", "year_s", "h_hi_cm", "r_escape", "time_slice",
                                          "tau_cutoff", "l_norm")] \
             + [(n, c_double_p) for n in ("mc_tau", "mc_age", "mc_val", "l_age", "l_val", "ratio_val")] \
             + [("galaxy", C.POINTER(RtShellGalaxy))]


SHELL_REGION_FIELDS = ("m_g", "m_new", "m_old", "m_co_hi", "h_old", "h_g_i", "tau_i", "l_bol", "sigma_g", "r_end")


class RtShellRegions(C.Structure):
    _fields_ = [("n", C.c_int32), ("pad_", C.c_int32)] + [(n, c_double_p) for n in SHELL_REGION_FIELDS]


class RtShellControl(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("v0", "t0", "rtol", "atol_v", "atol_t", "max_step")] \
             + [("n_out", C.c_int32), ("max_steps", C.c_int32)]


class RtShellResult(C.Structure):
    _fields_ = [(n, c_double_p) for n in ("v_max", "r_vmax", "t_vmax", "r_last", "v_last", "t_last")] \
             + [("status", C.POINTER(C.c_int32)), ("n_steps", C.POINTER(C.c_int32)), ("v_out", c_double_p), ("t_out", c_double_p)]


SHELL_REACHED_END, SHELL_STALLED, SHELL_ESCAPED, SHELL_STEP_LIMIT = 0, 1, 2, 3


class RtPhotonTrace(C.Structure):
    _fields_ = [("final_z", C.c_double), ("final_mu", C.c_double), ("fate", C.c_int32),
                ("n_scatter", C.c_int32), ("n_draws", C.c_int32), ("pad_", C.c_int32)]


EXPORTS = ["rt_abi_version", "rt_last_error", "rt_create", "rt_destroy", "rt_device_info", "rt_set_tables",
           "rt_run", "rt_run_device", "rt_replay", "rt_kernel_time", "rt_selftest_math", "rt_build_grain_tables",
           "rt_shell_velocity", "rt_shell_rhs", "rt_set_tables_multi", "rt_run_device_static", "rt_last_kernel",
           "rt_comm_unique_id", "rt_comm_init", "rt_comm_destroy", "rt_run_allreduce", "rt_emission_indices"]

_lib = None


def load_library(path=None):
    """Load librt_amd.so and declare argument types.  Raises OSError if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("RT_AMD_LIB") or LIB_PATH  # RT_AMD_LIB: A/B builds of the same ABI
    if not os.path.exists(path):
        raise OSError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.rt_abi_version.restype = C.c_int
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_last_kernel.restype = C.c_char_p
    lib.rt_last_kernel.argtypes = [vp]
    lib.rt_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_device_info.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.rt_set_tables.argtypes = [vp, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    lib.rt_set_tables_multi.argtypes = [vp, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    lib.rt_run.argtypes = [vp, C.POINTER(RtParams), C.POINTER(RtTallies)]
    lib.rt_run_device.argtypes = [vp, C.POINTER(RtParams), vp, vp, vp, vp, vp]
    lib.rt_run_device_static.argtypes = [vp, C.POINTER(RtParams), vp, vp, vp, vp, vp]
    lib.rt_comm_unique_id.argtypes = [vp]
    lib.rt_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.rt_comm_destroy.argtypes = [vp]
    lib.rt_run_allreduce.argtypes = [vp, C.POINTER(RtParams), vp, vp]
    lib.rt_emission_indices.argtypes = [vp, C.POINTER(RtParams), C.c_int32, C.c_uint64, C.c_int64, C.POINTER(C.c_int32), c_double_p]
    lib.rt_replay.argtypes = [vp, C.POINTER(RtParams), C.c_int64, C.c_int32, c_double_p, c_double_p, c_double_p,
                              c_double_p, c_double_p, c_double_p, C.POINTER(RtTallies), C.POINTER(RtPhotonTrace)]
    lib.rt_build_grain_tables.argtypes = [vp, C.POINTER(RtGrainSet), C.c_double, C.c_double, C.c_double, C.c_int32,
                                          C.c_int32, c_double_p, c_double_p, C.POINTER(RtGrainTables)]
    i32p = C.POINTER(C.c_int32)
    lib.rt_shell_velocity.argtypes = [vp, C.POINTER(RtShellModel), C.POINTER(RtShellRegions), C.POINTER(RtShellControl),
                                      C.POINTER(RtShellResult)]
    lib.rt_shell_rhs.argtypes = [vp, C.POINTER(RtShellModel), C.POINTER(RtShellRegions), C.c_int32, i32p, c_double_p,
                                 c_double_p, c_double_p, c_double_p, c_double_p]
    lib.rt_selftest_math.argtypes = [vp, C.c_int64, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    lib.rt_kernel_time.argtypes = [vp, c_double_p, C.POINTER(C.c_int64), C.c_int]
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if an entry point is missing
        if name not in ("rt_last_error", "rt_last_kernel"):
            getattr(lib, name).restype = C.c_int
    version = lib.rt_abi_version()
    if version == (RT_ABI_VERSION | 0x1000) and os.environ.get("RT_AMD_ALLOW_ABLATION") == "1":
        version = RT_ABI_VERSION  # a timing-experiment build (tools/ablation); its results carry no parity claim
    if version != RT_ABI_VERSION:
        raise OSError(f"{path}: ABI version {lib.rt_abi_version():#x} != {RT_ABI_VERSION:#x}")
    _lib = lib
    return lib
