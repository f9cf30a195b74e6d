"""ctypes binding of include/cpprob_hip.h (harness for tests and bench; not a compute path)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcpprob_hip.so")

ALG_SIS, ALG_SMC = 2, 4
MODEL_GAUSSIAN_UNKNOWN_MEAN, MODEL_GAUSSIAN_README, MODEL_LINEAR_GAUSSIAN_1D, MODEL_HMM3, MODEL_GAUSSIAN_2D_UNKNOWN_MEAN, MODEL_HMM_TABLE = 0, 1, 2, 3, 4, 5
RESAMPLE_SYSTEMATIC, RESAMPLE_STRATIFIED, RESAMPLE_MULTINOMIAL = 0, 1, 2
SCOPE_GLOBAL, SCOPE_ISLAND, SCOPE_EXCHANGE = 0, 1, 2
# cpprob_hip_config::flags (A/B forms; 0 = the measured optimum)
FLAG_FLOATING_POINT_STEP, FLAG_NO_SKIP_ROWS, FLAG_SIS_PER_TILE, FLAG_SIS_SEPARATE_READOUT, FLAG_WREL_STORED, FLAG_FP_TILE_PARTIALS, FLAG_WALK_READOUT = 1, 2, 4, 8, 16, 32, 64
FLAG_MULTINOMIAL_LITERAL, FLAG_REPEAT_IN_FLOATING_POINT, FLAG_PAIRED_STEP_LAUNCH = 128, 256, 512
N_KERNEL_CLASSES = 6
KERNEL_CLASS_NAMES = ["smc_step", "scan_partials", "smooth", "finalize", "sis", "resample"]

# every symbol include/cpprob_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "cpprob_hip_abi_version", "cpprob_hip_build_id", "cpprob_hip_device_count", "cpprob_hip_create", "cpprob_hip_destroy", "cpprob_hip_last_error",
    "cpprob_hip_stream", "cpprob_hip_sync", "cpprob_hip_set_hmm", "cpprob_hip_infer_begin", "cpprob_hip_infer_run", "cpprob_hip_infer_summary",
    "cpprob_hip_infer_stats", "cpprob_hip_infer_results", "cpprob_hip_infer_results_device", "cpprob_hip_infer_step_trace", "cpprob_hip_copy_values", "cpprob_hip_copy_ancestors",
    "cpprob_hip_copy_logw", "cpprob_hip_copy_paths", "cpprob_hip_smc_step_begin", "cpprob_hip_smc_step_end",
    "cpprob_hip_smc_finish", "cpprob_hip_smc_first_bad_generation", "cpprob_hip_smc_repair_begin", "cpprob_hip_smc_repair_end", "cpprob_hip_filter_masses", "cpprob_hip_exchange_plan", "cpprob_hip_exchange_pack", "cpprob_hip_exchange_commit", "cpprob_hip_exchange_setup", "cpprob_hip_exchange_transport",
    "cpprob_hip_exchange_pack_async", "cpprob_hip_exchange_commit_async", "cpprob_hip_exchange_status", "cpprob_hip_exchange_direct", "cpprob_hip_exchange_traffic", "cpprob_hip_exchange_store", "cpprob_hip_exchange_remote",
    "cpprob_hip_group_create_external", "cpprob_hip_group_traffic", "cpprob_hip_group_profile", "cpprob_hip_group_profile_read", "cpprob_hip_group_note", "cpprob_hip_group_unique_id", "cpprob_hip_group_create", "cpprob_hip_group_destroy",
    "cpprob_hip_group_last_error", "cpprob_hip_group_begin", "cpprob_hip_group_transport", "cpprob_hip_group_run", "cpprob_hip_group_sync", "cpprob_hip_group_size",
    "cpprob_hip_group_context", "cpprob_hip_group_results", "cpprob_hip_philox_blocks", "cpprob_hip_draw_normal", "cpprob_hip_draw_uniform_smallint",
    "cpprob_hip_draw_discrete", "cpprob_hip_draw_uniform_real", "cpprob_hip_draw_poisson", "cpprob_hip_logpdf_normal", "cpprob_hip_logpdf_uniform_real",
    "cpprob_hip_logpdf_poisson", "cpprob_hip_logpdf_uniform_smallint", "cpprob_hip_logpdf_discrete", "cpprob_hip_logsumexp_ess",
    "cpprob_hip_weighted_moments", "cpprob_hip_weighted_hist", "cpprob_hip_weighted_moments_columns", "cpprob_hip_weighted_hist_columns", "cpprob_hip_resample", "cpprob_hip_smc_bookkeep", "cpprob_hip_smc_bookkeep_fixed", "cpprob_hip_smc_bookkeep_fixed_rs", "cpprob_hip_generic_begin", "cpprob_hip_generic_begin_tiles", "cpprob_hip_generic_quantize", "cpprob_hip_generic_max", "cpprob_hip_generic_quantize_ref", "cpprob_hip_generic_totals", "cpprob_hip_generic_finish", "cpprob_hip_systematic_offset", "cpprob_hip_lineage_gather", "cpprob_hip_lineage_prepare", "cpprob_hip_readback_with_next_result", "cpprob_hip_lineage_moments", "cpprob_hip_lineage_hist", "cpprob_hip_gather_f64",
    "cpprob_hip_gather_i32", "cpprob_hip_profile_enable", "cpprob_hip_profile_read", "cpprob_hip_fastmath",
]


class CpprobHipError(RuntimeError):
    pass


EPRECISION = -5


class Config(C.Structure):
    _fields_ = [("algorithm", C.c_int32), ("model", C.c_int32), ("resampler", C.c_int32), ("resample_scope", C.c_int32),
                ("keep_history", C.c_int32), ("annex_kcols", C.c_int32), ("flags", C.c_uint32), ("fuse_max_tiles", C.c_int32),
                ("ess_threshold", C.c_double), ("seed", C.c_uint64), ("n_particles", C.c_uint64),
                ("particle_offset", C.c_uint64), ("n_global", C.c_uint64)]


class Summary(C.Structure):
    _fields_ = [("log_evidence", C.c_double), ("ess_final", C.c_double), ("log_norm", C.c_double), ("max_logw", C.c_double),
                ("n_predict", C.c_int32), ("stats_per_predict", C.c_int32), ("is_int", C.c_int32), ("n_resampled", C.c_int32),
                ("step_form", C.c_int32), ("n_requantised", C.c_int32)]


FORM_FLOAT, FORM_COUNTS, FORM_FIXED = 0, 1, 2


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class Collectives(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allgather", ALLGATHER_FN)]


class Traffic(C.Structure):
    _fields_ = [("records", C.c_uint64), ("payload_bytes", C.c_uint64), ("wire_bytes", C.c_uint64), ("collective_bytes", C.c_uint64),
                ("transport", C.c_int32), ("remote_lineages", C.c_int32), ("mailbox_collectives", C.c_int32), ("reserved", C.c_int32)]


GROUP_SENDRECV, GROUP_WORLD1_COLLECTIVES, GROUP_SHIP_LINEAGES, GROUP_LIBRARY_COLLECTIVES, GROUP_MAILBOX_COLLECTIVES = 1, 2, 4, 8, 16
TRANSPORT_NONE, TRANSPORT_DIRECT, TRANSPORT_SENDRECV = 0, 1, 2

_lib = None


def load_library(path=None):
    """Loads libcpprob_hip.so.  Import torch first when both live in one process so that the
    already-loaded libamdhip64.so.7 (same SONAME) is shared."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("CPPROB_HIP_LIB") or LIB_PATH      # CPPROB_HIP_LIB: alternative build of the library (A/B runs)
    if not os.path.exists(p):
        raise CpprobHipError("%s is missing: run `python -m cpprob_amd.build` (there is no CPU fallback)" % p)
    L = C.CDLL(p, mode=C.RTLD_GLOBAL)
    vp, u64, i64, i32, dbl, sz = C.c_void_p, C.c_uint64, C.c_int64, C.c_int32, C.c_double, C.c_size_t
    sig = {
        "cpprob_hip_abi_version": (C.c_int, []),
        "cpprob_hip_build_id": (C.c_char_p, []),
        "cpprob_hip_device_count": (C.c_int, []),
        "cpprob_hip_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "cpprob_hip_destroy": (None, [vp]),
        "cpprob_hip_last_error": (C.c_char_p, [vp]),
        "cpprob_hip_stream": (vp, [vp]),
        "cpprob_hip_sync": (C.c_int, [vp]),
        "cpprob_hip_set_hmm": (C.c_int, [vp, i32, C.POINTER(dbl), C.POINTER(dbl)]),
        "cpprob_hip_infer_begin": (C.c_int, [vp, C.POINTER(Config), C.POINTER(dbl), sz]),
        "cpprob_hip_infer_run": (C.c_int, [vp, u64]),
        "cpprob_hip_infer_summary": (C.c_int, [vp, C.POINTER(Summary)]),
        "cpprob_hip_infer_stats": (C.c_int, [vp, C.POINTER(dbl), sz]),
        "cpprob_hip_infer_results": (C.c_int, [vp, vp, vp, sz, vp, vp]),
        "cpprob_hip_infer_results_device": (C.c_int, [vp, vp, sz]),
        "cpprob_hip_infer_step_trace": (C.c_int, [vp, C.POINTER(dbl), C.POINTER(i32)]),
        "cpprob_hip_copy_values": (C.c_int, [vp, vp, sz]),
        "cpprob_hip_copy_ancestors": (C.c_int, [vp, vp, sz]),
        "cpprob_hip_copy_logw": (C.c_int, [vp, vp, sz]),
        "cpprob_hip_copy_paths": (C.c_int, [vp, vp, sz]),
        "cpprob_hip_smc_step_begin": (C.c_int, [vp, i32, u64, vp]),
        "cpprob_hip_smc_step_end": (C.c_int, [vp, i32, vp, i32, i32]),
        "cpprob_hip_smc_finish": (C.c_int, [vp]),
        "cpprob_hip_smc_first_bad_generation": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
        "cpprob_hip_smc_repair_begin": (C.c_int, [vp, C.c_int32, vp]),
        "cpprob_hip_smc_repair_end": (C.c_int, [vp, C.c_int32, vp, C.c_int32, C.c_int32, vp]),
        "cpprob_hip_filter_masses": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i32)]),
        "cpprob_hip_exchange_plan": (C.c_int, [vp, i32, i32, i32, vp, vp, vp, C.POINTER(i32)]),
        "cpprob_hip_exchange_pack": (C.c_int, [vp, i32, vp]),
        "cpprob_hip_exchange_commit": (C.c_int, [vp, i32, vp]),
        "cpprob_hip_exchange_setup": (C.c_int, [vp, i32, i32, vp, i32, u64]),
        "cpprob_hip_exchange_transport": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(u64), C.POINTER(u64)]),
        "cpprob_hip_exchange_pack_async": (C.c_int, [vp, i32]),
        "cpprob_hip_exchange_commit_async": (C.c_int, [vp, i32]),
        "cpprob_hip_exchange_status": (C.c_int, [vp, C.POINTER(i32), C.POINTER(u64)]),
        "cpprob_hip_group_unique_id": (C.c_int, [vp, sz]),
        "cpprob_hip_group_create": (C.c_int, [C.POINTER(i32), i32, i32, i32, vp, C.POINTER(vp)]),
        "cpprob_hip_group_destroy": (None, [vp]),
        "cpprob_hip_group_last_error": (C.c_char_p, [vp]),
        "cpprob_hip_group_begin": (C.c_int, [vp, C.POINTER(Config), C.POINTER(dbl), sz, vp]),
        "cpprob_hip_group_transport": (C.c_int, [vp, u64, i32, C.c_uint32]),
        "cpprob_hip_group_create_external": (C.c_int, [i32, i32, i32, C.POINTER(Collectives), C.POINTER(vp)]),
        "cpprob_hip_group_traffic": (C.c_int, [vp, C.POINTER(Traffic)]),
        "cpprob_hip_group_profile": (C.c_int, [vp, i32]),
        "cpprob_hip_group_profile_read": (C.c_int, [vp, C.POINTER(dbl)]),
        "cpprob_hip_group_note": (C.c_char_p, [vp]),
        "cpprob_hip_exchange_direct": (C.c_int, [vp, vp]),
        "cpprob_hip_exchange_store": (C.c_int, [vp, vp]),
        "cpprob_hip_exchange_remote": (C.c_int, [vp, vp]),
        "cpprob_hip_exchange_traffic": (C.c_int, [vp, vp, sz, C.POINTER(u64), C.POINTER(u64)]),
        "cpprob_hip_group_run": (C.c_int, [vp, u64]),
        "cpprob_hip_group_sync": (C.c_int, [vp]),
        "cpprob_hip_group_size": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "cpprob_hip_group_context": (vp, [vp, i32]),
        "cpprob_hip_group_results": (C.c_int, [vp, C.POINTER(Summary), C.POINTER(dbl), sz, C.POINTER(i32)]),
        "cpprob_hip_philox_blocks": (C.c_int, [vp, u64, u64, u64, sz, vp]),
        "cpprob_hip_draw_normal": (C.c_int, [vp, u64, u64, u64, dbl, dbl, sz, vp]),
        "cpprob_hip_draw_uniform_smallint": (C.c_int, [vp, u64, u64, u64, i64, i64, sz, vp]),
        "cpprob_hip_draw_discrete": (C.c_int, [vp, u64, u64, u64, C.POINTER(dbl), i32, sz, vp]),
        "cpprob_hip_draw_uniform_real": (C.c_int, [vp, u64, u64, u64, dbl, dbl, sz, vp]),
        "cpprob_hip_draw_poisson": (C.c_int, [vp, u64, u64, u64, dbl, sz, vp]),
        "cpprob_hip_logpdf_normal": (C.c_int, [vp, vp, vp, vp, sz, vp]),
        "cpprob_hip_logpdf_uniform_real": (C.c_int, [vp, vp, vp, vp, sz, vp]),
        "cpprob_hip_logpdf_poisson": (C.c_int, [vp, vp, vp, sz, vp]),
        "cpprob_hip_logpdf_uniform_smallint": (C.c_int, [vp, vp, i64, i64, sz, vp]),
        "cpprob_hip_logpdf_discrete": (C.c_int, [vp, vp, C.POINTER(dbl), i32, sz, vp]),
        "cpprob_hip_fastmath": (C.c_int, [vp, i32, vp, sz, vp, vp]),
        "cpprob_hip_logsumexp_ess": (C.c_int, [vp, vp, sz, C.POINTER(dbl)]),
        "cpprob_hip_weighted_moments": (C.c_int, [vp, vp, vp, sz, C.POINTER(dbl)]),
        "cpprob_hip_weighted_hist": (C.c_int, [vp, vp, vp, sz, i32, C.POINTER(dbl)]),
        "cpprob_hip_weighted_moments_columns": (C.c_int, [vp, vp, sz, sz, vp, sz, C.POINTER(dbl)]),
        "cpprob_hip_weighted_hist_columns": (C.c_int, [vp, vp, sz, sz, vp, sz, i32, C.POINTER(dbl), C.POINTER(dbl)]),
        "cpprob_hip_resample": (C.c_int, [vp, i32, vp, sz, u64, u64, u64, sz, u64, vp]),
        "cpprob_hip_smc_bookkeep": (C.c_int, [vp, i32, vp, sz, u64, i32, i32, dbl, vp, vp, vp, vp]),
        "cpprob_hip_smc_bookkeep_fixed": (C.c_int, [vp, vp, sz, u64, i32, i32, dbl, vp, vp, vp, vp]),
        "cpprob_hip_smc_bookkeep_fixed_rs": (C.c_int, [vp, i32, vp, sz, u64, i32, i32, dbl, vp, vp, vp, vp]),
        "cpprob_hip_generic_begin": (C.c_int, [vp, sz, vp]),
        "cpprob_hip_generic_begin_tiles": (C.c_int, [vp, sz, vp]),
        "cpprob_hip_generic_quantize": (C.c_int, [vp, i32, vp, sz]),
        "cpprob_hip_generic_max": (C.c_int, [vp, i32, vp, sz]),
        "cpprob_hip_generic_quantize_ref": (C.c_int, [vp, i32, vp, sz, dbl]),
        "cpprob_hip_generic_totals": (C.c_int, [vp, i32, sz, vp]),
        "cpprob_hip_generic_finish": (C.c_int, [vp, i32, sz, dbl, vp, vp, vp, vp]),
        "cpprob_hip_systematic_offset": (dbl, [u64, u64]),
        "cpprob_hip_lineage_gather": (C.c_int, [vp, vp, vp, i32, sz, vp, i32, vp, i32, vp]),
        "cpprob_hip_lineage_prepare": (C.c_int, [vp, vp, i32, i32]),
        "cpprob_hip_readback_with_next_result": (C.c_int, [vp, vp, vp, sz]),
        "cpprob_hip_lineage_moments": (C.c_int, [vp, vp, vp, i32, sz, vp, vp, i32, vp, vp]),
        "cpprob_hip_lineage_hist": (C.c_int, [vp, vp, vp, i32, sz, vp, vp, i32, vp, i32, vp, vp]),
        "cpprob_hip_gather_f64": (C.c_int, [vp, vp, vp, sz, vp]),
        "cpprob_hip_gather_i32": (C.c_int, [vp, vp, vp, sz, vp]),
        "cpprob_hip_profile_enable": (C.c_int, [vp, i32]),
        "cpprob_hip_profile_read": (C.c_int, [vp, C.POINTER(dbl), C.POINTER(i64), i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.cpprob_hip_abi_version() != 3:
        raise CpprobHipError("ABI version mismatch")
    if p == LIB_PATH:
        # the in-tree binary must be the build of the sources next to it (a stale .so that merely looks newer is refused)
        from . import build as B
        have, want = L.cpprob_hip_build_id().decode(), B.source_hash()
        if have != want:
            raise CpprobHipError("%s was built from other sources (build id %s, sources %s): run `python -m cpprob_amd.build`" % (p, have, want))
    if path is None:
        _lib = L
    return L


def _dptr(t):
    """Device pointer of a torch tensor (contiguous), or None.

    Stream contract (include/cpprob_hip.h): the engine's stream is non-blocking, so work torch queued on ITS stream for a
    tensor (the fill of torch.zeros, an H2D copy) is not ordered before what the engine then does with that memory.  The
    caller completes its tensors first -- torch.cuda.current_stream().synchronize() after creating them, once."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


class Engine:
    """One context = one device, one stream, one particle shard (include/cpprob_hip.h)."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.cpprob_hip_create(int(device), C.byref(h))
        if rc:
            msg = self.L.cpprob_hip_last_error(None)
            raise CpprobHipError("cpprob_hip_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self.h = h
        self.device = device
        self.cfg = None
        self.T = 0
        self.is_int = False
        self.K = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.cpprob_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            msg = self.L.cpprob_hip_last_error(self.h)
            e = CpprobHipError("cpprob_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
            e.code = rc
            raise e

    @property
    def stream_ptr(self):
        return self.L.cpprob_hip_stream(self.h)

    def sync(self):
        self._chk(self.L.cpprob_hip_sync(self.h))

    def set_hmm(self, means, trans):
        """The table of MODEL_HMM_TABLE: k = len(means) states (2..8), emission N(means[s], 1), transition rows trans[s] (weights)."""
        m = np.ascontiguousarray(means, np.float64)
        tr = np.ascontiguousarray(trans, np.float64)
        assert tr.shape == (len(m), len(m))
        self._chk(self.L.cpprob_hip_set_hmm(self.h, len(m), m.ctypes.data_as(C.POINTER(C.c_double)), tr.ctypes.data_as(C.POINTER(C.c_double))))

    # ---- cpprob::inference --------------------------------------------------------------
    def begin(self, algorithm, model, observes, n_particles, seed=12345, resampler=RESAMPLE_SYSTEMATIC, ess_threshold=2.0,
              particle_offset=0, n_global=None, scope=SCOPE_GLOBAL, keep_history=True, flags=0, fuse_max_tiles=0):
        """keep_history=False: filtering only -- two rows of the particle store and no ancestors (memory O(N) instead of
        O(N T)); stats() then holds every predict hit's statistics under ITS generation's weights, and values() / ancestors() /
        paths() raise."""
        obs = np.ascontiguousarray(observes, np.float64)
        self._begin_args = dict(algorithm=algorithm, model=model, observes=obs.copy(), n_particles=n_particles, seed=seed, resampler=resampler, ess_threshold=ess_threshold,
                                particle_offset=particle_offset, n_global=n_global, scope=scope, keep_history=keep_history, flags=flags, fuse_max_tiles=fuse_max_tiles)
        cfg = Config(algorithm, model, resampler, scope, 1 if keep_history else 0, 0, int(flags), int(fuse_max_tiles), float(ess_threshold), int(seed),
                     int(n_particles), int(particle_offset), int(n_particles if n_global is None else n_global))
        self._chk(self.L.cpprob_hip_infer_begin(self.h, C.byref(cfg), obs.ctypes.data_as(C.POINTER(C.c_double)), len(obs)))
        self.cfg = cfg
        gauss = model in (MODEL_GAUSSIAN_UNKNOWN_MEAN, MODEL_GAUSSIAN_README)
        self.T = 1 if gauss else len(obs)
        self.is_int = model in (MODEL_HMM3, MODEL_HMM_TABLE)
        self.K = 8 if model == MODEL_HMM_TABLE else (3 if self.is_int else 2)
        self.n = int(n_particles)
        return self

    def rebegin(self, extra_flags):
        """begin() again with the same arguments and further cpprob_hip_config::flags."""
        a = dict(self._begin_args)
        a["flags"] = int(a["flags"]) | int(extra_flags)
        return self.begin(**a)

    def run(self, run_index=0):
        self._chk(self.L.cpprob_hip_infer_run(self.h, int(run_index)))

    def summary(self):
        s = Summary()
        self._chk(self.L.cpprob_hip_infer_summary(self.h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in Summary._fields_}

    def stats(self):
        out = np.zeros((self.T, self.K))
        self._chk(self.L.cpprob_hip_infer_stats(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
        return out

    def results(self):
        """summary, stats and step trace in one call and one stream synchronisation (cpprob_hip_infer_results)."""
        s = Summary()
        out = np.zeros((self.T, self.K))
        ess = np.zeros(self.T)
        res = np.zeros(self.T, np.int32)
        self._chk(self.L.cpprob_hip_infer_results(self.h, C.byref(s), out.ctypes.data, out.size, ess.ctypes.data, res.ctypes.data))
        return {f: getattr(s, f) for f, _ in Summary._fields_}, out, ess, res

    def results_device(self, out):
        """{log_evidence, ess, log_norm, max_logw, stats...} of the enqueued run into a device tensor; no host sync."""
        self._chk(self.L.cpprob_hip_infer_results_device(self.h, _dptr(out), out.numel()))

    def step_trace(self):
        ess = np.zeros(self.T)
        res = np.zeros(self.T, np.int32)
        self._chk(self.L.cpprob_hip_infer_step_trace(self.h, ess.ctypes.data_as(C.POINTER(C.c_double)),
                                                     res.ctypes.data_as(C.POINTER(C.c_int32))))
        return ess, res

    def values(self):
        out = np.zeros((self.T, self.n), np.int32 if self.is_int else np.float64)
        self._chk(self.L.cpprob_hip_copy_values(self.h, out.ctypes.data, out.nbytes))
        return out

    def ancestors(self):
        out = np.zeros((self.T, self.n), np.int32)
        self._chk(self.L.cpprob_hip_copy_ancestors(self.h, out.ctypes.data, out.nbytes))
        return out

    def logw(self):
        out = np.zeros(self.n)
        self._chk(self.L.cpprob_hip_copy_logw(self.h, out.ctypes.data, out.nbytes))
        return out

    def paths(self):
        out = np.zeros((self.T, self.n), np.int32 if self.is_int else np.float64)
        self._chk(self.L.cpprob_hip_copy_paths(self.h, out.ctypes.data, out.nbytes))
        return out

    # ---- sharded SMC ---------------------------------------------------------------------
    def step_begin(self, t, local_totals, run_index=0):
        """local_totals: torch float64 tensor (>= 3) on this device, filled on the engine's stream."""
        self._chk(self.L.cpprob_hip_smc_step_begin(self.h, int(t), int(run_index), _dptr(local_totals)))

    def step_end(self, t, all_totals, world, rank):
        self._chk(self.L.cpprob_hip_smc_step_end(self.h, int(t), _dptr(all_totals), int(world), int(rank)))

    def finish(self):
        self._chk(self.L.cpprob_hip_smc_finish(self.h))

    def first_bad_generation(self):
        """After finish(): the first generation whose fixed-point weights lost their bits (-1: none) and the run's largest gap in nats."""
        g, gap = C.c_int32(-1), C.c_double(0.0)
        self._chk(self.L.cpprob_hip_smc_first_bad_generation(self.h, C.byref(g), C.byref(gap)))
        return int(g.value), float(gap.value)

    def repair_begin(self, g, local3):
        self._chk(self.L.cpprob_hip_smc_repair_begin(self.h, int(g), _dptr(local3)))

    def repair_end(self, g, all3, world, rank, local3):
        self._chk(self.L.cpprob_hip_smc_repair_end(self.h, int(g), _dptr(all3), int(world), int(rank), _dptr(local3)))

    # ---- exchange scope: exact global resampling with migration ----------------------------
    def exchange_plan(self, t, world, rank, shard_begin):
        """-> (do_resample, send_counts[world], recv_counts[world]) in lineage records of t + 1 values.  Host-synchronising."""
        sb = np.ascontiguousarray(shard_begin, dtype=np.uint64)
        send = np.zeros(world, np.uint64)
        recv = np.zeros(world, np.uint64)
        flag = C.c_int32(0)
        self._chk(self.L.cpprob_hip_exchange_plan(self.h, int(t), int(world), int(rank), sb.ctypes.data, send.ctypes.data, recv.ctypes.data, C.byref(flag)))
        return bool(flag.value), send, recv

    def exchange_traffic(self):
        """(records sent after each step [T], total records, total bytes) of this rank's last exchange-scope run on a fixed transport."""
        per = np.zeros(self.T, np.int64)
        rec, byt = C.c_uint64(0), C.c_uint64(0)
        self._chk(self.L.cpprob_hip_exchange_traffic(self.h, per.ctypes.data, per.size, C.byref(rec), C.byref(byt)))
        return per, int(rec.value), int(byt.value)

    def exchange_pack(self, t, send):
        self._chk(self.L.cpprob_hip_exchange_pack(self.h, int(t), _dptr(send) if send is not None else None))

    def exchange_commit(self, t, recv):
        self._chk(self.L.cpprob_hip_exchange_commit(self.h, int(t), _dptr(recv) if recv is not None else None))

    # ---- building blocks (torch tensors on this device carry the memory) -------------------
    def philox_blocks(self, seed, pid0, draw, out):
        self._chk(self.L.cpprob_hip_philox_blocks(self.h, seed, pid0, draw, out.numel() // 4, _dptr(out)))

    def draw_normal(self, seed, pid0, draw, mean, sigma, out):
        self._chk(self.L.cpprob_hip_draw_normal(self.h, seed, pid0, draw, mean, sigma, out.numel(), _dptr(out)))

    def draw_uniform_smallint(self, seed, pid0, draw, a, b, out):
        self._chk(self.L.cpprob_hip_draw_uniform_smallint(self.h, seed, pid0, draw, a, b, out.numel(), _dptr(out)))

    def draw_discrete(self, seed, pid0, draw, weights, out):
        w = (C.c_double * len(weights))(*weights)
        self._chk(self.L.cpprob_hip_draw_discrete(self.h, seed, pid0, draw, w, len(weights), out.numel(), _dptr(out)))

    def draw_uniform_real(self, seed, pid0, draw, a, b, out):
        self._chk(self.L.cpprob_hip_draw_uniform_real(self.h, seed, pid0, draw, a, b, out.numel(), _dptr(out)))

    def draw_poisson(self, seed, pid0, draw, mean, out):
        self._chk(self.L.cpprob_hip_draw_poisson(self.h, seed, pid0, draw, mean, out.numel(), _dptr(out)))

    def logpdf_normal(self, x, mean, sigma, out):
        self._chk(self.L.cpprob_hip_logpdf_normal(self.h, _dptr(x), _dptr(mean), _dptr(sigma), x.numel(), _dptr(out)))

    def logpdf_uniform_real(self, x, a, b, out):
        self._chk(self.L.cpprob_hip_logpdf_uniform_real(self.h, _dptr(x), _dptr(a), _dptr(b), x.numel(), _dptr(out)))

    def logpdf_poisson(self, x, mean, out):
        self._chk(self.L.cpprob_hip_logpdf_poisson(self.h, _dptr(x), _dptr(mean), x.numel(), _dptr(out)))

    def logpdf_uniform_smallint(self, x, a, b, out):
        self._chk(self.L.cpprob_hip_logpdf_uniform_smallint(self.h, _dptr(x), a, b, x.numel(), _dptr(out)))

    def logpdf_discrete(self, x, weights, out):
        w = (C.c_double * len(weights))(*weights)
        self._chk(self.L.cpprob_hip_logpdf_discrete(self.h, _dptr(x), w, len(weights), x.numel(), _dptr(out)))

    def fastmath(self, which, x, out0, out1=None):
        """which: 0 log01, 1 sincospi02 (out0 = sin, out1 = cos), 2 exp_nonpos, 3 fix_weight against reference 0; torch float64 tensors on this device."""
        self._chk(self.L.cpprob_hip_fastmath(self.h, int(which), _dptr(x), x.numel(), _dptr(out0), _dptr(out1)))

    def logsumexp_ess(self, logw):
        out = (C.c_double * 3)()
        self._chk(self.L.cpprob_hip_logsumexp_ess(self.h, _dptr(logw), logw.numel(), out))
        return tuple(out)

    def weighted_moments(self, x, logw):
        out = (C.c_double * 4)()
        self._chk(self.L.cpprob_hip_weighted_moments(self.h, _dptr(x), _dptr(logw), logw.numel(), out))
        return tuple(out)

    def weighted_hist(self, x, logw, k):
        out = (C.c_double * 8)()
        self._chk(self.L.cpprob_hip_weighted_hist(self.h, _dptr(x), _dptr(logw), logw.numel(), k, out))
        return np.array(out[:k])

    def weighted_moments_columns(self, x, logw):
        """x: [n_cols, n] device tensor (contiguous rows) -> array [n_cols, 4] of (mean, variance, logsumexp, ess)."""
        n_cols, n = int(x.shape[0]), int(x.shape[1])
        out = (C.c_double * (4 * n_cols))()
        self._chk(self.L.cpprob_hip_weighted_moments_columns(self.h, _dptr(x), n_cols, n, _dptr(logw), n, out))
        return np.array(out[:]).reshape(n_cols, 4)

    def weighted_hist_columns(self, x, logw, k):
        """x: [n_cols, n] int32 device tensor -> array [n_cols, k] of P(x = s)."""
        n_cols, n = int(x.shape[0]), int(x.shape[1])
        out = (C.c_double * (k * n_cols))()
        self._chk(self.L.cpprob_hip_weighted_hist_columns(self.h, _dptr(x), n_cols, n, _dptr(logw), n, k, out, None))
        return np.array(out[:]).reshape(n_cols, k)

    def lineage_gather(self, anc, resampled, cols, gens, out):
        """anc [T, n] int32, resampled [T] int32, cols [H, n] (fp64 / int32) recorded in generations gens[h]; out [H, n]: the traces."""
        T, n = int(anc.shape[0]), int(anc.shape[1])
        g = (C.c_int32 * len(gens))(*gens)
        self._chk(self.L.cpprob_hip_lineage_gather(self.h, _dptr(anc), _dptr(resampled), T, n, _dptr(cols), 0 if cols.dtype.is_floating_point else 1, g, len(gens), _dptr(out)))

    def lineage_moments(self, anc, resampled, cols, gens, logw):
        """-> array [H, 4] = {mean, variance, logsumexp, ess} of every record along the final particles' lineages."""
        T, n = int(anc.shape[0]), int(anc.shape[1])
        g = (C.c_int32 * len(gens))(*gens)
        out = (C.c_double * (4 * len(gens)))()
        self._chk(self.L.cpprob_hip_lineage_moments(self.h, _dptr(anc), _dptr(resampled), T, n, _dptr(cols), g, len(gens), _dptr(logw), out))
        return np.array(out[:]).reshape(len(gens), 4)

    def lineage_hist(self, anc, resampled, cols, gens, logw, k):
        """-> array [H, k] of P(record h = s) along the final particles' lineages."""
        T, n = int(anc.shape[0]), int(anc.shape[1])
        g = (C.c_int32 * len(gens))(*gens)
        out = (C.c_double * (k * len(gens)))()
        self._chk(self.L.cpprob_hip_lineage_hist(self.h, _dptr(anc), _dptr(resampled), T, n, _dptr(cols), g, len(gens), _dptr(logw), k, out, None))
        return np.array(out[:]).reshape(len(gens), k)

    def readback_with_next_result(self, src, out):
        """Hang a read-back of device tensor `src` into the numpy array `out` on the next lineage_moments / lineage_hist call (one wait for both)."""
        self._chk(self.L.cpprob_hip_readback_with_next_result(self.h, _dptr(src), out.ctypes.data, out.nbytes))

    def resample(self, kind, logw, seed, step, anc_out, j0=0, n_total_out=None):
        n_out = anc_out.numel()
        nt = logw.numel() if n_total_out is None else n_total_out
        self._chk(self.L.cpprob_hip_resample(self.h, kind, _dptr(logw), logw.numel(), seed, step, j0, n_out, nt, _dptr(anc_out)))

    def smc_bookkeep(self, kind, logw, seed, step, last, ess_frac, ess, resampled, log_z, anc):
        """Device-side SMC bookkeeping of one step (all tensors on this device; no host sync)."""
        self._chk(self.L.cpprob_hip_smc_bookkeep(self.h, kind, _dptr(logw), logw.numel(), seed, int(step), 1 if last else 0, float(ess_frac),
                                                 _dptr(ess), _dptr(resampled), _dptr(log_z), _dptr(anc)))

    def smc_bookkeep_fixed(self, logw, seed, step, last, ess_frac, ess, resampled, log_z, anc):
        """The same for systematic resampling on fixed-point weights (integer masses; cpprob_hip_smc_bookkeep_fixed)."""
        self._chk(self.L.cpprob_hip_smc_bookkeep_fixed(self.h, _dptr(logw), logw.numel(), seed, int(step), 1 if last else 0, float(ess_frac),
                                                       _dptr(ess), _dptr(resampled), _dptr(log_z), _dptr(anc)))

    def gather(self, src, idx, dst):
        import torch
        fn = self.L.cpprob_hip_gather_f64 if src.dtype == torch.float64 else self.L.cpprob_hip_gather_i32
        self._chk(fn(self.h, _dptr(src), _dptr(idx), idx.numel(), _dptr(dst)))

    def profile_enable(self, on=True):
        self._chk(self.L.cpprob_hip_profile_enable(self.h, 1 if on else 0))

    def profile_read(self, reset=True):
        ms = (C.c_double * N_KERNEL_CLASSES)()
        calls = (C.c_int64 * N_KERNEL_CLASSES)()
        self._chk(self.L.cpprob_hip_profile_read(self.h, ms, calls, 1 if reset else 0))
        return {KERNEL_CLASS_NAMES[k]: (ms[k], calls[k]) for k in range(N_KERNEL_CLASSES)}


class Group:
    """One joint population over several GPUs, driven by the library's own host code (cpprob_amd/csrc/group.hpp): the
    exchange scope's per-step protocol with RCCL collectives on each context's stream and no host synchronisation inside a
    run.  devices: this process's GPUs.  world > len(devices): one rank of a multi-process group (unique_id from
    Group.unique_id() on rank 0, distributed by the launcher).  All devices equal: loopback (every rank on that one GPU)."""

    def __init__(self, devices, world=None, first_rank=0, unique_id=None, allgather=None):
        """allgather(bytes_in) -> bytes of every rank in rank order: the caller's own collective (cpprob_hip_group_create_external);
        then devices = [this rank's GPU], world and first_rank (= rank) as given."""
        self.L = load_library()
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        world = len(devices) if world is None else int(world)
        h = C.c_void_p()
        if allgather is not None:
            def _cb(user, h_in, h_out, nbytes):
                try:
                    out = allgather(C.string_at(h_in, nbytes))
                    if len(out) != nbytes * world:
                        return 1
                    C.memmove(h_out, out, len(out))
                    return 0
                except Exception:        # noqa: reported as a failed collective
                    return 1
            self._cb = ALLGATHER_FN(_cb)              # (kept alive with the group)
            self._coll = Collectives(None, self._cb)
            rc = self.L.cpprob_hip_group_create_external(int(devices[0]), world, int(first_rank), C.byref(self._coll), C.byref(h))
        else:
            uid = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
            rc = self.L.cpprob_hip_group_create(devs, len(devices), world, int(first_rank), uid, C.byref(h))
        if rc:
            msg = self.L.cpprob_hip_group_last_error(None)
            raise CpprobHipError("cpprob_hip_group_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self.h = h
        self.world, self.n_local, self.first_rank = world, len(devices), int(first_rank)
        self.T = self.K = 0
        self.is_int = False

    @staticmethod
    def unique_id():
        L = load_library()
        buf = C.create_string_buffer(128)
        rc = L.cpprob_hip_group_unique_id(buf, 128)
        if rc:
            msg = L.cpprob_hip_group_last_error(None)
            raise CpprobHipError("cpprob_hip_group_unique_id failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        return buf.raw

    def close(self):
        if getattr(self, "h", None):
            self.L.cpprob_hip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            msg = self.L.cpprob_hip_group_last_error(self.h)
            raise CpprobHipError("cpprob_hip group error %d: %s" % (rc, msg.decode() if msg else "?"))

    def begin(self, algorithm, model, observes, n_particles, seed=12345, resampler=RESAMPLE_SYSTEMATIC, ess_threshold=2.0, shard_sizes=None, flags=0,
              keep_history=True):
        """n_particles = the whole population."""
        obs = np.ascontiguousarray(observes, np.float64)
        cfg = Config(algorithm, model, resampler, SCOPE_EXCHANGE, 1 if keep_history else 0, 0, int(flags), 0, float(ess_threshold), int(seed), int(n_particles), 0,
                     int(n_particles))
        ss = None
        if shard_sizes is not None:
            ss = np.ascontiguousarray(shard_sizes, np.uint64)
            assert len(ss) == self.world
        self._chk(self.L.cpprob_hip_group_begin(self.h, C.byref(cfg), obs.ctypes.data_as(C.POINTER(C.c_double)), len(obs),
                                                ss.ctypes.data if ss is not None else None))
        gauss = model in (MODEL_GAUSSIAN_UNKNOWN_MEAN, MODEL_GAUSSIAN_README)
        self.T = 1 if gauss else len(obs)
        self.is_int = model in (MODEL_HMM3, MODEL_HMM_TABLE)
        self.K = 8 if model == MODEL_HMM_TABLE else (3 if self.is_int else 2)
        self.n = int(n_particles)
        return self

    def transport(self, records_per_peer=0, all_peers=-1, flags=0):
        """Transport parameters of the next begin() (0 / -1: the defaults; flags: GROUP_SENDRECV, GROUP_WORLD1_COLLECTIVES)."""
        self._chk(self.L.cpprob_hip_group_transport(self.h, int(records_per_peer), int(all_peers), int(flags)))

    PHASES = ["step_and_totals", "allgather", "totals_handover", "pack", "barrier", "commit", "mailbox_wait"]

    def profile(self, on=True):
        """HIP events between the launches of every step of the following runs (cpprob_hip_group_profile)."""
        self._chk(self.L.cpprob_hip_group_profile(self.h, 1 if on else 0))

    def profile_read(self):
        """Microseconds per rank-step of the last (profiled) run, by phase; 'steps' = steps timed."""
        out = (C.c_double * 8)()
        self._chk(self.L.cpprob_hip_group_profile_read(self.h, out))
        d = {k: out[i] for i, k in enumerate(self.PHASES)}
        d["steps"] = int(out[7])
        return d

    def note(self):
        """Which collectives / transport the group settled on, and why."""
        return self.L.cpprob_hip_group_note(self.h).decode()

    def traffic(self):
        """Of the run results() last collected: dict of records, payload_bytes, wire_bytes, collective_bytes, transport."""
        t = Traffic()
        self._chk(self.L.cpprob_hip_group_traffic(self.h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in Traffic._fields_}

    def run(self, run_index=0):
        self._chk(self.L.cpprob_hip_group_run(self.h, int(run_index)))

    def sync(self):
        self._chk(self.L.cpprob_hip_group_sync(self.h))

    def results(self):
        """(stats[T, K], summary dict, reruns): the joint population's numbers, as one GPU holding all particles would report them."""
        s = Summary()
        out = np.zeros((self.T, self.K))
        rr = C.c_int32(0)
        self._chk(self.L.cpprob_hip_group_results(self.h, C.byref(s), out.ctypes.data_as(C.POINTER(C.c_double)), out.size, C.byref(rr)))
        return out, {f: getattr(s, f) for f, _ in Summary._fields_}, int(rr.value)

    def context(self, local_index):
        """The rank's context as an Engine view (not owned): shard-level read-outs (paths, values, logw)."""
        e = Engine.__new__(Engine)
        e.L = self.L
        e.h = C.c_void_p(self.L.cpprob_hip_group_context(self.h, int(local_index)))
        e.device = None
        e.T, e.K, e.is_int = self.T, self.K, self.is_int
        e.close = lambda: None
        return e
