"""ctypes front-end of oracle/cpprob_oracle.c (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODEL_GAUSSIAN_UNKNOWN_MEAN, MODEL_GAUSSIAN_README, MODEL_LINEAR_GAUSSIAN_1D, MODEL_HMM3, MODEL_GAUSSIAN_2D_UNKNOWN_MEAN = 0, 1, 2, 3, 4
RESAMPLE_SYSTEMATIC, RESAMPLE_STRATIFIED, RESAMPLE_MULTINOMIAL, RESAMPLE_MULTINOMIAL_LITERAL = 0, 1, 2, 3
RESAMPLE_DRAW_BASE = 1 << 40

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_up = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "_build", "liboracle.so")
    src = os.path.join(_HERE, "cpprob_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "_build/liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u64, i64, dbl, sz = C.c_uint64, C.c_int64, C.c_double, C.c_size_t
        L.orc_philox4x32_10.argtypes = [_up, _up, _up]
        L.orc_draw_block.argtypes = [u64, u64, u64, _up]
        L.orc_u01_53.restype = dbl; L.orc_u01_53.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_u01_32.restype = dbl; L.orc_u01_32.argtypes = [C.c_uint32]
        L.orc_box_muller.argtypes = [_up, _dp]
        L.orc_draw_word.restype = C.c_uint32; L.orc_draw_word.argtypes = [u64, u64, u64]
        L.orc_draw_std_normal.restype = dbl; L.orc_draw_std_normal.argtypes = [u64, u64, u64]
        L.orc_draw_u01_53.restype = dbl; L.orc_draw_u01_53.argtypes = [u64, u64, u64]
        L.orc_draw_normal.restype = dbl; L.orc_draw_normal.argtypes = [u64, u64, u64, dbl, dbl]
        L.orc_draw_smallint.restype = u64; L.orc_draw_smallint.argtypes = [u64, u64, u64, u64, u64]
        L.orc_draw_discrete.restype = u64; L.orc_draw_discrete.argtypes = [u64, u64, u64, _dp, C.c_int]
        L.orc_draw_uniform_real.restype = dbl; L.orc_draw_uniform_real.argtypes = [u64, u64, u64, dbl, dbl]
        L.orc_draw_poisson.restype = i64; L.orc_draw_poisson.argtypes = [u64, u64, u64, dbl]
        L.orc_normal_logpdf.restype = dbl; L.orc_normal_logpdf.argtypes = [dbl, dbl, dbl]
        L.orc_uniform_smallint_logpdf.restype = dbl; L.orc_uniform_smallint_logpdf.argtypes = [i64, i64, i64]
        L.orc_discrete_logpdf.restype = dbl; L.orc_discrete_logpdf.argtypes = [i64, _dp, C.c_int]
        L.orc_uniform_real_logpdf.restype = dbl; L.orc_uniform_real_logpdf.argtypes = [dbl, dbl, dbl]
        L.orc_poisson_logpdf.restype = dbl; L.orc_poisson_logpdf.argtypes = [i64, dbl]
        L.orc_model_num_predicts.restype = C.c_int; L.orc_model_num_predicts.argtypes = [C.c_int, sz]
        L.orc_sis.restype = C.c_int
        L.orc_sis.argtypes = [C.c_int, _dp, sz, u64, u64, u64, C.c_void_p, C.c_void_p, _dp]
        L.orc_sis_faithful.restype = C.c_int
        L.orc_sis_faithful.argtypes = [C.c_int, _dp, sz, u64, u64, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_logsumexp.restype = dbl; L.orc_logsumexp.argtypes = [_dp, u64]
        L.orc_weighted_moments.argtypes = [_dp, _dp, u64, _dp]
        L.orc_weighted_hist.argtypes = [_ip, _dp, u64, C.c_int, _dp]
        L.orc_resample.restype = C.c_int
        L.orc_resample.argtypes = [C.c_int, _dp, u64, u64, u64, u64, u64, u64, _ip, C.c_void_p]
        L.orc_resample_table_systematic.restype = C.c_int
        L.orc_resample_table_systematic.argtypes = [_ip, u64, _dp, _u64p, _u64p, C.c_int, u64, u64, u64, u64, u64, _ip]
        L.orc_resample_table_stratified.restype = C.c_int
        L.orc_resample_table_stratified.argtypes = [_ip, u64, _dp, _u64p, _u64p, C.c_int, u64, u64, u64, u64, u64, _ip]
        L.orc_resample_table_multinomial.restype = C.c_int
        L.orc_resample_table_multinomial.argtypes = [_ip, u64, _dp, u64, u64, u64, _ip]
        L.orc_fix_weight.restype = C.c_uint32; L.orc_fix_weight.argtypes = [dbl, dbl]
        L.orc_fix_weights.argtypes = [_dp, u64, dbl, _up]
        L.orc_resample_fixed_systematic.restype = C.c_int
        L.orc_resample_fixed_systematic.argtypes = [_up, u64, u64, u64, C.c_int, u64, u64, u64, u64, u64, _ip]
        L.orc_resample_fixed_stratified.restype = C.c_int
        L.orc_resample_fixed_stratified.argtypes = [_up, u64, u64, u64, C.c_int, u64, u64, u64, u64, u64, _ip]
        L.orc_resample_fixed_multinomial.restype = C.c_int
        L.orc_resample_fixed_multinomial.argtypes = [_up, u64, u64, u64, u64, u64, u64, u64, _ip]
        L.orc_multinomial_threshold.restype = u64; L.orc_multinomial_threshold.argtypes = [u64, u64, u64, u64]
        L.orc_resample_fixed_multinomial_strata.restype = C.c_int
        L.orc_resample_fixed_multinomial_strata.argtypes = [_up, u64, u64, u64, u64, _ip]
        L.orc_strata_levels.restype = C.c_int; L.orc_strata_levels.argtypes = [u64]
        L.orc_resample_fixed_multinomial_strata_shard.restype = C.c_int
        L.orc_resample_fixed_multinomial_strata_shard.argtypes = [_up, u64, u64, u64, u64, u64, u64, _ip]
        L.orc_resample_table_multinomial_shard.restype = C.c_int
        L.orc_resample_table_multinomial_shard.argtypes = [_ip, u64, _dp, _u64p, _u64p, C.c_int, u64, u64, u64, _ip]
        L.orc_strata_thresholds_fixed.restype = C.c_int
        L.orc_strata_thresholds_fixed.argtypes = [u64, u64, u64, u64, _u64p, _ip]
        L.orc_strata_thresholds_table.restype = C.c_int
        L.orc_strata_thresholds_table.argtypes = [dbl, u64, u64, u64, _dp, _ip]
        L.orc_table_cdf.restype = dbl; L.orc_table_cdf.argtypes = [_u64p, _dp]
        L.orc_multinomial_strata.argtypes = [u64, u64, u64, C.c_int, _up]
        L.orc_set_hmm.restype = C.c_int; L.orc_set_hmm.argtypes = [C.c_int, _dp, _dp]
        L.orc_smc.restype = C.c_int
        L.orc_smc.argtypes = [C.c_int, _dp, sz, u64, u64, C.c_int, dbl, C.c_void_p, C.c_void_p, _ip, _dp,
                              C.POINTER(dbl), _dp, _ip]
        L.orc_smc_ref.restype = C.c_int
        L.orc_smc_ref.argtypes = [C.c_int, _dp, sz, u64, u64, C.c_int, dbl, C.c_int, C.c_void_p, C.c_void_p, _ip, _dp,
                                  C.POINTER(dbl), _dp, _ip]
        L.orc_smc_filter.restype = C.c_int
        L.orc_smc_filter.argtypes = [C.c_int, _dp, sz, u64, u64, C.c_int, dbl, C.c_void_p, C.c_void_p, _ip, _dp,
                                     C.POINTER(dbl), _dp, _ip, _dp]
        L.orc_trace_lineage.argtypes = [_ip, sz, u64, _ip]
        L.orc_smoothing_real.argtypes = [_dp, _ip, _dp, sz, u64, _dp]
        L.orc_smoothing_int.argtypes = [_ip, _ip, _dp, sz, u64, C.c_int, _dp]
        _LIB = L
    return _LIB


def philox(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def draw_block(seed, group, draw):
    out = np.zeros(4, np.uint32)
    lib().orc_draw_block(seed, group, draw, out)
    return out


def box_muller(words):
    out = np.zeros(2)
    lib().orc_box_muller(np.asarray(words, np.uint32), out)
    return out


MODEL_HMM_TABLE = 5


def set_hmm(means, trans):
    """The table of MODEL_HMM_TABLE: k = len(means) states (2..8), emission N(means[s], 1), transition rows trans[s] (weights)."""
    means = np.ascontiguousarray(means, np.float64)
    trans = np.ascontiguousarray(trans, np.float64)
    assert trans.shape == (len(means), len(means))
    if lib().orc_set_hmm(len(means), means, trans.reshape(-1)):
        raise RuntimeError("orc_set_hmm failed")
    global _HMM_K
    _HMM_K = len(means)


_HMM_K = 3


def is_int_model(model):
    return model in (MODEL_HMM3, MODEL_HMM_TABLE)


def sis(model, obs, n, seed, pid0=0):
    """cpprob::inference(StateType::sis,...) in memory. Returns (values[n_pred][n], logw[n])."""
    obs = np.ascontiguousarray(obs, np.float64)
    T = lib().orc_model_num_predicts(model, len(obs))
    logw = np.zeros(n, np.float64)
    if is_int_model(model):
        vals = np.zeros((T, n), np.int32)
        rc = lib().orc_sis(model, obs, len(obs), n, seed, pid0, None, vals.ctypes.data, logw)
    else:
        vals = np.zeros((T, n), np.float64)
        rc = lib().orc_sis(model, obs, len(obs), n, seed, pid0, vals.ctypes.data, None, logw)
    if rc:
        raise RuntimeError("orc_sis failed rc=%d" % rc)
    return vals, logw


def sis_faithful(model, obs, n, seed, prefix, address, progress=False):
    obs = np.ascontiguousarray(obs, np.float64)
    rc = lib().orc_sis_faithful(model, obs, len(obs), n, seed, prefix.encode(), address.encode(), int(progress))
    if rc:
        raise RuntimeError("orc_sis_faithful failed rc=%d" % rc)


def logsumexp(logw):
    logw = np.ascontiguousarray(logw, np.float64)
    return lib().orc_logsumexp(logw, len(logw))


def weighted_moments(x, logw):
    """(mean, variance, log_norm, ess) as EmpiricalDistribution computes them."""
    out = np.zeros(4)
    lib().orc_weighted_moments(np.ascontiguousarray(x, np.float64), np.ascontiguousarray(logw, np.float64), len(logw), out)
    return out


def weighted_hist(x, logw, k):
    out = np.zeros(k)
    lib().orc_weighted_hist(np.ascontiguousarray(x, np.int32), np.ascontiguousarray(logw, np.float64), len(logw), k, out)
    return out


def resample(kind, logw, seed, step, j0=0, n_out=None, n_total_out=None):
    logw = np.ascontiguousarray(logw, np.float64)
    n_in = len(logw)
    n_out = n_in if n_out is None else n_out
    n_total_out = n_in if n_total_out is None else n_total_out
    anc = np.zeros(n_out, np.int32)
    rc = lib().orc_resample(kind, logw, n_in, seed, step, j0, n_out, n_total_out, anc, None)
    if rc:
        raise RuntimeError("orc_resample failed")
    return anc


def resample_table_systematic(x, e, seed, step, before=None, total=None, last_shard=True, j0=0, n_out=None, n_total_out=None, stratified=False):
    """Order-independent systematic resampling of a table-weight generation (integer prefix counts): ancestors of the
    outputs [j0, j0 + n_out) among the sources x (states 0..2, weights e[state]); -1 where the ancestor is on another shard."""
    x = np.ascontiguousarray(x, np.int32)
    e = np.ascontiguousarray(e, np.float64)
    cnt = np.bincount(x, minlength=3).astype(np.uint64)
    before = np.zeros(3, np.uint64) if before is None else np.ascontiguousarray(before, np.uint64)
    total = cnt if total is None else np.ascontiguousarray(total, np.uint64)
    n_out = len(x) if n_out is None else n_out
    n_total_out = len(x) if n_total_out is None else n_total_out
    anc = np.zeros(n_out, np.int32)
    fn = lib().orc_resample_table_stratified if stratified else lib().orc_resample_table_systematic
    rc = fn(x, len(x), e, before, total, int(bool(last_shard)), seed, step, j0, n_out, n_total_out, anc)
    if rc:
        raise RuntimeError("orc_resample_table_%s failed rc=%d" % ("stratified" if stratified else "systematic", rc))
    return anc


def resample_table_multinomial(x, e, seed, step, n_out=None):
    """Multinomial resampling (strata form) of a table-weight generation on integer prefix counts (orc_resample_table_multinomial)."""
    x = np.ascontiguousarray(x, np.int32)
    e = np.ascontiguousarray(e, np.float64)
    n_out = len(x) if n_out is None else n_out
    anc = np.zeros(n_out, np.int32)
    rc = lib().orc_resample_table_multinomial(x, len(x), e, seed, step, n_out, anc)
    if rc:
        raise RuntimeError("orc_resample_table_multinomial failed rc=%d" % rc)
    return anc


def resample_table_stratified(x, e, seed, step, **kw):
    """Stratified resampling of a table-weight generation on integer prefix counts (orc_resample_table_stratified)."""
    return resample_table_systematic(x, e, seed, step, stratified=True, **kw)


def fix_weights(logw, ref):
    """The fixed-point form's integer weights q_i = min(rint(exp(lw_i - ref) 2^32), 2^32 - 1) (orc_fix_weight)."""
    logw = np.ascontiguousarray(logw, np.float64)
    q = np.zeros(len(logw), np.uint32)
    lib().orc_fix_weights(logw, len(logw), float(ref), q)
    return q


def resample_fixed_systematic(q, seed, step, before=0, total=None, last_shard=True, j0=0, n_out=None, n_total_out=None):
    """Order-independent systematic resampling on integer weights: ancestors of the outputs [j0, j0 + n_out) among the sources q;
    -1 where the ancestor is on another shard."""
    q = np.ascontiguousarray(q, np.uint32)
    total = int(q.astype(np.uint64).sum()) if total is None else int(total)
    n_out = len(q) if n_out is None else n_out
    n_total_out = len(q) if n_total_out is None else n_total_out
    anc = np.zeros(n_out, np.int32)
    rc = lib().orc_resample_fixed_systematic(q, len(q), int(before), total, int(bool(last_shard)), seed, step, j0, n_out, n_total_out, anc)
    if rc:
        raise RuntimeError("orc_resample_fixed_systematic failed rc=%d" % rc)
    return anc


def resample_fixed_stratified(q, seed, step, before=0, total=None, last_shard=True, j0=0, n_out=None, n_total_out=None):
    """Stratified resampling on integer weights (orc_resample_fixed_stratified): ancestors of the outputs [j0, j0 + n_out) among the
    sources q; -1 where the ancestor is on another shard."""
    q = np.ascontiguousarray(q, np.uint32)
    total = int(q.astype(np.uint64).sum()) if total is None else int(total)
    n_out = len(q) if n_out is None else n_out
    n_total_out = len(q) if n_total_out is None else n_total_out
    anc = np.zeros(n_out, np.int32)
    rc = lib().orc_resample_fixed_stratified(q, len(q), int(before), total, int(bool(last_shard)), seed, step, j0, n_out, n_total_out, anc)
    if rc:
        raise RuntimeError("orc_resample_fixed_stratified failed rc=%d" % rc)
    return anc


def resample_fixed_multinomial(q, seed, step, before=0, total=None, j0=0, n_out=None):
    """Multinomial resampling on integer weights (orc_resample_fixed_multinomial): tau_j = floor(u_j C_N), ancestor = min{k : C_k > tau_j};
    -1 where the ancestor is on another shard."""
    q = np.ascontiguousarray(q, np.uint32)
    total = int(q.astype(np.uint64).sum()) if total is None else int(total)
    n_out = len(q) if n_out is None else n_out
    anc = np.zeros(n_out, np.int32)
    rc = lib().orc_resample_fixed_multinomial(q, len(q), int(before), total, seed, step, j0, n_out, anc)
    if rc:
        raise RuntimeError("orc_resample_fixed_multinomial failed rc=%d" % rc)
    return anc


def resample_fixed_multinomial_strata(q, seed, step, n_out=None):
    """Multinomial resampling on integer weights, strata form (orc_resample_fixed_multinomial_strata): N iid thresholds as counts per
    equal stratum (popcounts of Philox bits down a binary tree) and iid uniforms inside each stratum; ancestor = min{k : C_k > tau}."""
    q = np.ascontiguousarray(q, np.uint32)
    n_out = len(q) if n_out is None else n_out
    anc = np.zeros(n_out, np.int32)
    rc = lib().orc_resample_fixed_multinomial_strata(q, len(q), seed, step, n_out, anc)
    if rc:
        raise RuntimeError("orc_resample_fixed_multinomial_strata failed rc=%d" % rc)
    return anc


def resample_fixed_multinomial_strata_shard(q, seed, step, before, total, n_pop):
    """The strata form over shards of ONE population (orc_resample_fixed_multinomial_strata_shard): ancestors of all n_pop outputs
    among the sources q, which hold the mass range [before, before + sum q) of `total`; -1 where the ancestor is another shard's."""
    q = np.ascontiguousarray(q, np.uint32)
    anc = np.zeros(int(n_pop), np.int32)
    rc = lib().orc_resample_fixed_multinomial_strata_shard(q, len(q), int(before), int(total), seed, step, int(n_pop), anc)
    if rc:
        raise RuntimeError("orc_resample_fixed_multinomial_strata_shard failed rc=%d" % rc)
    return anc


def resample_table_multinomial_shard(x, e, seed, step, before, total, last_shard, n_pop):
    """... and the table form over shards (orc_resample_table_multinomial_shard): before / total = state counts [3]."""
    x = np.ascontiguousarray(x, np.int32)
    e = np.ascontiguousarray(e, np.float64)
    anc = np.zeros(int(n_pop), np.int32)
    rc = lib().orc_resample_table_multinomial_shard(x, len(x), e, np.ascontiguousarray(before, np.uint64), np.ascontiguousarray(total, np.uint64),
                                                    int(bool(last_shard)), seed, step, int(n_pop), anc)
    if rc:
        raise RuntimeError("orc_resample_table_multinomial_shard failed rc=%d" % rc)
    return anc


def strata_thresholds_fixed(total, seed, step, n_pop):
    """(tau[n_pop] uint64, stratum[n_pop]) of the strata form on integer masses."""
    tau = np.zeros(int(n_pop), np.uint64); st = np.zeros(int(n_pop), np.int32)
    if lib().orc_strata_thresholds_fixed(int(total), seed, step, int(n_pop), tau, st):
        raise RuntimeError("orc_strata_thresholds_fixed failed")
    return tau, st


def strata_thresholds_table(W, seed, step, n_pop):
    """(tau[n_pop] float64, stratum[n_pop]) of the strata form on the table CDF."""
    tau = np.zeros(int(n_pop), np.float64); st = np.zeros(int(n_pop), np.int32)
    if lib().orc_strata_thresholds_table(float(W), seed, step, int(n_pop), tau, st):
        raise RuntimeError("orc_strata_thresholds_table failed")
    return tau, st


def table_cdf(counts, e):
    return float(lib().orc_table_cdf(np.ascontiguousarray(counts, np.uint64), np.ascontiguousarray(e, np.float64)))


def multinomial_strata(seed, step, n_out, n_particles=None):
    """offs[0 .. K]: the first output of every stratum (K = 2^k strata, k = strata_levels(n_particles))."""
    k = lib().orc_strata_levels(n_out if n_particles is None else n_particles)
    offs = np.zeros((1 << k) + 1, np.uint32)
    lib().orc_multinomial_strata(seed, step, n_out, k, offs)
    return offs


def multinomial_threshold(seed, j, step, total):
    return int(lib().orc_multinomial_threshold(seed, j, step, int(total)))


def smoothing_linear(hist, anc, w, k=3):
    """StatsPrinter's numbers over the lineages for given LINEAR final weights (the fixed-point form's q)."""
    path = lineage(np.ascontiguousarray(anc))
    col = np.take_along_axis(hist, path, axis=1)
    w = np.asarray(w, np.float64)
    W = w.sum()
    if hist.dtype == np.int32:
        return np.stack([[(w * (col[t] == s2)).sum() / W for s2 in range(k)] for t in range(hist.shape[0])])
    m1 = (col * w).sum(axis=1) / W
    m2 = (col * col * w).sum(axis=1) / W
    return np.stack([m1, m2 - m1 * m1], axis=1)


def smc(model, obs, n, seed, resampler=RESAMPLE_SYSTEMATIC, ess_frac=2.0):
    """Returns dict(hist, anc, logw, log_z, ess, resampled, filter): filter[t] = predict hit t under generation t's own weights
    (P(x_t = s) or {mean, variance}) -- what a filtering-only run reports."""
    obs = np.ascontiguousarray(obs, np.float64)
    T = len(obs)
    anc = np.zeros((T, n), np.int32)
    logw = np.zeros(n)
    ess = np.zeros(T)
    res = np.zeros(T, np.int32)
    lz = C.c_double(0.0)
    if is_int_model(model):
        hist = np.zeros((T, n), np.int32)
        filt = np.zeros((T, 3 if model == MODEL_HMM3 else _HMM_K))
        rc = lib().orc_smc_filter(model, obs, T, n, seed, resampler, ess_frac, None, hist.ctypes.data, anc, logw,
                                  C.byref(lz), ess, res, filt)
    else:
        hist = np.zeros((T, n), np.float64)
        filt = np.zeros((T, 2))
        rc = lib().orc_smc_filter(model, obs, T, n, seed, resampler, ess_frac, hist.ctypes.data, None, anc, logw,
                                  C.byref(lz), ess, res, filt)
    if rc:
        raise RuntimeError("orc_smc failed rc=%d" % rc)
    return dict(hist=hist, anc=anc, logw=logw, log_z=lz.value, ess=ess, resampled=res, filter=filt)


REF_MODEL_BOUND, REF_STATEMENT_BOUND, REF_EXACT_MAX, REF_FLOATING_POINT = 0, 1, 2, 3


def smc_ref(model, obs, n, seed, ref_mode, resampler=RESAMPLE_SYSTEMATIC, ess_frac=2.0):
    """SMC with the fixed-point reference chosen as the unchanged-model path chooses it (cpprob/gpu.hpp: StepForm):
    REF_STATEMENT_BOUND = the observe statement's density at its mode, REF_EXACT_MAX = the generation's exact maximum."""
    obs = np.ascontiguousarray(obs, np.float64)
    T = len(obs)
    anc = np.zeros((T, n), np.int32)
    logw = np.zeros(n)
    ess = np.zeros(T)
    res = np.zeros(T, np.int32)
    lz = C.c_double(0.0)
    is_int = is_int_model(model)
    hist = np.zeros((T, n), np.int32 if is_int else np.float64)
    rc = lib().orc_smc_ref(model, obs, T, n, seed, resampler, ess_frac, ref_mode, None if is_int else hist.ctypes.data, hist.ctypes.data if is_int else None,
                           anc, logw, C.byref(lz), ess, res)
    if rc:
        raise RuntimeError("orc_smc_ref failed rc=%d" % rc)
    return dict(hist=hist, anc=anc, logw=logw, log_z=lz.value, ess=ess, resampled=res)


def smoothing(hist, anc, logw, k=3):
    T, n = hist.shape
    if hist.dtype == np.int32:
        out = np.zeros((T, k))
        lib().orc_smoothing_int(hist, anc, np.ascontiguousarray(logw), T, n, k, out.reshape(-1))
    else:
        out = np.zeros((T, 2))
        lib().orc_smoothing_real(hist, anc, np.ascontiguousarray(logw), T, n, out.reshape(-1))
    return out


def lineage(anc):
    T, n = anc.shape
    path = np.zeros((T, n), np.int32)
    lib().orc_trace_lineage(np.ascontiguousarray(anc), T, n, path)
    return path
