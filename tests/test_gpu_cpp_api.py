"""GPU tests of the C++14 drop-in layer: cpprob::inference called exactly like the reference calls it
(cpprob_amd/examples/cpprob_main.cpp, compiled by g++ -std=c++14, mirrors src/main.cpp), with
  * the hand-fused kernels (CPPROB_REGISTER_BUILTIN) and
  * the UNCHANGED model body compiled for the device (CPPROB_REGISTER_MODEL, trace replay for SMC),
checked against each other, the oracle, the exact posteriors and the reference's file grammar."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "cpprob_amd", "bin", "cpprob_main")
GOLD = os.path.join(ROOT, "tests", "golden")


def run_main(tmp_path, *args, expect_rc=0):
    cmd = [MAIN, "--model_folder", str(tmp_path)] + [str(a) for a in args]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if expect_rc is not None:
        assert p.returncode == expect_rc, p.stdout[-2000:] + p.stderr[-2000:]
    res = None
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            res = json.loads(line)
    return res, p.stdout, p.stderr


def obs_str(v):
    return "[" + " ".join(repr(float(x)) for x in v) + "]"


def read_dump(path, is_int):
    vals, lw = [], []
    for line in open(path):
        body, w = line.rsplit("]", 1)
        lw.append(float(w.strip().rstrip(")")))
        items = body[2:].replace("(", "").replace(")", "").split()
        vals.append([(int if is_int else float)(x) for x in items[1::2]])
    return np.array(vals).T, np.array(lw)


@pytest.mark.parametrize("generic", [False, True])
def test_gaussian_sis_like_the_readme(tmp_path, generic):
    """README.md:109-118 flow: inference(sis) then StatsPrinter."""
    n = 200000
    args = ["--model", "gaussian_unknown_mean", "--sis", "--observes", "3 4", "--n_samples", n, "--seed", 7, "--json", "--estimate"]
    res, out, _ = run_main(tmp_path, *(args + (["--generic"] if generic else [])))
    assert res["builtin"] == (not generic) and res["n"] == n
    p = res["predicts"][0]
    assert p["address"] == "Mu"
    assert abs(p["mean"] - 3.0833333) < 0.01 and abs(p["variance"] - 0.8333333) < 0.015
    # same particles as the oracle (same seed): per-particle parity through the dump file
    vals, lw = read_dump(str(tmp_path / "post_sis.real"), False)
    ov, olw = O.sis(O.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0], n, 7)
    np.testing.assert_allclose(vals, ov, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lw, olw, rtol=1e-12, atol=1e-12)
    assert open(str(tmp_path / "post_sis.ids")).read() == "Mu\n"
    assert not os.path.exists(str(tmp_path / "post_sis.int")) and not os.path.exists(str(tmp_path / "post_sis.any"))
    # StatsPrinter re-reads the files: same estimators as the in-memory result
    assert "Estimators for" in out and "Mu:" in out
    mean_line = [l for l in out.splitlines() if l.strip().startswith("Mean:")][0]
    assert abs(float(mean_line.split(":")[1]) - p["mean"]) < 1e-4
    # the reference's own parser accepts the file
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_serialization")
    if os.path.exists(ref):
        head = "".join(open(str(tmp_path / "post_sis.real")).readlines()[:200])
        parsed = subprocess.check_output([ref, "parse-real"], input=head.encode()).decode().splitlines()
        assert len(parsed) == 200 and "BAD" not in parsed


def test_generic_and_builtin_agree_hmm_smc(tmp_path):
    z = np.load(os.path.join(GOLD, "observations.npz"))
    n = 50000
    base = ["--model", "hmm16", "--smc", "--observes", obs_str(z["hmm16"]), "--n_samples", n, "--seed", 3, "--ess_threshold", 2.0, "--json"]
    rb, _, _ = run_main(tmp_path / "b" if False else tmp_path, *base, "--generated_file", "b")
    rg, _, _ = run_main(tmp_path, *base, "--generated_file", "g", "--generic")
    assert rb["builtin"] and not rg["builtin"]
    assert rb["n_resampled"] == rg["n_resampled"] == 15
    assert abs(rb["log_evidence"] - rg["log_evidence"]) < 1e-6
    pb = np.array([p["p"] + [0.0] * (3 - len(p["p"])) for p in rb["predicts"]])
    pg = np.array([p["p"] + [0.0] * (3 - len(p["p"])) for p in rg["predicts"]])
    assert pb.shape == (16, 3)
    np.testing.assert_allclose(pb, pg, atol=2e-3)
    assert np.abs(pb - z["hmm16_smooth"]).max() < 0.04
    # the two paths resample on DIFFERENT integer arithmetic (the built-in kernel: prefix counts of three table weights; the unchanged
    # model: fixed-point masses against the observe statement's bound), so their files agree up to rare CDF-boundary flips ...
    vb, lwb = read_dump(str(tmp_path / "b_smc.int"), True)
    vg, lwg = read_dump(str(tmp_path / "g_smc.int"), True)
    assert vb.shape == vg.shape == (16, n)
    assert np.mean(vb != vg) < 2e-3
    np.testing.assert_allclose(np.sort(lwb), np.sort(lwg), atol=1e-9)
    # ... and EACH equals the oracle's statement of its own arithmetic, trace for trace
    assert rg["step_form"] == 1 and rg["launches_per_step"] == 1 and rg["replay_window"] == 1 and rg["markov_crosscheck"] == 1
    r = O.smc_ref(O.MODEL_HMM3, z["hmm16"], n, 3, O.REF_STATEMENT_BOUND, O.RESAMPLE_SYSTEMATIC, 2.0)
    assert np.array_equal(vg, np.take_along_axis(r["hist"], O.lineage(r["anc"]), axis=1))
    np.testing.assert_allclose(lwg, r["logw"], atol=1e-12)
    assert abs(rg["log_evidence"] - r["log_z"]) < 1e-12
    rt = O.smc(O.MODEL_HMM3, z["hmm16"], n, 3, O.RESAMPLE_SYSTEMATIC, 2.0)
    assert np.array_equal(vb, np.take_along_axis(rt["hist"], O.lineage(rt["anc"]), axis=1))


@pytest.mark.parametrize("model,key,T,ess,oid,is_int", [("hmm16", "hmm16", 16, 2.0, O.MODEL_HMM3, True), ("hmm16", "hmm16", 16, 0.5, O.MODEL_HMM3, True),
                                                        ("linear_gaussian_1d25", "lgssm100", 25, 0.5, O.MODEL_LINEAR_GAUSSIAN_1D, False)])
@pytest.mark.parametrize("form,ref_mode,launches", [(1, O.REF_STATEMENT_BOUND, 1), (3, O.REF_STATEMENT_BOUND, 1), (2, O.REF_EXACT_MAX, 3), (0, O.REF_EXACT_MAX, 4)])
def test_unchanged_model_smc_equals_the_oracle_in_every_step_form(tmp_path, model, key, T, ess, oid, is_int, form, ref_mode, launches):
    """cpprob::inference(StateType::smc, <unchanged model>) (reference cpprob.hpp:173-203 with models.hpp:67-80,114-141 as the model):
    the resampling inside the model's own launch against the dry run's bounds (1; 3: four particles a lane behind one search), against
    exact maxima (2), and as separate bookkeeping launches (0) -- each the oracle's fixed-point SMC with the same reference rule: surviving traces array_equal."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    obs = z[key][:T]
    n = 40000
    res, _, _ = run_main(tmp_path, "--model", model, "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 3, "--ess_threshold", ess, "--json",
                         "--generic", "--generated_file", "g", "--step_form", form)
    assert res["step_form"] == form and res["launches_per_step"] == launches and res["replay_window"] == 1
    vg, lwg = read_dump(str(tmp_path / ("g_smc.int" if is_int else "g_smc.real")), is_int)
    r = O.smc_ref(oid, obs, n, 3, ref_mode, O.RESAMPLE_SYSTEMATIC, ess)
    paths = np.take_along_axis(r["hist"], O.lineage(r["anc"]), axis=1)
    assert res["n_resampled"] == int(r["resampled"].sum())
    if is_int:
        assert np.array_equal(vg, paths)
    else:
        np.testing.assert_allclose(vg, paths, rtol=0, atol=1e-10)     # lean vs libm normal draws (1e-11), the same ancestors
        assert np.array_equal(np.argsort(vg[-1], kind="stable"), np.argsort(paths[-1], kind="stable"))
    np.testing.assert_allclose(lwg, r["logw"], atol=1e-10)
    assert abs(res["log_evidence"] - r["log_z"]) < 1e-9


@pytest.mark.parametrize("model,key,T,ess,is_int,used", [("hmm16", "hmm16", 16, 2.0, True, 16), ("linear_gaussian_1d25", "lgssm100", 25, 0.5, False, 25),
                                                         ("hmm128", "hmm128", 128, 0.5, True, 112), ("linear_gaussian_1d100", "lgssm100", 100, 0.5, False, 87)])
def test_step_kernels_built_per_step_change_no_number(tmp_path, model, key, T, ess, is_int, used):
    """The step kernels built per step (cpprob/gpu.hpp: model_step_kernel_at -- one per step up to 32 observes, one every T / 8 steps
    beyond; the step's thresholds as compile-time facts, dead iterations folded away) against the run-time kernel that re-runs the model's
    loop from its first statement at every launch: the same dump, byte for byte, and the same evidence."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    obs = z[key][:T]
    n = 30000
    out = {}
    for tag, extra in (("built", []), ("runtime", ["--no_step_builds"])):
        res, _, _ = run_main(tmp_path, "--model", model, "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 5, "--ess_threshold", ess, "--json",
                             "--generic", "--generated_file", tag, "--step_form", 1, *extra)
        assert res["step_form"] == 1
        out[tag] = (res, open(str(tmp_path / (tag + ("_smc.int" if is_int else "_smc.real")))).read())
    assert out["built"][0]["step_builds_used"] == used and out["runtime"][0]["step_builds_used"] == 0
    assert out["built"][1] == out["runtime"][1]
    assert out["built"][0]["log_evidence"] == out["runtime"][0]["log_evidence"] and out["built"][0]["n_resampled"] == out["runtime"][0]["n_resampled"]
    if T <= 32:
        # ... and four particles a lane on the builds (model_step_kernel_quad_at: a call of the body is the live iteration alone)
        res, _, _ = run_main(tmp_path, "--model", model, "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 5, "--ess_threshold", ess, "--json",
                             "--generic", "--generated_file", "quad", "--step_form", 3)
        assert res["step_form"] == 3 and res["step_builds_used"] == used
        assert open(str(tmp_path / ("quad" + ("_smc.int" if is_int else "_smc.real")))).read() == out["runtime"][1] and res["log_evidence"] == out["runtime"][0]["log_evidence"]


@pytest.mark.parametrize("n", [1, 255, 1024, 1025, 3000])
def test_four_particles_a_lane_equals_one_particle_a_lane_on_ragged_populations(tmp_path, n):
    """Step form 3 (model_step_kernel_quad: a workgroup owns a 1024-particle tile, one search, the model body four times a lane) against
    form 1 on populations that end inside a tile, fill one exactly, and spill one particle into the next: the same posterior file."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    for model, key, T, ess, is_int in (("hmm16", "hmm16", 16, 2.0, True), ("linear_gaussian_1d25", "lgssm100", 25, 0.5, False)):
        obs = z[key][:T]
        out = {}
        for form in (1, 3):
            res, _, _ = run_main(tmp_path, "--model", model, "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 11, "--ess_threshold", ess, "--json",
                                 "--generic", "--generated_file", "f%d" % form, "--step_form", form)
            assert res["step_form"] in (form, 2)                      # (a single particle's generation may not fit the statement's bound: exact maxima then)
            v, lw = read_dump(str(tmp_path / ("f%d_smc.%s" % (form, "int" if is_int else "real"))), is_int)
            out[form] = (v, lw, res["log_evidence"], res["n_resampled"])
        assert np.array_equal(out[1][0], out[3][0]) and np.array_equal(out[1][1], out[3][1]) and out[1][2] == out[3][2] and out[1][3] == out[3][3]


@pytest.mark.parametrize("model,key,T,ess,oid,is_int", [("hmm16", "hmm16", 16, 2.0, O.MODEL_HMM3, True), ("linear_gaussian_1d25", "lgssm100", 25, 0.5, O.MODEL_LINEAR_GAUSSIAN_1D, False)])
@pytest.mark.parametrize("rname,rid", [("stratified", O.RESAMPLE_STRATIFIED), ("multinomial", O.RESAMPLE_MULTINOMIAL)])
def test_unchanged_model_smc_with_the_other_resamplers_equals_the_oracle(tmp_path, model, key, T, ess, oid, is_int, rname, rid):
    """Stratified and multinomial resampling of an unchanged model: the bookkeeping launches between two launches of the model body
    run on the same integer masses as the built-in models' step kernels (cpprob_hip_smc_bookkeep_fixed_rs: references from exact
    maxima, stratified comb / strata-form multinomial) -- surviving traces array_equal to the oracle's fixed-point SMC."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    obs = z[key][:T]
    n = 40000
    res, _, _ = run_main(tmp_path, "--model", model, "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 3, "--ess_threshold", ess, "--json",
                         "--generic", "--generated_file", "g", "--resampler", rname)
    assert res["step_form"] == 0 and res["launches_per_step"] == 4
    vg, lwg = read_dump(str(tmp_path / ("g_smc.int" if is_int else "g_smc.real")), is_int)
    r = O.smc_ref(oid, obs, n, 3, O.REF_EXACT_MAX, rid, ess)
    paths = np.take_along_axis(r["hist"], O.lineage(r["anc"]), axis=1)
    assert res["n_resampled"] == int(r["resampled"].sum())
    if is_int:
        assert np.array_equal(vg, paths)
    else:
        np.testing.assert_allclose(vg, paths, rtol=0, atol=1e-10)
        assert np.array_equal(np.argsort(vg[-1], kind="stable"), np.argsort(paths[-1], kind="stable"))
    np.testing.assert_allclose(lwg, r["logw"], atol=1e-10)
    assert abs(res["log_evidence"] - r["log_z"]) < 1e-9


def test_unchanged_model_step_falls_back_to_exact_maxima_when_a_generation_leaves_its_bound(tmp_path):
    """An observation 30 sigma from every state: the observe statement's bound lies > 6 nats above every particle, the integer
    weights would lose their bits -- the device flags the generation and the run is repeated against exact maxima (two launches per
    observe); the result is the oracle's exact-maximum form."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    obs = np.array(z["hmm16"], dtype=float)
    obs[5] = 31.0
    n = 30000
    res, _, _ = run_main(tmp_path, "--model", "hmm16", "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 11, "--ess_threshold", 2.0, "--json",
                         "--generic", "--generated_file", "g")
    assert res["step_form"] == 2 and res["launches_per_step"] == 3
    vg, lwg = read_dump(str(tmp_path / "g_smc.int"), True)
    r = O.smc_ref(O.MODEL_HMM3, obs, n, 11, O.REF_EXACT_MAX, O.RESAMPLE_SYSTEMATIC, 2.0)
    assert np.array_equal(vg, np.take_along_axis(r["hist"], O.lineage(r["anc"]), axis=1))
    assert abs(res["log_evidence"] - r["log_z"]) < 1e-9
    # the same as three shards of one joint population: every rank's launch takes the decision from all ranks' totals, every rank
    # sees the generation leave its bound, and the run is repeated on exact maxima -- the single-context answer, bit for bit
    many, _, _ = run_main(tmp_path, "--model", "hmm16", "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 11, "--ess_threshold", 2.0, "--json",
                          "--generic", "--generated_file", "m", "--devices", "0,0,0")
    # (traces, weights and ancestors are the same integers; the joint form's evidence takes its logarithms on the host -- glibc against
    #  the device's log: a few units in the last place, not a guaranteed tie)
    assert many["joint"] is True and many["step_form"] == 2 and abs(many["log_evidence"] - res["log_evidence"]) <= 16 * np.spacing(abs(res["log_evidence"]))
    vm, lwm = read_dump(str(tmp_path / "m_smc.int"), True)
    assert np.array_equal(vm, vg) and np.array_equal(lwm, lwg)


def test_unchanged_model_without_a_likelihood_bound_takes_exact_maxima(tmp_path):
    """random_scale samples the emission's standard deviation: the observe statement's density at its mode differs from trace to trace,
    the host probe sees that, and the step takes its references from exact maxima -- the same integers as the separate bookkeeping
    launches, so the two forms' files are identical."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    obs = z["lgssm100"][:12]
    n = 40000
    base = ["--model", "random_scale12", "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 6, "--ess_threshold", 0.5, "--json", "--generic"]
    a, _, _ = run_main(tmp_path, *base, "--generated_file", "a")
    b, _, _ = run_main(tmp_path, *base, "--generated_file", "b", "--step_form", 0)
    assert a["step_form"] == 2 and b["step_form"] == 0 and a["replay_window"] == 1 and a["markov_crosscheck"] == 1
    va, lwa = read_dump(str(tmp_path / "a_smc.real"), False)
    vb, lwb = read_dump(str(tmp_path / "b_smc.real"), False)
    assert np.array_equal(va, vb) and np.array_equal(lwa, lwb) and a["log_evidence"] == b["log_evidence"] and a["n_resampled"] == b["n_resampled"]
    assert 0 < a["n_resampled"] < 11


def test_unchanged_model_keeps_its_context_and_workspace_across_inference_calls(tmp_path):
    """--repeat calls cpprob::inference again and again, as src/main.cpp would in a loop: the first call creates the context and
    sizes the device workspace, the later ones find both (set-up in the tens of microseconds)."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    res, out, _ = run_main(tmp_path, "--model", "hmm16", "--smc", "--observes", obs_str(z["hmm16"]), "--n_samples", 200000, "--seed", 3, "--ess_threshold", 2.0,
                           "--json", "--generic", "--no_dump", "--repeat", 4)
    setups = [float(l.split("set-up")[1].split("ms")[0]) for l in out.splitlines() if l.startswith("run ")]
    assert len(setups) == 4 and setups[0] > 5.0 and max(setups[1:]) < 2.0
    assert res["workspace_grown"] is False and res["step_form"] == 1


def test_builtin_registration_keeps_its_context_across_inference_calls(tmp_path):
    """The same for the built-in registration (cpprob::gpu::ContextLease): a context made per call cost ~2 ms of stream and buffer
    creation around a 0.2 ms run; a later call finds the first one's -- and, same seed, returns the same numbers; a call with another
    model and size in between reconfigures it."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    res, out, _ = run_main(tmp_path, "--model", "hmm16", "--smc", "--observes", obs_str(z["hmm16"]), "--n_samples", 200000, "--seed", 3, "--ess_threshold", 2.0,
                           "--json", "--no_dump", "--repeat", 4)
    setups = [float(l.split("set-up")[1].split("ms")[0]) for l in out.splitlines() if l.startswith("run ")]
    assert res["builtin"] and len(setups) == 4 and setups[0] > 5.0 and max(setups[1:]) < 1.0
    once, _, _ = run_main(tmp_path, "--model", "hmm16", "--smc", "--observes", obs_str(z["hmm16"]), "--n_samples", 200000, "--seed", 6, "--ess_threshold", 2.0,
                          "--json", "--no_dump")
    # (--repeat moves the seed on by one a call: the fourth call on the kept context = a first call on a fresh one with that seed)
    assert once["log_evidence"] == res["log_evidence"] and once["predicts"] == res["predicts"]


def test_filtering_only_run_from_the_cpp_host(tmp_path):
    """cpprob::gpu::options().keep_history = false (CLI --filtering_only): the reference's call, an O(N) particle store, the
    filtering marginals instead of the whole-trace posterior, the same evidence."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    n = 400000
    base = ["--model", "hmm16", "--smc", "--observes", obs_str(z["hmm16"]), "--n_samples", n, "--seed", 3, "--ess_threshold", 2.0, "--json", "--no_dump"]
    keep, _, _ = run_main(tmp_path, *base)
    filt, _, _ = run_main(tmp_path, *base, "--filtering_only")
    assert filt["builtin"] and filt["log_evidence"] == keep["log_evidence"] and filt["n_resampled"] == 15
    pf = np.array([p["p"] + [0.0] * (3 - len(p["p"])) for p in filt["predicts"]])
    pk = np.array([p["p"] + [0.0] * (3 - len(p["p"])) for p in keep["predicts"]])
    assert np.abs(pf - z["hmm16_filter"]).max() < 6e-3 and np.abs(pk - z["hmm16_smooth"]).max() < 2e-2
    np.testing.assert_allclose(pf[-1], pk[-1], atol=1e-12)          # the last hit: filtering and smoothing coincide


def test_generic_lgssm_smc_ess_triggered(tmp_path):
    z = np.load(os.path.join(GOLD, "observations.npz"))
    obs = z["lgssm100"][:25]
    n = 40000
    base = ["--model", "linear_gaussian_1d25", "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 5, "--ess_threshold", 0.5, "--json", "--no_dump"]
    rb, _, _ = run_main(tmp_path, *base)
    rg, _, _ = run_main(tmp_path, *base, "--generic")
    assert 0 < rb["n_resampled"] < 24 and rb["n_resampled"] == rg["n_resampled"]
    mb = np.array([[p["mean"], p["variance"]] for p in rb["predicts"]])
    mg = np.array([[p["mean"], p["variance"]] for p in rg["predicts"]])
    np.testing.assert_allclose(mb, mg, atol=5e-3)
    from oracle import exact as E
    ms, ps, _, _, ll = E.kalman_rts(obs)
    assert abs(mg[-1, 0] - ms[-1]) < 0.03 and abs(mg[-1, 1] - ps[-1]) < 0.03
    assert abs(rg["log_evidence"] - ll) < 0.1 and abs(rb["log_evidence"] - rg["log_evidence"]) < 1e-6


def test_generic_sis_on_state_space_models(tmp_path):
    z = np.load(os.path.join(GOLD, "observations.npz"))
    n = 30000
    res, _, _ = run_main(tmp_path, "--model", "hmm16", "--sis", "--observes", obs_str(z["hmm16"]), "--n_samples", n, "--seed", 9, "--generic", "--json")
    vals, lw = read_dump(str(tmp_path / "post_sis.int"), True)
    ov, olw = O.sis(O.MODEL_HMM3, z["hmm16"], n, 9)
    assert np.array_equal(vals, ov)                                   # integer states: bit-exact
    np.testing.assert_allclose(lw, olw, rtol=1e-11, atol=1e-11)
    ref = np.array([O.weighted_hist(ov[t], olw, 3) for t in range(16)])
    got = np.array([p["p"] + [0.0] * (3 - len(p["p"])) for p in res["predicts"]])
    np.testing.assert_allclose(got, ref, atol=1e-9)


def test_cli_errors_like_the_reference(tmp_path):
    _, _, err = run_main(tmp_path, "--model", "hmm16", "--sis", "--observes", "[1 2 3]", expect_rc=1)
    assert "Could not parse the observations" in err
    _, _, err = run_main(tmp_path, "--model", "hmm16", "--sis", expect_rc=1)
    assert "exactly one of the options" in err


def test_generic_path_other_distributions(tmp_path):
    """uniform_real prior + poisson observes: only the generic device path can run it."""
    n = 400000
    res, _, _ = run_main(tmp_path, "--model", "poisson_rate", "--sis", "--observes", "3 5", "--n_samples", n, "--seed", 2, "--json", "--no_dump")
    assert not res["builtin"]
    lam = np.linspace(0.5, 10.0, 200001)
    post = lam ** 8 * np.exp(-2 * lam)              # flat prior on [0.5, 10], counts 3 and 5
    post /= np.trapezoid(post, lam)
    mean = np.trapezoid(lam * post, lam)
    var = np.trapezoid(lam ** 2 * post, lam) - mean ** 2
    p = res["predicts"][0]
    assert p["address"] == "Rate"
    assert abs(p["mean"] - mean) < 0.02 and abs(p["variance"] - var) < 0.05
    # evidence: integral of the likelihood against the flat prior
    from scipy.special import gammaln
    lik = np.exp(8 * np.log(lam) - 2 * lam - gammaln(4) - gammaln(6)) / 9.5
    assert abs(res["log_evidence"] - np.log(np.trapezoid(lik, lam))) < 0.02
    # smc on the same model: two observes -> one resampling opportunity, same posterior
    res2, _, _ = run_main(tmp_path, "--model", "poisson_rate", "--smc", "--observes", "3 5", "--n_samples", n, "--seed", 2, "--json", "--no_dump",
                          "--ess_threshold", 2.0)
    assert res2["n_resampled"] == 1
    assert abs(res2["predicts"][0]["mean"] - mean) < 0.03 and abs(res2["log_evidence"] - res["log_evidence"]) < 0.02


def test_rejection_sampling_model_through_generic_sis(tmp_path):
    """models.hpp:82-112 pattern: the prior is simulated by a rejection loop, so particles execute different numbers
    of sample statements.  SIS runs it on the device as it comes; the posterior is the conjugate one."""
    n = 300000
    res, _, _ = run_main(tmp_path, "--model", "gaussian_by_rejection", "--sis", "--observes", "3 4", "--n_samples", n, "--seed", 4, "--json", "--no_dump")
    p = res["predicts"][0]
    assert not res["builtin"] and p["address"] == "Mu"
    assert abs(p["mean"] - 3.0833333) < 0.012 and abs(p["variance"] - 0.8333333) < 0.02
    assert abs(res["ess"] / n - 0.344) < 0.02              # same importance-sampling efficiency as the direct prior draw
    # SMC keeps 4x the dry run's trace rows per particle; rejection loops are geometric (acceptance ~ 1/16 here), so some
    # particle overflows them: the run is repeated with 4x more rows until every trace fits -- never silently truncated
    r2, out, err = run_main(tmp_path, "--model", "gaussian_by_rejection", "--smc", "--observes", "3 4", "--n_samples", 200000, "--seed", 4, "--json",
                            "--no_dump")
    assert abs(r2["predicts"][0]["mean"] - 3.0833333) < 0.02 and abs(r2["predicts"][0]["variance"] - 0.8333333) < 0.03


def test_functor_model_is_found_by_type(tmp_path):
    """models::Gauss<>-style functor (reference models.hpp:51-65): registered with CPPROB_REGISTER_FUNCTOR."""
    n = 100000
    res, _, _ = run_main(tmp_path, "--model", "gauss_functor", "--sis", "--observes", "3 4", "--n_samples", n, "--seed", 7, "--json")
    vals, lw = read_dump(str(tmp_path / "post_sis.real"), False)
    ov, olw = O.sis(O.MODEL_GAUSSIAN_UNKNOWN_MEAN, [3.0, 4.0], n, 7)
    np.testing.assert_allclose(vals, ov, rtol=1e-12, atol=1e-12)       # same statements -> same stream as the function form
    np.testing.assert_allclose(lw, olw, rtol=1e-12, atol=1e-12)
    assert not res["builtin"]


def test_vector_valued_statements_gaussian_2d(tmp_path):
    """SURVEY 8(f) row 4, reference models.hpp:38-49: multivariate-normal sample, vector observe, NDArray predict.
    Per-particle parity with the oracle through the dump (NDArray grammar `(id [v0 v1])`), posterior against the
    conjugate answer, StatsPrinter's elementwise estimators; smc == sis for a model with one observe statement."""
    n = 200000
    y = [3.0, 4.5]
    res, out, _ = run_main(tmp_path, "--model", "gaussian_2d_unk_mean", "--sis", "--observes", obs_str(y), "--n_samples", n, "--seed", 11, "--json", "--estimate")
    assert res["builtin"] and res["n"] == n and len(res["predicts"]) == 1
    p = res["predicts"][0]
    assert p["address"] == "Mu" and len(p["mean_nd"]) == 2
    for d, (m0, s0) in enumerate([(1.0, 5.0), (2.0, 3.0)]):           # prior variances 5, 3; likelihood variance 2
        var = 1.0 / (1.0 / s0 + 1.0 / 2.0)
        mean = var * (m0 / s0 + y[d] / 2.0)
        assert abs(p["mean_nd"][d] - mean) < 0.012 and abs(p["variance_nd"][d] - var) < 0.02
    lines = open(str(tmp_path / "post_sis.real")).readlines()
    assert len(lines) == n and lines[0].startswith("([(0 [") and lines[0].count("[") == 2
    vals = np.array([[float(x) for x in l[l.index("(0 [") + 4: l.index("])")].split()] for l in lines]).T
    lw = np.array([float(l.rsplit("]", 1)[1].strip().rstrip(")")) for l in lines])
    ov, olw = O.sis(O.MODEL_GAUSSIAN_2D_UNKNOWN_MEAN, y, n, 11)
    np.testing.assert_allclose(vals, ov, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lw, olw, rtol=1e-12, atol=1e-12)
    assert open(str(tmp_path / "post_sis.ids")).read() == "Mu\n"
    mean_line = [l for l in out.splitlines() if l.strip().startswith("Mean:")][0]
    got = [float(x) for x in mean_line.split(":")[1].strip().strip("[]").split()]
    assert len(got) == 2 and abs(got[0] - p["mean_nd"][0]) < 1e-4 and abs(got[1] - p["mean_nd"][1]) < 1e-4
    res2, _, _ = run_main(tmp_path, "--model", "gaussian_2d_unk_mean", "--smc", "--observes", obs_str(y), "--n_samples", n, "--seed", 11, "--json", "--no_dump")
    assert res2["predicts"][0]["mean_nd"] == p["mean_nd"] and res2["n_resampled"] == 0


def test_replicates_give_error_bars(tmp_path):
    """Options::replicates (CLI --replicates): R seeds of the same inference, up to three runs in flight on their own contexts;
    the spread over the replicates is the Monte-Carlo error bar (seeds for error bars)."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    n, R = 200000, 6
    res, _, _ = run_main(tmp_path, "--model", "hmm16", "--smc", "--observes", obs_str(z["hmm16"]), "--n_samples", n, "--seed", 40, "--ess_threshold", "2.0",
                         "--json", "--no_dump", "--replicates", R)
    assert res["replicates"] == R and len(res["predict_mean"]) == 16 and len(res["predict_sd"]) == 16
    exact = z["hmm16_smooth"][:, 0]
    pm, sd = np.array(res["predict_mean"]), np.array(res["predict_sd"])
    assert (sd > 0).all() and sd.max() < 5e-3                       # ~1e-3 at 2e5 particles
    assert np.all(np.abs(pm - exact) < 5 * sd / np.sqrt(R) + 1e-3)  # the replicate mean sits within its own error bar of the truth
    assert abs(res["log_evidence_mean"] - float(z["hmm16_logz"])) < 0.01 and 0 < res["log_evidence_sd"] < 0.02
    # replicate 0 is the plain run with the same seed
    one, _, _ = run_main(tmp_path, "--model", "hmm16", "--smc", "--observes", obs_str(z["hmm16"]), "--n_samples", n, "--seed", 40, "--ess_threshold", "2.0",
                         "--json", "--no_dump")
    assert one["log_evidence"] == res["log_evidence"] and one["predicts"][0]["p"] == res["predicts"][0]["p"]


def test_cpp_host_shards_one_population_over_several_ranks(tmp_path):
    """SURVEY 8(e) from the C++14 host: cpprob::gpu::options().devices (cpprob_main --devices) makes cpprob::inference run ONE
    joint population over several ranks -- here three on this GPU (loopback transport; distinct devices take RCCL over xGMI) --
    through cpprob_hip_group_*: the answers, the dump and the estimators are those of the single-GPU call, bit for bit (hmm<16>,
    every-step schedule: integer prefix counts; continuous weights, ESS-triggered: fixed-point weights)."""
    obs = np.load(os.path.join(GOLD, "observations.npz"))["hmm16"]
    n = 150001
    common = ["--model", "hmm16", "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 3, "--ess_threshold", 2.0, "--json"]
    one, _, _ = run_main(tmp_path, *common, "--generated_file", "one")
    many, _, _ = run_main(tmp_path, *common, "--generated_file", "many", "--devices", "0,0,0")
    assert one["n_gpus"] == 1 and many["n_gpus"] == 3 and many["builtin"]
    assert many["log_evidence"] == one["log_evidence"] and many["n_resampled"] == one["n_resampled"] == 15
    for a, b in zip(one["predicts"], many["predicts"]):
        np.testing.assert_allclose(a["p"], b["p"], rtol=0, atol=1e-13)
    assert open(str(tmp_path / "one_smc.int")).read() == open(str(tmp_path / "many_smc.int")).read()       # every trace, every weight
    # continuous weights, ESS-triggered, two ranks
    lg = np.load(os.path.join(GOLD, "observations.npz"))["lgssm100"][:25]
    common = ["--model", "linear_gaussian_1d25", "--smc", "--observes", obs_str(lg), "--n_samples", 100000, "--seed", 5, "--ess_threshold", 0.5, "--json", "--no_dump"]
    one, _, _ = run_main(tmp_path, *common)
    two, _, _ = run_main(tmp_path, *common, "--devices", "0,0")
    assert two["n_gpus"] == 2 and two["n_resampled"] == one["n_resampled"]
    assert two["log_evidence"] == one["log_evidence"]                 # fixed-point weights: the two-rank run IS the one-GPU run
    for a, b in zip(one["predicts"], two["predicts"]):
        assert abs(a["mean"] - b["mean"]) < 1e-12 and abs(a["variance"] - b["variance"]) < 1e-12


@pytest.mark.parametrize("model,obs", [("hmm16", None), ("poisson_rate", "3 5"), ("gaussian_by_rejection", "3.0 4.0")])
def test_unchanged_model_shards_over_several_devices_under_sis(tmp_path, model, obs):
    """SURVEY 8(e) for the unchanged-model path: under StateType::sis the shards of a population need no communication, so
    cpprob::gpu::options().devices (cpprob_main --generic --devices) runs the model body for contiguous blocks of particles on every
    device -- global particle ids select the random streams -- and combines the shards by their evidence.  Three shards on this GPU:
    every trace and every weight of the dump equals the one-device run's (per-particle parity), the estimators agree to rounding.
    StateType::smc over several devices: the joint population (next test); --islands keeps the independent-runs form, a consistent
    estimator of the same posterior, checked against forward-backward -- and its dumped files carry every island's mass in the
    particles' log-weights, so that an estimate re-derived from the files is the one Result reports."""
    if obs is None:
        obs = obs_str(np.load(os.path.join(GOLD, "observations.npz"))["hmm16"])
    n = 50001
    common = ["--model", model, "--sis", "--observes", obs, "--n_samples", n, "--seed", 9, "--json", "--generic"]
    one, _, _ = run_main(tmp_path, *common, "--generated_file", "one")
    many, _, _ = run_main(tmp_path, *common, "--generated_file", "many", "--devices", "0,0,0")
    assert not one["builtin"] and not many["builtin"] and many["n_gpus"] == 3 and one["n_gpus"] == 1
    assert abs(many["log_evidence"] - one["log_evidence"]) < 1e-10 and abs(many["ess"] - one["ess"]) < 1e-6 * one["ess"]
    for a, b in zip(one["predicts"], many["predicts"]):
        if "p" in a:
            np.testing.assert_allclose(a["p"], b["p"], rtol=0, atol=1e-12)
        else:
            assert abs(a["mean"] - b["mean"]) < 1e-11 and abs(a["variance"] - b["variance"]) < 1e-11
    for ext in ("int", "real"):
        f1, f2 = tmp_path / ("one_sis." + ext), tmp_path / ("many_sis." + ext)
        assert f1.exists() == f2.exists()
        if f1.exists():
            assert open(str(f1)).read() == open(str(f2)).read()
    if model == "hmm16":
        z = np.load(os.path.join(GOLD, "observations.npz"))
        isl, _, _ = run_main(tmp_path, "--model", model, "--smc", "--observes", obs, "--n_samples", 150000, "--seed", 9, "--generic", "--devices", "0,0,0", "--json",
                             "--islands", "--generated_file", "isl")
        assert isl["n_gpus"] == 3 and not isl["builtin"] and isl["n_resampled"] > 0 and isl["joint"] is False
        assert abs(isl["log_evidence"] - float(z["hmm16_logz"])) < 0.02
        got = np.array([p["p"][:3] + [0.0] * (3 - len(p["p"][:3])) for p in isl["predicts"]])
        assert np.abs(got - z["hmm16_smooth"]).max() < 0.01
        vi, lwi = read_dump(str(tmp_path / "isl_smc.int"), True)
        w = np.exp(lwi - lwi.max())
        from_files = np.array([[(w * (vi[t] == s2)).sum() / w.sum() for s2 in range(3)] for t in range(16)])
        np.testing.assert_allclose(from_files, got, atol=1e-9)        # the files weigh the islands as the combined statistics do
        assert abs(lwi.max() + np.log(w.sum()) - np.log(150000) - isl["log_evidence"]) < 1e-9


@pytest.mark.parametrize("model,key,T,ess,is_int,n", [("hmm16", "hmm16", 16, 2.0, True, 50001), ("hmm16", "hmm16", 16, 0.5, True, 30000),
                                                     ("linear_gaussian_1d25", "lgssm100", 25, 0.5, False, 40000), ("random_scale12", "lgssm100", 12, 0.5, False, 30000)])
@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0"])
def test_unchanged_model_smc_over_several_ranks_is_the_single_device_population(tmp_path, model, key, T, ess, is_int, n, devices):
    """north_star: "particles shard across the GPUs ... global weight sum and ancestor redistribution" for models that "compile
    unchanged".  cpprob::inference(StateType::smc, <unchanged model>) with options().devices naming several ranks (here: all on this
    GPU, the loopback form) runs ONE joint population: every step resamples all particles together -- the receiving rank's launch
    searches and walks the masses of whichever rank owns an ancestor and reads its window from that rank's store -- and since masses
    are exact integers, every surviving trace, every log-weight and the evidence equal the single-context run's, bit for bit (uneven
    shards, shards that end inside a 256-particle block, ESS-triggered schedules, the exact-maximum form included)."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    obs = z[key][:T]
    base = ["--model", model, "--smc", "--observes", obs_str(obs), "--n_samples", n, "--seed", 3, "--ess_threshold", ess, "--json", "--generic"]
    one, _, _ = run_main(tmp_path, *base, "--generated_file", "one")
    many, _, _ = run_main(tmp_path, *base, "--generated_file", "many", "--devices", devices)
    assert many["joint"] is True and many["n_gpus"] == len(devices.split(",")) and not many["builtin"] and one["joint"] is False
    assert many["step_form"] == one["step_form"] == (2 if model == "random_scale12" else 1)
    ext = "int" if is_int else "real"
    v1, lw1 = read_dump(str(tmp_path / ("one_smc." + ext)), is_int)
    vm, lwm = read_dump(str(tmp_path / ("many_smc." + ext)), is_int)
    assert np.array_equal(v1, vm) and np.array_equal(lw1, lwm)
    # (the joint run takes the evidence's logarithms on the HOST -- every generation's in the exact-maximum form, the last one's otherwise -- and
    #  the one-device run on the device: the two libraries' log may differ in the last place, so the evidence is compared to a few units of it)
    assert abs(many["log_evidence"] - one["log_evidence"]) <= 16 * np.spacing(abs(one["log_evidence"]))
    assert many["n_resampled"] == one["n_resampled"]
    for a, b in zip(one["predicts"], many["predicts"]):
        if "p" in a:
            np.testing.assert_allclose(a["p"], b["p"], rtol=0, atol=1e-12)
        else:
            assert abs(a["mean"] - b["mean"]) < 1e-11 and abs(a["variance"] - b["variance"]) < 1e-11


def test_vector_statements_run_through_the_generic_device_path(tmp_path):
    """the model's DEVICE VIEW (cpprob/device_view_begin.hpp: the same source compiled with fixed-capacity
    containers) carries vector-valued sample / observe / predict through model_kernel -- per-particle parity with the oracle and
    with the built-in kernel of the same model."""
    n = 60000
    args = ["--model", "gaussian_2d_unk_mean", "--sis", "--observes", "[3 4]", "--n_samples", n, "--seed", 9, "--json"]
    gen, _, _ = run_main(tmp_path, *args, "--generic", "--generated_file", "gen")
    blt, _, _ = run_main(tmp_path, *args, "--generated_file", "blt")
    assert not gen["builtin"] and blt["builtin"]
    ov, olw = O.sis(O.MODEL_GAUSSIAN_2D_UNKNOWN_MEAN, [3.0, 4.0], n, 9)
    for name in ("gen", "blt"):
        lines = open(str(tmp_path / (name + "_sis.real"))).read().splitlines()
        assert len(lines) == n
        vals = np.array([[float(x) for x in l[l.index("[", 3) + 1: l.index("]")].split()] for l in lines])     # ([(0 [v0 v1])] logw)
        lw = np.array([float(l.rsplit("]", 1)[1].strip().rstrip(")")) for l in lines])
        np.testing.assert_allclose(vals.T, ov, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(lw, olw, rtol=1e-10, atol=1e-10)
    for a, b in zip(gen["predicts"], blt["predicts"]):
        np.testing.assert_allclose(a["mean_nd"], b["mean_nd"], rtol=1e-9)
        np.testing.assert_allclose(a["variance_nd"], b["variance_nd"], rtol=1e-8)
    assert abs(gen["log_evidence"] - blt["log_evidence"]) < 1e-9


def test_all_distr_model_and_addressless_predicts(tmp_path):
    """the mixed-distribution model of src/models/models.cpp:13-47 (normal, uniform_smallint, uniform_real,
    poisson, multivariate normal; every predict WITHOUT an address) through the generic path.  Addresses come from the call sites
    (src/cpprob/utils.cpp:71-128): five predicts -> five distinct ids; the posterior is each prior reweighted by its own density."""
    n = 400000
    res, _, _ = run_main(tmp_path, "--model", "all_distr", "--sis", "--observes", "0 0", "--n_samples", n, "--seed", 6, "--json", "--generated_file", "ad")
    assert not res["builtin"]
    ids = open(str(tmp_path / "ad_sis.ids")).read().splitlines()
    assert len(ids) == 5 and len(set(ids)) == 5 and all(a.startswith("[") and "all_distr" in a for a in ids)
    reals = [p for p in res["predicts"] if "mean" in p]
    ints = [p for p in res["predicts"] if "p" in p]
    assert len(reals) == 3 and len(ints) == 2
    # weight = product of the densities at the sampled values, which factorises: each marginal is its prior times its own density
    # normal(1, 2) reweighted by N(x; 1, 2): N(1, sd 2/sqrt 2)
    assert abs(reals[0]["mean"] - 1.0) < 0.02 and abs(reals[0]["variance"] - 2.0) < 0.03
    # uniform_real(2, 9.5) reweighted by a constant: unchanged
    assert abs(reals[1]["mean"] - 5.75) < 0.02 and abs(reals[1]["variance"] - 7.5 ** 2 / 12) < 0.05
    # 4 independent normals (means 1..4, sd 2,1,5,3) each reweighted by its own density: variance halves
    np.testing.assert_allclose(reals[2]["mean_nd"], [1, 2, 3, 4], atol=0.05)
    np.testing.assert_allclose(reals[2]["variance_nd"], [2.0, 0.5, 12.5, 4.5], rtol=0.03)
    # uniform_smallint{2..7} reweighted by a constant: uniform over 2..7
    np.testing.assert_allclose(ints[0]["p"][2:8], [1 / 6.0] * 6, atol=5e-3)
    # poisson(0.8) reweighted by its own pmf: p(k) ~ pmf(k)^2
    from scipy.stats import poisson
    pm = poisson.pmf(np.arange(8), 0.8) ** 2
    np.testing.assert_allclose(ints[1]["p"][:6], (pm / pm.sum())[:6], atol=5e-3)


@pytest.mark.parametrize("model,window", [("hmm16", 1), ("linear_gaussian_1d25", 1), ("second_order12", 2), ("running_mean12", -1)])
def test_markov_probe_finds_the_replay_window_and_changes_no_number(tmp_path, model, window):
    """Unchanged-model SMC: the host probe (cpprob/detail/host_trace.hpp) finds how many of its ancestor's samples a step depends on --
    1 for the first-order models, 2 for second_order, none finite for running_mean (the whole trace is replayed) -- and the windowed
    replay draws the same variates, weights and ancestors as the full replay: every reported number is identical."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    T = {"hmm16": 16, "linear_gaussian_1d25": 25}.get(model, 12)
    obs = z["hmm16"] if model == "hmm16" else z["lgssm100"][:T]
    common = ["--model", model, "--smc", "--observes", obs_str(obs), "--n_samples", 50000, "--seed", 8, "--ess_threshold", 0.5, "--generic", "--json", "--no_dump"]
    # (same bookkeeping between the launches on both sides -- step form 0 -- so that the replay is the only difference)
    a, _, _ = run_main(tmp_path, *common, "--step_form", 0)
    b, _, _ = run_main(tmp_path, *common, "--no_markov_probe")
    assert a["replay_window"] == window and b["replay_window"] == -1 and not a["builtin"]
    assert a["log_evidence"] == b["log_evidence"] and a["n_resampled"] == b["n_resampled"] and a["ess"] == b["ess"]
    for pa, pb in zip(a["predicts"], b["predicts"]):
        assert pa == pb
    # the default form (resampling inside the model's launch, its own reference rule): the same posterior up to Monte-Carlo error
    d, _, _ = run_main(tmp_path, *common)
    assert d["replay_window"] == window and d["step_form"] == (1 if window >= 0 else 0)
    assert abs(d["log_evidence"] - b["log_evidence"]) < 0.05
    for pd, pb in zip(d["predicts"], b["predicts"]):
        if "mean" in pd: assert abs(pd["mean"] - pb["mean"]) < 0.05
        else: assert np.abs(np.array(pd["p"]) - np.array(pb["p"][:len(pd["p"])])).max() < 0.02
    if window >= 0:
        # ... and four particles a lane behind one ancestor search (step form 3): the default form's arithmetic, every number the same
        q, _, _ = run_main(tmp_path, *common, "--step_form", 3)
        assert q["step_form"] == 3 and q["replay_window"] == window
        assert q["log_evidence"] == d["log_evidence"] and q["n_resampled"] == d["n_resampled"] and q["ess"] == d["ess"]
        for pq, pd in zip(q["predicts"], d["predicts"]):
            assert pq == pd


def test_device_pilot_refutes_a_window_the_host_probe_lets_through(tmp_path):
    """rare_memory is first order on almost every trace; a state beyond 3.6 drags the first state back in.  Whatever the host probe's
    handful of traces concluded, the window is certified on the device -- a pilot population under windowed and under full replay
    must agree bit for bit -- and is refuted here: the run replays whole traces, and every number is the full replay's.  The
    first-order models keep their certified window (the other tests of this file)."""
    z = np.load(os.path.join(GOLD, "observations.npz"))
    obs = 1.5 * z["lgssm100"][:12]
    common = ["--model", "rare_memory12", "--smc", "--observes", obs_str(obs), "--n_samples", 50000, "--seed", 8, "--ess_threshold", 0.5, "--generic", "--json", "--no_dump"]
    a, _, _ = run_main(tmp_path, *common)
    b, _, _ = run_main(tmp_path, *common, "--no_markov_probe")
    c, _, _ = run_main(tmp_path, *common, "--no_markov_crosscheck")
    assert a["replay_window"] == -1 and b["replay_window"] == -1
    assert a["log_evidence"] == b["log_evidence"] and a["n_resampled"] == b["n_resampled"] and a["ess"] == b["ess"]
    for pa, pb in zip(a["predicts"], b["predicts"]):
        assert pa == pb
    if c["replay_window"] >= 0:
        # the host probe alone was fooled (and its run's numbers are those of another model): the pilot is what caught it
        assert a["markov_crosscheck"] == -1 and c["log_evidence"] != b["log_evidence"]
    h, _, _ = run_main(tmp_path, "--model", "hmm16", "--smc", "--observes", obs_str(z["hmm16"]), "--n_samples", 20000, "--generic", "--json", "--no_dump")
    assert h["replay_window"] == 1 and h["markov_crosscheck"] == 1
