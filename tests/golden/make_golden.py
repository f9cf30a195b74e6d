#!/usr/bin/env python3
"""Regenerates every fixture in tests/golden/ (run in the build container; needs scipy,
hipcc for the rocRAND host engine, and /root/reference for the serialization samples).

  logpdf_grid.npz        grids of the reference's own tests (tests/cpprob/logpdf.cpp:23-35
                         normal, :61-78 uniform) with closed-form values from scipy
                         (the reference compares against log(boost::math::pdf), i.e. the
                         same closed form, eps 1e-8, :16)
  philox_rocrand.json    Philox4x32-10 words / Box-Muller / u01 from rocRAND's host-callable
                         engine (gen_philox_rocrand.cpp)
  serialization.json     lines printed by the REFERENCE's serialization.hpp (oracle/_ref)
                         with dump_predicts' flags (state.cpp:262-267) + the inputs
  observations.npz       synthetic observation vectors of BASELINE.json configs C3/C4/C5
                         (SURVEY 8(d) seeds) and their exact posteriors
  posteriors.json        analytic anchors (README.md:118; thesis p.85)
"""
import json
import os
import subprocess
import sys

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import exact as E  # noqa: E402


def logpdf_grid():
    rows = []
    for mean in range(-10, 10):
        for std in range(1, 20):
            for f in range(1, 20):
                for i in range(-10, 10):
                    rows.append((i, mean / float(f), float(std)))
    g = np.array(rows)
    normal_expected = stats.norm.logpdf(g[:, 0], loc=g[:, 1], scale=g[:, 2])
    rows = []
    for a in range(-10, 10):
        for b in range(a + 1, 10):
            for f in range(1, 20):
                for i in range(-10, 10):
                    rows.append((i, a / float(f), b / float(f)))
    u = np.array(rows)
    with np.errstate(divide="ignore"):
        uniform_expected = np.log(stats.uniform.pdf(u[:, 0], loc=u[:, 1], scale=u[:, 2] - u[:, 1]))
    # extra closed forms for the off-path functors (no reference test pins them)
    pk = np.arange(0, 20)
    pl = np.array([0.5, 1.0, 3.7, 9.0])
    pgrid = np.array([(k, l) for l in pl for k in pk], float)
    poisson_expected = stats.poisson.logpmf(pgrid[:, 0], pgrid[:, 1])
    np.savez_compressed(os.path.join(HERE, "logpdf_grid.npz"), normal_grid=g, normal_expected=normal_expected,
                        uniform_grid=u, uniform_expected=uniform_expected,
                        poisson_grid=pgrid, poisson_expected=poisson_expected)


def philox():
    exe = "/tmp/gen_philox_rocrand"
    subprocess.check_call(["hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", "-o", exe,
                           os.path.join(HERE, "gen_philox_rocrand.cpp")])
    out = subprocess.check_output([exe]).decode()
    json.loads(out)
    with open(os.path.join(HERE, "philox_rocrand.json"), "w") as f:
        f.write(out)


def serialization():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_serialization")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(7)
    real_cases, int_cases = [], []
    specials = [0.0, -0.0, 1.0, -1.5, 3.083333333333333, 1e-300, -2.5e+120, 123456789.123456789]
    for n in (0, 1, 2, 5):
        for _ in range(4):
            vals = [float(rng.choice(specials)) if rng.random() < 0.3 else float(rng.normal() * 10.0 ** int(rng.integers(-3, 4)))
                    for _ in range(n)]
            real_cases.append((vals, float(-abs(rng.normal()) * 10)))
            int_cases.append(([int(v) for v in rng.integers(0, 3, n)], float(-abs(rng.normal()) * 10)))

    def run(mode, cases):
        inp = "".join("%d %s %r\n" % (len(v), " ".join(repr(x) for x in v), lw) for v, lw in cases)
        return subprocess.check_output([exe, mode], input=inp.encode()).decode().splitlines()

    real_lines = run("print-real", real_cases)
    int_lines = run("print-int", int_cases)
    doc = {"real": [{"values": [x.hex() for x in v], "logw": lw.hex(), "line": ln}
                    for (v, lw), ln in zip(real_cases, real_lines)],
           "int": [{"values": v, "logw": lw.hex(), "line": ln} for (v, lw), ln in zip(int_cases, int_lines)]}
    with open(os.path.join(HERE, "serialization.json"), "w") as f:
        json.dump(doc, f, indent=1)


def observations():
    hmm16 = E.simulate_hmm(16, 20260101)
    lg100 = E.simulate_lgssm(100, 20260102)
    hmm128 = E.simulate_hmm(128, 20260103)
    g16, a16, ll16 = E.hmm_forward_backward(hmm16)
    g128, a128, ll128 = E.hmm_forward_backward(hmm128)
    ms, ps, mf, pf, lll = E.kalman_rts(lg100)
    np.savez_compressed(os.path.join(HERE, "observations.npz"),
                        hmm16=hmm16, hmm16_smooth=g16, hmm16_filter=a16, hmm16_logz=ll16,
                        hmm128=hmm128, hmm128_smooth=g128, hmm128_filter=a128, hmm128_logz=ll128,
                        lgssm100=lg100, lgssm100_smooth_mean=ms, lgssm100_smooth_var=ps,
                        lgssm100_filter_mean=mf, lgssm100_filter_var=pf, lgssm100_logz=lll)


def posteriors():
    doc = {
        "readme_gaussian_obs_3_4": {"mean": 2.32353, "variance": 1.05882, "source": "README.md:118",
                                    "derived": list(map(float, E.gaussian_posterior(1, 1.5, 2, [3, 4])))},
        "models_gaussian_obs_8_9": {"mean": 7.25, "variance": 5.0 / 6.0, "source": "thesis 6.1 p.85",
                                    "derived": list(map(float, E.gaussian_posterior(1, np.sqrt(5), np.sqrt(2), [8, 9])))},
        "models_gaussian_obs_3_4": {"mean": 3.0833333333333335, "variance": 0.8333333333333334, "source": "BASELINE.md (derived)",
                                    "derived": list(map(float, E.gaussian_posterior(1, np.sqrt(5), np.sqrt(2), [3, 4]))),
                                    "log_evidence": float(E.gaussian_log_evidence(1, np.sqrt(5), np.sqrt(2), [3, 4]))},
    }
    with open(os.path.join(HERE, "posteriors.json"), "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    logpdf_grid(); philox(); serialization(); observations(); posteriors()
    print("golden fixtures regenerated in", HERE)
