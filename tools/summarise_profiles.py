#!/usr/bin/env python3
"""Turns the raw output of tools/profile_round.sh (gpurun_out/<tag>_*) into the committed evidence under profiles/:
  profiles/<tag>_bench_n1.json                      the bench line (with PMC traffic filled in)
  profiles/<tag>_<workload>_<n>_kernel_stats.{csv,md}  rocprofv3 --kernel-trace --stats, resampling / non-resampling launches apart
  profiles/<tag>_pmc_traffic.json                   HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, gfx950-corrected) + issue counters
  profiles/<tag>_loopback.md                        the 8-GPU configs whole, eight loopback ranks on one GPU
usage: python tools/summarise_profiles.py r03b r03"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src_tag, out_tag = sys.argv[1], sys.argv[2]
G = "gpurun_out"


def newest(pat):
    fs = glob.glob(pat)
    return max(fs, key=os.path.getmtime) if fs else None


def counters(d, full_grid=None):
    """Per kernel and counter: (mean, launches).  full_grid: step kernels are counted at THIS grid size only -- bench.py's launch-floor
    run (the same kernels at 4096 particles) sits in the same process and would dilute a per-launch mean (it did in r01 .. r04: the
    committed traffic of the headline step was 8.3 MB where the population's launches move 14.1)."""
    f = newest("%s/%s/*/*counter_collection.csv" % (G, d))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            if full_grid is not None and "smc_step" in r["Kernel_Name"] and int(r["Grid_Size"]) != full_grid:
                continue
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d2.items()} for k, d2 in acc.items()}


pmc = {"_comment": "rocprofv3 --pmc, one counter set per run (tools/profile_round.sh).  hbm_bytes_per_launch_corrected = 2 x FETCH_SIZE KB x 1024 + "
                   "WRITE_SIZE KB x 1024 (MI355X_MICROARCH.md: FETCH_SIZE counts 64-byte requests in units that read half on gfx950; WRITE_SIZE in KB).  "
                   "Per-wave issue counters are SQ sums over the chip divided by the launch's wavefronts (workgroups x 4).", "workloads": {}}
ALGO = {"hmm16_smc": 56, "lgssm100_smc": 72, "hmm128_smc_ess": 56, "gaussian_sis": 16}
for wl, n in (("hmm16_smc", 1000000), ("hmm16_smc", 10000000), ("lgssm100_smc", 1250000), ("lgssm100_smc", 10000000), ("hmm128_smc_ess", 12500000)):
    fg = ((n + 1023) // 1024) * 256
    fe, wr, sq = counters("%s_pmc_%s_%d_FETCH_SIZE" % (src_tag, wl, n), fg), counters("%s_pmc_%s_%d_WRITE_SIZE" % (src_tag, wl, n), fg), counters("%s_pmc_%s_%d_SQ_WAVE_CYCLES" % (src_tag, wl, n), fg)
    rec = {}
    for k in fe:
        name = "step_kernel" if "smc_step" in k else ("smooth_kernel" if ("smooth" in k or "trace_readout" in k) else None)
        if not name:
            continue
        f = fe[k]["FETCH_SIZE"][0]
        w = wr.get(k, {}).get("WRITE_SIZE", (0, 0))[0]
        waves = ((n + 1023) // 1024) * 4
        e = {"kernel": k[:100], "launches": fe[k]["FETCH_SIZE"][1], "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "hbm_bytes_per_launch_corrected": 2 * f * 1024 + w * 1024}
        if k in sq:
            c = {kk: vv[0] for kk, vv in sq[k].items()}
            e.update({"valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0) / waves, "salu_insts_per_wave": c.get("SQ_INSTS_SALU", 0) / waves,
                      "wave_cycles_per_wave": c.get("SQ_WAVE_CYCLES", 0) / waves, "wait_frac": c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
                      "valu_issue_frac": c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)})
        rec[name] = e
    if rec:
        pmc["workloads"]["%s@%d" % (wl, n)] = rec
json.dump(pmc, open("profiles/%s_pmc_traffic.json" % out_tag, "w"), indent=1)

notes = []
for wl, n in (("hmm16_smc", 1000000), ("hmm16_smc", 10000000), ("lgssm100_smc", 1250000), ("lgssm100_smc", 10000000), ("hmm128_smc_ess", 12500000), ("gaussian_sis", 10000000)):
    d = "%s_prof_%s_%d" % (src_tag, wl, n)
    ks, tr = newest("%s/%s/*/*kernel_stats.csv" % (G, d)), newest("%s/%s/*/*kernel_trace.csv" % (G, d))
    if not ks:
        continue
    base = "profiles/%s_%s_%d_kernel_stats" % (out_tag, wl, n)
    shutil.copy(ks, base + ".csv")
    rows = list(csv.DictReader(open(ks)))
    md = "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --particles %d --steps 10 --warmup 2 --no-cpu-baseline --no-extras (MI355X, %s)\n\n" % (wl, n, out_tag)
    md += "| kernel | calls | total us | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n"
    for r in rows[:12]:
        md += "| `%s` | %s | %.1f | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"])
    if tr:
        # (the step launches of THIS population: bench.py's launch-floor run -- 4096 particles, four workgroups -- is in the same trace)
        full_grid = ((n + 1023) // 1024) * 256
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr))
                if ("smc_step" in r["Kernel_Name"] or (wl == "gaussian_sis" and "sis_" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"])) and (wl == "gaussian_sis" or int(r["Grid_Size_X"]) == full_grid)]
        if durs:
            import statistics
            srt = sorted(durs)
            # launches split at the widest gap of the sorted durations between the 10th and 90th percentile: the lighter group does not resample
            lo, hi = srt[len(srt) // 10], srt[(9 * len(srt)) // 10]
            thr = (lo + hi) / 2
            a, b = [x for x in durs if x < thr], [x for x in durs if x >= thr]
            by = ALGO[wl]
            light = {"hmm16_smc": 24, "hmm128_smc_ess": 24, "lgssm100_smc": 32, "gaussian_sis": 16}[wl]
            md += "\nStep launches in the kernel trace: %d, average %.2f us.  " % (len(durs), sum(durs) / len(durs))
            if a and b and hi > 1.3 * lo:
                md += "Launches that resample (>= %.1f us): %d, average %.2f us = %d B x %d / %.2f us = %.0f GB/s = **%.2f** of 8 TB/s; launches that do not (step 0, steps behind a generation that kept its weights): %d, average %.2f us = %d B x %d / that = %.0f GB/s = %.2f.\n" % (
                    thr, len(b), sum(b) / len(b), by, n, sum(b) / len(b), by * n / (sum(b) / len(b)) / 1e3, by * n / (sum(b) / len(b)) / 1e3 / 8000,
                    len(a), sum(a) / len(a), light, n, light * n / (sum(a) / len(a)) / 1e3, light * n / (sum(a) / len(a)) / 1e3 / 8000)
            else:
                md += "%d B x %d / %.2f us = %.0f GB/s = **%.2f** of 8 TB/s by the SURVEY 8(d) convention.\n" % (by, n, sum(durs) / len(durs), by * n / (sum(durs) / len(durs)) / 1e3, by * n / (sum(durs) / len(durs)) / 1e3 / 8000)
    bj = "%s/%s.json" % (G, d)
    if os.path.exists(bj):
        b = json.load(open(bj))
        md += "\nbench.py, same command un-profiled, same gpurun call: %.4g particles/s, %.4f ms per run, step %.2f us per launch by HIP events (events bracket the run's launches: gaps included), roofline.frac %.3f, bound %s, launch floor %.2f us.\n" % (
            b["value"], b["ms_per_step"], b["roofline"]["avg_launch_us"], b["roofline"]["frac"], b["roofline"]["bound"], b["roofline"]["launch_floor_us"])
    open(base + ".md", "w").write(md)
    notes.append(base + ".md")

# the other resamplers: kernel stats + the un-profiled line of the same call
for wl, n in (("hmm16_smc", 1000000), ("lgssm100_smc", 1250000)):
    for rs_name in ("stratified", "multinomial", "multinomial_literal"):
        d = "%s_prof_%s_%d_%s" % (src_tag, wl, n, rs_name)
        ks = newest("%s/%s/*/*kernel_stats.csv" % (G, d))
        if not ks:
            continue
        rows = list(csv.DictReader(open(ks)))
        md = "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --particles %d --resampler %s --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc (MI355X, %s)\n\n" % (wl, n, rs_name, out_tag)
        md += "| kernel | calls | total us | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n"
        for r in rows[:12]:
            md += "| `%s` | %s | %.1f | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:120], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"])
        bj = "%s/%s.json" % (G, d)
        if os.path.exists(bj) and os.path.getsize(bj):
            b = json.load(open(bj))
            md += "\nbench.py, same command un-profiled, same gpurun call: %.4g particles/s, %.4f ms per run, step %.2f us per launch by HIP events, step form: %s, posterior max abs err vs exact %.2e.\n" % (
                b["value"], b["ms_per_step"], b["roofline"]["avg_launch_us"], b["roofline"]["step_form"], b["posterior_max_abs_err_vs_exact"])
        open("profiles/%s_%s_%d_%s_kernel_stats.md" % (out_tag, wl, n, rs_name), "w").write(md)
        notes.append("profiles/%s_%s_%d_%s_kernel_stats.md" % (out_tag, wl, n, rs_name))

# the three resamplers' sharded form: eight loopback ranks of one population (multinomial: the strata form over cut strata)
for rs_name in ("multinomial", "systematic"):
    d = "%s_prof_loop8_%s" % (src_tag, rs_name)
    ks = newest("%s/%s/*/*kernel_stats.csv" % (G, d))
    if not ks:
        continue
    rows = list(csv.DictReader(open(ks)))
    md = "# rocprofv3 --kernel-trace --stats -- python3 bench.py --resampler %s --loopback-ranks 8 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc (MI355X, %s)\n\n" % (rs_name, out_tag)
    md += "hmm<16>, 10^6 particles as ONE population over eight loopback ranks of this GPU (one stream: the ranks' launches serialise).\n\n"
    md += "| kernel | calls | total us | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n"
    for r in rows[:12]:
        md += "| `%s` | %s | %.1f | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:120], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"])
    bj = "%s/%s.json" % (G, d)
    if os.path.exists(bj) and os.path.getsize(bj):
        b = json.load(open(bj))
        md += "\nbench.py, same command un-profiled, same gpurun call: %.4f ms per run, posterior max abs err vs exact %.2e; per rank-step (sum over the eight ranks' launches, us): %s.\n" % (
            b["ms_per_step"], b["posterior_max_abs_err_vs_exact"], {k: round(v, 1) for k, v in (b.get("rank_step_breakdown_us") or {}).items() if isinstance(v, float)})
    open("profiles/%s_hmm16_smc_1000000_loop8_%s_kernel_stats.md" % (out_tag, rs_name), "w").write(md)
    notes.append("profiles/%s_hmm16_smc_1000000_loop8_%s_kernel_stats.md" % (out_tag, rs_name))
for f in ("ab_step_builds.txt", "ab_quad_builds.txt", "ab_walk_r06.txt"):
    if os.path.exists("%s/%s" % (G, f)):
        shutil.copy("%s/%s" % (G, f), "profiles/%s_%s" % (out_tag, f))
        notes.append("profiles/%s_%s" % (out_tag, f))

bp = "%s/%s_bench.json" % (G, src_tag)
if os.path.exists(bp):
    b = json.load(open(bp))
    rec = pmc["workloads"].get("hmm16_smc@1000000", {}).get("step_kernel")
    if rec:
        b["roofline"]["traffic"] = rec["hbm_bytes_per_launch_corrected"]
        b["roofline"]["traffic_measured_in"] = "profiles/%s_pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command in the SAME gpurun call, gfx950-corrected)" % out_tag
        b["roofline"]["hbm_frac_measured"] = rec["hbm_bytes_per_launch_corrected"] / (b["roofline"]["avg_launch_us"] * 1e-6) / 1e9 / 8000.0
        b["roofline"]["valu_issue_frac"] = rec.get("valu_issue_frac"); b["roofline"]["wait_frac"] = rec.get("wait_frac")
    json.dump(b, open("profiles/%s_bench_n1.json" % out_tag, "w"))

for form in (1, 0, 3):
    d = "%s_generic_form%d" % (src_tag, form)
    ks = newest("%s/%s/*kernel_stats.csv" % (G, d)) or newest("%s/%s/*/*kernel_stats.csv" % (G, d))
    tr = newest("%s/%s/*kernel_trace.csv" % (G, d)) or newest("%s/%s/*/*kernel_trace.csv" % (G, d))
    if not ks:
        continue
    rows = list(csv.DictReader(open(ks)))
    gmd = "# rocprofv3 --kernel-trace --stats -- cpprob_main --generic --model hmm16 --smc --n_samples 1000000 --repeat 8 --step_form %d (MI355X, %s; tools/profile_generic.sh)\n\n" % (form, out_tag)
    gmd += "step form %d = %s.  Eight cpprob::inference calls + the Markov pilot (8192 particles, both replay forms) in one process.\n\n" % (form, {1: "the resampling inside the model's launch, one particle a lane (model_step_kernel_at: a build per step, no dead iteration; the run-time model_step_kernel where a step has none)", 0: "model launch + three bookkeeping launches (the r03 form)", 3: "the resampling inside the model's launch, four particles a lane behind one ancestor search (model_step_kernel_quad_at: a build per step, a call of the body is the live iteration alone; the engine's choice from 7e5 particles on)"}[form])
    gmd += "| kernel | calls | total us | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n"
    for r in rows[:10]:
        gmd += "| `%s` | %s | %.1f | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:120], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"])
    if tr:
        mk = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(tr)) if "model_" in r["Kernel_Name"])
        last = [x for _, x in mk[-16:]]
        gmd += "\nThe last call's 16 model launches, t = 0 .. 15 (us): %s -- sum %.1f us (r05, run-time kernel: launch t ran t dead iterations of the model's own loop before the live one: 14.0, then 25.7 + 0.39 t).\n" % (" ".join("%.1f" % x for x in last), sum(last))
    up = "%s/%s/unprofiled.json" % (G, d)
    if os.path.exists(up) and os.path.getsize(up):
        try:
            u = json.loads(open(up).read().strip())
            gmd += "\nThe same command un-profiled, same gpurun call: %.4f ms per run (the last call's device work, read-out included).\n" % (u["run_seconds"] * 1e3)
        except Exception:
            pass
    if form == 1:
        # issue counters of the fused step kernel: SQ sums over the chip, per wavefront (3907 workgroups x 4 at 10^6 particles)
        for sub, title in (("_pmc", "issue"), ("_pmc_lds", "LDS / scalar")):
            cc = newest("%s/%s%s/*counter_collection.csv" % (G, d, sub)) or newest("%s/%s%s/*/*counter_collection.csv" % (G, d, sub))
            if not cc:
                continue
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(cc)):
                if "model_step_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if acc:
                waves = ((1000000 + 255) // 256) * 4
                c = {k: sum(v) / len(v) for k, v in acc.items()}
                gmd += "\nPMC (%s counters, rocprofv3 --pmc in a run of its own; averages over %d launches of `model_step_kernel*`, per wavefront = / %d):\n\n| counter | per launch | per wavefront |\n|---|---|---|\n" % (title, len(next(iter(acc.values()))), waves)
                for k in sorted(c):
                    gmd += "| %s | %.4g | %.4g |\n" % (k, c[k], c[k] / waves)
                if "SQ_WAVE_CYCLES" in c:
                    gmd += "\nwait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f; valu_issue_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.2f; scalar instructions per vector instruction = %.2f.\n" % (
                        c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"], c.get("SQ_INSTS_SALU", 0) / max(c.get("SQ_INSTS_VALU", 1), 1))
    open("profiles/%s_generic_hmm16_form%d_kernel_stats.md" % (out_tag, form), "w").write(gmd)
    notes.append("profiles/%s_generic_hmm16_form%d_kernel_stats.md" % (out_tag, form))

md = "# The two 8-GPU configs (and the headline shape x 8) whole, eight loopback ranks on ONE MI355X (%s)\n\n" % out_tag
md += "`python bench.py --workload W --particles N --loopback-ranks 8` -- the library's multi-GPU driver with every rank's context on this GPU and one stream: the whole exchange protocol, program order instead of collectives, so a run is the SUM of the eight ranks' steps.\n\n"
md += "| workload | particles (whole) | ms per run | records per run | bytes on the links per run (direct stores) | reruns (settling / timed) | posterior max abs err vs exact |\n|---|---|---|---|---|---|---|\n"
for f, label in (("loop_c4", "configs[3] linear_gaussian_1d<100>, ESS < N/2"), ("loop_c5", "configs[4] hmm<128>, ESS < N/2"), ("loop_c3x8", "hmm<16> every step, 8 x 10^6")):
    p = "%s/%s_%s.json" % (G, src_tag, f)
    if os.path.exists(p) and os.path.getsize(p):
        b = json.load(open(p))
        t = b.get("exchange_traffic_per_run") or {}
        rr = b["config"]["exchange_reruns"]
        md += "| %s | %d | %.2f | %s | %s | %s / %s | %.2e |\n" % (label, b["config"]["n_global"], b["ms_per_step"], t.get("records"), t.get("wire_bytes"), rr.get("settling"), rr.get("timed_batch"), b["posterior_max_abs_err_vs_exact"])
open("profiles/%s_loopback.md" % out_tag, "w").write(md)
print("wrote", notes, "profiles/%s_pmc_traffic.json" % out_tag, "profiles/%s_loopback.md" % out_tag)
