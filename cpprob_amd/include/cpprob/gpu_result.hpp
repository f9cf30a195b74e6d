// Host-visible results and options of a device inference run (additions to the reference API; the
// reference returns nothing and leaves everything in files).
#ifndef CPPROB_COMPAT_GPU_RESULT_HPP
#define CPPROB_COMPAT_GPU_RESULT_HPP
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "cpprob_hip.h"

namespace cpprob {
namespace gpu {

struct Options {
    int device = 0;
    std::vector<int> devices;                         // more than one entry: ONE joint population sharded over these GPUs, resampled exactly as
                                                      // one GPU holding every particle would (exchange scope: per step an RCCL all-gather of the
                                                      // rank totals and the redistribution of offspring over xGMI; cpprob_hip_group_*).  All
                                                      // entries equal: every rank on that one GPU (loopback transport; a one-GPU machine's way
                                                      // to run the protocol).  Built-in models; unchanged models (CPPROB_REGISTER_MODEL): under
                                                      // StateType::sis the shards need no communication until their sums are combined (the
                                                      // one-device run, exactly); under StateType::smc every device runs an ISLAND of its own
                                                      // and the islands are combined by their evidence estimates.
    std::uint64_t seed = 12345;
    int resampler = CPPROB_HIP_RESAMPLE_SYSTEMATIC;   // smc
    double ess_threshold = 0.5;                       // smc: resample when ESS < threshold * N (thesis p.37); > 1: every step
    bool dump = true;                                 // write <file>.real/.int/.ids like the reference (state.cpp:193-202)
    std::size_t dump_max_particles = 0;               // 0 = all particles
    bool markov_probe = true;                         // smc, unchanged-model path: test on the host whether a step depends on more than the last few
                                                      // sampled values; a model that does not is replayed from that window only (O(T) instead of O(T^2))
    bool markov_crosscheck = true;                    // ... and certify the probe's window on the device before using it: a pilot population under windowed
                                                      // and under full replay must agree bit for bit (once per model and trace shape), else full replay
    bool prefer_builtin = true;                       // use the hand-fused kernels when the model is one of the built-ins
    bool keep_history = true;                         // smc, built-in models on one GPU: false = filtering only -- O(N) particle store instead of O(N T),
                                                      // every predict hit's numbers under its own generation's weights, no posterior files
                                                      // (cpprob_hip_config::keep_history)
    std::uint64_t particle_offset = 0;                // (set by the engine when it shards a population: global id of this shard's first particle)
    bool progress = false;
    bool islands = false;                             // smc over several ranks, unchanged models: independent SMC runs combined by their evidence instead of
                                                      // the joint population (the default wherever the model has a joint form)
    int step_form_override = -1;                      // smc, unchanged-model path: -1 the engine chooses; 0 model launch + separate bookkeeping launches,
                                                      // 1 the resampling inside the model's launch against the dry run's bounds, 2 ... against exact maxima
    bool step_builds = true;                          // smc, unchanged-model path: launch the step kernels built per step (CPPROB_REGISTER_MODEL_STEPS) where the model
                                                      // unit holds them for a step's thresholds (false: always the run-time kernel -- the A/B switch)
    int replicates = 1;                               // built-in models: R independent runs (seeds seed .. seed + R - 1), up to three in
                                                      // flight on separate contexts; Result then carries their spread (error bars)
    bool joint_across_devices = false;                // smc, unchanged models, `devices` naming DIFFERENT GPUs: run the joint population (its peer reads have
                                                      // only ever run between loopback ranks of one device) instead of islands; Result::joint_note says so
};

struct PredictStats {            // one per predict hit, StatsPrinter's numbers
    std::string address;
    bool is_int = false;
    double mean = 0, variance = 0;            // real predicts (vector-valued: component 0)
    std::vector<double> mean_nd, variance_nd; // real predicts: one entry per component (size 1 for a scalar)
    std::vector<double> probabilities;        // int predicts: P(x = s), s = 0..k-1
};

struct Result {
    std::size_t n_particles = 0;
    double log_evidence = 0, ess = 0, log_norm = 0;
    int n_resampled = 0;
    bool used_builtin = false;
    int n_gpus = 1;                           // ranks the population was sharded over
    int exchange_reruns = 0;
    int markov_crosscheck = 0;                // unchanged-model smc: 1 the device pilot certified the probe's window, -1 it refuted it (full replay), 0 not run
    int replay_window = -1;                   // unchanged-model smc: samples of the ancestor a step replays (-1: the whole trace)                  // multi-GPU: runs repeated with a larger lineage transport (results never depend on it)
    bool joint = false;                       // unchanged-model smc over several ranks: ONE joint population (false: islands combined by their evidence)
    int joint_flag = 0;                       // (a rank's share of a joint run: the flag its generations raised)
    std::string joint_note;                   // why a multi-device smc call of an unchanged model ran islands, or that its joint run is unvalidated on real links
    int step_form = 0;                        // unchanged-model smc: 0 separate bookkeeping launches, 1 fused step on bounded references, 2 fused step + exact-maximum pass,
                                              // 3 = 1 with four particles a lane behind one ancestor search (chosen for large populations of models with <= 32 observes)
    int launches_per_step = 0;                // unchanged-model smc: dependent launches per observe
    int step_builds_used = 0;                 // unchanged-model smc: launches of the last attempt that ran a step kernel built for its step (model_step_kernel_at)
    double setup_seconds = 0;                 // unchanged-model path: context / workspace / scratch set-up and the Markov pilot of THIS call (0.0x ms once the workspace is warm)
    bool workspace_grown = false;             // ... this call created or enlarged its device workspace
    double run_seconds = 0;                   // device work of the run (launch to synchronise), excluding allocation and dumps
    std::vector<PredictStats> predicts;       // real hits first (in trace order), then int hits
    std::vector<double> step_ess;             // smc: ESS after each observe
    // Options::replicates > 1: replicate 0 is what the fields above (and the dump) describe; across replicates:
    int n_replicates = 1;
    double log_evidence_mean = 0, log_evidence_sd = 0;        // log of the mean evidence (unbiased in Z); sd of the log estimates
    std::vector<std::vector<double>> replicate_values;        // [predict hit][replicate]: mean (real, component 0) or P(x = 0) (int)
    std::vector<double> predict_mean, predict_sd;             // per predict hit over the replicates
    double replicates_seconds = 0;                            // device time of all replicates together
};

inline Options& options() { static thread_local Options o; return o; }
inline Result& last_result() { static thread_local Result r; return r; }

}  // namespace gpu
}  // namespace cpprob
#endif
