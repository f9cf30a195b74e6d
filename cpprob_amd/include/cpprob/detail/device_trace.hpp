// Device side of the statement API (hipcc only): what cpprob::sample / observe / predict do when
// the model body runs on a GPU lane.  Replaces the reference's process-global trace record
// (StateInfer::trace_, src/cpprob/state.cpp:148; TraceInfer, include/cpprob/trace.hpp:34-63) by a
// per-lane record in LDS (the mutable part) and the kernel-argument segment (the rest), one particle per lane.
//
//   sample  #j : SIS -> a fresh draw (Philox block of (particle id, ordinal j), cpprob/detail/rng.hpp);
//                SMC step t -> the stored value of the lane's ancestor for j < n_stored (trace replay),
//                a fresh draw otherwise; every value is written to the lane's new trace column.
//   observe #m : log_w += logpdf for first_observe <= m; after observe #stop_after the lane is `done`
//                and every later statement is a no-op (SMC stops all particles at the same observe).
//   predict    : k-th real / int hit -> column k of the particle store (only when the pointers are set).
#ifndef CPPROB_COMPAT_DETAIL_DEVICE_TRACE_HPP
#define CPPROB_COMPAT_DETAIL_DEVICE_TRACE_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include <boost/random/discrete_distribution.hpp>
#include <boost/random/normal_distribution.hpp>
#include <boost/random/poisson_distribution.hpp>
#include <boost/random/uniform_real_distribution.hpp>
#include <boost/random/uniform_smallint.hpp>

#include "cpprob/detail/device_vector.hpp"
#include "cpprob/detail/fixed_mass.hpp"
#include "cpprob/detail/rng.hpp"
#include "cpprob/distributions/utils_distributions.hpp"

namespace cpprob {
namespace device {

constexpr int kLaneBlock = 256;              // threads per workgroup of model_kernel (one launch runs the whole model body: SIS, full replay)
constexpr int kStepBlock = 256;              // ... of model_step_kernel (windowed replay with the resampling inside the launch), one particle per lane:
constexpr int kStepWaves = kStepBlock / 64;  //     one entry of the mass hierarchy per workgroup
constexpr int kStepPass = 2;                 // source blocks the walk scans per pass (one particle per lane and block)
constexpr int kStepFetch = 6;                // source blocks whose weights are fetched at kernel entry ...
constexpr int kStepFetchBack = 2;            // ... starting this many blocks before the workgroup's own
constexpr int kStepProbe = 16;               // blocks whose boundaries the search's probe evaluates at once
constexpr int kStepProbeBack = 7;            // ... starting this many blocks before the workgroup's own

// What the fused SMC step of an unchanged model carries (cpprob/gpu.hpp: model_step_kernel): generation t-1's fixed-point masses
// for the ancestor search in the launch's prologue, and where generation t's go in its epilogue (cpprob/detail/fixed_mass.hpp).
struct StepCtrl2 { double ref_cur, gap_max; };            // device-resident between launches: R_{t-1}; the largest R_t - M_t of the run
struct FusedStep {
    cph::FHier f;                              // masses of generation t-1 (read); generation t's are published f.h.to_next words further
    const uint32_t* q_prev; uint32_t* q_next;  // [tiles * 1024] integer weights of generations t-1 / t
    double u0;                                 // systematic offset of the resampling before step t (Philox, evaluated on the host)
    double bound;                              // B_t: host-known upper bound of the step's incremental log-weight (exact_ref = 0)
    double ess_frac, n_pop;
    int32_t t, T, nb;
    int32_t may_carry;                         // 0: every step resamples (threshold > 1): no log-weight ever carries into a step
    int32_t exact_ref;                         // 1: no bound is known -- the epilogue publishes the tile maxima only and a separate launch quantises
    int32_t* anc_row;                          // [n] ancestors of generation t (written when the step resampled)
    StepCtrl2* ctrl; double* ess; int32_t* resampled; double* log_z;
    double gap_limit;                          // a generation whose heaviest particle sits further below its reference raises flag 5
};

// One shard of a JOINT population (cpprob/gpu.hpp: generic_joint): the resampling is the whole population's -- ancestors are searched
// in the hierarchy and the weights of whichever rank owns them and their windows are read from that rank's store through its mapping
// (peer access between the GPUs of a process; plain pointers when several ranks share one GPU) -- the migration is a PULL by the
// receiving rank's step launch: no packing launch on the sender, no remote writes.  The 24 bytes a rank contributes to a generation's
// totals are pulled the same way: a one-wavefront launch behind every step leaves them in the rank's memory and the next step's first
// wavefront reads every rank's (lane r = rank r: the all-gather is three loads a lane) and takes the generation's decision as a
// single device's prologue does.  What orders a rank's step t + 1 behind every rank's step t is a stream wait on the peers' events.
// In the exact-maximum form a pass over the log-weights lies between two launches anyway: there the ranks' host threads combine the
// numbers and hand the decision to the launch ready-made (ShardArgs::host_decided).
constexpr int kMaxShards = 16;
constexpr int kShardIndexBits = 26;                                    // an ancestor's code: rank << 26 | slot on that rank (shards up to 6.7e7 particles)
struct ShardPeer {                                                     // rank r's generation t-1 as every rank addresses it
    cph::FHier f; const uint32_t* q; const uint64_t* carry; int64_t n; int32_t nb, pad;
    const uint64_t* totals;                                            // {mass, squares, key of the maximum} of that rank's generation t-1 (cpprob_hip_generic_totals)
};
struct ShardArgs {
    int32_t world, rank;                                               // world = 0: a population of its own
    const ShardPeer* peers;                                            // [world], device memory of this rank
    double obound[kMaxShards + 1];                                     // o_r = first output owned by rank r's sources (o_world = N): G(mass before rank r)
    uint64_t before[kMaxShards];                                       // mass of the ranks before r
    uint64_t first[kMaxShards + 1];                                    // global id of rank r's first particle
    double inv, ref; int32_t resample;                                 // generation t-1's decision and generation t's reference, where the host took them
    int32_t host_decided;                                              // 1: obound / before / inv / ref / resample above are valid (exact-maximum form); 0: the launch
};                                                                     //    derives them from the ranks' totals itself (ShardPeer::totals): no host round trip between two steps

// What every statement of a launch reads and none writes: the kernel's first argument.  A statement fetches the fields it needs
// straight from the kernel-argument segment (scalar loads into scalar registers, wherever in the call tree it sits), so they cost
// no LDS and no vector registers.  model_kernel / model_step_kernel (cpprob/gpu.hpp) take this struct as their FIRST parameter.
struct LaunchArgs {
    int64_t n, ld;
    uint64_t seed;
    const int32_t* anc;            // ancestors of this generation (identity where the previous step did not resample); nullptr at step 0
    const int32_t* resampled_prev; // device flag: did the previous step resample?  (the decision is taken on the device)
    const double* logw_in;         // log-weights of the previous generation, carried over when it was not resampled
    double* logw_out;
    const uint64_t* trace_in; uint64_t* trace_out;       // stored sample values, row stride ld: the ancestor's / this lane's new column
    const int32_t* nstored_in; int32_t* nstored_out;
    double* pred_real; int32_t* pred_int;                 // predict columns, row stride ld (nullptr: do not record)
    int32_t first_observe;         // observes with a smaller index were weighted in earlier steps
    int32_t stop_after;            // index of the observe that ends this step (-1: run to completion)
    uint32_t trace_cap;            // rows of the trace buffers; a longer trace raises *overflow (rejection loops)
    int32_t* overflow;
    uint32_t pred_real_cap, pred_int_cap;                 // predict columns the host allocated (hits seen by the structural dry run)
    // windowed replay (models that passed the Markov probe): a step replays only the ancestor's last `win` samples, kept per
    // particle in `carry` rows (slot k <-> ordinal fresh_lo - win + k); older samples come back value-initialised, never from memory
    uint32_t windowed, win;
    int32_t fresh_lo;              // first ordinal this step draws itself (= samples before the previous observe)
    int32_t next_fresh;            // first ordinal the NEXT step draws (= samples before this step's observe): what carry_out must end with
    const uint64_t* carry_in; uint64_t* carry_out;
    uint64_t pid0;                 // global id of lane 0's particle (shards of one population draw from the population's streams)
    uint32_t lane_block;           // lanes per workgroup: kLaneBlock or kStepBlock (the lanes' LDS state is laid out by it)
    uint32_t fused;                // 1, model_step_kernel: the step's observe quantises the weight and publishes the tile's mass before it ends the wavefront;
                                   // 2 (kFusedQuad), model_step_kernel_quad: the step's observe ends the CALL (every later statement is dead), the kernel goes on
    FusedStep fs;
    ShardArgs sh;
};
typedef const LaunchArgs __attribute__((address_space(4))) * LaunchArgsPtr;
__device__ inline LaunchArgsPtr launch_args() { return (LaunchArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(); }

// What a lane's statements do write: counters, the running log-weight, the frontier flag -- in LDS, sized by the launch (dynamic:
// 36 bytes per lane under windowed replay, 64 + 1/8 otherwise), one workgroup's lanes side by side per field (a wavefront's access
// to a field is one conflict-free LDS instruction).  B = lanes per workgroup:
//   [ 0, 16B)  WinRec  windowed replay: four words per lane (below), a statement reads two of them
//   [16B, 24B) log_w   the step's incremental log-weight          [24B, 32B) carried  the log-weight the particle brought along
//   [32B, 36B) src     the lane's ancestor (its own index where the previous step did not resample)
//   [36B, 40B) pid     quad step only (model_step_kernel_quad): the particle the lane's CURRENT call of the model body runs
//   [36B, 64B) full replay: n_sample, n_observe, n_pred_real, n_pred_int, n_recorded, n_stored, done
//   [64B, ..)  full replay: per wavefront, the lanes that carry a particle
// Windowed replay (the model passed the Markov probe: statement counts do not depend on sampled values): every lane of a
// wavefront executes the same statements, so the counters are WAVE-UNIFORM -- packed words per lane, read back through the scalar
// unit (readfirstlane), and every test on them a scalar branch: a statement behind or before the live window costs one LDS round
// trip and a handful of scalar instructions.  The launch's own thresholds ride in the same 16 bytes (written once by begin_lane),
// so a dead statement waits for nothing but that one load -- no kernel-argument fetch behind it -- and each kind of statement
// touches ONE counter word and ONE threshold word (a CU has one scalar unit for its four SIMDs: the dead iterations of a model's
// loop are bound by scalar issue -- 65 instructions an iteration of hmm<N> cost 1.07 us per launch at 10^6 particles):
//   n_sample   samples executed                          lim_sample   first sample ordinal inside the window
//   n_other    bits 0..11 observes, 12..21 int predicts, 22..31 real predicts (kWinMaxObserves / kWinMaxPredicts: the host replays
//              the whole trace for models beyond them)
//   lim_other  bits 0..11 first_observe, 16..31 stop_after + 1 (0: run to completion)
// Four arrays of one word per lane (lane stride 4 bytes: conflict-free; the four words of a lane side by side put a wavefront's 64
// accesses to one of them on eight LDS banks: measured 25 % slower on the 100-observe linear-Gaussian model).
struct WinRec { uint32_t &n_sample, &n_other, &lim_sample, &lim_other; };
constexpr uint32_t kWObserve = 1u, kWPredInt = 1u << 12, kWPredReal = 1u << 22;
constexpr uint32_t kWinMaxObserves = 4095u, kWinMaxPredicts = 1023u;
__device__ inline uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// "some lane of the wavefront": one vector compare and a scalar branch.  Where the condition is a compile-time fact (the step kernels
// built for ONE step: cpprob/gpu.hpp, model_step_kernel_at) it is that fact -- a ballot of `true` is the execution mask, which the
// optimiser does not know to be non-zero, and everything behind the step's live observe would stay in the kernel as code nobody runs.
__device__ __forceinline__ bool wave_any(bool c)
{
    if (__builtin_constant_p(c)) return c;
    return __ballot(c) != 0ull;
}
__device__ inline char* lane_lds()
{
    extern __shared__ __attribute__((aligned(16))) char cpprob_lane_lds[];
    return cpprob_lane_lds;
}
__host__ __device__ inline size_t lane_lds_bytes(uint32_t lanes, bool windowed_only, bool quad = false) { return windowed_only ? (size_t)(quad ? 40 : 36) * lanes : (size_t)64 * lanes + lanes / 8 + 8; }
struct LaneState {                                   // (pointers to this lane's slots: address arithmetic on one scalar, B)
    double* log_w; double* carried; int32_t* src;
    uint32_t *n_sample, *n_observe, *n_pred_real, *n_pred_int, *n_recorded, *n_stored, *done;
    unsigned long long* active;
};
__device__ inline WinRec win_rec()
{
    uint32_t* p = reinterpret_cast<uint32_t*>(lane_lds()) + threadIdx.x;
    const uint32_t B = launch_args()->lane_block;
    return WinRec{p[0], p[B], p[2 * B], p[3 * B]};
}
__device__ inline double& lane_log_w() { return reinterpret_cast<double*>(lane_lds() + (size_t)16 * launch_args()->lane_block)[threadIdx.x]; }
__device__ inline double& lane_carried() { return reinterpret_cast<double*>(lane_lds() + (size_t)24 * launch_args()->lane_block)[threadIdx.x]; }
__device__ inline int32_t& lane_src() { return reinterpret_cast<int32_t*>(lane_lds() + (size_t)32 * launch_args()->lane_block)[threadIdx.x]; }
__device__ inline int32_t& lane_pid() { return reinterpret_cast<int32_t*>(lane_lds() + (size_t)36 * launch_args()->lane_block)[threadIdx.x]; }
constexpr uint32_t kFusedQuad = 2u;          // LaunchArgs::fused of model_step_kernel_quad
__device__ __forceinline__ LaneState lane_state()
{
    const size_t B = launch_args()->lane_block;
    char* p = lane_lds();
    const int l = threadIdx.x;
    LaneState s;
    s.log_w = reinterpret_cast<double*>(p + 16 * B) + l; s.carried = reinterpret_cast<double*>(p + 24 * B) + l;
    s.src = reinterpret_cast<int32_t*>(p + 32 * B) + l;
    uint32_t* u = reinterpret_cast<uint32_t*>(p + 36 * B) + l;
    s.n_sample = u; s.n_observe = u + B; s.n_pred_real = u + 2 * B; s.n_pred_int = u + 3 * B; s.n_recorded = u + 4 * B; s.n_stored = u + 5 * B; s.done = u + 6 * B;
    s.active = reinterpret_cast<unsigned long long*>(p + 64 * B) + l / 64;
    return s;
}
// = the particle id.  The fused step keeps every lane of its last workgroup alive (its wavefront reductions and barriers want all
// of them): lanes beyond the population redo the last particle -- same id, same variates, same stores -- and weigh nothing.
__device__ __forceinline__ int64_t lane_index()
{
    LaunchArgsPtr A = launch_args();
    if (A->fused == kFusedQuad) return (int64_t)lane_pid();           // (a lane runs the body for four particles, one after the other)
    const int64_t i = (int64_t)blockIdx.x * A->lane_block + threadIdx.x;
    return (A->fused && i >= A->n) ? A->n - 1 : i;
}

__device__ __forceinline__ void begin_lane(int32_t src, uint32_t n_stored, double carried)
{
    LaunchArgsPtr A = launch_args();
    lane_log_w() = 0.0; lane_carried() = carried; lane_src() = src;
    if (A->windowed) {
        const int32_t base = A->fresh_lo - (int32_t)A->win;
        const WinRec r = win_rec();
        r.n_sample = 0; r.n_other = 0;
        r.lim_sample = (uint32_t)(base > 0 ? base : 0);
        r.lim_other = (uint32_t)A->first_observe | ((uint32_t)(A->stop_after + 1) << 16);
        return;
    }
    const LaneState s = lane_state();
    *s.n_sample = 0; *s.n_observe = 0; *s.n_pred_real = 0; *s.n_pred_int = 0; *s.n_recorded = 0;
    *s.n_stored = n_stored; *s.done = 0;
    *s.active = __ballot(1);               // (every lane of the wavefront writes the same word)
}

// finish_trace() of one lane: the particle's log-weight (and, full replay, how many samples its trace holds)
__device__ __forceinline__ void finish_lane()
{
    LaunchArgsPtr A = launch_args();
    const int64_t i = lane_index();
    A->logw_out[i] = lane_carried() + lane_log_w();
    if (A->nstored_out) A->nstored_out[i] = (int32_t)*lane_state().n_recorded;
}

// 8-byte raw slots of the sample trace
template <class T> __device__ inline uint64_t to_raw(T v)
{
    if (std::is_floating_point<T>::value) { const double d = static_cast<double>(v); return __double_as_longlong(d); }
    return static_cast<uint64_t>(static_cast<int64_t>(v));
}
template <class T> __device__ inline T from_raw(uint64_t r)
{
    if (std::is_floating_point<T>::value) return static_cast<T>(__longlong_as_double(static_cast<long long>(r)));
    return static_cast<T>(static_cast<int64_t>(r));
}

// ---- variate generators: stand-ins for boost::random::X::operator()(get_rng()) -------------------------
template <class R>
__device__ inline R draw(const boost::random::normal_distribution<R>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    return static_cast<R>(d.mean() + d.sigma() * cph::draw_std_normal(seed, pid, j));
}
template <class I>
__device__ inline I draw(const boost::random::uniform_smallint<I>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    return static_cast<I>(d.a() + static_cast<I>(cph::smallint_from_word(cph::draw_word(seed, pid, j), 0, static_cast<uint64_t>(d.b() - d.a()))));
}
template <class I, class W>
__device__ inline I draw(const boost::random::discrete_distribution<I, W>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    const auto& w = d.weights();
    return static_cast<I>(cph::discrete_from_u_dyn(cph::u01_32(cph::draw_word(seed, pid, j)), w.p, static_cast<int>(w.n)));
}
template <class R>
__device__ inline R draw(const boost::random::uniform_real_distribution<R>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    return static_cast<R>(cph::draw_uniform_real(seed, pid, j, d.a(), d.b()));
}

template <class I, class R>
__device__ inline I draw(const boost::random::poisson_distribution<I, R>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    return static_cast<I>(cph::draw_poisson(seed, pid, j, static_cast<double>(d.mean())));
}

// ---- the fused SMC step (model_step_kernel, cpprob/gpu.hpp) ---------------------------------------------------------------------
// Shape of smc_step_fixed_kernel (csrc/step_fixed.hpp) with the model body in the middle: everything the prologue reads is
// addressed by the launch geometry and fetched at kernel entry in one round trip; the workgroup's first wavefront takes generation
// t-1's totals, decides (ESS), fixes the reference R_t and searches the hierarchy for the source blocks this workgroup's 256 outputs
// draw from; all four walk them; the lane then replays its ancestor's window and runs the step.  The step's observe statement --
// every wavefront of the workgroup arrives there: windowed replay is wave-uniform -- quantises the log-weight, reduces the
// workgroup's {mass, squares, maximum} and publishes them as its entry of generation t's hierarchy before it ends the wavefront.
// Geometry: the hierarchy's entries are 256-particle blocks, one per workgroup -- plain stores at level 0, fire-and-forget atomics
// above, so no workgroup waits for an atomic's return below 64^2 blocks -- and many small workgroups are resident per CU, so one's
// search and barriers run under the others' model bodies.  (Measured on the way here, hmm<16>, 10^6 particles, per launch: a
// 1024-lane workgroup per 1024-particle tile -- one resident per CU at the model's register count, every phase of its prologue
// exposed -- 42 us against the unfused launch's 13; four 256-lane workgroups adding into a shared tile entry and counting arrivals:
// 35 us, each workgroup's lifetime extended by two dependent atomic round trips.)
struct StepLocated { int c, c_last; uint64_t P; };           // first source block, last source block (>= n_blocks: not known), mass before block c
struct StepFound { StepLocated loc; double inv, ref; int resample; };
struct StepLds {
    int32_t slot[kStepBlock];                    // scatter slots of the workgroup's outputs
    uint64_t scan[2][kStepPass][kStepWaves];     // per-wave totals of the in-block scans, double-buffered across passes
    int32_t iscr[kStepWaves];
    uint64_t red[3 * kStepWaves];
    StepFound found;
    // a joint population's decision for this step (from the ranks' totals, or copied from the launch's arguments)
    double obound[kMaxShards + 1]; uint64_t before[kMaxShards]; double j_inv, j_ref; int j_resample;
};
__device__ __forceinline__ StepLds& step_lds()
{
    __shared__ __attribute__((aligned(16))) StepLds s_step;
    return s_step;
}

// Wavefront sums of 32-bit weights as TWO 16-bit halves: each half's sum over 64 lanes fits 22 bits, so a scan step is one
// v_add_u32_dpp per half (a 64-bit scan step is four instructions: the DPP moves carry no 64-bit add).
__device__ __forceinline__ uint64_t wave_incl_scan_q(uint32_t q)
{
    const uint32_t lo = cph::wave_incl_scan_u32(q & 0xffffu), hi = cph::wave_incl_scan_u32(q >> 16);
    return ((uint64_t)hi << 16) + lo;
}
__device__ __forceinline__ uint64_t wave_sum_q(uint32_t q)
{
    const uint32_t lo = cph::wave_sum_u32(q & 0xffffu), hi = cph::wave_sum_u32(q >> 16);
    return ((uint64_t)hi << 16) + lo;
}
using cph::wave_max_key;                                              // (the maximum of 64-bit keys as two 32-bit maxima: fixed_mass.hpp)

// What the search's probe reads: the hierarchy words of block cs = max(own - kStepProbeBack, 0)'s prefix and kStepProbe block entries.
struct StepProbeWords { uint64_t lvl[cph::kHierMaxLevels]; uint64_t we; };
__device__ __forceinline__ int step_probe_start(int own) { return own > kStepProbeBack ? own - kStepProbeBack : 0; }
__device__ __forceinline__ void step_probe_fetch(const cph::Hier& h, int own, int nb, StepProbeWords& w)
{
    const int cs = step_probe_start(own);
    cph::hier_prefix_fetch(h, cs, w.lvl);
    const int lane = threadIdx.x & 63;
    const int i = cs + (lane < kStepProbe ? lane : 0);
    w.we = h.lvl[0][i < nb ? i : nb - 1];
}

// The SEARCH (one wavefront; csrc/step_fixed.hpp: fixed_locate, with a wider probe over smaller entries): lane i evaluates the first
// output G owned by the sources from block cs + i on; the first source block of this workgroup's outputs is the last one whose G
// does not exceed the first output.  A miss descends the hierarchy from the top.
__device__ __forceinline__ StepLocated step_locate(const cph::FHier& f, const cph::FixedCdf& fc, int nb, double gj_first, double gj_last, int own, const StepProbeWords& pw)
{
    using namespace cph;
    const int lane = threadIdx.x & 63;
    const int cs = step_probe_start(own);
    const uint64_t Pc = fhier_prefix_sum(cs, pw.lvl);
    const uint64_t we = (lane < kStepProbe && cs + lane < nb) ? (pw.we & kMassMask) : 0ull;
    const uint64_t incl = wave_incl_scan_u64(we);
    const uint64_t x = Pc + incl - we;                                   // lanes 0..kStepProbe: the mass before block cs + lane
    const double gt = fc.g(x);
    const bool known = lane <= kStepProbe && cs + lane < nb;
    const unsigned long long m = __ballot(known && gt <= gj_first);
    const int i_lo = m ? (63 - __builtin_clzll(m)) : -1;
    StepLocated r;
    if ((i_lo >= 0 || cs == 0) && i_lo < kStepProbe) {
        const int i = i_lo < 0 ? 0 : i_lo;
        // (G is monotone in the block index: the blocks that start at or before the last output form a prefix of the window)
        const unsigned long long mh = __ballot(known && gt <= gj_last);
        const int i_hi = mh ? (63 - __builtin_clzll(mh)) : i;
        r.c = cs + i;
        r.P = read_lane_u64(x, i);
        r.c_last = (i_hi >= kStepProbe && cs + kStepProbe < nb) ? nb : cs + (i_hi > i ? i_hi : i);
        return r;
    }
    r.c = fhier_locate(f.h.table, f.h.copy, fc, gj_first, r.P);
    r.c_last = nb;
    return r;
}

// Inclusive prefix-max over the slots (visible on entry): within the wavefront by DPP, over the wavefronts in front of it by reading
// their slots directly -- no barrier of its own.
__device__ __forceinline__ int32_t step_prefix_max(StepLds& L)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t incl = cph::wave_incl_max_i32(L.slot[tid]);
    int32_t before = -1;
#pragma unroll
    for (int w2 = 0; w2 < kStepWaves - 1; ++w2) if (w2 < wv) before = max(before, L.slot[w2 * 64 + lane]);
    before = cph::wave_incl_max_i32(before);
    before = __builtin_amdgcn_readlane(before, 63);
    return max(incl, before);
}

// The WALK (csrc/step_fixed.hpp: fixed_walk, one source particle per lane and block, kStepPass blocks per pass): every source
// block that may own outputs of this workgroup rebuilds its prefix masses (one scan), each source with a non-empty range writes its
// index into the slot of its FIRST output, one prefix-max hands every output its ancestor.  Slots must hold -1 and be visible on
// entry.  Integers throughout: the same ancestors as any other tiling of the same masses.
struct StepFetched { uint32_t q[kStepFetch]; int first; };      // the weights of blocks first .. first + kStepFetch - 1 (lane's particle of each), fetched at entry
// One pass over ONE rank's sources: gj_first = the workgroup's first output (slots are relative to it), gj_last = the last output
// these sources may own (n_out = the workgroup's outputs: a source's range is cut there -- the lanes behind a shard's last particle
// redo that particle and must find ITS ancestor, not that of an output another workgroup owns); last_rank: the population's last source
// owns the rest; code_hi = the rank's bits of an ancestor's code.
// The final prefix-max (step_prefix_max, behind a barrier) is the caller's: a workgroup at a shard boundary makes several passes.
__device__ __forceinline__ void step_walk(const cph::FixedCdf& fc, const uint32_t* __restrict__ qprev, int64_t n, int nb, double gj_first, int n_out, double gj_last,
                                          bool last_rank, int32_t code_hi, const StepLocated& loc, const StepFetched& pf, StepLds& L)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int c = __builtin_amdgcn_readfirstlane(loc.c);
    const int c_last = __builtin_amdgcn_readfirstlane(loc.c_last);
    uint64_t P = loc.P;
    auto load_q = [&](int cc) -> uint32_t {                            // (uniform cc: a scalar switch over the fetched registers)
        const int k = cc - pf.first;
        uint32_t v = 0u;
        bool have = false;
#pragma unroll
        for (int j = 0; j < kStepFetch; ++j) if (k == j) { v = pf.q[j]; have = true; }
        if (!have && cc < nb) v = qprev[(int64_t)cc * kStepBlock + tid];
        return ((int64_t)cc * kStepBlock + tid < n && cc <= c_last) ? v : 0u;      // slots beyond the population (and blocks beyond the last source) weigh nothing
    };
    auto place = [&](double g) -> int { return (int)fmin(fmax(g - gj_first, 0.0), (double)n_out); };      // exact: integers
    uint32_t raw[kStepPass];
#pragma unroll
    for (int k = 0; k < kStepPass; ++k) raw[k] = load_q(c + k);
    int it = 0;
    while (c < nb && c <= c_last) {
        // (wave-uniform values -- the branches are made scalar so that the barrier inside the loop sits in uniform control flow)
        if (c_last >= nb && __builtin_amdgcn_readfirstlane(fc.g(P) > gj_last ? 1 : 0)) break;      // the last block is not known from the probe
        uint32_t raw_next[kStepPass];
#pragma unroll
        for (int k = 0; k < kStepPass; ++k) raw_next[k] = c + kStepPass <= c_last ? load_q(c + kStepPass + k) : 0u;
        uint64_t incl[kStepPass];
#pragma unroll
        for (int k = 0; k < kStepPass; ++k) { incl[k] = wave_incl_scan_q(raw[k]); if (lane == 63) L.scan[it & 1][k][wv] = incl[k]; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kStepPass; ++k) {
            uint64_t off = 0, tot = 0;
#pragma unroll
            for (int w2 = 0; w2 < kStepWaves; ++w2) { const uint64_t sw = L.scan[it & 1][k][w2]; if (w2 < wv) off += sw; tot += sw; }
            const uint64_t excl = P + off + incl[k] - raw[k];          // mass before this lane's particle
            const int64_t src = (int64_t)(c + k) * kStepBlock + tid;
            const int p_prev = place(fc.g(excl));
            int p = place(fc.g(excl + raw[k]));
            if (last_rank && src + 1 == n) p = place(fc.n_pop);         // the population's last source owns the rest
            if (p > p_prev && src < n) L.slot[p_prev] = (int32_t)src | code_hi;
            P += tot;
        }
        ++it;
#pragma unroll
        for (int k = 0; k < kStepPass; ++k) raw[k] = raw_next[k];
        c += kStepPass;
    }
}

// the hierarchy's view, field by field out of the kernel-argument segment (scalar loads)
__device__ __forceinline__ cph::FHier step_hier()
{
    LaunchArgsPtr A = launch_args();
    cph::FHier f;
#pragma unroll
    for (int l = 0; l < cph::kHierMaxLevels; ++l) { f.h.lvl[l] = A->fs.f.h.lvl[l]; f.h.n_ent[l] = A->fs.f.h.n_ent[l]; }
    f.h.n_lev = A->fs.f.h.n_lev; f.h.to_next = A->fs.f.h.to_next; f.h.to_clear = A->fs.f.h.to_clear;
    f.h.table = A->fs.f.h.table; f.h.copy = A->fs.f.h.copy;
    f.h.top = A->fs.f.h.top; f.h.top_n = A->fs.f.h.top_n; f.h.top_stride = A->fs.f.h.top_stride;
    f.q0 = A->fs.f.q0; f.m0 = A->fs.f.m0;
    return f;
}

// A joint population's resampling for this workgroup's outputs [g0, g0 + n_out): for every rank whose sources own some of them, the
// search in THAT rank's hierarchy and the walk over THAT rank's weights; the decision, the comb's scale and the ranks' offspring
// bounds come ready-made from the host (ShardArgs).  Returns the ancestor's code.
__device__ __forceinline__ int32_t step_resample_joint(const StepFetched& pf, StepLds& L)
{
    using namespace cph;
    LaunchArgsPtr A = launch_args();
    const int tid = threadIdx.x, wv = tid >> 6;
    const int world = A->sh.world, rank = A->sh.rank;
    const int bid = (int)blockIdx.x;
    const int64_t rem = A->n - (int64_t)bid * kStepBlock;
    const int n_out = rem < kStepBlock ? (int)rem : kStepBlock;
    const double g0 = (double)(A->sh.first[rank] + (uint64_t)bid * kStepBlock), g_end = g0 + (double)n_out;
    FixedCdf fc;
    fc.u0 = A->fs.u0; fc.n_pop = A->fs.n_pop; fc.inv = L.j_inv; fc.base = 0;
    for (int r = 0; r < world; ++r) {
        const double o_lo = L.obound[r], o_hi = L.obound[r + 1];
        if (o_hi <= g0 || o_lo >= g_end || o_hi <= o_lo) continue;     // (uniform) none of these outputs descends from rank r
        const ShardPeer* pr = A->sh.peers + r;
        const double lo = fmax(g0, o_lo), hi_last = fmin(g_end, o_hi) - 1.0;
        fc.base = L.before[r];
        const int nb_r = pr->nb;
        // the first output's ancestor is expected where the output itself sits in the population (equal shards: the same block of rank r)
        const double at = lo - (double)A->sh.first[r];
        const int guess = (int)fmin(fmax(floor(at * (1.0 / kStepBlock)), 0.0), (double)(nb_r - 1));
        if (wv == 0) {
            StepProbeWords pw;
            step_probe_fetch(pr->f.h, guess, nb_r, pw);
            const StepLocated loc = step_locate(pr->f, fc, nb_r, lo, hi_last, guess, pw);
            if (tid == 0) L.found.loc = loc;
        }
        __syncthreads();
        StepFetched none{};
        none.first = -(1 << 28);
        step_walk(fc, pr->q, pr->n, nb_r, g0, n_out, hi_last, r + 1 == world, r << kShardIndexBits, L.found.loc, r == rank ? pf : none, L);
        __syncthreads();                                               // (the next pass overwrites the hand-over)
    }
    return step_prefix_max(L);
}

// Prologue of model_step_kernel: the lane's ancestor and carried log-weight; bookkeeping of generation t-1 (one thread).
__device__ __forceinline__ void step_prologue()
{
    using namespace cph;
    LaunchArgsPtr A = launch_args();
    StepLds& L = step_lds();
    const int tid = threadIdx.x, wv = tid >> 6;
    const int nb = A->fs.nb, bid = (int)blockIdx.x, t = A->fs.t;
    const int64_t n = A->n;
    const int64_t i = lane_index();
    const FHier f = step_hier();
    StepFetched pf{};
    FTotWords tw{};
    StepProbeWords pw0{};
    double lw_carry = 0.0;
    pf.first = bid > kStepFetchBack ? bid - kStepFetchBack : 0;        // the sources of these outputs are expected around the workgroup's own block
    if (t > 0) {
        // (the weight arrays reach a tile beyond the last block: the loads need no bound)
        const uint32_t* qp = A->fs.q_prev + (int64_t)pf.first * kStepBlock + tid;
#pragma unroll
        for (int k = 0; k < kStepFetch; ++k) pf.q[k] = qp[k * kStepBlock];
        if (wv == 0) { ftot_fetch(f, tw); step_probe_fetch(f.h, bid, nb, pw0); }
        if (A->fs.may_carry) lw_carry = A->logw_in[i];                  // (speculative: a launch that resamples drops it)
    }
    int32_t anc = (int32_t)i;
    bool resample = false;
    const bool joint = A->sh.world != 0;
    if (joint) anc |= A->sh.rank << kShardIndexBits;                   // (an ancestor's code names its rank)
    if (t > 0 && joint) {
        L.slot[tid] = -1;
        if (wv == 0) {
            const int world = A->sh.world, lane = tid;
            if (A->sh.host_decided) {
                if (lane <= world) L.obound[lane] = A->sh.obound[lane];
                if (lane < world) L.before[lane] = A->sh.before[lane];
                if (lane == 0) { L.j_inv = A->sh.inv; L.j_ref = A->sh.ref; L.j_resample = A->sh.resample; }
            } else {
                // the generation's totals: every rank's 24 bytes, read where they lie (lane r = rank r)
                uint64_t s_r = 0, q_r = 0, m_r = 0;
                if (lane < world) {
                    // (system scope: the words lie in another rank's memory and were written since this device last read that line.
                    //  The RULE for every other peer read of this prologue -- the peers' hierarchy words, integer weights and carry
                    //  windows -- is the kernel boundary: they are written by launches that completed before this one began (stream waits
                    //  on the peers' events), this launch starts with invalidated L1 / L2 lines for memory it does not own, and they are
                    //  plain loads.  That rule has only ever been exercised between loopback ranks of ONE device, which is why
                    //  generic_joint_launcher (cpprob/gpu.hpp) runs islands across physical devices unless Options::joint_across_devices
                    //  asks for the joint form; the totals, read in the launch's first instructions, keep the stronger form.)
                    const uint64_t* tp = A->sh.peers[lane].totals;
                    s_r = __hip_atomic_load(tp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    q_r = __hip_atomic_load(tp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    m_r = __hip_atomic_load(tp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                const uint64_t incl = wave_incl_scan_u64(s_r);
                const uint64_t S = read_lane_u64(incl, 63), Q = wave_sum_u64(q_r);
                const double M = dkey_inv(wave_max_u64(m_r));
                const FixedDecision d = fixed_decide(S, Q, A->fs.n_pop, A->fs.ess_frac, true);
                const double r_t = fixed_reference(d.resample, M, A->fs.bound);
                FixedCdf fc;
                fc.u0 = A->fs.u0; fc.n_pop = A->fs.n_pop; fc.inv = d.inv; fc.base = 0;
                if (lane < world) { L.before[lane] = incl - s_r; L.obound[lane] = lane == 0 ? 0.0 : fc.g(incl - s_r); }
                if (lane == world) L.obound[lane] = A->fs.n_pop;
                if (lane == 0) { L.j_inv = d.inv; L.j_ref = r_t; L.j_resample = d.resample ? 1 : 0; }
                if (bid == 0 && lane == 0) {
                    // every rank keeps the run's books (the same numbers on each: they come from the same totals)
                    StepCtrl2* c = A->fs.ctrl;
                    const double ref_prev = c->ref_cur;
                    const double gap = d.W > 0.0 ? ref_prev - M : 1e300;
                    c->gap_max = t == 1 ? gap : fmax(c->gap_max, gap);
                    if (gap < 0.0) *A->overflow = 4;
                    else if (gap > A->fs.gap_limit && *A->overflow == 0) *A->overflow = 5;
                    A->fs.ess[t - 1] = d.ess;
                    A->fs.resampled[t - 1] = d.resample ? 1 : 0;
                    double lz = t == 1 ? 0.0 : *A->fs.log_z;
                    if (d.resample) lz += ref_prev + log(d.W / A->fs.n_pop);
                    *A->fs.log_z = lz;
                    c->ref_cur = r_t;
                }
            }
        }
        __syncthreads();
        resample = L.j_resample != 0;
        if (resample) {
            anc = step_resample_joint(pf, L);
            lw_carry = 0.0;
            if ((int64_t)bid * kStepBlock + tid < n) A->fs.anc_row[i] = anc;
        }
    } else if (t > 0) {
        L.slot[tid] = -1;
        const int64_t rem = n - (int64_t)bid * kStepBlock;
        const int n_out = rem < kStepBlock ? (int)rem : kStepBlock;
        const double gj_first = (double)((uint64_t)bid * kStepBlock), gj_last = gj_first + (double)(n_out - 1);
        FixedCdf fc;
        fc.u0 = A->fs.u0; fc.n_pop = A->fs.n_pop; fc.base = 0; fc.inv = 0.0;
        if (wv == 0) {
            // (where every step resamples -- a threshold above 1 -- the decision and the reference need the generation's mass alone:
            //  squares and maximum are the bookkeeping's, summed by workgroup 0 only: two wavefront reductions off everyone else's search)
            FTot tot;
            FixedDecision d;
            if (A->fs.may_carry || bid == 0) {
                tot = ftot_sum(f, tw);
                d = fixed_decide(tot.S, tot.Q, A->fs.n_pop, A->fs.ess_frac, true);        // (generation t-1 is never the last one here)
            } else {
                tot.S = wave_sum_u64((tid < f.h.top_n ? tw.s : 0ull) & kMassMask); tot.Q = 0; tot.M = 0.0;
                d.resample = tot.S > 0; d.inv = A->fs.n_pop / u64_to_double(tot.S); d.W = 0.0; d.Qd = 0.0; d.ess = 0.0;
            }
            fc.inv = d.inv;
            const double r_t = fixed_reference(d.resample, tot.M, A->fs.bound);
            if (bid == 0 && tid == 0) {
                // ESS (thesis p.37), decision and evidence of generation t-1, and how its weights sat against their reference
                StepCtrl2* c = A->fs.ctrl;
                const double ref_prev = c->ref_cur;
                const double gap = d.W > 0.0 ? ref_prev - tot.M : 1e300;
                c->gap_max = t == 1 ? gap : fmax(c->gap_max, gap);
                if (gap < 0.0) *A->overflow = 4;                          // the bound was not one: the host repeats the run on exact maxima
                else if (gap > A->fs.gap_limit && *A->overflow == 0) *A->overflow = 5;
                A->fs.ess[t - 1] = d.ess;
                A->fs.resampled[t - 1] = d.resample ? 1 : 0;
                double lz = t == 1 ? 0.0 : *A->fs.log_z;
                if (d.resample) lz += ref_prev + log(d.W / A->fs.n_pop);
                *A->fs.log_z = lz;
                if (!A->fs.exact_ref) c->ref_cur = r_t;
            }
            StepLocated loc{0, 0, 0};
            if (d.resample) loc = step_locate(f, fc, nb, gj_first, gj_last, bid, pw0);
            if (tid == 0) { L.found.loc = loc; L.found.inv = d.inv; L.found.ref = r_t; L.found.resample = d.resample ? 1 : 0; }
        }
        __syncthreads();                                               // slots reset, search results in place
        resample = L.found.resample != 0;
        if (resample) {
            fc.inv = L.found.inv;
            step_walk(fc, A->fs.q_prev, n, nb, gj_first, n_out, gj_last, true, 0, L.found.loc, pf, L);
            __syncthreads();
            anc = max(step_prefix_max(L), 0);
            lw_carry = 0.0;                                            // equal weights after resampling
            if ((int64_t)bid * kStepBlock + tid < n) A->fs.anc_row[i] = anc;
        }
    } else if (bid == 0 && tid == 0 && !A->fs.exact_ref && !(joint && A->sh.host_decided)) {
        A->fs.ctrl->ref_cur = A->fs.bound;                             // R_0 = B_0
    }
    begin_lane(anc, 0u, lw_carry);
}

// Epilogue of the fused step (every lane of the workgroup, in uniform control flow): observe #t's weight as an integer, the
// workgroup's totals as its entry of generation t's hierarchy.  exact_ref: the log-weights only -- maxima and masses follow
// in launches of their own, against the generation's exact maximum (cpprob_hip_generic_quantize).
__device__ __forceinline__ void step_epilogue()
{
    using namespace cph;
    LaunchArgsPtr A = launch_args();
    StepLds& L = step_lds();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int bid = (int)blockIdx.x;
    const int64_t i_raw = (int64_t)bid * kStepBlock + tid;
    const bool valid = i_raw < A->n;
    const int t = A->fs.t;
    double lw = lane_carried() + lane_log_w();
    if (!valid) lw = -INFINITY;                                        // padding lanes weigh nothing
    if (A->fs.exact_ref) {
        if (valid) A->logw_out[i_raw] = lw;
        return;
    }
    const double ref = t == 0 ? A->fs.bound : (A->sh.world ? L.j_ref : L.found.ref);
    const uint32_t q = fix_weight(lw, ref);
    const uint64_t s_w = wave_sum_q(q), q_w = wave_sum_q(cph::fix_square(q)), m_w = wave_max_key(dkey(lw));
    if (lane == 0) { L.red[wv] = s_w; L.red[kStepWaves + wv] = q_w; L.red[2 * kStepWaves + wv] = m_w; }
    __syncthreads();
    if (tid == 0) {
        uint64_t St = 0, Qt = 0, Mk = 0;
#pragma unroll
        for (int w2 = 0; w2 < kStepWaves; ++w2) { St += L.red[w2]; Qt += L.red[kStepWaves + w2]; Mk = umax64(Mk, L.red[2 * kStepWaves + w2]); }
        fhier_publish(step_hier(), bid, A->fs.nb, St, Qt, Mk);
    }
    A->fs.q_next[i_raw] = q;                                           // (the weight arrays are padded to whole tiles)
    if (valid && (A->fs.may_carry || t + 1 == A->fs.T)) A->logw_out[i_raw] = lw;
}

// what ends a lane's step: its log-weight to memory (and, fused step, its integer weight and the tile's totals)
__device__ __forceinline__ void finish_step()
{
    if (launch_args()->fused) step_epilogue(); else finish_lane();
}

// ---- the QUAD step (model_step_kernel_quad, cpprob/gpu.hpp): the model body for FOUR particles a lane behind ONE ancestor search ----
// Geometry of the library's fused step (csrc/step_fixed.hpp): a workgroup of 256 lanes owns a 1024-particle TILE -- one entry of the mass
// hierarchy, one search (cpprob/detail/fixed_mass.hpp: fixed_locate), one walk over the source tiles with four sources a lane
// (fixed_walk), one set of wavefront reductions and one publish -- and runs the model body four times: pass p of lane l runs particle
// tile * 1024 + 256 p + l, so that a pass's loads and stores (carried windows, predict columns, weights) are the contiguous rows they
// are in model_step_kernel.  What a pass leaves behind -- its integer weight, and the tile's running {mass, squares, maximum} -- rides
// in registers across the next call; the step's observe statement ends the CALL, not the wavefront: the rest of the body runs as dead
// statements (observe_impl), which is what this form pays for sharing the prologue: T - 1 dead iterations a call whatever the step.
struct QuadFound { cph::FLocated loc; double inv, ref; int resample; };
struct QuadLds {
    cph::FixedLds walk;                          // the tile's 1024 scatter slots -> ancestors, the walk's scan words
    QuadFound found;
    uint64_t red[3 * kStepWaves];
};
__device__ __forceinline__ QuadLds& quad_lds()
{
    __shared__ __attribute__((aligned(16))) QuadLds s_quad;
    return s_quad;
}
struct QuadStep {                                // a lane's registers across the four calls
    uint64_t s_l, q_l;                           // mass and squares of the passes so far
    double m_l;                                  // their largest log-weight
    double ref;                                  // R_t
    int64_t i_raw;                               // the current pass's slot (may lie beyond the population)
    bool resample;
};
constexpr int kQuadPasses = cph::kPPT;

// Prologue: generation t-1's decision and books (first wavefront), the tile's ancestors into the slots.
__device__ __forceinline__ void quad_prologue(QuadStep& qs)
{
    using namespace cph;
    LaunchArgsPtr A = launch_args();
    QuadLds& L = quad_lds();
    const int tid = threadIdx.x, wv = tid >> 6;
    const int nb = A->fs.nb, bid = (int)blockIdx.x, t = A->fs.t;
    const int64_t n = A->n;
    const FHier f = step_hier();
    qs.s_l = 0; qs.q_l = 0; qs.m_l = -INFINITY; qs.ref = A->fs.bound; qs.resample = false; qs.i_raw = 0;
    if (t == 0) {
        if (bid == 0 && tid == 0) A->fs.ctrl->ref_cur = A->fs.bound;   // R_0 = B_0
        return;
    }
    // everything the prologue reads is addressed by the launch geometry: the weights of the tile's own source tile and its neighbours,
    // the generation's totals and the probe's words, in one round trip
    const int64_t g0 = (int64_t)bid * kTile + (int64_t)tid * kPPT;
    const U4 zero = {0u, 0u, 0u, 0u};
    U4 q_0 = *reinterpret_cast<const U4*>(A->fs.q_prev + g0);
    U4 q_m1 = *reinterpret_cast<const U4*>(A->fs.q_prev + (bid > 0 ? g0 - kTile : g0));
    U4 q_p1 = *reinterpret_cast<const U4*>(A->fs.q_prev + (bid + 1 < nb ? g0 + kTile : g0));
    FTotWords tw{};
    ProbeWords pw0{};
    if (wv == 0) { ftot_fetch(f, tw); probe_fetch(f.h, bid, nb, pw0); }
    if (bid == 0) q_m1 = zero;
    if (bid + 1 >= nb) q_p1 = zero;
    int32_t neg[kPPT];
    lane_fill(neg, (int32_t)-1);
    store4(L.walk.slot, (int64_t)tid * kPPT, neg);
    const int64_t rem = n - (int64_t)bid * kTile;
    const int n_out = rem < kTile ? (int)rem : kTile;
    const double gj_first = (double)((uint64_t)bid * kTile);
    FixedCdf fc;
    fc.u0 = A->fs.u0; fc.n_pop = A->fs.n_pop; fc.base = 0; fc.inv = 0.0; fc.seed = 0; fc.draw = 0; fc.uid0 = 0;
    if (wv == 0) {
        // (as step_prologue: where every step resamples the decision needs the generation's mass alone)
        FTot tot;
        FixedDecision d;
        if (A->fs.may_carry || bid == 0) {
            tot = ftot_sum(f, tw);
            d = fixed_decide(tot.S, tot.Q, A->fs.n_pop, A->fs.ess_frac, true);
        } else {
            tot.S = wave_sum_u64((tid < f.h.top_n ? tw.s : 0ull) & kMassMask); tot.Q = 0; tot.M = 0.0;
            d.resample = tot.S > 0; d.inv = A->fs.n_pop / u64_to_double(tot.S); d.W = 0.0; d.Qd = 0.0; d.ess = 0.0;
        }
        fc.inv = d.inv;
        const double r_t = fixed_reference(d.resample, tot.M, A->fs.bound);
        if (bid == 0 && tid == 0) {
            StepCtrl2* c = A->fs.ctrl;
            const double ref_prev = c->ref_cur;
            const double gap = d.W > 0.0 ? ref_prev - tot.M : 1e300;
            c->gap_max = t == 1 ? gap : fmax(c->gap_max, gap);
            if (gap < 0.0) *A->overflow = 4;
            else if (gap > A->fs.gap_limit && *A->overflow == 0) *A->overflow = 5;
            A->fs.ess[t - 1] = d.ess;
            A->fs.resampled[t - 1] = d.resample ? 1 : 0;
            double lz = t == 1 ? 0.0 : *A->fs.log_z;
            if (d.resample) lz += ref_prev + log(d.W / A->fs.n_pop);
            *A->fs.log_z = lz;
            c->ref_cur = r_t;
        }
        FLocated loc{0, 0, 0};
        if (d.resample) loc = fixed_locate<kFixSystematic>(f, fc, nb, gj_first, n_out, bid, &pw0);
        if (tid == 0) { L.found.loc = loc; L.found.inv = d.inv; L.found.ref = r_t; L.found.resample = d.resample ? 1 : 0; }
    }
    __syncthreads();                                                   // slots reset, search results in place
    qs.resample = L.found.resample != 0;
    qs.ref = L.found.ref;
    if (qs.resample) {
        fc.inv = L.found.inv;
        int32_t anc[kPPT];
        fixed_walk<kFixSystematic>(fc, A->fs.q_prev, n, nb, true, gj_first, n_out, L.found.loc, bid, true, q_m1, q_0, q_p1, anc, L.walk);
#pragma unroll
        for (int k = 0; k < kPPT; ++k) anc[k] = max(anc[k], 0);       // padding outputs of the last tile
        // (every lane read its four slots in front of the walk's last barrier: they now hold the ancestors, for the passes to pick up)
        store4(L.walk.slot, (int64_t)tid * kPPT, anc);
        __syncthreads();
    }
}

// Pass p of a lane: which particle, its ancestor and carried log-weight; the lane's statement record starts over.
__device__ __forceinline__ void quad_begin(QuadStep& qs, int p)
{
    using namespace cph;
    LaunchArgsPtr A = launch_args();
    const int tid = threadIdx.x, bid = (int)blockIdx.x, t = A->fs.t;
    const int64_t n = A->n;
    const int64_t base = (int64_t)bid * kTile;
    qs.i_raw = base + (int64_t)p * kStepBlock + tid;
    // (slots beyond the population redo its last particle -- same id, same variates, same stores -- and weigh nothing)
    const int64_t i = qs.i_raw < n ? qs.i_raw : n - 1;
    int32_t anc = (int32_t)i;
    double lw_carry = 0.0;
    if (qs.resample) {
        anc = quad_lds().walk.slot[(int)(i - base)];
        if (qs.i_raw < n) A->fs.anc_row[i] = anc;
    } else if (t > 0 && A->fs.may_carry) lw_carry = A->logw_in[i];
    lane_pid() = (int32_t)i;
    begin_lane(anc, 0u, lw_carry);
}

// ... and what it leaves: the observe's weight as an integer, in memory and in the tile's running totals.
__device__ __forceinline__ void quad_end(QuadStep& qs)
{
    LaunchArgsPtr A = launch_args();
    const bool valid = qs.i_raw < A->n;
    double lw = lane_carried() + lane_log_w();
    if (!valid) lw = -INFINITY;
    const uint32_t q = cph::fix_weight(lw, qs.ref);
    qs.s_l += q; qs.q_l += cph::fix_square(q); qs.m_l = fmax(qs.m_l, lw);
    A->fs.q_next[qs.i_raw] = q;                                        // (the weight arrays are padded to whole tiles)
    if (valid && (A->fs.may_carry || A->fs.t + 1 == A->fs.T)) A->logw_out[qs.i_raw] = lw;
}

// Epilogue: the tile's totals as its entry of generation t's hierarchy.
__device__ __forceinline__ void quad_epilogue(const QuadStep& qs)
{
    using namespace cph;
    LaunchArgsPtr A = launch_args();
    QuadLds& L = quad_lds();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t s_w = wave_sum_u34(qs.s_l), q_w = wave_sum_u34(qs.q_l), m_w = wave_max_key(dkey(qs.m_l));      // (sums of four 32-bit terms)
    if (lane == 0) { L.red[wv] = s_w; L.red[kStepWaves + wv] = q_w; L.red[2 * kStepWaves + wv] = m_w; }
    __syncthreads();
    if (tid == 0) {
        uint64_t St = 0, Qt = 0, Mk = 0;
#pragma unroll
        for (int w2 = 0; w2 < kStepWaves; ++w2) { St += L.red[w2]; Qt += L.red[kStepWaves + w2]; Mk = umax64(Mk, L.red[2 * kStepWaves + w2]); }
        fhier_publish(step_hier(), (int)blockIdx.x, A->fs.nb, St, Qt, Mk);
    }
}

template <class Distribution>
__device__ __forceinline__ typename std::decay_t<Distribution>::result_type sample_impl(Distribution& distr)
{
    using R = typename std::decay_t<Distribution>::result_type;
    LaunchArgsPtr A = launch_args();
    if constexpr (is_dev_mvn<std::decay_t<Distribution>>::value) {
        // vector-valued sample (multivariate_normal.hpp:268-274): the components draw one after the other, each its own trace slot
        using T = typename R::value_type;
        const LaneState s = lane_state();
        R value;
        for (std::size_t i = 0; i < distr.size(); ++i) {
            const uint32_t j = (*s.n_sample)++;
            T v = T();
            if (!*s.done) {
                if (j < *s.n_stored && j < A->trace_cap) v = from_raw<T>(A->trace_in[(int64_t)j * A->ld + *s.src]);
                else v = static_cast<T>(distr.mean_at(i) + distr.sigma_at(i) * cph::draw_std_normal(A->seed, (uint64_t)lane_index() + A->pid0, (uint64_t)j));
                if (A->trace_out) {
                    if (j < A->trace_cap) A->trace_out[(int64_t)j * A->ld + lane_index()] = to_raw<T>(v);
                    else if (A->overflow) *A->overflow = 1;
                }
                *s.n_recorded = j + 1;
            }
            value.push_back(v);
        }
        return value;
    } else if constexpr (!std::is_arithmetic<R>::value) {
        // vector-valued statements own host containers: they reach the device through built-in kernels only
        // (cpprob::inference refuses to launch this path for them, host_engine.hpp)
        __builtin_trap();
    } else {
        if (A->windowed) {
            // (scalar control flow: see WinRec)
            const WinRec rec = win_rec();
            const uint32_t n_own = rec.n_sample;
            rec.n_sample = n_own + 1u;
            // (the test on the lanes' own words, made uniform by the ballot: one vector compare and a scalar branch -- the scalar
            //  unit, shared by the CU's four SIMDs, is what bounds the dead iterations; the ordinal goes through it on the live path only)
            if (wave_any(n_own < rec.lim_sample)) return R();     // older than the window: the step does not depend on it (host probe)
            const int32_t jj = (int32_t)wave_uniform(n_own);
            const int32_t base = A->fresh_lo - (int32_t)A->win;
            R v;
            if (jj < A->fresh_lo) {
                const int32_t src = lane_src();
                if (A->sh.world) {                                    // the ancestor's window lies in its rank's store
                    const ShardPeer* pr = A->sh.peers + (src >> kShardIndexBits);
                    v = from_raw<R>(pr->carry[(int64_t)(jj - base) * pr->n + (src & ((1 << kShardIndexBits) - 1))]);
                } else v = from_raw<R>(A->carry_in[(int64_t)(jj - base) * A->ld + src]);
            }
            else v = draw(distr, A->seed, (uint64_t)lane_index() + A->pid0, (uint64_t)jj);
            const int32_t out_base = A->next_fresh - (int32_t)A->win;
            if (A->carry_out && jj >= out_base && jj < A->next_fresh) A->carry_out[(int64_t)(jj - out_base) * A->ld + lane_index()] = to_raw<R>(v);
            return v;
        }
        const LaneState s = lane_state();
        const uint32_t j = (*s.n_sample)++;
        if (*s.done) return R();
        R value;
        if (j < *s.n_stored && j < A->trace_cap) value = from_raw<R>(A->trace_in[(int64_t)j * A->ld + *s.src]);   // (beyond the rows: the run is being repeated anyway)
        else value = draw(distr, A->seed, (uint64_t)lane_index() + A->pid0, (uint64_t)j);
        if (A->trace_out) {
            if (j < A->trace_cap) A->trace_out[(int64_t)j * A->ld + lane_index()] = to_raw<R>(value);
            else if (A->overflow) *A->overflow = 1;
        }
        *s.n_recorded = j + 1;
        return value;
    }
}

template <class Distribution, class X>
__device__ __forceinline__ void observe_impl(Distribution& distr, const X& x)
{
    constexpr bool vec = is_dev_mvn<std::decay_t<Distribution>>::value;
    if constexpr (!vec && !std::is_arithmetic<X>::value) {
        __builtin_trap();                                           // heap-backed vector types: compile the model's device view (cpprob/gpu.hpp)
    } else {
        // (vector-valued: ONE observe statement whose log-density is the sum over the components, utils_multivariate_normal.hpp:20-33)
        LaunchArgsPtr A = launch_args();
        if (A->windowed) {
            const WinRec rec = win_rec();
            const uint32_t own = rec.n_other, lim_own = rec.lim_other;
            rec.n_other = own + kWObserve;
            if (wave_any((own & 0xfffu) >= (lim_own & 0xfffu))) {
                const uint32_t w = wave_uniform(own), lim = wave_uniform(lim_own);
                const uint32_t m = w & 0xfffu;
                // (a lane whose counters differ from its wavefront's executed other statements: the counts DO depend on sampled values,
                //  the probe notwithstanding -- reported, and the host repeats the run with full replay; counters only grow, so the
                //  step's live observe sees whatever went apart before it)
                if (wave_any(own != w || rec.n_sample != wave_uniform(rec.n_sample)) && A->overflow) *A->overflow = 3;
                lane_log_w() += logpdf<std::decay_t<Distribution>>()(distr, x);     // StateInfer::increment_log_prob, state.cpp:212-223
                if (m + 1 == (lim >> 16)) {                                          // every lane of the wavefront is here
                    if (A->fused == kFusedQuad) {
                        // (the lane has more particles to run: every later statement of THIS call is made a dead one -- samples "older than the
                        //  window", observes and predicts "of an earlier step" -- and the body returns to the kernel's loop)
                        rec.lim_sample = 0xffffffffu; rec.lim_other = lim_own | 0xfffu;
                        return;
                    }
                    finish_step();
                    // The wavefront ends here.  In the step kernels (every branch on the way is a scalar one: uniform control flow) as the
                    // BUILTIN, which -- unlike an asm statement with a memory clobber -- lets the statement counters live in registers
                    // across the model's loop (linear_gaussian_1d<100>: 8.0 -> 7.0 ms).  model_kernel also compiles the full-replay
                    // statements, whose flow is not provably uniform: there the builtin becomes lane masking and the wavefront runs on
                    // with an empty mask (scalar loads from stale addresses: a memory fault at -O3) -- the instruction itself, then.
                    if (A->fused) { __builtin_amdgcn_endpgm(); __builtin_unreachable(); }
                    asm volatile("s_endpgm" ::: "memory");
                }
            }
            return;
        }
        const LaneState s = lane_state();
        const int32_t m = (int32_t)(*s.n_observe)++;
        if (!*s.done && m >= A->first_observe) {
            *s.log_w += logpdf<std::decay_t<Distribution>>()(distr, x);     // StateInfer::increment_log_prob, state.cpp:212-223
            if (m == A->stop_after) *s.done = 1;
        }
        // Every statement behind the frontier is a no-op: once ALL of the wavefront's lanes have passed it (models whose observes do
        // not depend on sampled values: all at this very statement) the wavefront is finished.  The test is wave-uniform (all lanes
        // or none enter the block); the exit is written as an instruction rather than the builtin, which the compiler would turn
        // into lane masking when it cannot prove the surrounding control flow uniform.
        if (A->stop_after >= 0) {
            const unsigned long long arrived = __ballot(*s.done != 0);
            if (__builtin_amdgcn_readfirstlane(arrived == *s.active ? 1 : 0)) { finish_lane(); asm volatile("s_endpgm" ::: "memory"); }
        }
    }
}

template <class T>
__device__ __forceinline__ void predict_impl(const T& x)
{
    using V = std::decay_t<T>;
    LaunchArgsPtr A = launch_args();
    if constexpr (std::is_integral<V>::value || std::is_floating_point<V>::value) {
        if (A->windowed) {
            constexpr bool is_int = std::is_integral<V>::value;
            const WinRec rec = win_rec();
            const uint32_t own = rec.n_other;
            rec.n_other = own + (is_int ? kWPredInt : kWPredReal);
            if (wave_any((own & 0xfffu) < (rec.lim_other & 0xfffu))) return;                    // an earlier step's hit: recorded by that step's launch
            const uint32_t w = wave_uniform(own);
            const uint32_t k = (w >> (is_int ? 12 : 22)) & 0x3ffu;
            if constexpr (is_int) {
                if (A->pred_int) { if (k < A->pred_int_cap) A->pred_int[(int64_t)k * A->ld + lane_index()] = static_cast<int32_t>(x); else if (A->overflow) *A->overflow = 2; }
            } else {
                if (A->pred_real) { if (k < A->pred_real_cap) A->pred_real[(int64_t)k * A->ld + lane_index()] = static_cast<double>(x); else if (A->overflow) *A->overflow = 2; }
            }
            return;
        }
    }
    const LaneState s = lane_state();
    if constexpr (std::is_integral<V>::value) {                         // state.hpp:312-318 -> predict_int_
        if (*s.done) return;
        const uint32_t k = (*s.n_pred_int)++;
        if (A->pred_int) {
            if (k < A->pred_int_cap) A->pred_int[(int64_t)k * A->ld + lane_index()] = static_cast<int32_t>(x);
            else if (A->overflow) *A->overflow = 2;                     // more predict hits than the dry run: data-dependent predicts
        }
    } else if constexpr (std::is_floating_point<V>::value) {            // state.hpp:320-326 -> predict_real_
        if (*s.done) return;
        const uint32_t k = (*s.n_pred_real)++;
        if (A->pred_real) {
            if (k < A->pred_real_cap) A->pred_real[(int64_t)k * A->ld + lane_index()] = static_cast<double>(x);
            else if (A->overflow) *A->overflow = 2;
        }
    } else if constexpr (is_dev_ndarray<V>::value) {                    // state.hpp:330-337: an NDArray goes to the real list; one column per component
        if (*s.done) return;
        for (std::size_t i = 0; i < x.size(); ++i) {
            const uint32_t k = (*s.n_pred_real)++;
            if (A->pred_real) {
                if (k < A->pred_real_cap) A->pred_real[(int64_t)k * A->ld + lane_index()] = static_cast<double>(x[i]);
                else if (A->overflow) *A->overflow = 2;
            }
        }
    } else {
        __builtin_trap();                                               // heap-backed vector types: compile the model's device view (cpprob/gpu.hpp)
    }
}

}  // namespace device
}  // namespace cpprob
#endif
