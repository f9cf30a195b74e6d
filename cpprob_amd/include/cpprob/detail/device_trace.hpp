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
constexpr int kStepBlock = 1024;             // ... of model_step_kernel (windowed replay with the resampling inside the launch): one 1024-particle
constexpr int kStepWaves = kStepBlock / 64;  //     tile of the mass hierarchy per workgroup, one particle per lane

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
    uint32_t fused;                // model_step_kernel: the step's observe quantises the weight and publishes the tile's mass before it ends the wavefront
    FusedStep fs;
};
typedef const LaunchArgs __attribute__((address_space(4))) * LaunchArgsPtr;
__device__ inline LaunchArgsPtr launch_args() { return (LaunchArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(); }

// What a lane's statements do write: counters, the running log-weight, the frontier flag -- in LDS, sized by the launch (dynamic:
// 36 bytes per lane under windowed replay, 64 + 1/8 otherwise), one workgroup's lanes side by side per field (a wavefront's access
// to a field is one conflict-free LDS instruction).  B = lanes per workgroup:
//   [ 0, 16B)  WinRec  windowed replay: {w, lim}, read by every statement in ONE 16-byte load
//   [16B, 24B) log_w   the step's incremental log-weight          [24B, 32B) carried  the log-weight the particle brought along
//   [32B, 36B) src     the lane's ancestor (its own index where the previous step did not resample)
//   [36B, 64B) full replay: n_sample, n_observe, n_pred_real, n_pred_int, n_recorded, n_stored, done
//   [64B, ..)  full replay: per wavefront, the lanes that carry a particle
// Windowed replay (the model passed the Markov probe: statement counts do not depend on sampled values): every lane of a
// wavefront executes the same statements, so the counters are WAVE-UNIFORM -- one packed word per lane, read back through the
// scalar unit (readfirstlane), and every test on it a scalar branch: a statement behind or before the live window costs one LDS
// round trip and a handful of scalar instructions.  The launch's own thresholds ride in the same 16 bytes (`lim`: written once
// by begin_lane), so a dead statement waits for nothing but that one load -- no kernel-argument fetch behind it.
//   w   bits 0..23 samples, 24..39 observes, 40..49 int predicts, 50..59 real predicts, 63 done
//   lim bits 0..23 first sample ordinal inside the window, 24..39 first_observe, 40..55 stop_after + 1 (0: run to completion)
struct alignas(16) WinRec { unsigned long long w, lim; };
constexpr unsigned long long kWSample = 1ull, kWObserve = 1ull << 24, kWPredInt = 1ull << 40, kWPredReal = 1ull << 50, kWDone = 1ull << 63;
__device__ inline unsigned long long wave_uniform(unsigned long long v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__device__ inline char* lane_lds()
{
    extern __shared__ __attribute__((aligned(16))) char cpprob_lane_lds[];
    return cpprob_lane_lds;
}
__host__ __device__ inline size_t lane_lds_bytes(uint32_t lanes, bool windowed_only) { return windowed_only ? (size_t)36 * lanes : (size_t)64 * lanes + lanes / 8 + 8; }
struct LaneState {                                   // (pointers to this lane's slots: address arithmetic on one scalar, B)
    WinRec* win; double* log_w; double* carried; int32_t* src;
    uint32_t *n_sample, *n_observe, *n_pred_real, *n_pred_int, *n_recorded, *n_stored, *done;
    unsigned long long* active;
};
__device__ inline WinRec& win_rec() { return reinterpret_cast<WinRec*>(lane_lds())[threadIdx.x]; }
__device__ inline double& lane_log_w() { return reinterpret_cast<double*>(lane_lds() + (size_t)16 * launch_args()->lane_block)[threadIdx.x]; }
__device__ inline double& lane_carried() { return reinterpret_cast<double*>(lane_lds() + (size_t)24 * launch_args()->lane_block)[threadIdx.x]; }
__device__ inline int32_t& lane_src() { return reinterpret_cast<int32_t*>(lane_lds() + (size_t)32 * launch_args()->lane_block)[threadIdx.x]; }
__device__ __forceinline__ LaneState lane_state()
{
    const size_t B = launch_args()->lane_block;
    char* p = lane_lds();
    const int l = threadIdx.x;
    LaneState s;
    s.win = reinterpret_cast<WinRec*>(p) + l;
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
    const int64_t i = (int64_t)blockIdx.x * A->lane_block + threadIdx.x;
    return (A->fused && i >= A->n) ? A->n - 1 : i;
}

__device__ __forceinline__ void begin_lane(int32_t src, uint32_t n_stored, double carried)
{
    LaunchArgsPtr A = launch_args();
    lane_log_w() = 0.0; lane_carried() = carried; lane_src() = src;
    if (A->windowed) {
        const int32_t base = A->fresh_lo - (int32_t)A->win;
        WinRec r;
        r.w = 0;
        r.lim = (unsigned long long)(uint32_t)(base > 0 ? base : 0) | ((unsigned long long)(uint32_t)A->first_observe << 24) |
                ((unsigned long long)(uint32_t)(A->stop_after + 1) << 40);
        win_rec() = r;
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
// t-1's totals, decides (ESS), fixes the reference R_t and searches the hierarchy for the source tiles this output tile draws from;
// all sixteen walk them (one particle per lane); the lane then replays its ancestor's window and runs the step.  The step's observe
// statement -- every wavefront of the workgroup arrives there: windowed replay is wave-uniform -- quantises the log-weight, reduces
// the tile's {mass, squares, maximum} and publishes them into generation t's hierarchy before it ends the wavefront.
struct StepFound { cph::FLocated loc; double inv, ref; int resample; };
struct StepLds {
    int32_t slot[kStepBlock];            // scatter slots of the output tile
    uint64_t scan[2][kStepWaves];        // per-wave totals of the in-tile scan, double-buffered across source tiles
    int32_t iscr[kStepWaves];
    uint64_t red[3 * kStepWaves];
    StepFound found;
};
__device__ inline StepLds& step_lds()
{
    __shared__ __attribute__((aligned(16))) StepLds s_step;
    return s_step;
}

// The WALK, one particle per lane (csrc/step_fixed.hpp: fixed_walk states it for four): every source tile that owns outputs of
// this tile rebuilds its prefix masses (one scan), each source with a non-empty range writes its index into the slot of its FIRST
// output, one prefix-max hands every output its ancestor.  Slots must hold -1 and be visible on entry.  Integers throughout: the
// same ancestors as any other tiling of the same masses.
__device__ __forceinline__ int32_t step_walk(const cph::FixedCdf& fc, const uint32_t* __restrict__ qprev, int64_t n, int nb, double gj_first, int n_out,
                                    const cph::FLocated& loc, int guess, uint32_t q_m1, uint32_t q_0, uint32_t q_p1, StepLds& L)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double gj_last = gj_first + (double)(n_out - 1);
    int c = __builtin_amdgcn_readfirstlane(loc.c), c_last = __builtin_amdgcn_readfirstlane(loc.c_last);
    uint64_t P = loc.P;
    auto load_q = [&](int cc) -> uint32_t {
        if (cc == guess) return q_0;
        if (cc == guess - 1) return q_m1;
        if (cc == guess + 1) return q_p1;
        return cc < nb ? qprev[(int64_t)cc * kStepBlock + tid] : 0u;
    };
    auto place = [&](double g) -> int { return (int)fmin(fmax(g - gj_first, 0.0), (double)kStepBlock); };      // exact: integers
    uint32_t raw = load_q(c);
    int it = 0;
    while (c < nb && c <= c_last) {
        // (wave-uniform values -- the branches are made scalar so that the barrier inside the loop sits in uniform control flow)
        if (c_last >= nb && __builtin_amdgcn_readfirstlane(fc.g(P) > gj_last ? 1 : 0)) break;      // the last tile is not known from the probe
        const uint32_t raw_next = c < c_last ? load_q(c + 1) : 0u;
        const bool edge = c == nb - 1;
        const int nvt = edge ? (int)(n - (int64_t)c * kStepBlock) : kStepBlock;     // valid particles of this tile (padding slots weigh 0)
        const uint64_t incl = cph::wave_incl_scan_u64((uint64_t)raw);
        if (lane == 63) L.scan[it & 1][wv] = incl;
        __syncthreads();
        const uint64_t s = lane < kStepWaves ? L.scan[it & 1][lane] : 0ull;
        const uint64_t off = cph::wave_sum_u64(lane < wv ? s : 0ull), tot = cph::wave_sum_u64(s);
        const uint64_t excl = P + off + incl - raw;                    // mass before this lane's particle
        const int p_prev = place(fc.g(excl));
        int p = place(fc.g(excl + raw));
        if (edge && tid + 1 == nvt) p = place(fc.n_pop);               // the population's last source owns the rest
        if (edge && tid + 1 > nvt) p = p_prev;                         // padding slots own nothing
        if (p > p_prev) L.slot[p_prev] = c * kStepBlock + tid;
        P += tot;
        ++it;
        raw = raw_next;
        ++c;
    }
    __syncthreads();
    const int32_t incl = cph::wave_incl_max_i32(L.slot[tid]);        // inclusive prefix-max over the slots
    if (lane == 63) L.iscr[wv] = incl;
    __syncthreads();
    int32_t before = (lane < kStepWaves && lane < wv) ? L.iscr[lane] : -1;
    before = cph::wave_incl_max_i32(before);
    before = __builtin_amdgcn_readlane(before, 63);
    return max(incl, before);
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
    uint32_t q_0 = 0u, q_m1 = 0u, q_p1 = 0u;
    FTotWords tw{};
    ProbeWords pw0{};
    double lw_carry = 0.0;
    const int guess = bid;                                             // the source tile this output tile is expected to start in
    if (t > 0) {
        const uint32_t* qp = A->fs.q_prev;
        const int64_t g0 = (int64_t)guess * kStepBlock + tid;
        q_0 = qp[g0];
        q_m1 = qp[guess > 0 ? g0 - kStepBlock : g0];
        q_p1 = qp[guess + 1 < nb ? g0 + kStepBlock : g0];
        if (wv == 0) { ftot_fetch(f, tw); probe_fetch(f.h, guess, nb, pw0); }
        if (A->fs.may_carry) lw_carry = A->logw_in[i];                  // (speculative: a launch that resamples drops it)
    }
    int32_t anc = (int32_t)i;
    bool resample = false;
    if (t > 0) {
        if (guess == 0) q_m1 = 0u;
        if (guess + 1 >= nb) q_p1 = 0u;
        L.slot[tid] = -1;
        const int64_t rem = n - (int64_t)bid * kStepBlock;
        const int n_out = rem < kStepBlock ? (int)rem : kStepBlock;
        const double gj_first = (double)((uint64_t)bid * kStepBlock);
        FixedCdf fc;
        fc.u0 = A->fs.u0; fc.n_pop = A->fs.n_pop; fc.base = 0; fc.inv = 0.0;
        if (wv == 0) {
            const FTot tot = ftot_sum(f, tw);
            const FixedDecision d = fixed_decide(tot.S, tot.Q, A->fs.n_pop, A->fs.ess_frac, true);        // (generation t-1 is never the last one here)
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
            FLocated loc{0, 0, 0};
            if (d.resample) loc = fixed_locate(f, fc, nb, gj_first, n_out, guess, &pw0);
            if (tid == 0) { L.found.loc = loc; L.found.inv = d.inv; L.found.ref = r_t; L.found.resample = d.resample ? 1 : 0; }
        }
        __syncthreads();                                               // slots reset, search results in place
        resample = L.found.resample != 0;
        if (resample) {
            fc.inv = L.found.inv;
            anc = step_walk(fc, A->fs.q_prev, n, nb, gj_first, n_out, L.found.loc, guess, q_m1, q_0, q_p1, L);
            anc = max(anc, 0);
            lw_carry = 0.0;                                            // equal weights after resampling
            if ((int64_t)bid * kStepBlock + tid < n) __builtin_nontemporal_store(anc, A->fs.anc_row + i);
        }
    } else if (bid == 0 && tid == 0 && !A->fs.exact_ref) {
        A->fs.ctrl->ref_cur = A->fs.bound;                             // R_0 = B_0
    }
    begin_lane(anc, 0u, lw_carry);
}

// Epilogue of the fused step (every lane of the workgroup, in uniform control flow): observe #t's weight as an integer, the
// tile's totals into generation t's hierarchy.  exact_ref: the tile's maximum only -- the masses follow in a launch of their own,
// against the generation's exact maximum (cpprob_hip_generic_quantize).
__device__ __forceinline__ void step_epilogue()
{
    using namespace cph;
    LaunchArgsPtr A = launch_args();
    StepLds& L = step_lds();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t i_raw = (int64_t)blockIdx.x * kStepBlock + tid;
    const bool valid = i_raw < A->n;
    const int t = A->fs.t;
    double lw = lane_carried() + lane_log_w();
    if (!valid) lw = -INFINITY;                                        // padding lanes weigh nothing
    const FHier f = step_hier();
    if (A->fs.exact_ref) {
        const uint64_t m_w = wave_max_u64(dkey(lw));
        if (lane == 0) L.red[wv] = m_w;
        __syncthreads();
        if (tid == 0) {
            uint64_t Mk = 0;
#pragma unroll
            for (int w = 0; w < kStepWaves; ++w) Mk = umax64(Mk, L.red[w]);
            FHier fn = f;                                              // the copy written
            for (int l = 0; l < kHierMaxLevels; ++l) fn.h.lvl[l] = f.h.lvl[l] + f.h.to_next;
            fn.m0 = f.m0 + f.h.to_next; fn.q0 = f.q0 + f.h.to_next; fn.h.to_clear = f.h.to_clear - f.h.to_next;
            bbf_publish_max(fn, (int)blockIdx.x, A->fs.nb, Mk);
        }
        if (valid) A->logw_out[i_raw] = lw;
        return;
    }
    const double ref = t == 0 ? A->fs.bound : L.found.ref;
    const uint32_t q = fix_weight(lw, ref);
    const uint64_t s_w = wave_sum_u64((uint64_t)q), q_w = wave_sum_u64((uint64_t)(q >> 16) * (uint64_t)(q >> 16)), m_w = wave_max_u64(dkey(lw));
    if (lane == 0) { L.red[wv] = s_w; L.red[kStepWaves + wv] = q_w; L.red[2 * kStepWaves + wv] = m_w; }
    __syncthreads();
    if (tid == 0) {
        uint64_t St = 0, Qt = 0, Mk = 0;
#pragma unroll
        for (int w = 0; w < kStepWaves; ++w) { St += L.red[w]; Qt += L.red[kStepWaves + w]; Mk = umax64(Mk, L.red[2 * kStepWaves + w]); }
        fhier_publish(f, (int)blockIdx.x, A->fs.nb, St, Qt, Mk);
    }
    A->fs.q_next[i_raw] = q;                                           // (the weight arrays are padded to whole tiles)
    if (valid && (A->fs.may_carry || t + 1 == A->fs.T)) A->logw_out[i_raw] = lw;
}

// what ends a lane's step: its log-weight to memory (and, fused step, its integer weight and the tile's totals)
__device__ __forceinline__ void finish_step()
{
    if (launch_args()->fused) step_epilogue(); else finish_lane();
}

template <class Distribution>
__device__ inline typename std::decay_t<Distribution>::result_type sample_impl(Distribution& distr)
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
            WinRec& rec = win_rec();
            const WinRec own = rec;
            const unsigned long long w = wave_uniform(own.w), lim = wave_uniform(own.lim);
            rec.w = w + kWSample;
            if (w & kWDone) return R();
            const int32_t jj = (int32_t)(w & 0xffffffu);
            if (jj < (int32_t)(lim & 0xffffffu)) return R();              // older than the window: the step does not depend on it (host probe)
            const int32_t base = A->fresh_lo - (int32_t)A->win;
            R v;
            if (jj < A->fresh_lo) v = from_raw<R>(A->carry_in[(int64_t)(jj - base) * A->ld + lane_src()]);
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
__device__ inline void observe_impl(Distribution& distr, const X& x)
{
    constexpr bool vec = is_dev_mvn<std::decay_t<Distribution>>::value;
    if constexpr (!vec && !std::is_arithmetic<X>::value) {
        __builtin_trap();                                           // heap-backed vector types: compile the model's device view (cpprob/gpu.hpp)
    } else {
        // (vector-valued: ONE observe statement whose log-density is the sum over the components, utils_multivariate_normal.hpp:20-33)
        LaunchArgsPtr A = launch_args();
        if (A->windowed) {
            WinRec& rec = win_rec();
            const WinRec own = rec;
            const unsigned long long w = wave_uniform(own.w), lim = wave_uniform(own.lim);
            rec.w = w + kWObserve;
            // (a lane whose counters differ from its wavefront's executed other statements: the counts DO depend on sampled values,
            //  the probe notwithstanding -- reported, and the host repeats the run with full replay)
            if (__ballot(own.w != w) != 0ull && A->overflow) *A->overflow = 3;
            const uint32_t m = (uint32_t)((w >> 24) & 0xffffu);
            if (!(w & kWDone) && m >= (uint32_t)((lim >> 24) & 0xffffu)) {
                lane_log_w() += logpdf<std::decay_t<Distribution>>()(distr, x);     // StateInfer::increment_log_prob, state.cpp:212-223
                if (m + 1 == (uint32_t)((lim >> 40) & 0xffffu)) { finish_step(); asm volatile("s_endpgm" ::: "memory"); }      // every lane of the wavefront is here
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
__device__ inline void predict_impl(const T& x)
{
    using V = std::decay_t<T>;
    LaunchArgsPtr A = launch_args();
    if constexpr (std::is_integral<V>::value || std::is_floating_point<V>::value) {
        if (A->windowed) {
            constexpr bool is_int = std::is_integral<V>::value;
            WinRec& rec = win_rec();
            const WinRec own = rec;
            const unsigned long long w = wave_uniform(own.w), lim = wave_uniform(own.lim);
            if (w & kWDone) return;
            rec.w = w + (is_int ? kWPredInt : kWPredReal);
            if ((uint32_t)((w >> 24) & 0xffffu) < (uint32_t)((lim >> 24) & 0xffffu)) return;           // an earlier step's hit: recorded by that step's launch
            const uint32_t k = (uint32_t)((w >> (is_int ? 40 : 50)) & 0x3ffu);
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
