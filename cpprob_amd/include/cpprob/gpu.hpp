// gpu.hpp -- turns an UNCHANGED CPProb model function into MI355X kernels (hipcc, C++17 only).
//
// A model translation unit looks like this:
//
//     #include "cpprob/gpu.hpp"
//     #pragma clang force_cuda_host_device begin
//     #include "models/models.hpp"                  // the reference's model source, untouched
//     #pragma clang force_cuda_host_device end
//     CPPROB_REGISTER_MODEL(models::hmm<16>);
//
// and from then on `cpprob::inference(StateType::sis | smc, models::hmm<16>, observes, n, file)` -- the
// reference's call, from any C++14 host code -- runs the model body on the GPU, one particle per lane.
//
//   SIS : one launch; every lane runs the model to completion (cpprob.hpp:194-201 for all i at once).
//   SMC : one launch per observe (trace replay): the lane re-runs the model from
//         the top, `sample` statements already executed by its ancestor return the stored values, the
//         next ones draw fresh, the step's observe adds the incremental weight and ends the lane.
//         Between launches one C-ABI building block (cpprob_hip_smc_bookkeep) normalises the weights, tests the ESS and
//         resamples ON THE DEVICE -- no host synchronisation inside a run; the stored sample values of the chosen
//         ancestors are carried along by the replay itself (each step rewrites the full trace), so the
//         final launch regenerates every predict of every surviving particle.
// Cost: O(T^2) statement executions for T observes -- the price of not editing the model; the built-in
// kernels (CPPROB_REGISTER_BUILTIN) are the O(T) fast path for the models they cover.
// Restriction: the number and order of observe / predict statements must not depend on sampled values.  The
// number of sample statements may (rejection-sampling loops): SIS runs them as they come; SMC keeps 4x the
// dry run's count of trace rows per particle and reports an error if a particle needs more.
#ifndef CPPROB_COMPAT_GPU_HPP
#define CPPROB_COMPAT_GPU_HPP
#if !defined(__HIPCC__) && !defined(__HIP__)
#error "cpprob/gpu.hpp needs hipcc (-std=c++17 --offload-arch=gfx950); host-only code includes cpprob/cpprob.hpp"
#endif

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <limits>
#include <thread>
#include <map>
#include <mutex>
#include <stdexcept>
#include <tuple>
#include <vector>

#include <boost/math/distributions/normal.hpp>

#include "cpprob/cpprob.hpp"
#include "cpprob/detail/device_vector.hpp"
#include "cpprob/distributions/multivariate_normal.hpp"
#include "cpprob/ndarray.hpp"

namespace cpprob {
namespace gpu {

using ModelKernelArgs = device::LaunchArgs;     // (cpprob/detail/device_trace.hpp: the statements read it from the kernel-argument segment)

// How the kernel reaches the model body: a function (by address, as a template argument) or a functor
// (by type, default-constructed on the lane -- e.g. the reference's models::Gauss<>, models.hpp:51-65).
template <class FP, FP F>
struct FunctionCaller {
    using observes_t = tuple_observes_t<FP>;
    CPPROB_HD __attribute__((always_inline)) static void call(const observes_t& obs) { call_f_tuple(F, obs); }
};
template <class Functor>
struct FunctorCaller {
    using observes_t = tuple_observes_t<Functor>;
    CPPROB_HD __attribute__((always_inline)) static void call(const observes_t& obs) { call_f_tuple(Functor{}, obs); }
};

template <class Caller, class Tuple>
__global__ __launch_bounds__(device::kLaneBlock) void model_kernel(ModelKernelArgs a, const Tuple* __restrict__ observes)
{
    // (what the statements may take for granted in this kernel: they read the launch's mode from the kernel-argument segment)
    { const uint32_t fused = device::launch_args()->fused, lanes = device::launch_args()->lane_block; __builtin_assume(fused == 0u); __builtin_assume(lanes == (uint32_t)device::kLaneBlock); }
    { const int32_t world = device::launch_args()->sh.world; __builtin_assume(world == 0); }
    const int64_t i = (int64_t)blockIdx.x * device::kLaneBlock + threadIdx.x;
    if (i >= a.n) return;
    const bool resampled = a.resampled_prev && *a.resampled_prev != 0;
    const int64_t src = (a.anc && resampled) ? (int64_t)a.anc[i] : i;
    const double carried = (!resampled && a.logw_in) ? a.logw_in[i] : 0.0;   // equal weights after resampling
    device::begin_lane((int32_t)src, a.nstored_in ? (uint32_t)a.nstored_in[src] : 0u, carried);
    Caller::call(*observes);                                      // the model body, cpprob.hpp:199
    device::finish_lane();                                        // finish_trace(): the particle's log_w_
}

// One SMC step of a model under windowed replay, the resampling INSIDE the launch (cpprob/detail/device_trace.hpp: step_prologue /
// step_epilogue; the shape of the built-in smc_step_fixed_kernel with the model body where its propagate4 / loglik sit): ancestors by
// the integer comb over generation t-1's fixed-point masses, the ancestor's window replayed, the step's statements, the new weight
// quantised and the workgroup's mass published -- one launch per observe instead of four.
// Two builds of it: FULL asks the compiler for eight wavefronts a SIMD (64 registers: 2048 workgroups resident, two passes of 3907 at
// 10^6 particles instead of three) and is used when that costs a few spilled registers; models whose body needs more (fp64 Box-Muller:
// ~90) run the build that takes what it needs (the host asks the runtime which is which: step_kernel_full).
// ... and a third for a shard of a joint population (JOINT: the statements' and the prologue's sharded branches exist only there).
template <class Caller, class Tuple, bool JOINT>
__device__ __forceinline__ void model_step_body(const Tuple* __restrict__ observes);
template <class Caller, class Tuple>
__global__ __launch_bounds__(device::kStepBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void model_step_kernel_full(ModelKernelArgs a, const Tuple* __restrict__ observes)
{
    (void)a;
    model_step_body<Caller, Tuple, false>(observes);
}
template <class Caller, class Tuple>
__global__ __launch_bounds__(device::kStepBlock) void model_step_kernel(ModelKernelArgs a, const Tuple* __restrict__ observes)
{
    (void)a;                                                      // (the statements read it where it lies: the kernel-argument segment)
    model_step_body<Caller, Tuple, false>(observes);
}
template <class Caller, class Tuple>
__global__ __launch_bounds__(device::kStepBlock) void model_step_kernel_joint(ModelKernelArgs a, const Tuple* __restrict__ observes)
{
    (void)a;
    model_step_body<Caller, Tuple, true>(observes);
}
template <class Caller, class Tuple, bool JOINT>
__device__ __forceinline__ void model_step_body(const Tuple* __restrict__ observes)
{
    {
        const uint32_t fused = device::launch_args()->fused, win = device::launch_args()->windowed, lanes = device::launch_args()->lane_block;
        __builtin_assume(fused == 1u); __builtin_assume(win == 1u); __builtin_assume(lanes == (uint32_t)device::kStepBlock);
        const int32_t world = device::launch_args()->sh.world;
        if (JOINT) __builtin_assume(world > 0); else __builtin_assume(world == 0);
    }
    device::step_prologue();
    Caller::call(*observes);
    device::step_epilogue();                                      // the run's last step: the body ran to completion
}

// Step kernels built for ONE step, or for the steps FROM an ordinal on (CPPROB_REGISTER_MODEL_STEPS, in a translation unit of its own).
// model_step_kernel re-runs the model's loop from its first statement at every launch and tests every statement against the launch's
// thresholds at run time: 0.4 - 0.7 us per dead iteration and launch, half of linear_gaussian_1d<100>'s run (O(T^2)).
// With the step's thresholds as compile-time FACTS (__builtin_assume on the launch's arguments: the host launches a kernel only where
// they hold) and the model's loop fully unrolled, the statement counters -- LDS words the optimiser forwards from store to load -- are
// constants, a dead statement's test folds, its value feeds only dead statements, and the iteration vanishes with its table loads:
//   EXACT   first_observe == FO, the window's ordinals == A FO .. A (FO + 1): the kernel is the step alone -- prologue, ONE live
//           iteration, the publishing observe -- no loop, no dead statement (hmm<16>: 2 200 instructions against 3 600 with a loop);
//   FROM    first_observe >= FO: the iterations below FO are gone, the others keep their run-time tests (models of many observes:
//           one build per step costs the compiler ~a minute each at 128 observes -- GVN's store-to-load forwarding over the unrolled
//           body -- so they get a build every T / 8 steps and run < T / 8 dead iterations a launch instead of t).
// A = sample statements per observe (the models.hpp state-space models: 1), WIN = the replay window the host's probe finds (1).
template <class Caller, class Tuple, int FO, int A, int WIN, bool EXACT, bool LAST>
__global__ __launch_bounds__(device::kStepBlock) void model_step_kernel_at(ModelKernelArgs a, const Tuple* __restrict__ observes)
{
    (void)a;
    {
        const int32_t fo = device::launch_args()->first_observe, sa = device::launch_args()->stop_after, fl = device::launch_args()->fresh_lo, nf = device::launch_args()->next_fresh;
        const uint32_t w = device::launch_args()->win;
        __builtin_assume(w == (uint32_t)WIN);
        if (EXACT) { __builtin_assume(fo == FO); __builtin_assume(sa == (LAST ? -1 : FO)); __builtin_assume(fl == A * FO); __builtin_assume(nf == A * (FO + 1)); }
        else { __builtin_assume(fo >= FO); __builtin_assume(fo < 4096); __builtin_assume(fl >= A * FO); __builtin_assume(fl < (1 << 20)); __builtin_assume(nf >= A * (FO + 1)); __builtin_assume(nf < (1 << 20)); }
    }
    model_step_body<Caller, Tuple, false>(observes);
}
template <class Caller, class Tuple, int FO, int A, int WIN, bool LAST>
__global__ void model_step_kernel_quad_at(ModelKernelArgs a, const Tuple* __restrict__ observes);      // (below, with the quad step)
// the builds a model unit registered for Caller (one table per model, shared by every translation unit of the library)
struct StepKernelEntry { int fo, a, win; bool exact, last; const void* fn; const void* quad_fn; };      // quad_fn: the four-a-lane build (exact builds only)
template <class Caller>
inline std::vector<StepKernelEntry>& step_kernels() { static std::vector<StepKernelEntry> v; return v; }
// the build to launch for a step whose thresholds are these (nullptr: model_step_kernel)
template <class Caller>
const void* step_kernel_for(const ModelKernelArgs& a, bool quad = false)
{
    const StepKernelEntry* best = nullptr;
    for (const StepKernelEntry& e : step_kernels<Caller>()) {
        if (e.win != (int)a.win) continue;
        if (e.exact) {
            if (a.first_observe == e.fo && a.stop_after == (e.last ? -1 : e.fo) && a.fresh_lo == e.a * e.fo && a.next_fresh == e.a * (e.fo + 1)) return quad ? e.quad_fn : e.fn;
        } else if (quad) {
            continue;
        } else if (a.first_observe >= e.fo && a.first_observe < 4096 && a.fresh_lo >= e.a * e.fo && a.next_fresh >= e.a * (e.fo + 1) && a.fresh_lo < (1 << 20) && a.next_fresh < (1 << 20)) {
            if (!best || e.fo > best->fo) best = &e;
        }
    }
    return best ? best->fn : nullptr;
}
constexpr int kStepExactMax = 32;               // observes up to which every step gets a build of its own
constexpr int kStepFromBuilds = 8;              // beyond: a build every ceil(T / 8) steps
template <class Caller, int T, int A, int WIN, int PART, int PARTS, int I>
void register_step_build()
{
    using Tuple = typename Caller::observes_t;
    constexpr bool exact = T <= kStepExactMax;
    constexpr int G = (T + kStepFromBuilds - 1) / kStepFromBuilds;
    constexpr int fo = exact ? I : I * G;
    if constexpr (I % PARTS == PART && fo < T && (exact || I > 0)) {
        constexpr bool last = exact && fo == T - 1;
        const void* quad_fn = nullptr;
        if constexpr (exact) quad_fn = reinterpret_cast<const void*>(&model_step_kernel_quad_at<Caller, Tuple, fo, A, WIN, last>);
        step_kernels<Caller>().push_back(StepKernelEntry{fo, A, WIN, exact, last, reinterpret_cast<const void*>(&model_step_kernel_at<Caller, Tuple, fo, A, WIN, exact, last>), quad_fn});
    }
}
template <class Caller, int T, int A, int WIN, int PART, int PARTS, int... I>
bool register_step_builds(std::integer_sequence<int, I...>)
{
    (register_step_build<Caller, T, A, WIN, PART, PARTS, I>(), ...);
    return true;
}

// The QUAD step (cpprob/detail/device_trace.hpp: quad_prologue ...): a workgroup owns a 1024-particle tile as the library's fused
// kernels do -- ONE search, one walk with four sources a lane, one publish -- and every lane runs the model body four times.
template <class Caller, class Tuple>
__global__ __launch_bounds__(device::kStepBlock) void model_step_kernel_quad(ModelKernelArgs a, const Tuple* __restrict__ observes)
{
    (void)a;
    {
        const uint32_t fused = device::launch_args()->fused, win = device::launch_args()->windowed, lanes = device::launch_args()->lane_block;
        __builtin_assume(fused == device::kFusedQuad); __builtin_assume(win == 1u); __builtin_assume(lanes == (uint32_t)device::kStepBlock);
        const int32_t world = device::launch_args()->sh.world;
        __builtin_assume(world == 0);
    }
    device::QuadStep qs;
    device::quad_prologue(qs);
#pragma unroll 1
    for (int p = 0; p < device::kQuadPasses; ++p) {
        device::quad_begin(qs, p);
        Caller::call(*observes);
        device::quad_end(qs);
    }
    device::quad_epilogue(qs);
}

// ... built for ONE step (model_step_kernel_at, EXACT): the call of the model body is then the step's live iteration alone -- the dead
// iterations in front of it fold away, and so do those BEHIND it (the step's observe ends the CALL by making every later statement a
// dead one: store-to-load forwarding turns that into compile-time facts too), which this form, unlike the one-a-lane step, used to run.
template <class Caller, class Tuple, int FO, int A, int WIN, bool LAST>
__global__ __launch_bounds__(device::kStepBlock) void model_step_kernel_quad_at(ModelKernelArgs a, const Tuple* __restrict__ observes)
{
    (void)a;
    {
        const uint32_t fused = device::launch_args()->fused, win = device::launch_args()->windowed, lanes = device::launch_args()->lane_block;
        __builtin_assume(fused == device::kFusedQuad); __builtin_assume(win == 1u); __builtin_assume(lanes == (uint32_t)device::kStepBlock);
        const int32_t world = device::launch_args()->sh.world;
        __builtin_assume(world == 0);
        const int32_t fo = device::launch_args()->first_observe, sa = device::launch_args()->stop_after, fl = device::launch_args()->fresh_lo, nf = device::launch_args()->next_fresh;
        const uint32_t w = device::launch_args()->win;
        __builtin_assume(w == (uint32_t)WIN); __builtin_assume(fo == FO); __builtin_assume(sa == (LAST ? -1 : FO)); __builtin_assume(fl == A * FO); __builtin_assume(nf == A * (FO + 1));
    }
    device::QuadStep qs;
    device::quad_prologue(qs);
#pragma unroll 1
    for (int p = 0; p < device::kQuadPasses; ++p) {
        device::quad_begin(qs, p);
        Caller::call(*observes);
        device::quad_end(qs);
    }
    device::quad_epilogue(qs);
}

// which build of the step kernel this model runs: the eight-wavefront build unless it spills more than a handful of registers
template <class Caller, class Tuple>
bool step_kernel_full()
{
    static const bool full = [] {
        hipFuncAttributes fa{};
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&model_step_kernel_full<Caller, Tuple>)) != hipSuccess) return false;
        return fa.localSizeBytes <= 96;                           // (scratch bytes per lane: spilled registers)
    }();
    return full;
}

template <class Tuple> struct observes_bytewise_copyable;
template <class... A> struct observes_bytewise_copyable<std::tuple<A...>>
{
    static constexpr bool value = (std::is_trivially_copyable<A>::value && ...);
};

inline void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// ---- a context and ONE block of device memory per device, kept across cpprob::inference calls ------------------------------------
// The reference's call is `inference(...)`, again and again (src/main.cpp:96-100): a context, its stream and every buffer of a run
// created per call cost more than the run (hipMalloc of ~20 buffers: milliseconds).  A workspace is leased for the length of an
// attempt; the block grows to the largest run seen and is carved anew by every attempt.  Workspaces live until
// cpprob::gpu::release_workspaces() (or the process ends: they are deliberately not torn down by static destructors, which run when
// the HIP runtime may already be gone).
struct Workspace {
    int device;
    Context ctx;
    char* block = nullptr; std::size_t cap = 0;
    explicit Workspace(int d) : device(d), ctx(d) {}
    ~Workspace() { if (block) (void)hipFree(block); }
    bool reserve(std::size_t bytes)                                   // true: (re)allocated
    {
        if (bytes <= cap) return false;
        hip_check(hipSetDevice(device), "hipSetDevice");
        if (block) { hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize"); (void)hipFree(block); block = nullptr; cap = 0; }
        const std::size_t want = bytes + bytes / 8;
        if (hipMalloc(reinterpret_cast<void**>(&block), want) != hipSuccess) throw std::runtime_error("hipMalloc of " + std::to_string(want) + " bytes failed");
        cap = want;
        return true;
    }
};
struct WorkspacePool { std::mutex mu; std::vector<std::unique_ptr<Workspace>> idle; };
inline WorkspacePool& workspace_pool() { static WorkspacePool* p = new WorkspacePool; return *p; }
inline void release_workspaces()
{
    WorkspacePool& p = workspace_pool();
    std::lock_guard<std::mutex> lock(p.mu);
    p.idle.clear();
}
class WorkspaceLease {
public:
    explicit WorkspaceLease(int device)
    {
        WorkspacePool& p = workspace_pool();
        {
            std::lock_guard<std::mutex> lock(p.mu);
            for (auto it = p.idle.begin(); it != p.idle.end(); ++it)
                if ((*it)->device == device) { w_ = std::move(*it); p.idle.erase(it); break; }
        }
        if (!w_) { w_.reset(new Workspace(device)); fresh_ = true; }
    }
    ~WorkspaceLease()
    {
        WorkspacePool& p = workspace_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        p.idle.push_back(std::move(w_));
    }
    Workspace& operator*() const { return *w_; }
    Workspace* operator->() const { return w_.get(); }
    bool fresh() const { return fresh_; }
private:
    std::unique_ptr<Workspace> w_;
    bool fresh_ = false;
};
// the attempt's buffers: registered with their sizes, then carved out of the workspace's block (256-byte aligned)
class Carver {
public:
    template <class T> void add(T** p, std::size_t count) { items_.push_back(Item{reinterpret_cast<void**>(p), count * sizeof(T)}); }
    bool commit(Workspace& w)
    {
        std::size_t total = 0;
        for (const Item& it : items_) total += (it.bytes + 255) / 256 * 256;
        const bool grown = w.reserve(total ? total : 256);
        std::size_t off = 0;
        for (const Item& it : items_) { *it.p = it.bytes ? w.block + off : nullptr; off += (it.bytes + 255) / 256 * 256; }
        return grown;
    }
private:
    struct Item { void** p; std::size_t bytes; };
    std::vector<Item> items_;
};

// how an SMC attempt resamples
enum class StepForm {
    unfused,        // model launch + cpprob_hip_smc_bookkeep(_fixed): full replay, the other resamplers, the Markov pilot
    fused_bounded,  // model_step_kernel, fixed-point weights against the dry run's per-observe bounds: ONE launch per observe
    fused_exact,    // model_step_kernel + cpprob_hip_generic_quantize (maximum pass, masses) against the generation's exact maximum: three launches per observe
    fused_quad      // model_step_kernel_quad: fused_bounded with four particles a lane behind one ancestor search (1024-particle tiles)
};
constexpr double kFixGapLimit = 6.0;            // nats a generation's heaviest particle may sit below its reference (csrc/step_fixed.hpp: the same contract)

// the launch's view of the library's mass hierarchy (include/cpprob_hip.h: cpprob_hip_generic_layout): copy `read`, where the copy
// written / the copy cleared sit relative to it
inline cph::FHier fused_view(const cpprob_hip_generic_layout& L, int read, int next, int clear)
{
    cph::FHier f{};
    auto lvl = [&](int copy, int l) { return L.hier + (std::size_t)copy * L.per_copy + L.lvl_off[l < L.n_lev ? l : 0]; };
    for (int l = 0; l < cph::kHierMaxLevels; ++l) { f.h.lvl[l] = lvl(read, l); f.h.n_ent[l] = L.n_ent[l]; }
    f.h.n_lev = L.n_lev; f.h.table = static_cast<const cph::HierTable*>(L.table); f.h.copy = read;
    const int top = L.n_lev - 1;
    f.h.top = lvl(read, top); f.h.top_n = L.n_ent[top]; f.h.top_stride = top == 0 ? 1 : cph::kHierStride;
    f.h.to_next = (int64_t)(next - read) * (int64_t)L.per_copy; f.h.to_clear = (int64_t)(clear - read) * (int64_t)L.per_copy;
    f.q0 = L.hier + (std::size_t)read * L.per_copy + L.q0_off;
    f.m0 = L.hier + (std::size_t)read * L.per_copy + L.m0_off;
    return f;
}

// One attempt with S trace rows per particle; returns 0, or -- nothing in res is valid then -- 1 when some particle needed more rows,
// 3 when the lanes of a wavefront executed different statements under windowed replay (the model's statement counts do depend on
// sampled values: the caller repeats the run with full replay), 4 when a generation's weights did not fit the reference they were
// taken against (fused_bounded: a bound that was not one, or one so loose that the integers lost their bits: the caller repeats the
// run in the fused_exact form).
template <class Caller>
int generic_attempt(StateType algorithm, const void* observes_v, std::size_t n, const detail::TraceStructure& st, const Options& opt,
                     Result& res, HostStore* store, const std::size_t S, StepForm form)
{
    using Tuple = typename Caller::observes_t;
    // (std::tuple itself is not trivially copyable in libstdc++ even when every element is: check the elements)
    static_assert(observes_bytewise_copyable<Tuple>::value,
                  "CPPROB_REGISTER_MODEL: the observes tuple is copied to the device bytewise, so every model argument must be trivially "
                  "copyable (arithmetic types, std::array of them); models with std::vector / NDArray arguments own host heap memory");
    const auto t_setup = std::chrono::steady_clock::now();
    // (what an earlier attempt of the same call reported is not this attempt's: a refuted window must not outlive its refutation)
    res.replay_window = -1; res.step_ess.clear(); res.n_resampled = 0; res.step_builds_used = 0;
    WorkspaceLease ws(opt.device);
    Context& ctx = ws->ctx;
    hip_check(hipSetDevice(opt.device), "hipSetDevice");
    hipStream_t stream = static_cast<hipStream_t>(cpprob_hip_stream(ctx.get()));
    const bool smc = algorithm == StateType::smc;
    // (SIS without a dump: the predict columns sit a whole number of 1024-particle tiles apart, so the statistics pass reads them where they
    //  lie -- cpprob_hip_weighted_*_columns copies columns of any other stride into padded scratch first: 80 MB each way at 10^7 particles)
    const int64_t ld = (!smc && !store) ? (int64_t)((n + cph::kTile - 1) / cph::kTile * cph::kTile) : (int64_t)n;
    const size_t n_real = st.real_rows(), n_int = st.int_ids.size();        // a vector-valued real predict owns one column per component
    const int T = (int)st.n_observe;
    const bool windowed = smc && st.window >= 0;
    const uint32_t w = (uint32_t)std::max(1, st.window);
    if (!(windowed && opt.resampler == CPPROB_HIP_RESAMPLE_SYSTEMATIC && n <= (std::size_t(64) * 64 * 64 * device::kStepBlock) && T > 0)) form = StepForm::unfused;
    const bool fused = form != StepForm::unfused;
    const bool quad = form == StepForm::fused_quad;

    // what the host reads when the run is over, side by side -- one copy: [T] step ESS, log evidence | [T] resampling decisions, flag word
    const size_t tail_doubles = (size_t)T + 1, tail_ints = (size_t)T + 1, tail_bytes = tail_doubles * sizeof(double) + tail_ints * sizeof(int32_t);
    Tuple* d_obs = nullptr;
    double *d_real = nullptr, *d_logw0 = nullptr, *d_logw1 = nullptr, *d_real_gen = nullptr;
    int32_t *d_int = nullptr, *d_anc = nullptr, *d_ns0 = nullptr, *d_ns1 = nullptr, *d_anc_all = nullptr, *d_int_gen = nullptr;
    uint64_t *d_tr0 = nullptr, *d_tr1 = nullptr, *d_c0 = nullptr, *d_c1 = nullptr;
    unsigned char* d_tail = nullptr;
    Carver carve;
    carve.add(&d_obs, 1); carve.add(&d_tail, tail_bytes);
    carve.add(&d_real, n_real * (size_t)ld); carve.add(&d_logw0, n); carve.add(&d_logw1, smc ? n : 0);
    carve.add(&d_int, n_int * (size_t)ld); carve.add(&d_anc, smc && !fused ? n : 0);
    carve.add(&d_ns0, smc && !windowed ? n : 0); carve.add(&d_ns1, smc && !windowed ? n : 0);
    carve.add(&d_tr0, smc && !windowed ? S * n : 0); carve.add(&d_tr1, smc && !windowed ? S * n : 0);
    // (windowed replay's buffers)
    carve.add(&d_c0, windowed ? (size_t)w * n : 0); carve.add(&d_c1, windowed ? (size_t)w * n : 0);
    carve.add(&d_anc_all, windowed ? (size_t)T * n : 0);
    carve.add(&d_real_gen, windowed ? n_real * n : 0); carve.add(&d_int_gen, windowed ? n_int * n : 0);
    const bool grown = carve.commit(*ws);
    hip_check(hipMemcpyAsync(d_obs, observes_v, sizeof(Tuple), hipMemcpyHostToDevice, stream), "copy observes");
    if ((size_t)ld > n) {                                           // the columns' slots behind n: finite (they weigh nothing in the statistics pass)
        if (n_real) hip_check(hipMemset2DAsync(d_real + n, (size_t)ld * sizeof(double), 0, ((size_t)ld - n) * sizeof(double), n_real, stream), "hipMemset2DAsync");
        if (n_int) hip_check(hipMemset2DAsync(d_int + n, (size_t)ld * sizeof(int32_t), 0, ((size_t)ld - n) * sizeof(int32_t), n_int, stream), "hipMemset2DAsync");
    }

    double* const d_ess_p = reinterpret_cast<double*>(d_tail);
    double* const d_logz_p = d_ess_p + T;
    int32_t* const d_res_p = reinterpret_cast<int32_t*>(d_tail + tail_doubles * sizeof(double));
    int32_t* const d_overflow_p = d_res_p + T;
    hip_check(hipMemsetAsync(d_tail, 0, tail_bytes, stream), "hipMemsetAsync");   // on the launches' own (non-blocking) stream
    std::vector<unsigned char> h_tail(tail_bytes);
    auto read_tail = [&]() {
        hip_check(hipMemcpyAsync(h_tail.data(), d_tail, tail_bytes, hipMemcpyDeviceToHost, stream), "copy the run's tail");
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    };
    const double* const h_ess_p = reinterpret_cast<const double*>(h_tail.data());
    const int32_t* const h_res_p = reinterpret_cast<const int32_t*>(h_tail.data() + tail_doubles * sizeof(double));
    double* logw[2] = {d_logw0, d_logw1};
    int32_t* ns[2] = {d_ns0, d_ns1};
    uint64_t* tr[2] = {d_tr0, d_tr1};
    const dim3 grid((unsigned)((n + device::kLaneBlock - 1) / device::kLaneBlock)), block(device::kLaneBlock);
    const size_t lane_lds = device::lane_lds_bytes(device::kLaneBlock, false);

    ModelKernelArgs a{};
    a.n = (int64_t)n; a.ld = ld; a.seed = opt.seed; a.trace_cap = (uint32_t)S; a.overflow = d_overflow_p; a.pid0 = opt.particle_offset;
    a.pred_real_cap = (uint32_t)n_real; a.pred_int_cap = (uint32_t)n_int; a.lane_block = device::kLaneBlock;
    // SMC bookkeeping between two launches of the model body, on the device: every resampler runs on fixed-point weights (integer
    // masses: three short launches, cpprob_amd/csrc/bookkeep_fixed.hpp; multinomial: the strata form, its counts in front); beyond
    // 2^28 particles the floating-point CDF
    auto bookkeep = [&](const double* lw, int t, bool last, int32_t* anc_out, double* ess, int32_t* resd, double* logz) {
        if (n <= (std::size_t(1) << 28))
            ctx.check(cpprob_hip_smc_bookkeep_fixed_rs(ctx.get(), opt.resampler, lw, n, opt.seed, t, last ? 1 : 0, opt.ess_threshold, ess, resd, logz, anc_out), "cpprob_hip_smc_bookkeep_fixed_rs");
        else
            ctx.check(cpprob_hip_smc_bookkeep(ctx.get(), opt.resampler, lw, n, opt.seed, t, last ? 1 : 0, opt.ess_threshold, ess, resd, logz, anc_out), "cpprob_hip_smc_bookkeep");
    };
    cpprob_hip_generic_layout lay{};
    if (fused) {
        // (sizes at the first call; clears are stream-ordered)
        if (quad) ctx.check(cpprob_hip_generic_begin_tiles(ctx.get(), n, &lay), "cpprob_hip_generic_begin_tiles");
        else ctx.check(cpprob_hip_generic_begin(ctx.get(), n, &lay), "cpprob_hip_generic_begin");
    } else if (smc && T > 0) {
        // The context sizes its own scratch (hierarchy of sums, integer weights, normalisation partials) at the first call that needs it:
        // one bookkeeping pass and one normalisation over zeroed log-weights before the clock starts -- allocation, like the buffers above.
        // (Their outputs are overwritten: step 0 starts the evidence, the flag word is cleared again.)
        hip_check(hipMemsetAsync(d_logw0, 0, n * sizeof(double), stream), "hipMemsetAsync");
        bookkeep(d_logw0, 0, true, d_anc, d_ess_p, d_res_p, d_logz_p);
        double warm[3];
        ctx.check(cpprob_hip_logsumexp_ess(ctx.get(), d_logw0, n, warm), "cpprob_hip_logsumexp_ess");      // (synchronises)
        hip_check(hipMemsetAsync(d_tail, 0, tail_bytes, stream), "hipMemsetAsync");
    }
    if (windowed && !store && T > 0) {                             // (the read-out's table of which record belongs to which generation: known from the dry run)
        std::vector<int32_t> g;
        for (int s2 : (n_real ? st.real_row_step : st.int_hit_step)) g.push_back(std::min(s2, T - 1));
        if (!g.empty()) ctx.check(cpprob_hip_lineage_prepare(ctx.get(), g.data(), (int32_t)g.size(), T), "cpprob_hip_lineage_prepare");
    }
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    const auto t_start = std::chrono::steady_clock::now();
    res.setup_seconds += std::chrono::duration<double>(t_start - t_setup).count();
    res.workspace_grown = res.workspace_grown || grown || ws.fresh();
    double log_z = 0.0;
    int cur = 0, n_resampled = 0;
    bool stats_on_walk = false;                                       // windowed SMC without a particle store: the read-out rode the lineage walk
    std::vector<double> walk_real, walk_int;
    double walk_lse_ess[2] = {0.0, 0.0};
    bool smc_log_z_done = false;
    res.step_ess.clear();
    res.launches_per_step = 1;
    if (!smc) {
        a.logw_out = logw[0]; a.pred_real = d_real; a.pred_int = d_int; a.first_observe = 0; a.stop_after = -1;
        hipLaunchKernelGGL((model_kernel<Caller, Tuple>), grid, block, lane_lds, stream, a, (const Tuple*)d_obs);
        hip_check(hipGetLastError(), "model_kernel");
    } else if (windowed) {
        // Windowed replay: the host probe found that a step depends on its ancestor's last `w` samples only.  Per step ONE gather of
        // w carried values and one row of new ones -- the traffic of the hand-fused kernels -- instead of re-reading and re-writing
        // the whole trace; every launch records the predicts of its own step, ancestors are kept per step, and the traces are read
        // out once at the end by walking the lineages (cpprob_hip_lineage_gather).
        uint64_t* carry[2] = {d_c0, d_c1};
        a.windowed = 1; a.win = w;
        const dim3 sgrid((unsigned)((n + device::kStepBlock - 1) / device::kStepBlock)), sblock(device::kStepBlock);
        const dim3 qgrid((unsigned)((n + cph::kTile - 1) / cph::kTile));
        const size_t step_lds = device::lane_lds_bytes(device::kStepBlock, true, quad);
        const bool step_full = fused && !quad && step_kernel_full<Caller, Tuple>();
        if (fused) {
            a.lane_block = device::kStepBlock; a.fused = quad ? device::kFusedQuad : 1u;
            a.fs.ess_frac = opt.ess_threshold; a.fs.n_pop = (double)n; a.fs.T = T; a.fs.nb = lay.blocks;
            a.fs.may_carry = opt.ess_threshold > 1.0 ? 0 : 1; a.fs.exact_ref = form == StepForm::fused_exact ? 1 : 0;
            a.fs.ctrl = static_cast<device::StepCtrl2*>(lay.ctrl); a.fs.ess = d_ess_p; a.fs.resampled = d_res_p; a.fs.log_z = d_logz_p;
            a.fs.gap_limit = kFixGapLimit;
            res.launches_per_step = form == StepForm::fused_exact ? 3 : 1;
        } else res.launches_per_step = 4;
        for (int t = 0; t < T; ++t) {
            const bool last = t + 1 == T;
            a.logw_in = t > 0 ? logw[cur] : nullptr;
            a.logw_out = logw[cur ^ 1];
            a.carry_in = t > 0 ? carry[cur] : nullptr; a.carry_out = last ? nullptr : carry[cur ^ 1];
            a.fresh_lo = t > 0 ? (int32_t)st.samples_before_observe[(size_t)t - 1] : 0;
            a.next_fresh = (int32_t)st.samples_before_observe[(size_t)t];
            a.pred_real = d_real_gen; a.pred_int = d_int_gen;
            a.first_observe = t; a.stop_after = last ? -1 : t;
            if (fused) {
                a.fs.f = fused_view(lay, (t + 2) % 3, t % 3, (t + 1) % 3);
                a.fs.q_prev = lay.q[(t + 1) & 1]; a.fs.q_next = lay.q[t & 1];
                a.fs.u0 = cpprob_hip_systematic_offset(opt.seed, (uint64_t)t);
                a.fs.bound = (form == StepForm::fused_bounded || quad) ? st.observe_bound[(size_t)t] : 0.0;
                a.fs.t = t; a.fs.anc_row = d_anc_all + (size_t)t * n;
                if (const void* qat = (quad && opt.step_builds) ? step_kernel_for<Caller>(a, true) : nullptr) {
                    const Tuple* obs_p = d_obs;
                    void* kargs[2] = {&a, &obs_p};
                    hip_check(hipLaunchKernel(qat, qgrid, sblock, kargs, step_lds, stream), "model_step_kernel_quad_at");
                    ++res.step_builds_used;
                }
                else if (quad) hipLaunchKernelGGL((model_step_kernel_quad<Caller, Tuple>), qgrid, sblock, step_lds, stream, a, (const Tuple*)d_obs);
                else if (const void* at = opt.step_builds ? step_kernel_for<Caller>(a) : nullptr) {
                    // (a build for this step: its thresholds are the kernel's compile-time facts -- checked above, where it was chosen)
                    const Tuple* obs_p = d_obs;
                    void* kargs[2] = {&a, &obs_p};
                    hip_check(hipLaunchKernel(at, sgrid, sblock, kargs, step_lds, stream), "model_step_kernel_at");
                    ++res.step_builds_used;
                }
                else if (step_full) hipLaunchKernelGGL((model_step_kernel_full<Caller, Tuple>), sgrid, sblock, step_lds, stream, a, (const Tuple*)d_obs);
                else hipLaunchKernelGGL((model_step_kernel<Caller, Tuple>), sgrid, sblock, step_lds, stream, a, (const Tuple*)d_obs);
                hip_check(hipGetLastError(), "model_step_kernel");
                cur ^= 1;
                if (form == StepForm::fused_exact) ctx.check(cpprob_hip_generic_quantize(ctx.get(), t, logw[cur], n), "cpprob_hip_generic_quantize");
                continue;
            }
            a.anc = t > 0 ? d_anc_all + (size_t)t * n : nullptr;
            a.resampled_prev = t > 0 ? d_res_p + (t - 1) : nullptr;
            hipLaunchKernelGGL((model_kernel<Caller, Tuple>), grid, block, lane_lds, stream, a, (const Tuple*)d_obs);
            hip_check(hipGetLastError(), "model_kernel");
            cur ^= 1;
            bookkeep(logw[cur], t, last, last ? d_anc : d_anc_all + (size_t)(t + 1) * n, d_ess_p, d_res_p, d_logz_p);
        }
        if (fused) ctx.check(cpprob_hip_generic_finish(ctx.get(), T, n, kFixGapLimit, d_ess_p, d_res_p, d_logz_p, d_overflow_p), "cpprob_hip_generic_finish");
        // traces: hit h was recorded in the slots of generation step(h); follow every final particle's lineage back to it.  When
        // nobody asked for the traces themselves, StatsPrinter's numbers are taken on that walk (cpprob_hip_lineage_moments / _hist).
        auto gens = [&](const std::vector<int>& steps) { std::vector<int32_t> g; for (int s2 : steps) g.push_back(std::min(s2, T - 1)); return g; };
        bool tail_read = false;
        if (!store) {
            // (the run's tail -- step ESS, evidence, decisions, flag word -- rides the LAST read-out call's result: one stream synchronisation)
            auto ride_tail = [&]() {
                ctx.check(cpprob_hip_readback_with_next_result(ctx.get(), d_tail, h_tail.data(), tail_bytes), "cpprob_hip_readback_with_next_result");
                tail_read = true;
            };
            if (n_real) {
                const std::vector<int32_t> g = gens(st.real_row_step);
                walk_real.resize(4 * n_real);
                if (!n_int) ride_tail();
                ctx.check(cpprob_hip_lineage_moments(ctx.get(), d_anc_all, d_res_p, T, n, d_real_gen, g.data(), (int32_t)g.size(), logw[cur], walk_real.data()), "cpprob_hip_lineage_moments");
            }
            if (n_int) {
                const std::vector<int32_t> g = gens(st.int_hit_step);
                walk_int.resize(8 * n_int);
                ride_tail();
                ctx.check(cpprob_hip_lineage_hist(ctx.get(), d_anc_all, d_res_p, T, n, d_int_gen, g.data(), (int32_t)g.size(), logw[cur], 8, walk_int.data(), walk_lse_ess), "cpprob_hip_lineage_hist");
            }
            stats_on_walk = true;
        } else {
        if (n_real) {
            const std::vector<int32_t> g = gens(st.real_row_step);
            ctx.check(cpprob_hip_lineage_gather(ctx.get(), d_anc_all, d_res_p, T, n, d_real_gen, 0, g.data(), (int32_t)g.size(), d_real), "cpprob_hip_lineage_gather");
        }
        if (n_int) {
            const std::vector<int32_t> g = gens(st.int_hit_step);
            ctx.check(cpprob_hip_lineage_gather(ctx.get(), d_anc_all, d_res_p, T, n, d_int_gen, 1, g.data(), (int32_t)g.size(), d_int), "cpprob_hip_lineage_gather");
        }
        }
        if (!tail_read) read_tail();
        res.step_ess.assign(h_ess_p, h_ess_p + T);
        log_z = h_ess_p[T];
        for (int t = 0; t < T; ++t) n_resampled += h_res_p[t];
        smc_log_z_done = true;
        res.replay_window = (int)w;
    } else {
        res.launches_per_step = 4;
        for (int t = 0; t < T; ++t) {
            const bool last = t + 1 == T;
            a.anc = t > 0 ? d_anc : nullptr;
            a.resampled_prev = t > 0 ? d_res_p + (t - 1) : nullptr;
            a.logw_in = t > 0 ? logw[cur] : nullptr;
            a.logw_out = logw[cur ^ 1];
            a.trace_in = t > 0 ? tr[cur] : nullptr; a.trace_out = tr[cur ^ 1];
            a.nstored_in = t > 0 ? ns[cur] : nullptr; a.nstored_out = ns[cur ^ 1];
            a.pred_real = last ? d_real : nullptr; a.pred_int = last ? d_int : nullptr;
            a.first_observe = t; a.stop_after = last ? -1 : t;
            hipLaunchKernelGGL((model_kernel<Caller, Tuple>), grid, block, lane_lds, stream, a, (const Tuple*)d_obs);
            hip_check(hipGetLastError(), "model_kernel");
            cur ^= 1;
            // normalise, ESS test (thesis p.37), evidence, ancestors of the next generation: all on the device
            bookkeep(logw[cur], t, last, d_anc, d_ess_p, d_res_p, d_logz_p);
        }
        read_tail();
        res.step_ess.assign(h_ess_p, h_ess_p + T);
        log_z = h_ess_p[T];
        for (int t = 0; t < T; ++t) n_resampled += h_res_p[t];
        smc_log_z_done = true;
    }
    res.step_form = fused ? (form == StepForm::fused_exact ? 2 : (quad ? 3 : 1)) : 0;
    fill_predict_names(res, st);
    double lse_ess[2] = {0.0, 0.0};
    bool have_norm = false;
    // StatsPrinter's numbers of every predict hit: all columns of a kind in one device pass against the final weights
    // (SIS: the run's tail rides the last statistics call's result -- one launch into pinned memory, one wait)
    bool sis_tail_read = false;
    auto ride_sis_tail = [&]() {
        if (smc || stats_on_walk) return;
        ctx.check(cpprob_hip_readback_with_next_result(ctx.get(), d_tail, h_tail.data(), tail_bytes), "cpprob_hip_readback_with_next_result");
        sis_tail_read = true;
    };
    if (n_real) {
        std::vector<double> o4(4 * n_real);
        if (stats_on_walk) o4 = walk_real;
        else {
            if (!n_int) ride_sis_tail();
            ctx.check(cpprob_hip_weighted_moments_columns(ctx.get(), d_real, n_real, (size_t)ld, logw[cur], n, o4.data()), "cpprob_hip_weighted_moments_columns");
        }
        lse_ess[0] = o4[2]; lse_ess[1] = o4[3]; have_norm = true;
        for (size_t k = 0, row = 0; k < st.real_ids.size(); ++k) {
            PredictStats& p = res.predicts[k];
            for (size_t d = 0; d < st.real_width[k]; ++d, ++row) { p.mean_nd.push_back(o4[4 * row]); p.variance_nd.push_back(o4[4 * row + 1]); }
            p.mean = p.mean_nd[0]; p.variance = p.variance_nd[0];
        }
    }
    const size_t n_real_hits = st.real_ids.size();
    if (n_int) {
        std::vector<double> h(8 * n_int);
        if (stats_on_walk) { h = walk_int; if (!have_norm) { lse_ess[0] = walk_lse_ess[0]; lse_ess[1] = walk_lse_ess[1]; } }
        else {
            ride_sis_tail();
            ctx.check(cpprob_hip_weighted_hist_columns(ctx.get(), d_int, n_int, (size_t)ld, logw[cur], n, 8, h.data(), have_norm ? nullptr : lse_ess), "cpprob_hip_weighted_hist_columns");
        }
        have_norm = true;
        for (size_t k = 0; k < n_int; ++k) {
            int top = 8;
            while (top > 1 && h[8 * k + top - 1] == 0.0) --top;
            res.predicts[n_real_hits + k].probabilities.assign(h.begin() + 8 * k, h.begin() + 8 * k + top);
        }
    }
    if (!have_norm) {                                                  // (a model without predicts: the weights' own pass)
        double o3[3];
        ctx.check(cpprob_hip_logsumexp_ess(ctx.get(), logw[cur], n, o3), "cpprob_hip_logsumexp_ess");
        lse_ess[0] = o3[1]; lse_ess[1] = o3[2];
    }
    if (!smc_log_z_done) log_z += lse_ess[0] - std::log((double)n);   // SIS: evidence = mean weight
    res.n_particles = n; res.log_evidence = log_z; res.log_norm = lse_ess[0]; res.ess = lse_ess[1]; res.n_resampled = n_resampled; res.used_builtin = false;
    res.run_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();   // (everything StatsPrinter prints; the calls above synchronised)
    if (store) {
        store->n = n;
        store->logw.resize(n); store->real.resize(n_real * n); store->ints.resize(n_int * n);
        hip_check(hipMemcpyAsync(store->logw.data(), logw[cur], n * sizeof(double), hipMemcpyDeviceToHost, stream), "copy logw");
        if (n_real) hip_check(hipMemcpyAsync(store->real.data(), d_real, n_real * n * sizeof(double), hipMemcpyDeviceToHost, stream), "copy real predicts");
        if (n_int) hip_check(hipMemcpyAsync(store->ints.data(), d_int, n_int * n * sizeof(int32_t), hipMemcpyDeviceToHost, stream), "copy int predicts");
    }
    if (!smc && !sis_tail_read) read_tail(); else hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");    // (SMC: the tail was read when the last launch had been issued)
    const int32_t overflow = h_res_p[T];
    if (overflow == 2)
        throw std::runtime_error("cpprob::inference: a particle executed more predict statements than the model's dry run did; the number "
                                 "and order of observe / predict statements must not depend on sampled values on the device path");
    if (overflow == 3) return 3;
    if (overflow == 4 || overflow == 5) return 4;
    return overflow != 0 ? 1 : 0;
}

template <class Caller>
void generic_launcher(StateType algorithm, const void* observes_v, std::size_t n, const detail::TraceStructure& st, const Options& opt,
                      Result& res, HostStore* store)
{
    // trace rows: the structural dry run saw st.n_sample sample statements; particles of models with data-dependent loops
    // (rejection sampling) may execute more -- start with head-room, and on overflow repeat the run with 4x the rows
    std::size_t S = 4 * st.n_sample + 16;
    detail::TraceStructure st_full = st;
    st_full.window = -1;
    const detail::TraceStructure* use = &st;
    // (windowed replay packs its statement counters into one word per lane: cpprob/detail/device_trace.hpp, WinRec)
    if (st.n_observe > device::kWinMaxObserves || st.real_rows() > device::kWinMaxPredicts || st.int_ids.size() > device::kWinMaxPredicts) use = &st_full;
    res.markov_crosscheck = 0;
    if (algorithm == StateType::smc && use->window >= 0 && opt.markov_crosscheck) {
        // The host probe saw a handful of traces; a dependence on older samples that shows on one trace in a thousand passes it.  So
        // the window is certified on the device before it is used: a pilot population under windowed replay and under full replay
        // -- same particle ids, same seed, same bookkeeping between the launches, hence the same variates, weights and ancestors if
        // the window is right -- must agree in every bit of every weight and predict.  Once per model, trace shape and window in
        // this process.
        static std::mutex mu;
        static std::map<std::tuple<std::size_t, std::size_t, int>, bool> certified;
        const auto key = std::make_tuple((std::size_t)st.n_observe, (std::size_t)st.n_sample, st.window);
        bool known = false, ok = false;
        { std::lock_guard<std::mutex> lock(mu); const auto it = certified.find(key); if (it != certified.end()) { known = true; ok = it->second; } }
        if (!known) {
            const std::size_t n_pilot = std::min<std::size_t>(n, 8192);
            Result r_win, r_full;
            HostStore s_win, s_full;
            // (a pilot that runs out of trace rows -- rejection-sampling loops -- says nothing about the window: more rows, again)
            std::size_t Sp = S;
            int rc_win = 1, rc_full = 1;
            for (int k = 0; k < 4 && (rc_win == 1 || rc_full == 1); ++k, Sp *= 4) {
                rc_win = generic_attempt<Caller>(algorithm, observes_v, n_pilot, st, opt, r_win, &s_win, Sp, StepForm::unfused);
                rc_full = (rc_win == 0 || rc_win == 1) ? generic_attempt<Caller>(algorithm, observes_v, n_pilot, st_full, opt, r_full, &s_full, Sp, StepForm::unfused) : rc_win;
            }
            auto same = [](const std::vector<double>& x, const std::vector<double>& y) {
                return x.size() == y.size() && (x.empty() || std::memcmp(x.data(), y.data(), x.size() * sizeof(double)) == 0);
            };
            const bool decided = (rc_win == 0 && rc_full == 0) || rc_win == 3;
            ok = rc_win == 0 && rc_full == 0 && same(s_win.logw, s_full.logw) && same(s_win.real, s_full.real) && s_win.ints == s_full.ints &&
                 std::memcmp(&r_win.log_evidence, &r_full.log_evidence, sizeof(double)) == 0;
            res.setup_seconds += r_win.setup_seconds + r_win.run_seconds + r_full.setup_seconds + r_full.run_seconds;
            if (decided) { std::lock_guard<std::mutex> lock(mu); certified[key] = ok; }      // (only a verdict is remembered: a refuted window, or a certified one)
        }
        res.markov_crosscheck = ok ? 1 : -1;
        if (!ok) use = &st_full;
    }
    StepForm form = st.bounds_fixed ? StepForm::fused_bounded : StepForm::fused_exact;
    // (four particles a lane pays where the population fills the chip several times over and the model's loop is short -- tools/
    //  ab_quad_threshold.sh, ms per run, one a lane / four a lane: hmm<16> 1.5e6 0.697 / 0.637, 3e6 1.22 / 1.11, 6e6 2.24 / 1.94, 1e7 3.78 / 3.27;
    //  linear_gaussian_1d<25> 2e6 1.54 / 1.64, 3e6 2.21 / 2.18, 6e6 4.36 / 4.01; at 10^6 it is latency bound, at T >= 100 its dead iterations
    //  cost more than the shared search saves: profiles/r05_notes.md section 8)
    //  With the step kernels built per step (CPPROB_REGISTER_MODEL_STEPS, <= 32 observes: one build a step) a call of the body is the live
    //  iteration alone, in front of the step's observe AND behind it, and four a lane wins from ~7e5 particles on (tools/ab_quad_builds.sh, one /
    //  four a lane: hmm<16> 3e5 0.256 / 0.277, 1e6 0.440 / 0.395, 3e6 1.03 / 0.81, 1e7 3.07 / 2.38; linear_gaussian_1d<25> 3e5 0.36 / 0.41, 1e6 0.70 / 0.64, 3e6 1.68 / 1.36).
    bool exact_builds = false;
    if (opt.step_builds) for (const StepKernelEntry& e : step_kernels<Caller>()) exact_builds = exact_builds || (e.exact && e.quad_fn);
    const std::size_t quad_from = exact_builds ? std::size_t(700000) : std::size_t(st.n_observe <= 16 ? 1500000 : 3000000);
    if (form == StepForm::fused_bounded && st.n_observe <= 32 && n >= quad_from) form = StepForm::fused_quad;
    if (opt.step_form_override >= 0) {
        if (opt.step_form_override > 3) throw std::invalid_argument("cpprob::gpu::Options::step_form_override: 0 (unfused), 1 (bounded), 2 (exact maximum) or 3 (bounded, four particles a lane)");
        form = static_cast<StepForm>(opt.step_form_override);
    }
    bool rows_exhausted = false;
    for (int attempt = 0; attempt < 6; ++attempt) {
        const int rc = generic_attempt<Caller>(algorithm, observes_v, n, *use, opt, res, store, S, form);
        if (rc == 0) return;
        if (rc == 3) use = &st_full;                               // the device saw what the host probe did not: replay the whole trace
        else if (rc == 4) {
            // a generation did not fit its reference: bounded -> exact maxima -> the three-launch form (its own maximum pass); a
            // generation that has no mass in ANY form (every likelihood -inf) is the model's, not a form's
            if (form == StepForm::fused_bounded || form == StepForm::fused_quad) form = StepForm::fused_exact;
            else if (form == StepForm::fused_exact) form = StepForm::unfused;
            else throw std::runtime_error("cpprob::inference(smc): a generation carries no mass (every particle's likelihood is zero at some observe statement)");
        }
        else { S *= 4; rows_exhausted = true; }
    }
    if (!rows_exhausted) throw std::runtime_error("cpprob::inference(smc): no step form completed the run (replay window and references were refuted in turn)");
    throw std::runtime_error("cpprob::inference(smc): a particle executed more than " + std::to_string(S / 4) + " sample statements "
                             "(data-dependent loop, e.g. rejection sampling, that rarely terminates); use StateType::sis for this model");
}

// ---- ONE joint population over several GPUs (or several ranks on one: loopback) ------------------------------------------------------
// cpprob::inference(StateType::smc, <unchanged model>, ...) with options().devices = {d0, d1, ...}: particle i lives on rank
// floor(i R / N) (contiguous shards, global ids select the random streams), every step resamples the WHOLE population -- the same
// integers as one GPU holding every particle (the masses are exact, the comb is evaluated on exact integers: cpprob/detail/
// fixed_mass.hpp), so the traces are the single-device run's, bit for bit.  One host thread per rank.  Per step and rank: the step
// launch (its prologue PULLS: it searches the hierarchy and walks the weights of whichever rank owns an output's ancestor and reads
// the ancestor's window from that rank's store -- peer access / one address space; nothing is packed or sent) and a one-wavefront launch
// that leaves the rank's {mass, squares, maximum} in its memory.  The NEXT step's first wavefront reads all ranks' 24 bytes and takes
// the generation's decision -- ESS, resample or not, the comb's scale, the ranks' offspring bounds, the next reference, the evidence --
// exactly as one GPU's prologue does; the launch is ordered behind every rank's previous step by stream waits on the peers' events
// (two events a rank; the threads only poll one another's enqueue counters).  Nothing returns to the host between two steps.  Where a
// model's likelihood has no bound (exact-maximum form) the reference needs a pass between two launches: there the ranks' threads
// all-gather maxima and totals through the host (two round trips a step) and take the decision with the device's arithmetic (IEEE fma /
// ceil on exact integers: the same bits).  What this buys is the joint estimator for ANY registered model, with no code of the
// model's own.
struct JointBarrier {
    std::mutex mu; std::condition_variable cv; int world, waiting = 0; unsigned long long generation = 0; bool broken = false;
    std::atomic<bool> failed{false};                                   // (what a thread polling a peer's progress looks at)
    explicit JointBarrier(int w) : world(w) {}
    void wait()
    {
        std::unique_lock<std::mutex> lock(mu);
        if (broken) throw std::runtime_error("another rank of the joint population failed");
        const unsigned long long g = generation;
        if (++waiting == world) { waiting = 0; ++generation; cv.notify_all(); return; }
        cv.wait(lock, [&] { return generation != g || broken; });
        if (broken) throw std::runtime_error("another rank of the joint population failed");
    }
    void fail() { failed.store(true); std::lock_guard<std::mutex> lock(mu); broken = true; cv.notify_all(); }
};
struct JointRankPointers {
    cpprob_hip_generic_layout lay{}; const uint64_t* carry[2] = {nullptr, nullptr};
    const int32_t* anc_all = nullptr; const double* real_gen = nullptr; const int32_t* int_gen = nullptr; std::size_t n = 0;
    const uint64_t* totals = nullptr;                                  // two generations' {mass, squares, key of the maximum}: 3 words by the step's parity
    hipEvent_t stepped[2] = {nullptr, nullptr};                        // by the step's parity: the rank's step launch + totals launch have completed
    std::atomic<int> recorded{0};                                      // steps whose event the rank's thread has recorded
};
struct JointShared {
    int world; std::size_t n_total; std::vector<std::size_t> begin;
    JointBarrier bar;
    std::vector<JointRankPointers> ptrs;
    std::vector<std::array<std::uint64_t, 3>> totals[2];               // by the step's parity
    std::vector<Result> rr; std::vector<HostStore> hs; std::vector<std::string> errs;
    std::vector<int> devices;
    JointShared(int w, std::size_t n) : world(w), n_total(n), begin((std::size_t)w + 1, 0), bar(w), ptrs((std::size_t)w), rr((std::size_t)w), hs((std::size_t)w), errs((std::size_t)w)
    { totals[0].resize((std::size_t)w); totals[1].resize((std::size_t)w); }
};
// lineage of one rank's final particles through every rank's per-step records (ancestor codes: rank << kShardIndexBits | slot)
struct JointStore { const int32_t* anc_all; const double* real_gen; const int32_t* int_gen; int64_t n; };
template <class V>
__global__ __launch_bounds__(256) void joint_lineage_gather_kernel(const JointStore* __restrict__ stores, int rank, const int32_t* __restrict__ resampled, int T, int64_t n,
                                                                   const int32_t* __restrict__ hit_gen, int H, V* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int32_t code = (int32_t)i | (rank << device::kShardIndexBits);
    int h = H - 1;
    for (int t = T - 1; t >= 0; --t) {
        const JointStore st = stores[code >> device::kShardIndexBits];
        const int64_t idx = code & ((1 << device::kShardIndexBits) - 1);
        for (; h >= 0 && hit_gen[h] == t; --h) {
            if (std::is_same<V, double>::value) out[(int64_t)h * n + i] = (V)st.real_gen[(int64_t)h * st.n + idx];
            else out[(int64_t)h * n + i] = (V)st.int_gen[(int64_t)h * st.n + idx];
        }
        if (t > 0 && resampled[t - 1]) code = st.anc_all[(int64_t)t * st.n + idx];
    }
}

// the device's decision arithmetic on the host (cpprob/detail/fixed_mass.hpp: u64_to_double, fixed_decide, FixedCdf::g): exact
// integers, one rounding per IEEE operation -- the same bits on both sides
inline double joint_u64_to_double(std::uint64_t c) { return std::fma((double)(std::uint32_t)(c >> 32), 4294967296.0, (double)(std::uint32_t)c); }
inline double joint_key_inv(std::uint64_t k)
{
    if (k == 0) return -std::numeric_limits<double>::infinity();
    union { double d; std::uint64_t u; } c;
    c.u = (k >> 63) ? (k & ~(1ull << 63)) : ~k;
    return c.d;
}

template <class Caller>
void generic_joint_rank(JointShared& sh, int rank, StateType algorithm, const void* observes_v, const detail::TraceStructure& st, const Options& opt, StepForm form)
{
    using Tuple = typename Caller::observes_t;
    (void)algorithm;
    const int world = sh.world;
    const std::size_t n = sh.begin[(std::size_t)rank + 1] - sh.begin[(std::size_t)rank], N = sh.n_total;
    const int device_id = sh.devices[(std::size_t)rank];
    Result& res = sh.rr[(std::size_t)rank];
    const auto t_setup = std::chrono::steady_clock::now();
    WorkspaceLease ws(device_id);
    Context& ctx = ws->ctx;
    hip_check(hipSetDevice(device_id), "hipSetDevice");
    for (int r = 0; r < world; ++r)                                   // every rank reads every rank's store
        if (sh.devices[(std::size_t)r] != device_id) {
            const hipError_t e = hipDeviceEnablePeerAccess(sh.devices[(std::size_t)r], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hip_check(e, "hipDeviceEnablePeerAccess");
            (void)hipGetLastError();
        }
    hipStream_t stream = static_cast<hipStream_t>(cpprob_hip_stream(ctx.get()));
    const size_t n_real = st.real_rows(), n_int = st.int_ids.size();
    const int T = (int)st.n_observe;
    const uint32_t w = (uint32_t)std::max(1, st.window);
    const bool exact = form == StepForm::fused_exact;

    Tuple* d_obs = nullptr;
    double *d_real = nullptr, *d_logw0 = nullptr, *d_logw1 = nullptr, *d_real_gen = nullptr;
    int32_t *d_int = nullptr, *d_anc_all = nullptr, *d_int_gen = nullptr, *d_res = nullptr, *d_flag = nullptr, *d_hit_real = nullptr, *d_hit_int = nullptr;
    uint64_t *d_c0 = nullptr, *d_c1 = nullptr, *d_tot = nullptr;
    double* d_book = nullptr;                                          // the launches' own books (device-decided form): T x ESS, the evidence
    device::ShardPeer* d_peers = nullptr;
    JointStore* d_stores = nullptr;
    Carver carve;
    carve.add(&d_obs, 1); carve.add(&d_res, (size_t)T + 1); carve.add(&d_flag, 1); carve.add(&d_tot, 8);
    carve.add(&d_book, (size_t)T + 1);
    carve.add(&d_real, n_real * n); carve.add(&d_logw0, n); carve.add(&d_logw1, n); carve.add(&d_int, n_int * n);
    carve.add(&d_c0, (size_t)w * n); carve.add(&d_c1, (size_t)w * n); carve.add(&d_anc_all, (size_t)T * n);
    carve.add(&d_real_gen, n_real * n); carve.add(&d_int_gen, n_int * n);
    carve.add(&d_peers, (size_t)6 * world); carve.add(&d_stores, (size_t)world);
    carve.add(&d_hit_real, n_real + 1); carve.add(&d_hit_int, n_int + 1);
    const bool grown = carve.commit(*ws);
    hip_check(hipMemcpyAsync(d_obs, observes_v, sizeof(Tuple), hipMemcpyHostToDevice, stream), "copy observes");
    hip_check(hipMemsetAsync(d_flag, 0, sizeof(int32_t), stream), "hipMemsetAsync");
    cpprob_hip_generic_layout lay{};
    ctx.check(cpprob_hip_generic_begin(ctx.get(), n, &lay), "cpprob_hip_generic_begin");
    double* logw[2] = {d_logw0, d_logw1};
    uint64_t* carry[2] = {d_c0, d_c1};
    {
        JointRankPointers& me = sh.ptrs[(std::size_t)rank];
        me.lay = lay; me.carry[0] = d_c0; me.carry[1] = d_c1; me.anc_all = d_anc_all; me.real_gen = d_real_gen; me.int_gen = d_int_gen; me.n = n;
        me.totals = d_tot; me.recorded.store(0);
        for (int k = 0; k < 2; ++k) hip_check(hipEventCreateWithFlags(&me.stepped[k], hipEventDisableTiming), "hipEventCreateWithFlags");
    }
    struct EventGuard { hipEvent_t* e; ~EventGuard() { for (int k = 0; k < 2; ++k) if (e[k]) (void)hipEventDestroy(e[k]); } } event_guard{sh.ptrs[(std::size_t)rank].stepped};
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    sh.bar.wait();                                                     // every rank's buffers exist
    // the ranks' stores as this rank's launches address them: step t reads hierarchy copy (t + 2) % 3, weights q[(t + 1) & 1] and
    // windows carry[t & 1] -- six combinations, uploaded once
    std::vector<device::ShardPeer> peers((size_t)6 * world);
    std::vector<JointStore> stores((size_t)world);
    for (int k = 0; k < 6; ++k)
        for (int r = 0; r < world; ++r) {
            const JointRankPointers& p = sh.ptrs[(size_t)r];
            device::ShardPeer& e = peers[(size_t)k * world + r];
            e.f = fused_view(p.lay, (k + 2) % 3, k % 3, (k + 1) % 3); e.q = p.lay.q[(k + 1) & 1]; e.carry = p.carry[k & 1]; e.n = (int64_t)p.n; e.nb = p.lay.blocks; e.pad = 0;
            e.totals = p.totals + 4 * ((k + 1) & 1);                    // step t reads generation t-1's
        }
    for (int r = 0; r < world; ++r) { const JointRankPointers& p = sh.ptrs[(size_t)r]; stores[(size_t)r] = JointStore{p.anc_all, p.real_gen, p.int_gen, (int64_t)p.n}; }
    hip_check(hipMemcpyAsync(d_peers, peers.data(), peers.size() * sizeof(device::ShardPeer), hipMemcpyHostToDevice, stream), "copy the peer tables");
    hip_check(hipMemcpyAsync(d_stores, stores.data(), stores.size() * sizeof(JointStore), hipMemcpyHostToDevice, stream), "copy the peer tables");
    auto gens = [&](const std::vector<int>& steps) { std::vector<int32_t> g; for (int s2 : steps) g.push_back(std::min(s2, T - 1)); return g; };
    const std::vector<int32_t> g_real = gens(st.real_row_step), g_int = gens(st.int_hit_step);
    if (n_real) hip_check(hipMemcpyAsync(d_hit_real, g_real.data(), n_real * sizeof(int32_t), hipMemcpyHostToDevice, stream), "copy the hit table");
    if (n_int) hip_check(hipMemcpyAsync(d_hit_int, g_int.data(), n_int * sizeof(int32_t), hipMemcpyHostToDevice, stream), "copy the hit table");
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    const auto t_start = std::chrono::steady_clock::now();
    res.setup_seconds = std::chrono::duration<double>(t_start - t_setup).count();
    res.workspace_grown = grown || ws.fresh();

    ModelKernelArgs a{};
    a.n = (int64_t)n; a.ld = (int64_t)n; a.seed = opt.seed; a.overflow = d_flag; a.pid0 = sh.begin[(size_t)rank];
    a.pred_real_cap = (uint32_t)n_real; a.pred_int_cap = (uint32_t)n_int;
    a.windowed = 1; a.win = w; a.lane_block = device::kStepBlock; a.fused = 1;
    a.fs.ess_frac = opt.ess_threshold; a.fs.n_pop = (double)N; a.fs.T = T; a.fs.nb = lay.blocks;
    a.fs.may_carry = opt.ess_threshold > 1.0 ? 0 : 1; a.fs.exact_ref = exact ? 1 : 0; a.fs.gap_limit = kFixGapLimit;
    a.sh.world = world; a.sh.rank = rank;
    for (int r = 0; r <= world; ++r) a.sh.first[r] = sh.begin[(size_t)r];
    const dim3 sgrid((unsigned)((n + device::kStepBlock - 1) / device::kStepBlock)), sblock(device::kStepBlock);
    const size_t step_lds = device::lane_lds_bytes(device::kStepBlock, true);
    std::array<std::uint64_t, 3> h_tot{};
    auto gather_totals = [&](int t) -> std::array<std::vector<std::uint64_t>, 3> {
        ctx.check(cpprob_hip_generic_totals(ctx.get(), t, n, d_tot + 4 * (t & 1)), "cpprob_hip_generic_totals");
        hip_check(hipMemcpyAsync(h_tot.data(), d_tot + 4 * (t & 1), 3 * sizeof(std::uint64_t), hipMemcpyDeviceToHost, stream), "copy the shard's totals");
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
        sh.totals[t & 1][(size_t)rank] = h_tot;
        sh.bar.wait();                                                // the generation's all-gather; every rank's step t is complete behind it
        std::array<std::vector<std::uint64_t>, 3> all;
        for (int k = 0; k < 3; ++k) { all[(size_t)k].resize((size_t)world); for (int r = 0; r < world; ++r) all[(size_t)k][(size_t)r] = sh.totals[t & 1][(size_t)r][(size_t)k]; }
        return all;
    };
    // generation t-1's decision (taken identically by every rank) and the run's bookkeeping
    std::vector<double> ess((size_t)T, 0.0);
    std::vector<int32_t> resd((size_t)T + 1, 0);
    double lz = 0.0, ref_prev = 0.0, gap_max = 0.0;
    int flag = 0;
    bool resample = false;
    double M_prev = 0.0;
    a.sh.host_decided = exact ? 1 : 0;
    a.fs.ctrl = static_cast<device::StepCtrl2*>(lay.ctrl); a.fs.ess = d_book; a.fs.log_z = d_book + T; a.fs.resampled = d_res;
    if (!exact) hip_check(hipMemsetAsync(d_res, 0, ((size_t)T + 1) * sizeof(int32_t), stream), "hipMemsetAsync");
    // one generation's totals summed on the host, and what follows from them (the device's arithmetic: fixed_decide)
    struct Totals { std::uint64_t S = 0, Q = 0; double M = 0.0, Sd = 0.0, W = 0.0, ess = 0.0; };
    auto sum_totals = [&](const std::array<std::vector<std::uint64_t>, 3>& all) {
        Totals g; std::uint64_t key = 0;
        for (int r = 0; r < world; ++r) { g.S += all[0][(size_t)r]; g.Q += all[1][(size_t)r]; key = std::max(key, all[2][(size_t)r]); }
        g.M = joint_key_inv(key);
        g.Sd = joint_u64_to_double(g.S); g.W = g.Sd * (1.0 / 4294967296.0);
        const double Qd = joint_u64_to_double(g.Q) * (1.0 / 4294967296.0), e = g.W * g.W / Qd;
        g.ess = e > (double)N ? (double)N : e;
        return g;
    };
    for (int t = 0; t < T; ++t) {
        const bool last = t + 1 == T;
        a.logw_in = t > 0 ? logw[t & 1] : nullptr; a.logw_out = logw[(t + 1) & 1];
        a.carry_in = t > 0 ? carry[t & 1] : nullptr; a.carry_out = last ? nullptr : carry[(t + 1) & 1];
        a.fresh_lo = t > 0 ? (int32_t)st.samples_before_observe[(size_t)t - 1] : 0;
        a.next_fresh = (int32_t)st.samples_before_observe[(size_t)t];
        a.pred_real = d_real_gen; a.pred_int = d_int_gen;
        a.first_observe = t; a.stop_after = last ? -1 : t;
        a.fs.f = fused_view(lay, (t + 2) % 3, t % 3, (t + 1) % 3);
        a.fs.q_prev = lay.q[(t + 1) & 1]; a.fs.q_next = lay.q[t & 1];
        a.fs.u0 = cpprob_hip_systematic_offset(opt.seed, (uint64_t)t);
        a.fs.bound = exact ? 0.0 : st.observe_bound[(size_t)t];
        a.fs.t = t; a.fs.anc_row = d_anc_all + (size_t)t * n;
        a.sh.peers = d_peers + (size_t)(t % 6) * world;
        if (!exact) {
            // bounded form: nothing comes back between two steps.  The step's first wavefront reads every rank's totals of generation
            // t-1 and takes the decision itself; what it needs is that every rank's step t-1 has completed: a stream wait on their events.
            hipLaunchKernelGGL((model_step_kernel_joint<Caller, Tuple>), sgrid, sblock, step_lds, stream, a, (const Tuple*)d_obs);
            hip_check(hipGetLastError(), "model_step_kernel");
            ctx.check(cpprob_hip_generic_totals(ctx.get(), t, n, d_tot + 4 * (t & 1)), "cpprob_hip_generic_totals");
            hip_check(hipEventRecord(sh.ptrs[(size_t)rank].stepped[t & 1], stream), "hipEventRecord");
            sh.ptrs[(size_t)rank].recorded.store(t + 1, std::memory_order_release);
            if (last) continue;
            for (int r = 0; r < world; ++r) {
                if (r == rank) continue;
                // (a peer's thread is an enqueue or two away: poll.  Its record of step t + 2 on the same event comes behind its own poll
                //  of THIS rank's step t + 1, i.e. behind this wait: two events a rank are enough)
                while (sh.ptrs[(size_t)r].recorded.load(std::memory_order_acquire) < t + 1) {
                    if (sh.bar.failed.load()) throw std::runtime_error("another rank of the joint population failed");
                    std::this_thread::yield();
                }
                hip_check(hipStreamWaitEvent(stream, sh.ptrs[(size_t)r].stepped[t & 1], 0), "hipStreamWaitEvent");
            }
            continue;
        }
        a.sh.resample = resample ? 1 : 0;
        a.sh.ref = 0.0;
        hipLaunchKernelGGL((model_step_kernel_joint<Caller, Tuple>), sgrid, sblock, step_lds, stream, a, (const Tuple*)d_obs);
        hip_check(hipGetLastError(), "model_step_kernel");
        // the generation's exact maximum over every rank first, then the masses against it
        ctx.check(cpprob_hip_generic_max(ctx.get(), t, logw[(t + 1) & 1], n), "cpprob_hip_generic_max");
        const auto mx = gather_totals(t);
        std::uint64_t key = 0;
        for (int r = 0; r < world; ++r) key = std::max(key, mx[2][(size_t)r]);
        const double ref_gen = joint_key_inv(key);
        ctx.check(cpprob_hip_generic_quantize_ref(ctx.get(), t, logw[(t + 1) & 1], n, ref_gen), "cpprob_hip_generic_quantize_ref");
        sh.bar.wait();                                                // (the totals' buffer of this parity is written again below)
        const auto all = gather_totals(t);
        const Totals g = sum_totals(all);
        resample = !last && g.ess < opt.ess_threshold * (double)N;
        const double gap = g.W > 0.0 ? ref_gen - g.M : 1e300;
        gap_max = t == 0 ? gap : std::max(gap_max, gap);
        if (gap < 0.0) flag = 4; else if (gap > kFixGapLimit && flag == 0) flag = 5;
        ess[(size_t)t] = g.ess; resd[(size_t)t] = resample ? 1 : 0;
        if (resample || last) lz += ref_gen + std::log(g.W / (double)N);
        M_prev = g.M; ref_prev = ref_gen;
        if (resample) {
            const double inv = (double)N / g.Sd, u0 = cpprob_hip_systematic_offset(opt.seed, (uint64_t)t + 1);
            a.sh.inv = inv;
            std::uint64_t before = 0;
            for (int r = 0; r < world; ++r) {
                a.sh.before[r] = before;
                a.sh.obound[r] = std::ceil(std::fma(joint_u64_to_double(before), inv, -u0));
                before += all[0][(size_t)r];
            }
            a.sh.obound[0] = 0.0;                                      // (u0 < 1: ceil(-u0) is -0 or 0)
            a.sh.obound[world] = (double)N;
        }
    }
    (void)M_prev;
    if (!exact) {
        // the last generation's books: its totals and what the launches kept come back once, behind the last step
        std::vector<double> book((size_t)T + 1);
        device::StepCtrl2 ctrl{};
        const int tl = T - 1;
        hip_check(hipMemcpyAsync(h_tot.data(), d_tot + 4 * (tl & 1), 3 * sizeof(std::uint64_t), hipMemcpyDeviceToHost, stream), "copy the shard's totals");
        hip_check(hipMemcpyAsync(book.data(), d_book, ((size_t)T + 1) * sizeof(double), hipMemcpyDeviceToHost, stream), "copy the run's books");
        hip_check(hipMemcpyAsync(resd.data(), d_res, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost, stream), "copy the decisions");
        hip_check(hipMemcpyAsync(&ctrl, lay.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, stream), "copy the run's books");
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
        sh.totals[tl & 1][(size_t)rank] = h_tot;
        sh.bar.wait();
        std::array<std::vector<std::uint64_t>, 3> all;
        for (int k = 0; k < 3; ++k) { all[(size_t)k].resize((size_t)world); for (int r = 0; r < world; ++r) all[(size_t)k][(size_t)r] = sh.totals[tl & 1][(size_t)r][(size_t)k]; }
        const Totals g = sum_totals(all);
        ref_prev = T > 1 ? ctrl.ref_cur : st.observe_bound[0];
        const double gap = g.W > 0.0 ? ref_prev - g.M : 1e300;
        gap_max = T > 1 ? std::max(ctrl.gap_max, gap) : gap;
        if (gap < 0.0) flag = 4; else if (gap > kFixGapLimit) flag = 5;
        for (int t = 0; t < tl; ++t) ess[(size_t)t] = book[(size_t)t];
        ess[(size_t)tl] = g.ess; resd[(size_t)tl] = 0;
        lz = (T > 1 ? book[(size_t)T] : 0.0) + (ref_prev + std::log(g.W / (double)N));
    }
    (void)ref_prev;
    int32_t dev_flag = 0;
    hip_check(hipMemcpyAsync(&dev_flag, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, stream), "copy the flag word");
    if (exact) hip_check(hipMemcpyAsync(d_res, resd.data(), ((size_t)T + 1) * sizeof(int32_t), hipMemcpyHostToDevice, stream), "copy the decisions");
    // traces: follow every final particle of this rank back through whichever rank recorded each step
    const dim3 ggrid((unsigned)((n + 255) / 256)), gblock(256);
    if (n_real) hipLaunchKernelGGL((joint_lineage_gather_kernel<double>), ggrid, gblock, 0, stream, (const JointStore*)d_stores, rank, (const int32_t*)d_res, T, (int64_t)n, (const int32_t*)d_hit_real, (int)n_real, d_real);
    if (n_int) hipLaunchKernelGGL((joint_lineage_gather_kernel<int32_t>), ggrid, gblock, 0, stream, (const JointStore*)d_stores, rank, (const int32_t*)d_res, T, (int64_t)n, (const int32_t*)d_hit_int, (int)n_int, d_int);
    hip_check(hipGetLastError(), "joint_lineage_gather_kernel");
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    double* const lw_final = logw[T & 1];
    fill_predict_names(res, st);
    double lse_ess[2] = {0.0, 0.0};
    bool have_norm = false;
    if (n_real) {
        std::vector<double> o4(4 * n_real);
        ctx.check(cpprob_hip_weighted_moments_columns(ctx.get(), d_real, n_real, n, lw_final, n, o4.data()), "cpprob_hip_weighted_moments_columns");
        lse_ess[0] = o4[2]; lse_ess[1] = o4[3]; have_norm = true;
        for (size_t k = 0, row = 0; k < st.real_ids.size(); ++k) {
            PredictStats& p = res.predicts[k];
            for (size_t d = 0; d < st.real_width[k]; ++d, ++row) { p.mean_nd.push_back(o4[4 * row]); p.variance_nd.push_back(o4[4 * row + 1]); }
            p.mean = p.mean_nd[0]; p.variance = p.variance_nd[0];
        }
    }
    if (n_int) {
        std::vector<double> h(8 * n_int);
        ctx.check(cpprob_hip_weighted_hist_columns(ctx.get(), d_int, n_int, n, lw_final, n, 8, h.data(), have_norm ? nullptr : lse_ess), "cpprob_hip_weighted_hist_columns");
        have_norm = true;
        for (size_t k = 0; k < n_int; ++k) res.predicts[st.real_ids.size() + k].probabilities.assign(h.begin() + 8 * k, h.begin() + 8 * k + 8);
    }
    if (!have_norm) {
        double o3[3];
        ctx.check(cpprob_hip_logsumexp_ess(ctx.get(), lw_final, n, o3), "cpprob_hip_logsumexp_ess");
        lse_ess[0] = o3[1]; lse_ess[1] = o3[2];
    }
    int n_resampled = 0;
    for (int t = 0; t < T; ++t) n_resampled += resd[(size_t)t];
    res.n_particles = n; res.log_evidence = lz; res.log_norm = lse_ess[0]; res.ess = lse_ess[1]; res.n_resampled = n_resampled; res.used_builtin = false;
    res.step_ess = ess; res.replay_window = (int)w; res.step_form = exact ? 2 : 1; res.launches_per_step = exact ? 4 : 2;
    res.joint_flag = dev_flag == 2 || dev_flag == 3 ? dev_flag : (dev_flag == 4 || flag == 4 ? 4 : (dev_flag == 5 || flag == 5 ? 5 : 0));
    res.run_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    HostStore& hs = sh.hs[(size_t)rank];
    hs.n = n; hs.logw.resize(n); hs.real.resize(n_real * n); hs.ints.resize(n_int * n);
    hip_check(hipMemcpyAsync(hs.logw.data(), lw_final, n * sizeof(double), hipMemcpyDeviceToHost, stream), "copy logw");
    if (n_real) hip_check(hipMemcpyAsync(hs.real.data(), d_real, n_real * n * sizeof(double), hipMemcpyDeviceToHost, stream), "copy real predicts");
    if (n_int) hip_check(hipMemcpyAsync(hs.ints.data(), d_int, n_int * n * sizeof(int32_t), hipMemcpyDeviceToHost, stream), "copy int predicts");
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    sh.bar.wait();                                                     // nobody's store is released while a peer may still read it
}

// returns 0, or 3 / 4 as generic_attempt does (the caller falls back / repeats on exact maxima); rr / hs of `sh` hold the ranks' shares
template <class Caller>
int generic_joint_attempt(StateType algorithm, const void* observes_v, std::size_t n, const detail::TraceStructure& st, const Options& opt, StepForm form, JointShared& sh)
{
    std::vector<std::thread> th;
    for (int r = 0; r < sh.world; ++r)
        th.emplace_back([&, r] {
            try { generic_joint_rank<Caller>(sh, r, algorithm, observes_v, st, opt, form); }
            catch (const std::exception& ex) { sh.errs[(size_t)r] = ex.what(); sh.bar.fail(); }
        });
    for (auto& t : th) t.join();
    std::string first;
    for (int r = 0; r < sh.world; ++r)
        if (!sh.errs[(size_t)r].empty() && sh.errs[(size_t)r] != "another rank of the joint population failed") { first = "shard " + std::to_string(r) + ": " + sh.errs[(size_t)r]; break; }
    if (first.empty()) for (int r = 0; r < sh.world; ++r) if (!sh.errs[(size_t)r].empty()) { first = sh.errs[(size_t)r]; break; }
    if (!first.empty()) throw std::runtime_error("cpprob::inference (joint population): " + first);
    (void)n;
    int flag = 0;
    for (const Result& r : sh.rr) flag = std::max(flag, r.joint_flag);
    if (flag == 2)
        throw std::runtime_error("cpprob::inference: a particle executed more predict statements than the model's dry run did; the number "
                                 "and order of observe / predict statements must not depend on sampled values on the device path");
    return flag == 3 ? 3 : (flag ? 4 : 0);
}

// Type-erased entry of the joint run (cpprob/detail/host_engine.hpp calls it when options().devices names several ranks): false when this
// model / these options have no joint form (no replay window, another resampler) -- the caller then runs islands and says so.
template <class Caller>
bool generic_joint_launcher(StateType algorithm, const void* observes_v, std::size_t n, const detail::TraceStructure& st, const Options& opt, Result& res, HostStore* store)
{
    const int world = (int)opt.devices.size();
    if (algorithm != StateType::smc || st.window < 0 || opt.resampler != CPPROB_HIP_RESAMPLE_SYSTEMATIC || world < 2 || world > device::kMaxShards) return false;
    if (st.n_observe > device::kWinMaxObserves || st.real_rows() > device::kWinMaxPredicts || st.int_ids.size() > device::kWinMaxPredicts) return false;
    if ((n + world - 1) / world >= (std::size_t(1) << device::kShardIndexBits) || n >= (std::size_t(1) << 31)) return false;
    if (n < (std::size_t)world) return false;                          // (a shard would be empty: the caller runs islands)
    // Ranks on DIFFERENT physical devices: every peer read of the step's prologue (totals, hierarchy words, integer weights, carry
    // windows) relies on a kernel boundary refreshing lines another device rewrote, and on peer access covering workspace blocks
    // allocated before it was enabled.  Neither has met a real link (every run so far: loopback ranks of one device), so the joint
    // form is offered there only on request (Options::joint_across_devices) and the caller runs islands otherwise -- said in Result.
    {
        bool distinct = false;
        for (int r = 1; r < world; ++r) distinct = distinct || opt.devices[(size_t)r] != opt.devices[0];
        if (distinct && !opt.joint_across_devices) { res.joint_note = "joint population across physical devices is unvalidated on real links: islands (set Options::joint_across_devices to run it)"; return false; }
        if (distinct) res.joint_note = "joint population across physical devices: not yet validated against the single-device run on real links";
    }
    // the replay window is certified on one device first (generic_launcher's pilot) -- once per model, trace shape and window in this
    // process: a later call reuses the verdict instead of running 8192 particles again
    {
        static std::mutex jmu;
        static std::map<std::tuple<std::size_t, std::size_t, int>, int> verdict;           // (Caller is part of the function's identity)
        const auto key = std::make_tuple((std::size_t)st.n_observe, (std::size_t)st.n_sample, st.window);
        int known = 0, cross = 0;
        { std::lock_guard<std::mutex> lock(jmu); const auto it = verdict.find(key); if (it != verdict.end()) { known = 1; cross = it->second; } }
        if (!known) {
            Options po = opt; po.devices.clear(); po.device = opt.devices[0]; po.dump = false;
            Result pr;
            generic_launcher<Caller>(algorithm, observes_v, std::min<std::size_t>(n, 8192), st, po, pr, nullptr);
            cross = pr.replay_window < 0 ? -2 : pr.markov_crosscheck;
            std::lock_guard<std::mutex> lock(jmu); verdict[key] = cross;
        }
        if (cross == -2) return false;                                 // the pilot refuted the window: no joint form for this model
        res.markov_crosscheck = cross;
    }
    StepForm form = st.bounds_fixed ? StepForm::fused_bounded : StepForm::fused_exact;
    if (opt.step_form_override == 2) form = StepForm::fused_exact;
    for (int attempt = 0; attempt < 2; ++attempt) {
        JointShared sh(world, n);
        sh.devices = opt.devices;
        for (int r = 0; r < world; ++r) sh.begin[(size_t)r + 1] = sh.begin[(size_t)r] + n / (size_t)world + ((size_t)r < n % (size_t)world ? 1 : 0);
        const int rc = generic_joint_attempt<Caller>(algorithm, observes_v, n, st, opt, form, sh);
        if (rc == 3) return false;
        if (rc == 4 && form == StepForm::fused_bounded) { form = StepForm::fused_exact; continue; }
        combine_shards(sh.rr, store ? &sh.hs : nullptr, sh.begin, n, st, false, res, store);
        const Result& r0 = sh.rr[0];
        res.log_evidence = r0.log_evidence; res.n_resampled = r0.n_resampled; res.step_ess = r0.step_ess; res.replay_window = r0.replay_window;
        res.step_form = r0.step_form; res.launches_per_step = r0.launches_per_step; res.joint = true; res.n_gpus = world;
        double setup = 0; bool grown = false;
        for (const Result& r : sh.rr) { setup = std::max(setup, r.setup_seconds); grown = grown || r.workspace_grown; }
        res.setup_seconds = setup; res.workspace_grown = grown;
        return true;
    }
    throw std::runtime_error("cpprob::inference (joint population): a generation's weights left even their exact maximum's range");
}

// Device view (cpprob/detail/device_vector.hpp): the launcher receives the HOST function's observes tuple (std::vector elements) and
// hands the device the view's tuple (fixed-capacity elements), converted element by element.
template <class To, class From, std::size_t... I>
To convert_observes(const From& from, std::index_sequence<I...>)
{
    return To(typename std::tuple_element<I, To>::type(std::get<I>(from))...);
}
template <class HostTuple, class Caller>
void generic_launcher_view(StateType algorithm, const void* observes_v, std::size_t n, const detail::TraceStructure& st, const Options& opt,
                           Result& res, HostStore* store)
{
    using DevTuple = typename Caller::observes_t;
    static_assert(std::tuple_size<HostTuple>::value == std::tuple_size<DevTuple>::value, "host model and device view take the same number of arguments");
    const DevTuple dev = convert_observes<DevTuple>(*static_cast<const HostTuple*>(observes_v), std::make_index_sequence<std::tuple_size<DevTuple>::value>{});
    generic_launcher<Caller>(algorithm, &dev, n, st, opt, res, store);
}

template <class HostFP, HostFP H, class DevFP, DevFP D>
bool register_model_view(const char* name)
{
    Entry e; e.name = name; e.generic = &generic_launcher_view<tuple_observes_t<HostFP>, FunctionCaller<DevFP, D>>; e.generic_vectors = true;   // (vector-valued statements: no replay window, no joint form)
    add_release_hook(&release_workspaces);
    return add_entry(Key{reinterpret_cast<const void*>(H), 0}, e);
}

template <class FP, FP F>
bool register_model(const char* name)
{
    Entry e; e.name = name; e.generic = &generic_launcher<FunctionCaller<FP, F>>; e.generic_joint = &generic_joint_launcher<FunctionCaller<FP, F>>;
    add_release_hook(&release_workspaces);
    return add_entry(Key{reinterpret_cast<const void*>(F), 0}, e);
}

// functor models (e.g. models::Gauss<>): keyed by type, called through a default-constructed instance
template <class Functor>
bool register_functor(const char* name)
{
    Entry e; e.name = name; e.generic = &generic_launcher<FunctorCaller<Functor>>; e.generic_joint = &generic_joint_launcher<FunctorCaller<Functor>>;
    add_release_hook(&release_workspaces);
    return add_entry(Key{nullptr, typeid(Functor).hash_code()}, e);
}

}  // namespace gpu
}  // namespace cpprob

#define CPPROB_REGISTER_MODEL(fn) \
    static const bool CPPROB_PP_CAT(cpprob_reg_model_, __LINE__) = ::cpprob::gpu::register_model<decltype(&fn), &fn>(#fn)
// Step kernels built per step for a model of n_observes observe statements, samples_per_observe sample statements each (see
// model_step_kernel_at).  In a translation unit of its OWN, compiled with
//     -mllvm -unroll-threshold=2000000 -mllvm -inline-threshold=10000000 -mllvm -amdgpu-inline-max-bb=1000000
//     -mllvm -memdep-block-scan-limit=2000 -mllvm -memdep-block-number-limit=2000
// (the model's loops fully unrolled, the model inlined whatever its size, store-to-load forwarding across the unrolled body); `part`
// of `parts`: the builds are dealt over that many units, which compile side by side (cpprob_amd/build.py).
#define CPPROB_REGISTER_MODEL_STEPS(fn, n_observes, samples_per_observe, part, parts) \
    static const bool CPPROB_PP_CAT(cpprob_reg_steps_, __LINE__) = ::cpprob::gpu::register_step_builds< \
        ::cpprob::gpu::FunctionCaller<decltype(&fn), &fn>, n_observes, samples_per_observe, 1, part, parts>(std::make_integer_sequence<int, (n_observes <= ::cpprob::gpu::kStepExactMax ? n_observes : ::cpprob::gpu::kStepFromBuilds)>{})
// host function, its instantiation inside namespace cpprob_device_view (see cpprob/device_view_begin.hpp)
#define CPPROB_REGISTER_MODEL_VIEW(host_fn, device_fn) \
    static const bool CPPROB_PP_CAT(cpprob_reg_view_, __LINE__) = ::cpprob::gpu::register_model_view<decltype(&host_fn), &host_fn, decltype(&device_fn), &device_fn>(#host_fn)
#define CPPROB_REGISTER_FUNCTOR(type) \
    static const bool CPPROB_PP_CAT(cpprob_reg_functor_, __LINE__) = ::cpprob::gpu::register_functor<type>(#type)

#endif
