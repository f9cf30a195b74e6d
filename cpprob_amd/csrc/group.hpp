// Multi-GPU driver of one joint population: one context per GPU, the exchange scope's per-step protocol (include/cpprob_hip.h)
// issued from C++ with no host synchronisation inside a run -- what cpprob::inference calls when cpprob::gpu::options().devices names
// several GPUs, and what bench.py runs under torchrun.
//
// Collectives (who carries the per-step all-gather of 3 doubles per rank and the run's final all-reduce)
//   RCCL (xGMI)  dlopen()ed (the copy the process already holds, else /opt/rocm's), on each context's own stream, no host in the loop.
//                In-process form: one communicator per local GPU (ncclCommInitAll) driven by one host thread each;
//                one-process-per-GPU form: ncclCommInitRank from a unique id the launcher distributes.
//   external     the caller's own collectives over host buffers (MPI, gloo, ...): cpprob_hip_group_create_external.  Host-synchronising.
//   loopback     every rank's context on ONE device and ONE stream; program order is the only synchronisation.  This is how a one-GPU
//                box exercises the whole protocol, shard layouts and capacities included (RCCL refuses duplicate devices).
//   mailboxes    (the PER-STEP collectives of RCCL and external groups, wherever the ranks can map each other's memory) stores into
//                the peers' mailboxes and a spin on this rank's own: device_collectives.hpp.  The library / the caller's all-gather
//                then carries the set-up exchanges and the run's final all-reduce only.
// Transports (who moves the migrating lineages)
//   direct       the packing kernel of the SENDING rank stores each record straight into the receiving rank's buffer -- peer access
//                inside one process, hipIpc mappings between processes, plain pointers in loopback -- so the bytes that cross xGMI are
//                the records themselves: records x (t + 1) x value size, nothing on a step that does not resample.  What orders the
//                receiver's commit behind the senders' stores is a second, one-double all-gather per step (stream-ordered after the
//                packing kernel on every rank: a rank's contribution cannot arrive before its stores are released).
//   send/recv    ncclSend / ncclRecv of the fixed-capacity peer segments: the fall-back where peers cannot map each other's memory
//                (counts must be host constants, so capacity travels, not records).
// Included by cpprob_hip.hip (it uses the context's internals).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <mutex>
#include <thread>

#include "device_collectives.hpp"

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string where;
};

// One RCCL per process: prefer the copy that is already mapped (a PyTorch process has its own), so that the two never coexist.
RcclApi* rccl_api(std::string& err)
{
    static RcclApi api;
    static bool tried = false;
    static std::string first_err;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (api.lib) return &api;
    if (tried) { err = first_err; return nullptr; }
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !api.lib; ++pass)
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (api.lib) { api.where = std::string(n) + (pass == 0 ? " (already loaded)" : ""); break; }
        }
    if (!api.lib) { first_err = err = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "?"); return nullptr; }
    bool ok = true;
    auto sym = [&](const char* s) -> void* { void* p = dlsym(api.lib, s); if (!p) { ok = false; first_err = std::string("RCCL lacks ") + s; } return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { err = first_err; dlclose(api.lib); api.lib = nullptr; return nullptr; }
    return &api;
}

// {raw weighted sums ..., overflow flag, records sent, bytes sent, mailbox time-outs} of a finished sharded run into one buffer: what the final all-reduce carries
constexpr int kJointExtra = 4;
// Filtering-only shards append their per-step masses (the joint normalisers are the sums over ranks); where the statistics are the
// joint population's already (count form: they come from the all-gathered totals) only rank 0 contributes them to the sum.
__global__ void pack_joint_kernel(const double* __restrict__ stats, int n_stats, const cph::ExchangePlan* __restrict__ plan, int has_traffic, double* __restrict__ out,
                                  const double* __restrict__ masses, int n_masses, int contribute_stats, const int32_t* __restrict__ dc_status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_stats) out[i] = contribute_stats ? stats[i] : 0.0;
    if (i < n_masses) out[n_stats + kJointExtra + i] = masses ? masses[i] : 0.0;
    // (the flag word's three bits travel as three base-128 digits, so that the all-reduce's SUM over <= 63 ranks keeps them apart)
    if (i == 0) {
        const int ov = plan ? plan->overflow : 0;
        out[n_stats] = (double)((ov & 1) + 128 * ((ov >> 1) & 1) + 16384 * ((ov >> 2) & 1) + 2097152 * ((ov >> 3) & 1));
        out[n_stats + 1] = (plan && has_traffic) ? (double)plan->run_records : 0.0;     // (integers below 2^53: exact in any order)
        out[n_stats + 2] = (plan && has_traffic) ? (double)plan->run_bytes : 0.0;
        out[n_stats + 3] = (dc_status && *dc_status != 0) ? 1.0 : 0.0;                  // a mailbox collective of this rank gave up waiting
    }
}

// loopback collectives: every rank lives on this device and this stream
__global__ void loop_allgather_kernel(double* const* __restrict__ locals, double* const* __restrict__ alls, int world)
{
    const int i = threadIdx.x;                        // world * 3 <= 192 threads
    if (i >= 3 * world) return;
    const double v = locals[i / 3][i % 3];
    for (int q = 0; q < world; ++q) alls[q][i] = v;
}
__global__ void loop_allreduce_kernel(double* const* __restrict__ bufs, int world, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int q = 0; q < world; ++q) s += bufs[q][i];   // rank order: bitwise reproducible
    __syncthreads();
    for (int q = 0; q < world; ++q) bufs[q][i] = s;
}

enum { kCollLoopback = 0, kCollRccl = 1, kCollExternal = 2 };
enum { kTransportNone = 0, kTransportDirect = 1, kTransportSendRecv = 2 };

}  // namespace

struct cpprob_hip_group {
    int world = 0, first_rank = 0;
    std::vector<cpprob_hip_ctx*> ctx;                  // local ranks first_rank .. first_rank + n_local - 1
    int coll = kCollRccl;
    bool loopback() const { return coll == kCollLoopback; }
    RcclApi* rccl = nullptr;
    std::vector<ncclComm_t> comm;
    cpprob_hip_collectives ext{};                      // kCollExternal: the caller's collectives (host buffers, blocking)
    std::string err;
    // run configuration
    cpprob_hip_config cfg{};
    std::vector<double> obs;
    std::vector<uint64_t> shard_begin;
    bool begun = false, exchange = false;
    int T = 0, K = 0, n_stats = 0;
    int all_peers = 0; uint64_t cap = 0; int annex_kcols = 0;
    uint64_t user_cap = 0; int user_all_peers = -1; uint32_t user_flags = 0;   // cpprob_hip_group_transport: the caller's choice (0 / -1: the defaults)
    // transport of the migrating lineages
    int transport = kTransportNone;
    bool world1_collectives = false;                   // diagnostic: a group of one still issues every collective and exchanges with itself
    std::vector<void*> ipc_open;                       // peers' receive buffers mapped into this process (hipIpcOpenMemHandle)
    bool remote = false;                               // remote lineages: every rank addresses every rank's particle store
    std::string transport_note;                        // why the direct transport was not taken
    // mailbox collectives (device_collectives.hpp)
    bool dev_coll = false, dc_tried = false; std::string dc_note; uint64_t dc_serial = 0;
    // phase timing (cpprob_hip_group_profile): HIP events between the launches of a step, per local rank (loopback: one stream, rank 0's)
    bool phase_profile = false; std::vector<std::vector<hipEvent_t>> ph_ev; std::vector<int> ph_used; int ph_steps = 0;
    std::vector<cph::Mailbox*> d_box; std::vector<cph::MailboxPeers*> d_box_peers; std::vector<int32_t*> d_dc_status;
    std::vector<void*> ipc_box_open;
    // per local rank device buffers
    std::vector<double*> d_local, d_all, d_joint, d_bar;
    double* const* d_ptr_locals = nullptr; double* const* d_ptr_alls = nullptr; double* const* d_ptr_joints = nullptr;   // loopback: device arrays of pointers
    std::vector<hipStream_t> own_stream;               // loopback: the streams the contexts were created with
    uint64_t last_run = 0; bool ran = false;
    int reruns = 0;
    int repair_gen = -1;                               // >= 0: the run being enqueued resumes behind this generation, requantised against its exact maximum
    int n_requantised = 0;                             // generations repaired in the run results() last collected
    cpprob_hip_traffic traffic{};                      // of the run cpprob_hip_group_results last collected
    // worker threads (RCCL, several local ranks)
    std::vector<std::thread> workers;
    std::mutex m; std::condition_variable cv_job, cv_done;
    uint64_t job_gen = 0; uint64_t job_run = 0; int job_pending = 0; bool stopping = false;
    std::vector<int> job_rc; std::vector<std::string> job_err;
};

namespace {

// (worker threads report through a thread-local message; the group's own string is written by the calling thread only)
thread_local std::string tl_group_err;
int gfail(cpprob_hip_group* g, int code, const std::string& msg)
{
    (void)g;
    tl_group_err = msg;
    g_last_error = msg;
    return code;
}
int gkeep(cpprob_hip_group* g, int rc)                  // calling thread: make the message the group's
{
    if (rc && g) g->err = tl_group_err.empty() ? g_last_error : tl_group_err;
    return rc;
}

#define GHIP_TRY(g, expr)                                                                                              \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess) return gkeep(g, gfail(g, CPPROB_HIP_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e__))); \
    } while (0)

#define NCCL_TRY(g, expr)                                                                                                        \
    do {                                                                                                                         \
        ncclResult_t r__ = (expr);                                                                                               \
        if (r__ != ncclSuccess) return gfail(g, CPPROB_HIP_EDEVICE, std::string(#expr) + ": " + (g)->rccl->GetErrorString(r__)); \
    } while (0)

// ---- collectives of local rank i (RCCL: stream-ordered; external: through host buffers, blocking) ----
int coll_allgather(cpprob_hip_group* g, int i, const double* d_in, double* d_out, size_t n_doubles)
{
    cpprob_hip_ctx* c = g->ctx[(size_t)i];
    if (g->coll == kCollRccl) {
        NCCL_TRY(g, g->rccl->AllGather(d_in, d_out, n_doubles, ncclDouble, g->comm[(size_t)i], c->stream));
        return 0;
    }
    std::vector<double> in(n_doubles), out(n_doubles * (size_t)g->world);
    HIP_TRY(c, hipMemcpyAsync(in.data(), d_in, n_doubles * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (g->ext.allgather(g->ext.user, in.data(), out.data(), n_doubles * sizeof(double))) return gfail(g, CPPROB_HIP_EDEVICE, "the caller's all-gather failed");
    HIP_TRY(c, hipMemcpyAsync(d_out, out.data(), out.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));           // (out is a local)
    return 0;
}
int coll_allreduce_sum(cpprob_hip_group* g, int i, double* d_buf, size_t n_doubles)
{
    cpprob_hip_ctx* c = g->ctx[(size_t)i];
    if (g->coll == kCollRccl) {
        NCCL_TRY(g, g->rccl->AllReduce(d_buf, d_buf, n_doubles, ncclDouble, ncclSum, g->comm[(size_t)i], c->stream));
        return 0;
    }
    // (rank order: the caller's all-gather, summed here -- bitwise reproducible whatever the caller's library does)
    std::vector<double> in(n_doubles), all(n_doubles * (size_t)g->world);
    HIP_TRY(c, hipMemcpyAsync(in.data(), d_buf, n_doubles * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (g->ext.allgather(g->ext.user, in.data(), all.data(), n_doubles * sizeof(double))) return gfail(g, CPPROB_HIP_EDEVICE, "the caller's all-gather failed");
    for (size_t k = 0; k < n_doubles; ++k) { double sum = 0.0; for (int r = 0; r < g->world; ++r) sum += all[(size_t)r * n_doubles + k]; in[k] = sum; }
    HIP_TRY(c, hipMemcpyAsync(d_buf, in.data(), n_doubles * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}
// host bytes of every rank, rank order (set-up time only): RCCL stages them through device memory
int coll_allgather_host(cpprob_hip_group* g, int i, const void* h_in, void* h_out, size_t bytes)
{
    cpprob_hip_ctx* c = g->ctx[(size_t)i];
    if (g->coll == kCollExternal) {
        if (g->ext.allgather(g->ext.user, h_in, h_out, bytes)) return gfail(g, CPPROB_HIP_EDEVICE, "the caller's all-gather failed");
        return 0;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    char* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, bytes * ((size_t)g->world + 1)));
    int rc = 0;
    if (hipMemcpyAsync(d, h_in, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = gfail(g, CPPROB_HIP_EDEVICE, "hipMemcpyAsync (set-up all-gather)");
    if (!rc) { const ncclResult_t r = g->rccl->AllGather(d, d + bytes, bytes, ncclInt8, g->comm[(size_t)i], c->stream); if (r != ncclSuccess) rc = gfail(g, CPPROB_HIP_EDEVICE, std::string("ncclAllGather: ") + g->rccl->GetErrorString(r)); }
    if (!rc && hipMemcpyAsync(h_out, d + bytes, bytes * (size_t)g->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = gfail(g, CPPROB_HIP_EDEVICE, "hipMemcpyAsync (set-up all-gather)");
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = gfail(g, CPPROB_HIP_EDEVICE, "hipStreamSynchronize (set-up all-gather)");
    (void)hipFree(d);
    return rc;
}

bool group_filtering(const cpprob_hip_group* g) { return g->cfg.keep_history == 0 && g->cfg.algorithm == CPPROB_HIP_ALG_SMC; }
size_t group_joint_len(const cpprob_hip_group* g) { return (size_t)g->n_stats + kJointExtra + (group_filtering(g) ? (size_t)g->T : 0); }
int group_record_len(const cpprob_hip_group* g, int t) { return group_filtering(g) ? 1 : t + 1; }      // values per migrating record after step t
void launch_pack_joint(cpprob_hip_group* g, cpprob_hip_ctx* c, int rank, bool traffic, double* d_joint, hipStream_t st)
{
    const size_t li = (size_t)(rank - g->first_rank);
    const int32_t* dc_status = (g->dev_coll && li < g->d_dc_status.size()) ? g->d_dc_status[li] : nullptr;
    const bool filt = group_filtering(g);
    const bool joint_already = filt && c->counts_mode;
    const int n_threads = (int)group_joint_len(g);
    hipLaunchKernelGGL(pack_joint_kernel, dim3((unsigned)((n_threads + 255) / 256)), dim3(256), 0, st, (const double*)c->d_stats, g->n_stats,
                       g->exchange ? (const cph::ExchangePlan*)c->d_xplan : nullptr, traffic ? 1 : 0, d_joint,
                       (filt && !joint_already) ? (const double*)c->d_filter_w : nullptr, filt ? g->T : 0, (!joint_already || rank == 0) ? 1 : 0, dc_status);
}

// the exchange that follows step t, as seen by local rank i (send/recv transport): its peers' fixed-capacity segments
int rccl_exchange(cpprob_hip_group* g, int i, int t)
{
    cpprob_hip_ctx* c = g->ctx[(size_t)i];
    void* d_send = nullptr; void* d_recv = nullptr; int32_t np = 0; int32_t peers[cph::kWorldSlots]; uint64_t cap = 0, bpv = 0;
    if (int rc = cpprob_hip_exchange_transport(c, &d_send, &d_recv, &np, peers, &cap, &bpv)) return gfail(g, rc, cpprob_hip_last_error(c));
    const size_t seg = (size_t)cap * (size_t)group_record_len(g, t) * (size_t)bpv;
    if (np == 0) return 0;
    NCCL_TRY(g, g->rccl->GroupStart());
    for (int s = 0; s < np; ++s) {
        NCCL_TRY(g, g->rccl->Send(static_cast<const char*>(d_send) + (size_t)s * seg, seg, ncclInt8, peers[s], g->comm[(size_t)i], c->stream));
        NCCL_TRY(g, g->rccl->Recv(static_cast<char*>(d_recv) + (size_t)s * seg, seg, ncclInt8, peers[s], g->comm[(size_t)i], c->stream));
    }
    NCCL_TRY(g, g->rccl->GroupEnd());
    return 0;
}

// ---- mailbox collectives of local rank i (device_collectives.hpp): one short launch on the rank's stream ----
constexpr long long kDcTimeoutTicks = 500000000ll;          // 5 s of the 100 MHz wall clock
void dc_allgather(cpprob_hip_group* g, int i, int t, int phases, hipStream_t st)
{
    const int rank = g->first_rank + i;
    const unsigned long long seq = (unsigned long long)(g->dc_serial << 32) | (unsigned long long)(uint32_t)(t + 1);
    hipLaunchKernelGGL(cph::dc_allgather_kernel, dim3(1), dim3(cph::kWave), 0, st, reinterpret_cast<const unsigned long long*>(g->d_local[(size_t)i]),
                       (const cph::MailboxPeers*)g->d_box_peers[(size_t)i], g->d_box[(size_t)i], g->world, rank, t & 1, seq, phases,
                       reinterpret_cast<unsigned long long*>(g->d_all[(size_t)i]), g->d_dc_status[(size_t)i], kDcTimeoutTicks);
}
void dc_barrier(cpprob_hip_group* g, int i, int t, int phases, hipStream_t st)
{
    const int rank = g->first_rank + i;
    const unsigned long long seq = (unsigned long long)(g->dc_serial << 32) | (unsigned long long)(uint32_t)(t + 1);
    hipLaunchKernelGGL(cph::dc_barrier_kernel, dim3(1), dim3(cph::kWave), 0, st, (const cph::MailboxPeers*)g->d_box_peers[(size_t)i], g->d_box[(size_t)i], g->world, rank,
                       t & 1, seq, phases, g->d_dc_status[(size_t)i], kDcTimeoutTicks);
}

// One whole run of local rank i (RCCL: every collective is stream-ordered, nothing waits on the host).
// ---- phase timing of a run: which part of a rank-step the time goes to (what the first multi-GPU run has to explain) ----
// boundaries of one step, in launch order: before the step | after step + shard totals (+ the all-gather where the totals launch
// carries it) | after a separate all-gather | after step_end (plan hand-over) | after the packing launch | after the barrier | after the commit
enum { kPhStepTotals = 0, kPhGather, kPhStepEnd, kPhPack, kPhBarrier, kPhCommit, kPhCount };
constexpr int kPhMarks = kPhCount + 1;
void ph_mark(cpprob_hip_group* g, int i, hipStream_t st)
{
    if (!g->phase_profile) return;
    std::vector<hipEvent_t>& ev = g->ph_ev[(size_t)i];
    int& used = g->ph_used[(size_t)i];
    if ((size_t)used == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ev.push_back(e); }
    (void)hipEventRecord(ev[(size_t)used++], st);
}
void ph_begin(cpprob_hip_group* g)
{
    if (!g->phase_profile) return;
    g->ph_ev.resize(g->ctx.size()); g->ph_used.assign(g->ctx.size(), 0); g->ph_steps = 0;
    for (size_t i = 0; i < g->d_dc_status.size(); ++i)
        if (g->d_dc_status[i]) { (void)hipSetDevice(g->ctx[i]->device); (void)hipMemsetAsync(g->d_dc_status[i] + 2, 0, sizeof(unsigned long long), g->ctx[i]->stream); }
}

int run_rank(cpprob_hip_group* g, int i, uint64_t run_index)
{
    cpprob_hip_ctx* c = g->ctx[(size_t)i];
    const int rank = g->first_rank + i, world = g->world;
    HIP_TRY(c, hipSetDevice(c->device));
    const bool sis = g->cfg.algorithm == CPPROB_HIP_ALG_SIS;
    // (a group of one has nobody to gather from or to exchange with: its own totals ARE the gathered totals -- unless the caller
    //  asked for every collective to run anyway, which is how a one-GPU machine exercises them)
    const bool talk = world > 1 || g->world1_collectives;
    // mailbox collectives: the shard-totals launch of the integer forms posts and waits itself (the floating-point form's totals come
    // out of its scan launch: a launch of their own gathers them)
    const bool gather_in_totals = talk && g->dev_coll && !sis;
    c->x_gather_on = gather_in_totals;
    if (gather_in_totals) {
        c->x_gather = cph::TotalsGather{(const cph::MailboxPeers*)g->d_box_peers[(size_t)i], g->d_box[(size_t)i], world, rank, 0, 0ull,
                                        reinterpret_cast<unsigned long long*>(g->d_all[(size_t)i]), g->d_dc_status[(size_t)i], kDcTimeoutTicks};
        c->x_gather_serial = (unsigned long long)g->dc_serial;
    }
    struct GatherOff { cpprob_hip_ctx* c; ~GatherOff() { c->x_gather_on = false; c->x_gather_tshift = 0; } } gather_off{c};
    // A run resumed behind a repaired generation r: two all-gathers of 24 bytes (the ranks' exact maxima, then their totals of the
    // requantised generation) stand where step r stood, and the collectives of this enqueue are numbered from there (the mailboxes'
    // slots alternate collective by collective: v = 0 the maxima, 1 generation r's totals, t - r + 1 step t's).
    const int rg = sis ? -1 : g->repair_gen;
    const int vshift = rg >= 0 ? rg - 1 : 0;
    c->x_gather_tshift = vshift;
    if (rg >= 0) {
        if (int rc = cpprob_hip_smc_repair_begin(c, rg, g->d_local[(size_t)i])) return gfail(g, rc, cpprob_hip_last_error(c));
        if (talk && g->dev_coll) dc_allgather(g, i, 0, cph::kDcPost | cph::kDcWait, c->stream);
        else if (talk) { if (int rc = coll_allgather(g, i, g->d_local[(size_t)i], g->d_all[(size_t)i], 3)) return rc; }
        if (int rc = cpprob_hip_smc_repair_end(c, rg, talk ? g->d_all[(size_t)i] : g->d_local[(size_t)i], world, rank, g->d_local[(size_t)i])) return gfail(g, rc, cpprob_hip_last_error(c));
    }
    for (int t = sis ? g->T - 1 : (rg >= 0 ? rg : 0); t < g->T; ++t) {
        const bool timed = g->phase_profile && g->exchange && t + 1 < g->T && rg < 0;       // (the steps that run every phase)
        const bool resumed = t == rg;                                                     // (generation rg exists: its totals are in d_local)
        if (timed) ph_mark(g, i, c->stream);
        if (!resumed) { if (int rc = cpprob_hip_smc_step_begin(c, t, run_index, g->d_local[(size_t)i])) return gfail(g, rc, cpprob_hip_last_error(c)); }
        if (timed) ph_mark(g, i, c->stream);
        if (gather_in_totals && c->x_gather_done && !resumed) {}
        else if (talk && g->dev_coll) dc_allgather(g, i, t - vshift, cph::kDcPost | cph::kDcWait, c->stream);
        else if (talk) { if (int rc = coll_allgather(g, i, g->d_local[(size_t)i], g->d_all[(size_t)i], 3)) return rc; }
        if (timed) ph_mark(g, i, c->stream);
        if (int rc = cpprob_hip_smc_step_end(c, t, talk ? g->d_all[(size_t)i] : g->d_local[(size_t)i], world, rank)) return gfail(g, rc, cpprob_hip_last_error(c));
        if (timed) ph_mark(g, i, c->stream);
        if (g->exchange && t + 1 < g->T) {
            // (a context without peers plans at most: cpprob_hip_exchange_pack_async / _commit_async launch nothing for it)
            if (int rc = cpprob_hip_exchange_pack_async(c, t)) return gfail(g, rc, cpprob_hip_last_error(c));
            if (timed) ph_mark(g, i, c->stream);
            if (!talk) {}
            else if (g->transport == kTransportDirect) {
                // the records are already where they belong; what remains is the order: no rank may commit before every rank's
                // packing kernel has completed, and a rank contributes to this all-gather only behind its own packing kernel
                if (g->dev_coll) dc_barrier(g, i, t - vshift, cph::kDcPost | cph::kDcWait, c->stream);
                else if (int rc = coll_allgather(g, i, g->d_bar[(size_t)i], g->d_bar[(size_t)i] + 1, 1)) return rc;
            } else {
                if (int rc = rccl_exchange(g, i, t)) return rc;
            }
            if (timed) ph_mark(g, i, c->stream);
            if (int rc = cpprob_hip_exchange_commit_async(c, t)) return gfail(g, rc, cpprob_hip_last_error(c));
            if (timed) { ph_mark(g, i, c->stream); if (i == 0) ++g->ph_steps; }
        }
    }
    if (int rc = cpprob_hip_smc_finish(c)) return gfail(g, rc, cpprob_hip_last_error(c));
    launch_pack_joint(g, c, rank, g->exchange && talk && g->T > 1, g->d_joint[(size_t)i], c->stream);
    if (int rc = coll_allreduce_sum(g, i, g->d_joint[(size_t)i], group_joint_len(g))) return rc;
    HIP_TRY(c, hipGetLastError());
    return 0;
}

// One whole run of all ranks on one device and one stream: the phases of all ranks interleave in program order.
int loopback_run(cpprob_hip_group* g, uint64_t run_index)
{
    const int world = g->world;
    cpprob_hip_ctx* c0 = g->ctx[0];
    HIP_TRY(c0, hipSetDevice(c0->device));
    hipStream_t st = c0->stream;
    const bool sis = g->cfg.algorithm == CPPROB_HIP_ALG_SIS;
    auto gather = [&](int v) {
        if (g->dev_coll) {
            // (mailboxes on one stream: every rank posts, then every rank finds what it waits for already there)
            for (int r = 0; r < world; ++r) dc_allgather(g, r, v, cph::kDcPost, st);
            for (int r = 0; r < world; ++r) dc_allgather(g, r, v, cph::kDcWait, st);
        } else hipLaunchKernelGGL(loop_allgather_kernel, dim3(1), dim3(192), 0, st, g->d_ptr_locals, g->d_ptr_alls, world);
    };
    // (a run resumed behind a repaired generation: run_rank states the protocol)
    const int rg = sis ? -1 : g->repair_gen;
    const int vshift = rg >= 0 ? rg - 1 : 0;
    if (rg >= 0) {
        for (int r = 0; r < world; ++r)
            if (int rc = cpprob_hip_smc_repair_begin(g->ctx[(size_t)r], rg, g->d_local[(size_t)r])) return gfail(g, rc, cpprob_hip_last_error(g->ctx[(size_t)r]));
        gather(0);
        for (int r = 0; r < world; ++r)
            if (int rc = cpprob_hip_smc_repair_end(g->ctx[(size_t)r], rg, g->d_all[(size_t)r], world, r, g->d_local[(size_t)r])) return gfail(g, rc, cpprob_hip_last_error(g->ctx[(size_t)r]));
    }
    for (int t = sis ? g->T - 1 : (rg >= 0 ? rg : 0); t < g->T; ++t) {
        // (phase timing of a loopback run: the SUM over the ranks' launches of each phase -- they share the one stream)
        const bool timed = g->phase_profile && g->exchange && t + 1 < g->T && rg < 0;
        if (timed) ph_mark(g, 0, st);
        if (t != rg)
            for (int r = 0; r < world; ++r)
                if (int rc = cpprob_hip_smc_step_begin(g->ctx[(size_t)r], t, run_index, g->d_local[(size_t)r])) return gfail(g, rc, cpprob_hip_last_error(g->ctx[(size_t)r]));
        if (timed) ph_mark(g, 0, st);
        gather(t - vshift);
        if (timed) ph_mark(g, 0, st);
        for (int r = 0; r < world; ++r)
            if (int rc = cpprob_hip_smc_step_end(g->ctx[(size_t)r], t, g->d_all[(size_t)r], world, r)) return gfail(g, rc, cpprob_hip_last_error(g->ctx[(size_t)r]));
        if (timed) ph_mark(g, 0, st);
        if (g->exchange && t + 1 < g->T) {
            // direct transport: every rank's packing kernel stores into the other ranks' receive buffers; program order is the barrier
            for (int r = 0; r < world; ++r)
                if (int rc = cpprob_hip_exchange_pack_async(g->ctx[(size_t)r], t)) return gfail(g, rc, cpprob_hip_last_error(g->ctx[(size_t)r]));
            if (timed) ph_mark(g, 0, st);
            if (g->transport == kTransportSendRecv) {
                // (the fixed-capacity segments of the send/recv transport, as copies: A/B against the direct stores)
                for (int r = 0; r < world; ++r) {
                    cpprob_hip_ctx* c = g->ctx[(size_t)r];
                    const size_t seg = (size_t)c->x_cap * (size_t)group_record_len(g, t) * c->ssz;
                    for (size_t s = 0; s < c->x_peers.size(); ++s) {
                        cpprob_hip_ctx* p = g->ctx[(size_t)c->x_peers[s]];
                        size_t ps = 0;                                   // the slot the peer keeps for rank r
                        while (ps < p->x_peers.size() && p->x_peers[ps] != r) ++ps;
                        if (ps == p->x_peers.size()) return gfail(g, CPPROB_HIP_EDEVICE, "loopback transport: asymmetric peer sets");
                        HIP_TRY(c, hipMemcpyAsync(static_cast<char*>(p->d_xrecv) + ps * seg, static_cast<const char*>(c->d_xsend) + s * seg, seg, hipMemcpyDeviceToDevice, st));
                    }
                }
            }
            if (g->dev_coll && g->transport == kTransportDirect) {
                for (int r = 0; r < world; ++r) dc_barrier(g, r, t - vshift, cph::kDcPost, st);
                for (int r = 0; r < world; ++r) dc_barrier(g, r, t - vshift, cph::kDcWait, st);
            }
            if (timed) ph_mark(g, 0, st);
            for (int r = 0; r < world; ++r)
                if (int rc = cpprob_hip_exchange_commit_async(g->ctx[(size_t)r], t)) return gfail(g, rc, cpprob_hip_last_error(g->ctx[(size_t)r]));
            if (timed) { ph_mark(g, 0, st); ++g->ph_steps; }
        }
    }
    for (int r = 0; r < world; ++r) {
        cpprob_hip_ctx* c = g->ctx[(size_t)r];
        if (int rc = cpprob_hip_smc_finish(c)) return gfail(g, rc, cpprob_hip_last_error(c));
        launch_pack_joint(g, c, r, g->exchange && g->T > 1, g->d_joint[(size_t)r], st);
    }
    hipLaunchKernelGGL(loop_allreduce_kernel, dim3((unsigned)((group_joint_len(g) + 255) / 256)), dim3(256), 0, st, g->d_ptr_joints, world, (int)group_joint_len(g));
    HIP_TRY(c0, hipGetLastError());
    return 0;
}

void group_worker(cpprob_hip_group* g, int i)
{
    uint64_t seen = 0;
    for (;;) {
        uint64_t run;
        {
            std::unique_lock<std::mutex> lock(g->m);
            g->cv_job.wait(lock, [&] { return g->stopping || g->job_gen != seen; });
            if (g->stopping) return;
            seen = g->job_gen; run = g->job_run;
        }
        tl_group_err.clear();
        const int rc = run_rank(g, i, run);
        {
            std::lock_guard<std::mutex> lock(g->m);
            g->job_rc[(size_t)i] = rc;
            if (rc) g->job_err[(size_t)i] = tl_group_err.empty() ? g_last_error : tl_group_err;
            if (--g->job_pending == 0) g->cv_done.notify_all();
        }
    }
}

void close_direct(cpprob_hip_group* g)
{
    g->remote = false;
    for (auto* c : g->ctx) { (void)cpprob_hip_exchange_remote(c, nullptr); (void)cpprob_hip_exchange_direct(c, nullptr); }
    for (void* p : g->ipc_open) (void)hipIpcCloseMemHandle(p);
    g->ipc_open.clear();
}

// After the contexts are set up: let every rank's packing kernel store into its peers' receive buffers, where the machine allows
// it.  Loopback: plain pointers.  Several GPUs of one process: peer access.  One process per GPU: hipIpc handles, all-gathered.
// Every rank takes the same decision (the availability flags are all-gathered); anything short of "every rank can reach every
// peer" selects the send/recv transport for the whole group.
int setup_direct(cpprob_hip_group* g)
{
    const int n_local = (int)g->ctx.size(), world = g->world;
    g->transport = kTransportNone; g->transport_note.clear();
    if (!g->exchange) return 0;
    if (world == 1 && !g->world1_collectives) return 0;
    if (g->user_flags & CPPROB_HIP_GROUP_SENDRECV) {
        if (g->coll == kCollExternal) return gfail(g, CPPROB_HIP_EUNSUPPORTED, "external collectives carry no point-to-point transport: the lineages move by direct stores");
        g->transport = kTransportSendRecv; g->transport_note = "send/recv transport requested";
        return 0;
    }
    if (world == n_local) {
        // every rank in this process
        bool ok = true;
        if (!g->loopback()) {
            for (int i = 0; i < n_local && ok; ++i)
                for (int j = 0; j < n_local && ok; ++j) {
                    if (i == j) continue;
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, g->ctx[(size_t)i]->device, g->ctx[(size_t)j]->device) != hipSuccess || !can) { ok = false; break; }
                    (void)hipSetDevice(g->ctx[(size_t)i]->device);
                    const hipError_t e = hipDeviceEnablePeerAccess(g->ctx[(size_t)j]->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
                    (void)hipGetLastError();
                }
        }
        if (!ok) { g->transport = kTransportSendRecv; g->transport_note = "no peer access between the group's devices"; return 0; }
        for (int i = 0; i < n_local; ++i) {
            void* tab[cph::kWorldSlots] = {nullptr};
            for (int r = 0; r < world; ++r) tab[r] = g->ctx[(size_t)r]->d_xrecv;
            if (int rc = cpprob_hip_exchange_direct(g->ctx[(size_t)i], tab)) return gfail(g, rc, cpprob_hip_last_error(g->ctx[(size_t)i]));
        }
        g->transport = kTransportDirect;
        return 0;
    }
    // one rank per process: map the peers' receive buffers through hipIpc handles
    cpprob_hip_ctx* c = g->ctx[0];
    HIP_TRY(c, hipSetDevice(c->device));
    struct Rec { hipIpcMemHandle_t h; int32_t ok; int32_t pad; };
    Rec mine{};
    mine.ok = hipIpcGetMemHandle(&mine.h, c->d_xrecv) == hipSuccess ? 1 : 0;
    (void)hipGetLastError();
    std::vector<Rec> all((size_t)world);
    if (int rc = coll_allgather_host(g, 0, &mine, all.data(), sizeof(Rec))) return rc;
    bool ok = true;
    for (int r = 0; r < world; ++r) ok = ok && all[(size_t)r].ok;
    void* tab[cph::kWorldSlots] = {nullptr};
    int32_t opened = 1;
    if (ok) {
        for (int r : c->x_peers) {
            if (r == g->first_rank) { tab[r] = c->d_xrecv; continue; }          // (diagnostic self-peer)
            void* ptr = nullptr;
            if (hipIpcOpenMemHandle(&ptr, all[(size_t)r].h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { opened = 0; (void)hipGetLastError(); break; }
            g->ipc_open.push_back(ptr);
            tab[r] = ptr;
        }
    }
    // every rank must have mapped every peer, or nobody uses the mappings
    std::vector<int32_t> flags((size_t)world);
    const int32_t my_flag = ok ? opened : 0;
    if (int rc = coll_allgather_host(g, 0, &my_flag, flags.data(), sizeof(int32_t))) return rc;
    for (int r = 0; r < world; ++r) ok = ok && flags[(size_t)r];
    if (!ok) {
        close_direct(g);
        if (g->coll == kCollExternal) return gfail(g, CPPROB_HIP_EUNSUPPORTED, "external collectives need the direct transport, and a peer's receive buffer cannot be mapped (hipIpc)");
        g->transport = kTransportSendRecv; g->transport_note = "hipIpc mapping of a peer's receive buffer failed";
        return 0;
    }
    if (int rc = cpprob_hip_exchange_direct(c, tab)) return gfail(g, rc, cpprob_hip_last_error(c));
    g->transport = kTransportDirect;
    return 0;
}

// Remote lineages on top of the direct transport: every rank learns where every rank's particle store sits (same process: the
// pointers; other processes: hipIpc mappings of values / ancestors / origin table).  All or nobody: the availability is all-gathered.
int setup_remote(cpprob_hip_group* g)
{
    const int n_local = (int)g->ctx.size(), world = g->world;
    g->remote = false;
    if (g->transport != kTransportDirect || !g->exchange || g->cfg.keep_history == 0 || (g->user_flags & CPPROB_HIP_GROUP_SHIP_LINEAGES)) return 0;
    std::vector<cpprob_hip_store> st((size_t)world);
    if (world == n_local) {
        for (int r = 0; r < world; ++r)
            if (int rc = cpprob_hip_exchange_store(g->ctx[(size_t)r], &st[(size_t)r])) return gfail(g, rc, cpprob_hip_last_error(g->ctx[(size_t)r]));
        for (int i = 0; i < n_local; ++i)
            if (int rc = cpprob_hip_exchange_remote(g->ctx[(size_t)i], st.data())) return gfail(g, rc, cpprob_hip_last_error(g->ctx[(size_t)i]));
        g->remote = true;
        return 0;
    }
    cpprob_hip_ctx* c = g->ctx[0];
    HIP_TRY(c, hipSetDevice(c->device));
    struct Rec { hipIpcMemHandle_t hv, ha, ho, ht0, ht1; uint64_t rs, ld; int32_t ok, words; };
    Rec mine{};
    cpprob_hip_store own{};
    mine.ok = cpprob_hip_exchange_store(c, &own) == 0 ? 1 : 0;
    if (mine.ok) {
        mine.ok = hipIpcGetMemHandle(&mine.hv, const_cast<void*>(own.d_values)) == hipSuccess && hipIpcGetMemHandle(&mine.ha, const_cast<void*>(own.d_ancestors)) == hipSuccess &&
                  hipIpcGetMemHandle(&mine.ho, const_cast<void*>(own.d_origin)) == hipSuccess ? 1 : 0;
        (void)hipGetLastError();
        mine.rs = own.row_stride; mine.ld = own.n_local_columns;
        // (trace words are optional: every rank or none -- a rank that cannot export them switches them off for the group)
        mine.words = (own.d_trace[0] && own.d_trace[1] && hipIpcGetMemHandle(&mine.ht0, const_cast<void*>(own.d_trace[0])) == hipSuccess &&
                      hipIpcGetMemHandle(&mine.ht1, const_cast<void*>(own.d_trace[1])) == hipSuccess) ? 1 : 0;
        (void)hipGetLastError();
    }
    std::vector<Rec> all((size_t)world);
    if (int rc = coll_allgather_host(g, 0, &mine, all.data(), sizeof(Rec))) return rc;
    bool ok = true, words = true;
    for (int r = 0; r < world; ++r) { ok = ok && all[(size_t)r].ok; words = words && all[(size_t)r].words; }
    int32_t opened = 1;
    const size_t n_before = g->ipc_open.size();
    if (ok) {
        for (int r = 0; r < world && opened; ++r) {
            if (r == g->first_rank) { st[(size_t)r] = own; if (!words) { st[(size_t)r].d_trace[0] = nullptr; st[(size_t)r].d_trace[1] = nullptr; } continue; }
            void* pv = nullptr; void* pa = nullptr; void* po = nullptr;
            if (hipIpcOpenMemHandle(&pv, all[(size_t)r].hv, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { opened = 0; break; }
            g->ipc_open.push_back(pv);
            if (hipIpcOpenMemHandle(&pa, all[(size_t)r].ha, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { opened = 0; break; }
            g->ipc_open.push_back(pa);
            if (hipIpcOpenMemHandle(&po, all[(size_t)r].ho, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { opened = 0; break; }
            g->ipc_open.push_back(po);
            st[(size_t)r].d_values = pv; st[(size_t)r].d_ancestors = pa; st[(size_t)r].d_origin = po;
            st[(size_t)r].row_stride = all[(size_t)r].rs; st[(size_t)r].n_local_columns = all[(size_t)r].ld;
            st[(size_t)r].d_trace[0] = nullptr; st[(size_t)r].d_trace[1] = nullptr;
            if (words) {
                void* p0 = nullptr; void* p1 = nullptr;
                if (hipIpcOpenMemHandle(&p0, all[(size_t)r].ht0, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { opened = 0; break; }
                g->ipc_open.push_back(p0);
                if (hipIpcOpenMemHandle(&p1, all[(size_t)r].ht1, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { opened = 0; break; }
                g->ipc_open.push_back(p1);
                st[(size_t)r].d_trace[0] = p0; st[(size_t)r].d_trace[1] = p1;
            }
        }
        (void)hipGetLastError();
    }
    std::vector<int32_t> flags((size_t)world);
    const int32_t my_flag = ok ? opened : 0;
    if (int rc = coll_allgather_host(g, 0, &my_flag, flags.data(), sizeof(int32_t))) return rc;
    for (int r = 0; r < world; ++r) ok = ok && flags[(size_t)r];
    if (!ok) {
        // (the direct transport stays; migrants keep taking their lineages along)
        while (g->ipc_open.size() > n_before) { (void)hipIpcCloseMemHandle(g->ipc_open.back()); g->ipc_open.pop_back(); }
        g->transport_note = "remote lineages unavailable: a rank's particle store cannot be mapped (hipIpc)";
        return 0;
    }
    if (int rc = cpprob_hip_exchange_remote(c, st.data())) return gfail(g, rc, cpprob_hip_last_error(c));
    g->remote = true;
    return 0;
}

// Mailboxes for the per-step collectives: one per rank, mapped into every rank like the direct transport's buffers, tried once per
// group and proven by a round trip before any run relies on them.  All or nobody (the outcome is all-gathered).
int setup_mailboxes(cpprob_hip_group* g)
{
    if (g->dc_tried) {                                   // (asking for the library's collectives later switches the mailboxes off; nothing switches them back on)
        if (g->user_flags & CPPROB_HIP_GROUP_LIBRARY_COLLECTIVES) { g->dev_coll = false; g->dc_note = "library collectives requested"; }
        return 0;
    }
    g->dc_tried = true; g->dev_coll = false;
    const int n_local = (int)g->ctx.size(), world = g->world;
    const bool talk = world > 1 || g->world1_collectives;
    if (!talk) return 0;
    if (g->user_flags & CPPROB_HIP_GROUP_LIBRARY_COLLECTIVES) { g->dc_note = "library collectives requested"; return 0; }
    if (g->loopback() && !(g->user_flags & CPPROB_HIP_GROUP_MAILBOX_COLLECTIVES)) return 0;       // (one launch gathers for every loopback rank)
    g->d_box.assign((size_t)n_local, nullptr); g->d_box_peers.assign((size_t)n_local, nullptr); g->d_dc_status.assign((size_t)n_local, nullptr);
    bool ok = true;
    for (int i = 0; i < n_local && ok; ++i) {
        cpprob_hip_ctx* c = g->ctx[(size_t)i];
        GHIP_TRY(g, hipSetDevice(c->device));
        void* p = nullptr;
        // (fine-grained: the peers' stores and this rank's spinning loads meet in memory, not in a cache)
        if (hipExtMallocWithFlags(&p, sizeof(cph::Mailbox), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            // ranks on ONE device and ONE stream (loopback) meet in that device's cache hierarchy: ordinary memory serves them.  Between
            // devices a spin on coarse-grained memory may never see the peer's store (it would end in the wait's time-out, run after run):
            // without fine-grained memory the mailboxes stay off and the library's collectives carry the steps -- said in the traffic record
            if (!g->loopback() || hipMalloc(&p, sizeof(cph::Mailbox)) != hipSuccess) { (void)hipGetLastError(); ok = false; g->dc_note = "no fine-grained device memory for the mailboxes: the library's collectives carry the steps"; break; }
        }
        g->d_box[(size_t)i] = static_cast<cph::Mailbox*>(p);
        GHIP_TRY(g, hipMemset(p, 0, sizeof(cph::Mailbox)));
        GHIP_TRY(g, hipMalloc(&g->d_box_peers[(size_t)i], sizeof(cph::MailboxPeers)));
        GHIP_TRY(g, hipMalloc(&g->d_dc_status[(size_t)i], 4 * sizeof(int32_t)));         // {status, -, ticks the waits spun: 64 bits}
        GHIP_TRY(g, hipMemset(g->d_dc_status[(size_t)i], 0, 4 * sizeof(int32_t)));
    }
    std::vector<cph::MailboxPeers> tab((size_t)n_local);
    if (ok && world == n_local) {
        if (!g->loopback())
            for (int i = 0; i < n_local && ok; ++i)
                for (int j = 0; j < n_local && ok; ++j) {
                    if (i == j) continue;
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, g->ctx[(size_t)i]->device, g->ctx[(size_t)j]->device) != hipSuccess || !can) { ok = false; break; }
                    (void)hipSetDevice(g->ctx[(size_t)i]->device);
                    const hipError_t e = hipDeviceEnablePeerAccess(g->ctx[(size_t)j]->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
                    (void)hipGetLastError();
                }
        for (int i = 0; i < n_local; ++i)
            for (int r = 0; r < world; ++r) tab[(size_t)i].box[r] = g->d_box[(size_t)r];
    } else if (world != n_local) {
        // one rank per process: hipIpc handles of the mailboxes, all-gathered
        cpprob_hip_ctx* c = g->ctx[0];
        struct Rec { hipIpcMemHandle_t h; int32_t ok; int32_t pad; };
        Rec mine{};
        mine.ok = (ok && hipIpcGetMemHandle(&mine.h, g->d_box[0]) == hipSuccess) ? 1 : 0;
        (void)hipGetLastError();
        std::vector<Rec> all((size_t)world);
        if (int rc = coll_allgather_host(g, 0, &mine, all.data(), sizeof(Rec))) return rc;
        for (int r = 0; r < world; ++r) ok = ok && all[(size_t)r].ok;
        int32_t opened = ok ? 1 : 0;
        if (ok) {
            GHIP_TRY(g, hipSetDevice(c->device));
            for (int r = 0; r < world; ++r) {
                if (r == g->first_rank) { tab[0].box[r] = g->d_box[0]; continue; }
                void* ptr = nullptr;
                if (hipIpcOpenMemHandle(&ptr, all[(size_t)r].h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { opened = 0; (void)hipGetLastError(); break; }
                g->ipc_box_open.push_back(ptr);
                tab[0].box[r] = static_cast<cph::Mailbox*>(ptr);
            }
        }
        std::vector<int32_t> flags((size_t)world);
        if (int rc = coll_allgather_host(g, 0, &opened, flags.data(), sizeof(int32_t))) return rc;
        for (int r = 0; r < world; ++r) ok = ok && flags[(size_t)r];
    }
    if (ok)
        for (int i = 0; i < n_local; ++i) {
            GHIP_TRY(g, hipSetDevice(g->ctx[(size_t)i]->device));
            GHIP_TRY(g, hipMemcpy(g->d_box_peers[(size_t)i], &tab[(size_t)i], sizeof(cph::MailboxPeers), hipMemcpyHostToDevice));
        }
    // the proof: one all-gather of {rank + 1, 7, 9} through the mailboxes, checked on the host
    int32_t proven = ok ? 1 : 0;
    if (ok) {
        g->dc_serial = 1;
        std::vector<unsigned long long> got((size_t)3 * (size_t)world);
        for (int i = 0; i < n_local; ++i) {
            const unsigned long long w[3] = {(unsigned long long)(g->first_rank + i + 1), 7ull, 9ull};
            GHIP_TRY(g, hipSetDevice(g->ctx[(size_t)i]->device));
            GHIP_TRY(g, hipMemcpy(g->d_local[(size_t)i], w, sizeof w, hipMemcpyHostToDevice));
        }
        const int t_probe = 0x7ffffffd;                         // (a step number no run uses: the step field of a sequence number is 32 bits wide)
        if (g->loopback()) {
            for (int i = 0; i < n_local; ++i) dc_allgather(g, i, t_probe, cph::kDcPost, g->ctx[0]->stream);
            for (int i = 0; i < n_local; ++i) dc_allgather(g, i, t_probe, cph::kDcWait, g->ctx[0]->stream);
        } else {
            for (int i = 0; i < n_local; ++i) { GHIP_TRY(g, hipSetDevice(g->ctx[(size_t)i]->device)); dc_allgather(g, i, t_probe, cph::kDcPost | cph::kDcWait, g->ctx[(size_t)i]->stream); }
        }
        for (int i = 0; i < n_local; ++i) {
            cpprob_hip_ctx* c = g->ctx[(size_t)i];
            GHIP_TRY(g, hipSetDevice(c->device));
            int32_t st = 0;
            GHIP_TRY(g, hipMemcpyAsync(got.data(), g->d_all[(size_t)i], got.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            GHIP_TRY(g, hipMemcpyAsync(&st, g->d_dc_status[(size_t)i], sizeof st, hipMemcpyDeviceToHost, c->stream));
            GHIP_TRY(g, hipStreamSynchronize(c->stream));
            if (st != 0) proven = 0;
            for (int r = 0; r < world; ++r)
                if (got[(size_t)3 * r] != (unsigned long long)(r + 1) || got[(size_t)3 * r + 1] != 7ull || got[(size_t)3 * r + 2] != 9ull) proven = 0;
            GHIP_TRY(g, hipMemset(g->d_dc_status[(size_t)i], 0, sizeof(int32_t)));
            GHIP_TRY(g, hipMemset(g->d_local[(size_t)i], 0, 4 * sizeof(double)));
        }
    }
    if (world != n_local) {
        std::vector<int32_t> flags((size_t)world);
        if (int rc = coll_allgather_host(g, 0, &proven, flags.data(), sizeof(int32_t))) return rc;
        for (int r = 0; r < world; ++r) proven = proven && flags[(size_t)r];
    }
    g->dev_coll = proven != 0;
    if (!g->dev_coll) g->dc_note = ok ? "the mailbox round trip failed: the library's collectives carry the steps" : "a rank's mailbox cannot be mapped: the library's collectives carry the steps";
    return 0;
}

int group_begin_contexts(cpprob_hip_group* g)
{
    const int n_local = (int)g->ctx.size();
    close_direct(g);                                     // (the peers' buffers may be reallocated below)
    for (int i = 0; i < n_local; ++i) {
        const int rank = g->first_rank + i;
        cpprob_hip_config c = g->cfg;
        c.n_particles = g->shard_begin[(size_t)rank + 1] - g->shard_begin[(size_t)rank];
        c.particle_offset = g->shard_begin[(size_t)rank];
        c.n_global = g->shard_begin[(size_t)g->world];
        c.resample_scope = g->exchange ? CPPROB_HIP_SCOPE_EXCHANGE : CPPROB_HIP_SCOPE_GLOBAL;
        c.annex_kcols = g->annex_kcols;
        cpprob_hip_ctx* x = g->ctx[(size_t)i];
        if (int rc = cpprob_hip_infer_begin(x, &c, g->obs.data(), g->obs.size())) return gfail(g, rc, cpprob_hip_last_error(x));
        if (g->exchange) {
            const int peers_mode = (g->world == 1 && g->world1_collectives) ? 2 : g->all_peers;     // 2: this rank is its own peer (diagnostic)
            if (int rc = cpprob_hip_exchange_setup(x, g->world, rank, g->shard_begin.data(), peers_mode, g->cap)) return gfail(g, rc, cpprob_hip_last_error(x));
        }
    }
    if (int rc = setup_direct(g)) return rc;
    if (int rc = setup_remote(g)) return rc;
    if (g->exchange && g->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL && !g->remote && (g->world > 1 || g->world1_collectives))
        return gfail(g, CPPROB_HIP_EUNSUPPORTED, "multinomial resampling of a joint population moves its migrants by remote lineages (every rank addresses every rank's particle "
                                                  "store); this group cannot: " + (g->transport_note.empty() ? std::string("the segment transports were requested") : g->transport_note));
    return 0;
}

int group_enqueue(cpprob_hip_group* g, uint64_t run_index)
{
    ph_begin(g);
    ++g->dc_serial;                                     // (every rank enqueues the same runs in the same order: the mailboxes' sequence numbers agree)
    if (g->loopback()) return loopback_run(g, run_index);
    if (g->ctx.size() == 1) return run_rank(g, 0, run_index);
    {
        std::lock_guard<std::mutex> lock(g->m);
        g->job_run = run_index; g->job_pending = (int)g->ctx.size(); ++g->job_gen;
        std::fill(g->job_rc.begin(), g->job_rc.end(), 0);
    }
    g->cv_job.notify_all();
    std::unique_lock<std::mutex> lock(g->m);
    g->cv_done.wait(lock, [&] { return g->job_pending == 0; });
    for (size_t i = 0; i < g->ctx.size(); ++i)
        if (g->job_rc[i]) return gfail(g, g->job_rc[i], g->job_err[i]);
    return 0;
}

// bytes the transport put on the links in one run (host-side bookkeeping: the sizes of the send/recv transport are host constants)
double sendrecv_wire_bytes(const cpprob_hip_group* g)
{
    double total = 0.0;
    const double bpv = (double)g->ctx[0]->ssz;
    for (int r = 0; r < g->world; ++r) {
        int np = 0;
        for (int q = 0; q < g->world; ++q) if (q != r && (g->all_peers || q == r - 1 || q == r + 1)) ++np;
        if (g->world == 1 && g->world1_collectives) np = 1;
        for (int t = 0; t + 1 < g->T; ++t) total += (double)np * (double)g->cap * (double)group_record_len(g, t) * bpv;
    }
    return total;
}

}  // namespace

extern "C" {

int cpprob_hip_group_unique_id(void* out, size_t n_bytes)
{
    if (!out || n_bytes < NCCL_UNIQUE_ID_BYTES) return fail(nullptr, CPPROB_HIP_EINVAL, "need a buffer of 128 bytes");
    std::string err;
    RcclApi* api = rccl_api(err);
    if (!api) return fail(nullptr, CPPROB_HIP_EDEVICE, err);
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, CPPROB_HIP_EDEVICE, std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
    std::memcpy(out, &id, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

static int group_create_impl(const int32_t* devices, int32_t n_local, int32_t world, int32_t first_rank, const void* unique_id,
                             const cpprob_hip_collectives* ext, cpprob_hip_group** out)
{
    if (!out || !devices) return fail(nullptr, CPPROB_HIP_EINVAL, "NULL argument");
    *out = nullptr;
    if (n_local < 1 || world < n_local || world > cph::kMaxWorld || first_rank < 0 || first_rank + n_local > world)
        return fail(nullptr, CPPROB_HIP_EINVAL, "need 1 <= n_local <= world <= 63 and first_rank + n_local <= world");
    if (!ext && world > n_local && (n_local != 1 || !unique_id)) return fail(nullptr, CPPROB_HIP_EINVAL, "ranks in other processes: one GPU per process and the group's unique id");
    if (ext && (n_local != 1 || !ext->allgather)) return fail(nullptr, CPPROB_HIP_EINVAL, "external collectives: one rank per group handle and an all-gather callback");
    cpprob_hip_group* g = new cpprob_hip_group();
    g->world = world; g->first_rank = first_rank;
    bool same = n_local > 1;
    for (int i = 1; i < n_local; ++i) same = same && devices[i] == devices[0];
    g->coll = ext ? kCollExternal : ((same && world == n_local) ? kCollLoopback : kCollRccl);
    if (ext) g->ext = *ext;
    if (g->coll == kCollRccl && world == n_local)
        for (int i = 0; i < n_local; ++i)
            for (int j = 0; j < i; ++j)
                if (devices[i] == devices[j]) { delete g; return fail(nullptr, CPPROB_HIP_EINVAL, "a device may appear once (RCCL) or every rank sits on the same device (loopback)"); }
    auto bail = [&](int rc) { const std::string e = g->err; cpprob_hip_group_destroy(g); g_last_error = e; return rc; };
    for (int i = 0; i < n_local; ++i) {
        cpprob_hip_ctx* c = nullptr;
        if (int rc = cpprob_hip_create(devices[i], &c)) { g->err = g_last_error; return bail(rc); }
        g->ctx.push_back(c);
    }
    if (g->loopback()) {
        // one stream for every rank: program order is the only synchronisation the loopback collectives need
        for (auto* c : g->ctx) g->own_stream.push_back(c->stream);
        for (auto* c : g->ctx) c->stream = g->ctx[0]->stream;
    } else if (g->coll == kCollRccl) {
        std::string err;
        g->rccl = rccl_api(err);
        if (!g->rccl) { g->err = err; return bail(CPPROB_HIP_EDEVICE); }
        g->comm.assign((size_t)n_local, nullptr);
        ncclResult_t r;
        if (world == n_local) {
            std::vector<int> devs(devices, devices + n_local);
            r = g->rccl->CommInitAll(g->comm.data(), n_local, devs.data());
        } else {
            ncclUniqueId id;
            std::memcpy(&id, unique_id, NCCL_UNIQUE_ID_BYTES);
            (void)hipSetDevice(devices[0]);
            r = g->rccl->CommInitRank(&g->comm[0], world, id, first_rank);
        }
        if (r != ncclSuccess) { g->err = std::string("RCCL communicator: ") + g->rccl->GetErrorString(r); g->comm.clear(); return bail(CPPROB_HIP_EDEVICE); }
    }
    *out = g;
    return 0;
}

int cpprob_hip_group_create(const int32_t* devices, int32_t n_local, int32_t world, int32_t first_rank, const void* unique_id, cpprob_hip_group** out)
{
    return group_create_impl(devices, n_local, world, first_rank, unique_id, nullptr, out);
}

int cpprob_hip_group_create_external(int32_t device, int32_t world, int32_t rank, const cpprob_hip_collectives* collectives, cpprob_hip_group** out)
{
    if (!collectives) return fail(nullptr, CPPROB_HIP_EINVAL, "NULL argument");
    return group_create_impl(&device, 1, world, rank, nullptr, collectives, out);
}

void cpprob_hip_group_destroy(cpprob_hip_group* g)
{
    if (!g) return;
    if (!g->workers.empty()) {
        { std::lock_guard<std::mutex> lock(g->m); g->stopping = true; }
        g->cv_job.notify_all();
        for (auto& w : g->workers) w.join();
    }
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->ctx[i]->device);
        (void)hipStreamSynchronize(g->ctx[i]->stream);
    }
    close_direct(g);
    for (void* p : g->ipc_box_open) (void)hipIpcCloseMemHandle(p);
    for (size_t i = 0; i < g->d_box.size(); ++i) {
        (void)hipSetDevice(g->ctx[i]->device);
        if (g->d_box[i]) (void)hipFree(g->d_box[i]);
        if (g->d_box_peers[i]) (void)hipFree(g->d_box_peers[i]);
        if (g->d_dc_status[i]) (void)hipFree(g->d_dc_status[i]);
    }
    for (size_t i = 0; i < g->comm.size(); ++i)
        if (g->comm[i]) (void)g->rccl->CommDestroy(g->comm[i]);
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->ctx[i]->device);
        if (i < g->d_local.size()) { (void)hipFree(g->d_local[i]); (void)hipFree(g->d_all[i]); (void)hipFree(g->d_joint[i]); (void)hipFree(g->d_bar[i]); }
        if (g->loopback() && i < g->own_stream.size()) g->ctx[i]->stream = g->own_stream[i];
    }
    if (g->d_ptr_locals) { (void)hipFree((void*)g->d_ptr_locals); (void)hipFree((void*)g->d_ptr_alls); (void)hipFree((void*)g->d_ptr_joints); }
    for (auto* c : g->ctx) cpprob_hip_destroy(c);
    delete g;
}

const char* cpprob_hip_group_last_error(const cpprob_hip_group* g) { return g ? g->err.c_str() : g_last_error.c_str(); }

int cpprob_hip_group_begin(cpprob_hip_group* g, const cpprob_hip_config* cfg, const double* h_observes, size_t n_observes, const uint64_t* h_shard_sizes)
{
    if (!g || !cfg || !h_observes) return gkeep(g, gfail(g, CPPROB_HIP_EINVAL, "NULL argument"));
    if (cfg->n_particles < (uint64_t)g->world) return gkeep(g, gfail(g, CPPROB_HIP_EINVAL, "fewer particles than ranks"));
    g->begun = false;                                   // (a failed begin leaves no half-updated configuration behind an earlier one)
    g->cfg = *cfg;
    g->obs.assign(h_observes, h_observes + n_observes);
    g->shard_begin.assign((size_t)g->world + 1, 0);
    uint64_t largest = 0;
    for (int r = 0; r < g->world; ++r) {
        // contiguous shards: particle i lives on rank floor(i * world / N) unless the caller names the sizes
        const uint64_t base = cfg->n_particles / (uint64_t)g->world, rem = cfg->n_particles % (uint64_t)g->world;
        const uint64_t sz = h_shard_sizes ? h_shard_sizes[r] : base + ((uint64_t)r < rem ? 1 : 0);
        if (sz == 0) return gkeep(g, gfail(g, CPPROB_HIP_EINVAL, "empty shard"));
        g->shard_begin[(size_t)r + 1] = g->shard_begin[(size_t)r] + sz;
        largest = std::max(largest, sz);
    }
    if (g->shard_begin[(size_t)g->world] != cfg->n_particles) return gkeep(g, gfail(g, CPPROB_HIP_EINVAL, "shard sizes do not add up to n_particles"));
    // one joint population, every resampler: systematic and stratified (one interval of outputs per rank) and thesis Alg. 1's multinomial
    // in its strata form (one interval + the strata the ranks' boundaries cut: strata_cut.hpp; histories kept, remote lineages).  The
    // literal multinomial form and a filtering-only multinomial run keep the global scope (shard-local resampling, carried mass).
    const bool strata = cfg->resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL && !(cfg->flags & (CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL | CPPROB_HIP_FLAG_FLOATING_POINT_STEP)) && cfg->keep_history != 0;
    g->exchange = cfg->algorithm == CPPROB_HIP_ALG_SMC && (cfg->resampler == CPPROB_HIP_RESAMPLE_SYSTEMATIC || cfg->resampler == CPPROB_HIP_RESAMPLE_STRATIFIED || strata);
    // transport defaults: the two neighbours, room for the O(sqrt(N)) outputs a rank's offspring interval leaves its shard by
    g->all_peers = 0;
    const uint64_t guess = (uint64_t)(8.0 * std::sqrt((double)cfg->n_particles)) / kTile * kTile + 4 * kTile;
    g->cap = std::min<uint64_t>(largest, guess);
    g->annex_kcols = 0;
    if (g->user_cap) g->cap = g->user_cap;                // cpprob_hip_group_transport
    if (g->user_all_peers >= 0) g->all_peers = g->user_all_peers;
    g->world1_collectives = (g->user_flags & CPPROB_HIP_GROUP_WORLD1_COLLECTIVES) && g->world == 1 && !g->loopback();
    const int n_local = (int)g->ctx.size();
    if (g->d_local.empty()) { g->d_local.assign((size_t)n_local, nullptr); g->d_all.assign((size_t)n_local, nullptr); g->d_joint.assign((size_t)n_local, nullptr); g->d_bar.assign((size_t)n_local, nullptr); }
    if (int rc = group_begin_contexts(g)) return gkeep(g, rc);
    g->T = g->ctx[0]->T; g->K = g->ctx[0]->K; g->n_stats = g->T * g->K;
    for (int i = 0; i < n_local; ++i) {
        cpprob_hip_ctx* c = g->ctx[(size_t)i];
        GHIP_TRY(g, hipSetDevice(c->device));
        if (g->d_local[(size_t)i]) { (void)hipFree(g->d_local[(size_t)i]); (void)hipFree(g->d_all[(size_t)i]); (void)hipFree(g->d_joint[(size_t)i]); (void)hipFree(g->d_bar[(size_t)i]); }
        const size_t nj = group_joint_len(g);
        GHIP_TRY(g, hipMalloc(&g->d_local[(size_t)i], 4 * sizeof(double)));
        GHIP_TRY(g, hipMalloc(&g->d_all[(size_t)i], 3 * (size_t)g->world * sizeof(double)));
        GHIP_TRY(g, hipMalloc(&g->d_joint[(size_t)i], nj * sizeof(double)));
        GHIP_TRY(g, hipMalloc(&g->d_bar[(size_t)i], ((size_t)g->world + 1) * sizeof(double)));
        GHIP_TRY(g, hipMemset(g->d_local[(size_t)i], 0, 4 * sizeof(double)));
        GHIP_TRY(g, hipMemset(g->d_all[(size_t)i], 0, 3 * (size_t)g->world * sizeof(double)));
        GHIP_TRY(g, hipMemset(g->d_joint[(size_t)i], 0, nj * sizeof(double)));
        GHIP_TRY(g, hipMemset(g->d_bar[(size_t)i], 0, ((size_t)g->world + 1) * sizeof(double)));
    }
    if (int rc = setup_mailboxes(g)) return gkeep(g, rc);    // (once per group; uses the buffers above for its proof)
    if (g->loopback()) {
        GHIP_TRY(g, hipSetDevice(g->ctx[0]->device));
        if (g->d_ptr_locals) { (void)hipFree((void*)g->d_ptr_locals); (void)hipFree((void*)g->d_ptr_alls); (void)hipFree((void*)g->d_ptr_joints); }
        void *pl = nullptr, *pa = nullptr, *pj = nullptr;
        GHIP_TRY(g, hipMalloc(&pl, (size_t)n_local * sizeof(double*)));
        GHIP_TRY(g, hipMalloc(&pa, (size_t)n_local * sizeof(double*)));
        GHIP_TRY(g, hipMalloc(&pj, (size_t)n_local * sizeof(double*)));
        GHIP_TRY(g, hipMemcpy(pl, g->d_local.data(), (size_t)n_local * sizeof(double*), hipMemcpyHostToDevice));
        GHIP_TRY(g, hipMemcpy(pa, g->d_all.data(), (size_t)n_local * sizeof(double*), hipMemcpyHostToDevice));
        GHIP_TRY(g, hipMemcpy(pj, g->d_joint.data(), (size_t)n_local * sizeof(double*), hipMemcpyHostToDevice));
        g->d_ptr_locals = static_cast<double* const*>(pl); g->d_ptr_alls = static_cast<double* const*>(pa); g->d_ptr_joints = static_cast<double* const*>(pj);
    } else if (n_local > 1 && g->workers.empty()) {
        g->job_rc.assign((size_t)n_local, 0); g->job_err.assign((size_t)n_local, "");
        for (int i = 0; i < n_local; ++i) g->workers.emplace_back(group_worker, g, i);
    }
    g->begun = true; g->ran = false; g->reruns = 0;
    return 0;
}

int cpprob_hip_group_transport(cpprob_hip_group* g, uint64_t records_per_peer, int32_t all_peers, uint32_t flags)
{
    if (!g) return fail(nullptr, CPPROB_HIP_EINVAL, "group is NULL");
    g->user_cap = records_per_peer; g->user_all_peers = all_peers < 0 ? -1 : (all_peers ? 1 : 0); g->user_flags = flags;
    return 0;
}

int cpprob_hip_group_run(cpprob_hip_group* g, uint64_t run_index)
{
    if (!g) return fail(nullptr, CPPROB_HIP_EINVAL, "group is NULL");
    if (!g->begun) return gkeep(g, gfail(g, CPPROB_HIP_ESTATE, "cpprob_hip_group_begin has not been called"));
    g->repair_gen = -1; g->n_requantised = 0;
    if (int rc = group_enqueue(g, run_index)) return gkeep(g, rc);
    g->last_run = run_index; g->ran = true;
    return 0;
}

int cpprob_hip_group_sync(cpprob_hip_group* g)
{
    if (!g) return fail(nullptr, CPPROB_HIP_EINVAL, "group is NULL");
    for (auto* c : g->ctx)
        if (int rc = cpprob_hip_sync(c)) return gfail(g, rc, cpprob_hip_last_error(c));
    return 0;
}

int cpprob_hip_group_size(const cpprob_hip_group* g, int32_t* world, int32_t* n_local, int32_t* first_rank)
{
    if (!g) return fail(nullptr, CPPROB_HIP_EINVAL, "group is NULL");
    if (world) *world = g->world;
    if (n_local) *n_local = (int32_t)g->ctx.size();
    if (first_rank) *first_rank = g->first_rank;
    return 0;
}

cpprob_hip_ctx* cpprob_hip_group_context(cpprob_hip_group* g, int32_t local_index)
{
    return (g && local_index >= 0 && (size_t)local_index < g->ctx.size()) ? g->ctx[(size_t)local_index] : nullptr;
}

int cpprob_hip_group_results(cpprob_hip_group* g, cpprob_hip_summary* out, double* h_stats, size_t n_doubles, int32_t* h_reruns)
{
    if (!g) return fail(nullptr, CPPROB_HIP_EINVAL, "group is NULL");
    if (!g->ran) return gkeep(g, gfail(g, CPPROB_HIP_ESTATE, "no finished run"));
    if (h_stats && n_doubles < (size_t)g->n_stats) return gkeep(g, gfail(g, CPPROB_HIP_EINVAL, "h_stats too small"));
    std::vector<double> joint(group_joint_len(g));
    int enlargements = 0;
    for (int attempt = 0;; ++attempt) {
        if (int rc = cpprob_hip_group_sync(g)) return gkeep(g, rc);
        cpprob_hip_ctx* c = g->ctx[0];
        GHIP_TRY(g, hipSetDevice(c->device));
        GHIP_TRY(g, hipMemcpy(joint.data(), g->d_joint[0], joint.size() * sizeof(double), hipMemcpyDeviceToHost));
        // First the flag that is the same on every rank by construction (all-reduced with the run's results): some rank's mailbox
        // wait gave up.  That rank planned its steps on stale totals, so its gap, masses and squares may differ from its peers' -- every
        // rank discards the run, the library's collectives carry the steps from here on, and the run is repeated on them.  (Testing
        // the per-rank precision gap first could send the ranks down different branches -- one into begin + a host all-gather of
        // hipIpc records, another into a 24-byte all-gather: mismatched collectives, a hang where a fall-back was meant.)
        if (joint[(size_t)g->n_stats + 3] != 0.0) {
            if (attempt >= 6) return gkeep(g, gfail(g, CPPROB_HIP_EDEVICE, "run repeated too often"));
            g->dev_coll = false; g->dc_note = "a mailbox collective timed out: the library's collectives carry the steps";
            for (size_t i = 0; i < g->d_dc_status.size(); ++i) { GHIP_TRY(g, hipSetDevice(g->ctx[i]->device)); GHIP_TRY(g, hipMemset(g->d_dc_status[i], 0, sizeof(int32_t))); }
            for (auto* x : g->ctx) x->fixed_check_pending = false;
            ++g->reruns;
            g->repair_gen = -1; g->n_requantised = 0;
            if (int rc = group_enqueue(g, g->last_run)) return gkeep(g, rc);
            continue;
        }
        if (joint[(size_t)g->n_stats] != 0.0) {
            // some rank's transport was too small (every rank sees the same all-reduced flags and takes the same decision): repeat
            // the run with what overflowed enlarged -- the annex (x4), the peer segments (x4, up to a whole shard), the peer list
            // (every rank).  Results do not depend on the transport parameters, only their validity does.
            // (first of all: a run whose transport overflowed holds nothing worth repairing -- and a run resumed behind a repaired generation keeps
            //  the sticky flag of the steps in front of it, so only a run that did not overflow is ever repaired)
            if (++enlargements > 8) return gkeep(g, gfail(g, CPPROB_HIP_EDEVICE, "exchange transport still too small after eight enlargements"));
            const long long v = (long long)joint[(size_t)g->n_stats];
            const bool seg = v % 128 != 0, peers = (v / 128) % 128 != 0, annex = (v / 16384) % 128 != 0;
            if (v / 2097152 != 0)
                return gkeep(g, gfail(g, CPPROB_HIP_EUNSUPPORTED, "multinomial resampling: a stratum cut by a rank boundary holds more thresholds than the cut table (8192; a stratum's "
                                                                   "count is Binomial(N, 1/K) with mean <= 256)"));
            uint64_t largest = 0;
            for (int r = 0; r < g->world; ++r) largest = std::max(largest, g->shard_begin[(size_t)r + 1] - g->shard_begin[(size_t)r]);
            if (peers) g->all_peers = 1;
            if (seg) g->cap = std::min<uint64_t>(largest, std::max<uint64_t>(g->cap * 4, 16384));
            if (annex || (!seg && !peers)) {
                // (0 = the context's default: max(a sixteenth of the shard, sqrt(N) T, four tiles) -- cpprob_hip_infer_begin)
                const uint64_t mixed = (uint64_t)(std::sqrt((double)g->cfg.n_particles) * (double)g->T) / 1024 + (g->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL ? (uint64_t)g->T * 3 / 2 : 0);
                const int in_use = g->annex_kcols > 0 ? g->annex_kcols : (int)std::max<uint64_t>(std::max<uint64_t>(4, ((largest + 1023) / 1024) / 16), mixed);
                g->annex_kcols = in_use * 4;
            }
            if (int rc = group_begin_contexts(g)) return gkeep(g, rc);
            ++g->reruns;
            g->repair_gen = -1; g->n_requantised = 0;
            for (auto* x : g->ctx) x->fixed_check_pending = false;
            if (int rc = group_enqueue(g, g->last_run)) return gkeep(g, rc);
            continue;
        }
        // fixed-point form: did the weights keep their bits (cpprob_hip.hip: settle_fixed)?  On a run whose collectives all completed
        // every rank holds the same gap and the same first offending generation (they come from the all-gathered totals), so every rank
        // takes the same decision
        bool imprecise = false;
        int first_bad = -1;
        if (c->fixed_check_pending) {
            cph::StepCtrl hc{};
            GHIP_TRY(g, hipMemcpy(&hc, c->d_ctrl, sizeof hc, hipMemcpyDeviceToHost));
            imprecise = !(hc.fix_gap <= kFixGapLimit);
            first_bad = hc.first_bad;
        }
        for (auto* x : g->ctx) x->fixed_check_pending = false;
        if (imprecise && c->keep && !(g->cfg.flags & CPPROB_HIP_FLAG_REPEAT_IN_FLOATING_POINT) && first_bad >= 0 && first_bad < g->T && first_bad > g->repair_gen) {
            // Repair, in integers, as a single context does (cpprob_hip.hip: settle_fixed): the first offending generation is weighed
            // again against the POPULATION's exact maximum and the steps behind it run again -- the sharded run stays the one-GPU
            // run, bit for bit.  Every rank holds the same books, so every rank enters the same collectives.
            if (g->n_requantised > g->T) return gkeep(g, gfail(g, CPPROB_HIP_EDEVICE, "run repaired too often"));
            g->repair_gen = first_bad;
            const int rc = group_enqueue(g, g->last_run);
            if (rc) { g->repair_gen = -1; return gkeep(g, rc); }
            ++g->n_requantised;
            continue;
        }
        if (imprecise && g->cfg.resampler != CPPROB_HIP_RESAMPLE_SYSTEMATIC && g->exchange) {
            // (the floating-point form of the exchange scope plans systematic offspring intervals only)
            g->repair_gen = -1;
            return gkeep(g, gfail(g, CPPROB_HIP_EPRECISION, "a generation's heaviest particle sat more than 6 nats below its fixed-point reference and could not be repaired in the run"));
        }
        if (imprecise) {
            if (attempt >= 6) return gkeep(g, gfail(g, CPPROB_HIP_EDEVICE, "run repeated too often"));
            g->repair_gen = -1;
            g->cfg.flags |= CPPROB_HIP_FLAG_FLOATING_POINT_STEP;
            if (int rc = group_begin_contexts(g)) return gkeep(g, rc);
            ++g->reruns;
            if (int rc = group_enqueue(g, g->last_run)) return gkeep(g, rc);
            continue;
        }
        break;
    }
    g->repair_gen = -1;
    if (h_reruns) *h_reruns = g->reruns;
    g->traffic.records = (uint64_t)joint[(size_t)g->n_stats + 1];
    g->traffic.payload_bytes = (uint64_t)joint[(size_t)g->n_stats + 2];
    g->traffic.transport = g->transport; g->traffic.remote_lineages = g->remote ? 1 : 0;
    const bool talk = g->world > 1 || g->world1_collectives;
    const double steps = (g->cfg.algorithm == CPPROB_HIP_ALG_SIS) ? 1.0 : (double)g->T;
    // (mailboxes: three words and a sequence number per rank pair and step)
    g->traffic.collective_bytes = talk ? (uint64_t)((double)g->world * (double)g->world * ((g->dev_coll ? 4.0 : 3.0) * 8.0 * steps + ((double)g->n_stats + kJointExtra) * 8.0)) : 0;
    g->traffic.mailbox_collectives = g->dev_coll ? 1 : 0;
    if (g->transport == kTransportDirect) {
        g->traffic.wire_bytes = g->traffic.payload_bytes;
        if (talk && (!g->loopback() || g->dev_coll)) g->traffic.collective_bytes += (uint64_t)((double)g->world * (double)g->world * 8.0 * std::max(0.0, steps - 1.0));   // the ordering all-gather
    } else if (g->transport == kTransportSendRecv) g->traffic.wire_bytes = (uint64_t)sendrecv_wire_bytes(g);
    else g->traffic.wire_bytes = 0;
    cpprob_hip_summary s{};
    if (int rc = cpprob_hip_infer_summary(g->ctx[0], &s)) return gkeep(g, gfail(g, rc, cpprob_hip_last_error(g->ctx[0])));
    if (out) *out = s;
    if (h_stats && group_filtering(g)) {
        // filtering-only shards: predict hit t under generation t's own weights -- the joint population's numbers as they are (count
        // form), or the ranks' raw sums over the ranks' masses of that generation
        const bool joint_already = g->ctx[0]->counts_mode;
        for (int t = 0; t < g->T; ++t) {
            const double W = joint_already ? 1.0 : joint[(size_t)g->n_stats + kJointExtra + (size_t)t];
            if (s.is_int || joint_already) { for (int k = 0; k < g->K; ++k) h_stats[t * g->K + k] = joint[(size_t)(t * g->K + k)] / W; }
            else {
                const double mean = joint[(size_t)(t * g->K)] / W;
                h_stats[t * g->K] = mean;
                h_stats[t * g->K + 1] = joint[(size_t)(t * g->K + 1)] / W - mean * mean;
            }
        }
    } else if (h_stats) {
        // StatsPrinter's numbers from the all-reduced un-normalised sums (relative to exp(max_logw))
        const double W = std::exp(s.log_norm - s.max_logw);
        for (int t = 0; t < g->T; ++t) {
            if (s.is_int) { for (int k = 0; k < g->K; ++k) h_stats[t * g->K + k] = joint[(size_t)(t * g->K + k)] / W; }
            else {
                const double mean = joint[(size_t)(t * g->K)] / W;
                h_stats[t * g->K] = mean;
                h_stats[t * g->K + 1] = joint[(size_t)(t * g->K + 1)] / W - mean * mean;      // raw_moment(2) - mean^2, empirical_distribution.hpp:78-81
            }
        }
    }
    return 0;
}

int cpprob_hip_group_profile(cpprob_hip_group* g, int32_t on)
{
    if (!g) return fail(nullptr, CPPROB_HIP_EINVAL, "group is NULL");
    g->phase_profile = on != 0;
    if (!on) {
        for (size_t i = 0; i < g->ph_ev.size(); ++i) { if (i < g->ctx.size()) (void)hipSetDevice(g->ctx[i]->device); for (hipEvent_t e : g->ph_ev[i]) (void)hipEventDestroy(e); }
        g->ph_ev.clear(); g->ph_used.clear();
    }
    return 0;
}

int cpprob_hip_group_profile_read(cpprob_hip_group* g, double* h_out8)
{
    if (!g || !h_out8) return fail(nullptr, CPPROB_HIP_EINVAL, "NULL argument");
    for (int k = 0; k < 8; ++k) h_out8[k] = 0.0;
    if (!g->phase_profile || !g->ran) return gkeep(g, gfail(g, CPPROB_HIP_ESTATE, "no profiled run (cpprob_hip_group_profile, then a run)"));
    if (int rc = cpprob_hip_group_sync(g)) return gkeep(g, rc);
    const int steps = g->ph_steps;
    h_out8[7] = (double)steps;
    if (steps == 0) return 0;
    for (size_t i = 0; i < g->ph_ev.size(); ++i) {
        if ((int)g->ph_used[i] < steps * kPhMarks) continue;
        GHIP_TRY(g, hipSetDevice(g->ctx[i]->device));
        double sum[kPhCount] = {0, 0, 0, 0, 0, 0};
        for (int s2 = 0; s2 < steps; ++s2)
            for (int k = 0; k < kPhCount; ++k) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, g->ph_ev[i][(size_t)(s2 * kPhMarks + k)], g->ph_ev[i][(size_t)(s2 * kPhMarks + k + 1)]) == hipSuccess) sum[k] += ms;
            }
        for (int k = 0; k < kPhCount; ++k) h_out8[k] = std::max(h_out8[k], sum[k] * 1e3 / steps);      // us per rank-step, the slowest local rank
    }
    for (size_t i = 0; i < g->d_dc_status.size(); ++i) {
        if (!g->d_dc_status[i]) continue;
        unsigned long long ticks = 0;
        GHIP_TRY(g, hipSetDevice(g->ctx[i]->device));
        GHIP_TRY(g, hipMemcpy(&ticks, g->d_dc_status[i] + 2, sizeof ticks, hipMemcpyDeviceToHost));
        h_out8[6] = std::max(h_out8[6], (double)ticks * 0.01 / steps);                                  // 100 MHz wall clock -> us per rank-step
    }
    return 0;
}

const char* cpprob_hip_group_note(const cpprob_hip_group* g)
{
    if (!g) return "";
    static thread_local std::string note;
    note = std::string("collectives: ") + (g->coll == kCollLoopback ? "loopback (one device, one stream: program order)" : (g->coll == kCollExternal ? "the caller's all-gather" : "RCCL"));
    note += g->dev_coll ? "; per-step collectives: mailbox stores over the peer mappings" : "; per-step collectives: the library's";
    if (!g->dc_note.empty()) note += " (" + g->dc_note + ")";
    note += std::string("; migrants: ") + (g->transport == kTransportDirect ? (g->remote ? "direct stores, remote lineages" : "direct stores, lineages shipped") : (g->transport == kTransportSendRecv ? "send/recv segments" : "none"));
    return note.c_str();
}

int cpprob_hip_group_traffic(cpprob_hip_group* g, cpprob_hip_traffic* out)
{
    if (!g || !out) return fail(nullptr, CPPROB_HIP_EINVAL, "NULL argument");
    if (!g->ran) return gkeep(g, gfail(g, CPPROB_HIP_ESTATE, "no finished run"));
    *out = g->traffic;
    return 0;
}

}  // extern "C"
