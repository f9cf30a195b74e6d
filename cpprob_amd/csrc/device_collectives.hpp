// Mailbox collectives: the per-step all-gather of an exchange-scope run as stores into the peers' memory.
//
// A step of the exchange scope needs two things from the other ranks: their 24 bytes of totals (before the plan) and the knowledge
// that their packing kernels have completed (before the next step reads its annex).  Through a collective library that is two
// latency-bound calls per step -- tens of microseconds each on xGMI, against an 8 us step kernel -- and through a caller-supplied host
// all-gather it is two host round trips.  The direct transport already maps every rank's memory into every rank (peer access, hipIpc):
// the same mappings carry a mailbox per rank, and a collective becomes one short launch --
//     post   lane r stores this rank's words into rank r's mailbox, then the step's sequence number (release, system scope);
//     wait   lane r spins on the sequence number rank r left in THIS rank's mailbox (acquire, system scope), then reads its words.
// Nothing on the host, no library call inside a run.  Slots alternate with the step's parity: a rank can be at most one collective
// ahead of a peer (its next wait needs that peer's next post), so two copies never collide.  Sequence numbers never repeat within a
// group (run serial << 32 | step + 1), so a stale slot is never mistaken for a fresh one.  A wait that sees nothing for `timeout` ticks
// of the 100 MHz wall clock gives up, sets a sticky status bit and lets the run finish on whatever it holds: the bit travels with the
// run's final all-reduce and the driver repeats the run on the library's collectives (group.hpp).
// Loopback ranks (one device, one stream) post in one pass and wait in the next: program order has already delivered everything.
#pragma once
#include "exchange.hpp"

namespace cph {

struct Mailbox {
    unsigned long long tot[2][kWorldSlots][4];      // [parity][sender]: three words and the sequence number
    unsigned long long bar[2][kWorldSlots];         // [parity][sender]: the sequence number of a completed exchange
};
struct MailboxPeers { Mailbox* box[kWorldSlots]; };  // every rank's mailbox as THIS device addresses it (its own included)

enum { kDcPost = 1, kDcWait = 2 };
constexpr int kDcTimedOut = 8;                       // status bit

__device__ __forceinline__ bool dc_spin(const unsigned long long* flag, unsigned long long seq, long long timeout, const int32_t* status)
{
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;       // (sticky: a run that lost a peer does not wait again)
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
        if (wall_clock64() - t0 > timeout) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    return true;
}

// how long the wavefront's wait took (its slowest lane), added to the rank's clock of spun ticks: status + 2, 64 bits (group.hpp reads it
// for cpprob_hip_group_profile_read: what a step spends waiting for its peers, as opposed to launching)
__device__ __forceinline__ void dc_waited(int32_t* status, long long t_begin)
{
    const long long dt = wall_clock64() - t_begin;                 // (behind the reconvergence of the lanes' spins)
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(status + 2), (unsigned long long)dt);
}

// all-gather of three 64-bit words per rank; one wavefront, lane r = rank r
__device__ __forceinline__ void dc_allgather_body(unsigned long long w0, unsigned long long w1, unsigned long long w2, const MailboxPeers* __restrict__ peers, Mailbox* mine,
                                                  int world, int rank, int parity, unsigned long long seq, int phases,
                                                  unsigned long long* __restrict__ all_out, int32_t* status, long long timeout)
{
    const int lane = threadIdx.x;
    if ((phases & kDcPost) && lane < world) {
        unsigned long long* slot = peers->box[lane]->tot[parity][rank];
        __hip_atomic_store(slot + 0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(slot + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(slot + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(slot + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (phases & kDcWait) {
        bool ok = true;
        const long long tb = wall_clock64();
        if (lane < world) {
            const unsigned long long* slot = mine->tot[parity][lane];
            ok = dc_spin(slot + 3, seq, timeout, status);
            // (whatever the slot holds when the wait gave up: the run is repeated, its numbers are never read)
            for (int k = 0; k < 3; ++k) all_out[3 * lane + k] = __hip_atomic_load(slot + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (!ok) atomicOr(status, kDcTimedOut);
        dc_waited(status, tb);
    }
}

__global__ __launch_bounds__(kWave) void dc_allgather_kernel(const unsigned long long* __restrict__ local3, const MailboxPeers* __restrict__ peers, Mailbox* mine,
                                                             int world, int rank, int parity, unsigned long long seq, int phases,
                                                             unsigned long long* __restrict__ all_out, int32_t* status, long long timeout)
{
    dc_allgather_body(local3[0], local3[1], local3[2], peers, mine, world, rank, parity, seq, phases, all_out, status, timeout);
}

// The shard-totals launch of a step (counts_totals_kernel / fixed_totals_kernel: one wavefront sums the hierarchy's top level) carrying
// the all-gather of what it has just summed: a rank with a stream of its own posts and waits where it stands -- one launch a step less.
// (Loopback ranks share a stream: they post in one pass and wait in the next, dc_allgather_kernel.)
struct TotalsGather {
    const MailboxPeers* peers; Mailbox* mine; int world, rank, parity; unsigned long long seq;
    unsigned long long* all_out; int32_t* status; long long timeout;
};

__global__ __launch_bounds__(kWave) void counts_totals_gather_kernel(Hier h, double n_local, double* __restrict__ out, TotalsGather d)
{
    const Cnt2 t = hier_total(h);
    const double v0 = (double)t.n0, v1 = (double)t.n1;
    if (threadIdx.x == 0) { out[0] = v0; out[1] = v1; out[2] = n_local; }
    dc_allgather_body((unsigned long long)__double_as_longlong(v0), (unsigned long long)__double_as_longlong(v1), (unsigned long long)__double_as_longlong(n_local),
                      d.peers, d.mine, d.world, d.rank, d.parity, d.seq, kDcPost | kDcWait, d.all_out, d.status, d.timeout);
}

__global__ __launch_bounds__(kWave) void fixed_totals_gather_kernel(FHier f, uint64_t* __restrict__ out, TotalsGather d)
{
    const FTot t = ftot(f);
    const uint64_t km = dkey(t.M);
    if (threadIdx.x == 0) { out[0] = t.S; out[1] = t.Q; out[2] = km; }
    dc_allgather_body(t.S, t.Q, km, d.peers, d.mine, d.world, d.rank, d.parity, d.seq, kDcPost | kDcWait, d.all_out, d.status, d.timeout);
}

// "every rank's packing kernel of this step has completed": stream-ordered behind the local packing launch on every rank
__global__ __launch_bounds__(kWave) void dc_barrier_kernel(const MailboxPeers* __restrict__ peers, Mailbox* mine, int world, int rank, int parity,
                                                           unsigned long long seq, int phases, int32_t* status, long long timeout)
{
    const int lane = threadIdx.x;
    // The migrants were stored into the peers' (coarse-grained) particle stores by the packing launch in front of this one; the flag
    // lives in the peers' fine-grained mailboxes.  What orders the two for the peer: the packing launch has completed (stream order),
    // and a system-scope release fence here writes back whatever of those stores this device still holds before the flag leaves --
    // stated explicitly rather than left to the release store's own scope, because the two allocations differ in coherence and this
    // path has never met a real link.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if ((phases & kDcPost) && lane < world)
        __hip_atomic_store(&peers->box[lane]->bar[parity][rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (phases & kDcWait) {
        bool ok = true;
        const long long tb = wall_clock64();
        if (lane < world) ok = dc_spin(&mine->bar[parity][lane], seq, timeout, status);
        if (!ok) atomicOr(status, kDcTimedOut);
        dc_waited(status, tb);
    }
}

}  // namespace cph
