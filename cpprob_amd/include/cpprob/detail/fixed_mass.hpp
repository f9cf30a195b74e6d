// Fixed-point particle masses and their 64-ary hierarchy: the device-side arithmetic every systematic-resampling step shares --
// the built-in kernels of libcpprob_hip (csrc/step_counts.hpp, csrc/step_fixed.hpp, csrc/bookkeep_fixed.hpp) and the step kernel
// an UNCHANGED model is compiled into (cpprob/gpu.hpp: the resampling search in the launch's prologue, quantise + publish in its
// epilogue).  One statement of the arithmetic (csrc/step_fixed.hpp's header comment; the CPU restatement the parity tests compare with
// states the same): q_i = min(rint(exp(lw_i - R) 2^32), 2^32 - 1), exact 64-bit prefix masses, G_k = ceil(fma(double(C_k), N / double(C_N), -u0)).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "cpprob/detail/fastmath.hpp"
#include "cpprob/detail/rng.hpp"
#include "cpprob/detail/wave.hpp"

namespace cph {

constexpr int kHierMaxLevels = 3;
// 64-bit words between entries of the levels >= 1: two 128-byte lines each.  A fixed-point entry is three words: {mass | arrivals}
// on the first line, {squares, maximum key} on the second (kHierQ, kHierM).  Where the words sit decides what the publishing atomics
// cost: the unchanged-model step (3907 workgroups at 10^6 particles, three atomics each) paid 5.7 us per launch with all three on
// one line and 4.3 with one line per word, and nothing measurable with this split (profiles/r04_notes.md); the count form's single
// word uses the first line.
#ifndef CPPROB_HIER_STRIDE
#define CPPROB_HIER_STRIDE 32
#endif
constexpr int kHierStride = CPPROB_HIER_STRIDE;
#ifndef CPPROB_HIER_Q
#define CPPROB_HIER_Q 16
#endif
constexpr int kHierQ = CPPROB_HIER_Q, kHierM = CPPROB_HIER_Q + 1;        // word offsets of the squares / the maximum key inside an entry
constexpr int kHierArrM = kHierM + 1;                  // arrivals of the exact-maximum pass
constexpr int64_t kCountsMaxTiles = 64LL * 64 * 64;
constexpr uint64_t kCntMask = (1ull << 28) - 1;

struct HierTable {                             // device-resident: every copy, every level (run-time indexed by the rare paths)
    uint64_t* lvl[3][kHierMaxLevels];
    int n_ent[kHierMaxLevels];                 // entries per level; n_ent[0] = tiles
    int n_lev;                                 // levels in use: the last one has <= 64 entries
};
struct Hier {                                  // what a launch carries: the copy it reads, by level (compile-time indices only -- a
    const uint64_t* lvl[kHierMaxLevels];       // run-time index into a kernel-argument array would send the struct through scratch
    int n_ent[kHierMaxLevels];                 // memory) and where the other two copies sit relative to it
    int n_lev;
    int64_t to_next, to_clear;                 // word offsets from the copy read to the copy written / the copy cleared
    const HierTable* table; int copy;          // the same hierarchy in device memory
    const uint64_t* top; int top_n, top_stride; // the last level in use (<= 64 entries): the generation's totals
};
// (levels beyond n_lev point at level 0, so that a load from any level is a load from valid memory: the fetches below carry no
//  branch, and the compiler lets them travel together instead of waiting for each in turn)
__device__ __forceinline__ constexpr int hier_stride(int level) { return level == 0 ? 1 : kHierStride; }

// ---- 32-bit wavefront sums / scans (one instruction per DPP step) -------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += dpp_u32<kDppRowShr1>(v);
    v += dpp_u32<kDppRowShr2>(v);
    v += dpp_u32<kDppRowShr4>(v);
    v += dpp_u32<kDppRowShr8>(v);
    v += dpp_u32<kDppRowBcast15, 0xA>(v);
    v += dpp_u32<kDppRowBcast31, 0xC>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), kWave - 1); }


// Exclusive prefix counts at tile c (wave-uniform result; every lane of the calling wave takes part): per level, the entries that
// precede c's block inside its parent block.  Two halves: the loads (unconditional, clamped -- issued at kernel entry, long before
// anything needs them) and the masked sum.
__device__ __forceinline__ void hier_prefix_fetch(const Hier& h, int c, uint64_t (&w)[kHierMaxLevels])
{
    const int lane = lane_id();
#pragma unroll
    for (int l = 0; l < kHierMaxLevels; ++l) {
        const int blk = c >> (6 * l);                           // c's block at this level (0 at the levels not in use)
        const int first = (blk >> 6) << 6;                      // first block of the parent
        w[l] = h.lvl[l][(int64_t)(first + (lane < (blk & 63) ? lane : 0)) * hier_stride(l)];
    }
}

// What a probe of the tiles around `at` reads: the hierarchy words of tile cs = max(at - 1, 0)'s prefix and four tile entries.
struct ProbeWords { uint64_t lvl[kHierMaxLevels]; uint64_t we; };
__device__ __forceinline__ void probe_fetch(const Hier& h, int at, int nb, ProbeWords& w)
{
    const int cs = at > 0 ? at - 1 : 0;
    hier_prefix_fetch(h, cs, w.lvl);
    const int lane = lane_id();
    const int i = cs + (lane < 4 ? lane : 0);
    w.we = h.lvl[0][i < nb ? i : nb - 1];
}


// 6 nats = 9 of the 32 bits: the heaviest particle of every generation carries at least a 23-bit weight (a gap opens where an
// observation lies more than ~3.5 standard deviations from EVERY particle: populations of thousands and more never see one).  A
// generation beyond it is requantised against its exact maximum before anything is drawn from it (cpprob_hip.hip: settle_fixed).
constexpr double kFixGapLimit = 6.0;
constexpr double kFixScale = 4294967296.0;                  // 2^32
constexpr double kFixInv = 1.0 / 4294967296.0;
constexpr uint64_t kMassMask = (1ull << 56) - 1;

// ---- 64-bit wavefront sums / scans / maxima -----------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t read_lane_u64(uint64_t v, int lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v)
{
    v += dpp_u64<kDppRowShr1>(v);
    v += dpp_u64<kDppRowShr2>(v);
    v += dpp_u64<kDppRowShr4>(v);
    v += dpp_u64<kDppRowShr8>(v);
    v += dpp_u64<kDppRowBcast15, 0xA>(v);
    v += dpp_u64<kDppRowBcast31, 0xC>(v);
    return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) { return read_lane_u64(wave_incl_scan_u64(v), kWave - 1); }
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v)
{
    v = umax64(v, dpp_u64<kDppRowShr1>(v));
    v = umax64(v, dpp_u64<kDppRowShr2>(v));
    v = umax64(v, dpp_u64<kDppRowShr4>(v));
    v = umax64(v, dpp_u64<kDppRowShr8>(v));
    v = umax64(v, dpp_u64<kDppRowBcast15, 0xA>(v));
    v = umax64(v, dpp_u64<kDppRowBcast31, 0xC>(v));
    return read_lane_u64(v, kWave - 1);
}

// Values below 2^34 -- a lane's four 32-bit weights, four squares of 16-bit halves -- as two 17-bit halves: each half's sum over 64
// lanes fits 23 bits, so a scan step is ONE v_add_u32_dpp per half (a 64-bit step is two DPP moves and a two-instruction add).
__device__ __forceinline__ uint64_t wave_incl_scan_u34(uint64_t v)
{
    const uint32_t lo = wave_incl_scan_u32((uint32_t)v & 0x1ffffu), hi = wave_incl_scan_u32((uint32_t)(v >> 17));
    return ((uint64_t)hi << 17) + lo;
}
__device__ __forceinline__ uint64_t wave_sum_u34(uint64_t v)
{
    const uint32_t lo = wave_sum_u32((uint32_t)v & 0x1ffffu), hi = wave_sum_u32((uint32_t)(v >> 17));
    return ((uint64_t)hi << 17) + lo;
}
// ... and the maximum of 64-bit keys as two 32-bit maxima: the high words, then the low words of the lanes that hold the high maximum
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, dpp_u32<kDppRowShr1>(v));
    v = max(v, dpp_u32<kDppRowShr2>(v));
    v = max(v, dpp_u32<kDppRowShr4>(v));
    v = max(v, dpp_u32<kDppRowShr8>(v));
    v = max(v, dpp_u32<kDppRowBcast15, 0xA>(v));
    v = max(v, dpp_u32<kDppRowBcast31, 0xC>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, kWave - 1);
}
__device__ __forceinline__ uint64_t wave_max_key(uint64_t k)
{
    const uint32_t hi = wave_max_u32((uint32_t)(k >> 32));
    const uint32_t lo = wave_max_u32((uint32_t)(k >> 32) == hi ? (uint32_t)k : 0u);
    return ((uint64_t)hi << 32) | lo;
}

// order-preserving key of a double (an unsigned maximum of keys is the maximum of the doubles; 0 lies below every key: "empty")
__host__ __device__ __forceinline__ uint64_t dkey(double x)
{
    union { double d; uint64_t u; } c; c.d = x;
    return (c.u >> 63) ? ~c.u : (c.u | (1ull << 63));
}
__host__ __device__ __forceinline__ double dkey_inv(uint64_t k)
{
    if (k == 0) return -INFINITY;
    union { double d; uint64_t u; } c;
    c.u = (k >> 63) ? (k & ~(1ull << 63)) : ~k;
    return c.d;
}
// an exact integer below 2^64 as the nearest double (one rounding: what a C cast does)
__device__ __forceinline__ double u64_to_double(uint64_t c) { return fma((double)(uint32_t)(c >> 32), 4294967296.0, (double)(uint32_t)c); }

// the weight as an integer (see the header of this file); lw <= ref, or -inf (padding slots): 0
__device__ __forceinline__ uint32_t fix_weight(double lw, double ref)
{
    const double e = exp_nonpos(fmax(lw - ref, -1000.0));
    const double s = rint(e * kFixScale);
    return s >= 4294967295.0 ? 0xffffffffu : (uint32_t)s;
}

// ---- the hierarchy's fixed-point view ------------------------------------------------------------------------------------------
struct FHier {
    Hier h;                                    // the S words: laid out, rotated and searched like the count hierarchy
    const uint64_t* q0; const uint64_t* m0;    // tiles' Q and M keys of the copy read (the copy written sits h.to_next words further)
};
struct FTotWords { uint64_t s, q, m; };
struct FTot { uint64_t S, Q; double M; };      // a generation's totals: mass, sum of squared 16-bit weights, largest log-weight

__device__ __forceinline__ void ftot_fetch(const FHier& f, FTotWords& w)
{
    const int lane = lane_id();
    const int64_t i = (int64_t)(lane < f.h.top_n ? lane : 0) * f.h.top_stride;
    w.s = f.h.top[i];
    w.q = f.h.n_lev == 1 ? f.q0[i] : f.h.top[i + kHierQ];
    w.m = f.h.n_lev == 1 ? f.m0[i] : f.h.top[i + kHierM];
}
__device__ __forceinline__ FTot ftot_sum(const FHier& f, FTotWords w)
{
    if (lane_id() >= f.h.top_n) { w.s = 0; w.q = 0; w.m = 0; }
    FTot t;
    t.S = wave_sum_u64(w.s & kMassMask); t.Q = wave_sum_u64(w.q); t.M = dkey_inv(wave_max_key(w.m));
    return t;
}
__device__ __forceinline__ FTot ftot(const FHier& f) { FTotWords w; ftot_fetch(f, w); return ftot_sum(f, w); }

// exclusive prefix mass at tile c (wave-uniform; every lane of the calling wave takes part)
__device__ __forceinline__ uint64_t fhier_prefix_sum(int c, const uint64_t (&w)[kHierMaxLevels])
{
    const int lane = lane_id();
    uint64_t s = 0;
#pragma unroll
    for (int l = 0; l < kHierMaxLevels; ++l) s += lane < ((c >> (6 * l)) & 63) ? (w[l] & kMassMask) : 0ull;
    return wave_sum_u64(s);
}

// Which outputs a prefix of the sources owns, on integer masses.  base = mass of the shards that precede this one (0 on one GPU).
//   systematic (kFixSystematic): one shared offset u0; the sources up to inclusive mass C own the outputs below
//        G = ceil(fma(double(C), N / double(C_N), -u0));
//   stratified (kFixStratified): output j sits at j + u_j, u_j = the 32-bit uniform of OUTPUT j (word id & 3 of Philox block id >> 2,
//        id = uid0 + j, draw kResampleDrawBase + step); the sources up to C reach H = double(C) * (N / double(C_N)) (one rounded
//        product) and own the outputs with j + u_j < H -- a prefix, because j + u_j increases with j:
//        A = F + [u_F < H - F],  F = floor(H)   (H - F and the comparison are exact);
//   multinomial (kFixMultinomial: binned, kFixMultinomialLiteral: one search per output): thresholds, not a comb -- csrc/step_fixed.hpp.
// Both are functions of the exact integer C alone, so tiles, wavefronts and shards may evaluate them in any order.
constexpr int kFixSystematic = 0, kFixStratified = 1, kFixMultinomial = 2, kFixMultinomialLiteral = 3;
constexpr uint64_t kResampleDrawBase2 = kResampleDrawBase + (1ull << 39);   // multinomial, strata form: the outputs' uniforms inside their strata
constexpr uint64_t kResampleDrawBase3 = kResampleDrawBase + (1ull << 38);   // ... and the bits that split the thresholds over the strata
constexpr int kStrataTiles = 3;   // source tiles of an output tile staged side by side by the strata form (more: in turn)
// K = 2^k strata, the smallest power of two >= FOUR times the number of tiles (128 .. 256 thresholds a stratum on average): the strata
// that overhang an output tile's ends add <= half a tile of mass to what its thresholds span, so its sources are two tiles, rarely three
// -- one staged group (kStrataTiles).  With K ~ the number of tiles the span reached four and five tiles: a second group in the
// workgroups the launch then waited for (profiles/r06_notes.md section 8).
constexpr int kStrataPerTileLog2 = 2;
__host__ __device__ inline int strata_levels(int64_t nb) { int k = 0; while (((int64_t)1 << k) < nb) ++k; return k + kStrataPerTileLog2; }
struct FixedCdf {
    double inv, u0, n_pop; uint64_t base;
    uint64_t seed, draw, uid0;                 // stratified: the run's Philox key, the resampling's draw index, the id of output 0
    // first output NOT owned by the sources up to a LOCAL inclusive mass C = first output owned by the sources that follow
    __device__ __forceinline__ double g(uint64_t C) const { return ceil(fma(u64_to_double(base + C), inv, -u0)); }
    __device__ __forceinline__ double h(uint64_t C) const { return __dmul_rn(u64_to_double(base + C), inv); }
    __device__ __forceinline__ double first_stratified(uint64_t C) const
    {
        const double H = h(C), F = floor(H);
        if (F >= n_pop) return n_pop;
        const double u = u01_32(draw_word(seed, uid0 + (uint64_t)F, draw));
        return u < H - F ? F + 1.0 : F;
    }
    template <int RS>
    __device__ __forceinline__ double first(uint64_t C) const
    {
        if constexpr (RS == kFixStratified) return first_stratified(C);
        else return g(C);
    }
};

// This tile's words of generation t's hierarchy, added into the levels above, and the entries of the third copy this tile is
// responsible for clearing.  One thread.  (step_counts.hpp: hier_publish -- here a block's line carries three words, and the last
// tile of a block reads the two it did not get back from its own add.)  Two parts: the levels >= 1 (fhier_forward), and the
// tile's own entries in front of them (fhier_publish) -- a tile that is produced by several workgroups adds into its entries
// itself and lets its last arriver forward (cpprob/detail/device_trace.hpp: the unchanged-model step).
__device__ __forceinline__ void fhier_forward(const FHier& f, int bid, int nb, uint64_t S, uint64_t Q, uint64_t mkey)
{
    const Hier& h = f.h;
    uint64_t* l1 = const_cast<uint64_t*>(h.lvl[1]);
    uint64_t* l2 = const_cast<uint64_t*>(h.lvl[2]);
    using ull = unsigned long long;
    const int b1 = bid >> 6, b2 = bid >> 12;
    if (h.n_lev == 2) {
        ull* e = reinterpret_cast<ull*>(l1 + h.to_next + (int64_t)b1 * kHierStride);
        atomicAdd(e, (ull)S); atomicAdd(e + kHierQ, (ull)Q); atomicMax(e + kHierM, (ull)mkey);
    } else if (h.n_lev == 3) {
        ull* e = reinterpret_cast<ull*>(l1 + h.to_next + (int64_t)b1 * kHierStride);
        atomicAdd(e + kHierQ, (ull)Q); atomicMax(e + kHierM, (ull)mkey);
        // both have been PERFORMED (device-scope atomics execute where every XCD sees them; the counter waits for their
        // acknowledgement) before this tile's arrival is counted -- no cache write-back: a __threadfence() here flushes the whole
        // L2 of dirty particle rows once per workgroup (measured: 695 us per step at 10^7 particles instead of 60)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const ull old = atomicAdd(e, (ull)(S + (1ull << 56)));
        const int tiles_in_block = nb - (b1 << 6) < 64 ? nb - (b1 << 6) : 64;
        if ((int)(old >> 56) == tiles_in_block - 1) {
            const uint64_t totS = (old + S) & kMassMask;
            const ull totQ = atomicAdd(e + kHierQ, (ull)0), totM = atomicMax(e + kHierM, (ull)0);       // (read where the adds were performed)
            ull* e2 = reinterpret_cast<ull*>(l2 + h.to_next + (int64_t)b2 * kHierStride);
            atomicAdd(e2, (ull)totS); atomicAdd(e2 + kHierQ, totQ); atomicMax(e2 + kHierM, totM);
        }
    }
    if (h.n_lev >= 2 && (b1 << 6) == bid) { uint64_t* e = l1 + h.to_clear + (int64_t)b1 * kHierStride; e[0] = 0; e[kHierQ] = 0; e[kHierM] = 0; }
    if (h.n_lev >= 3 && (b2 << 12) == bid) { uint64_t* e = l2 + h.to_clear + (int64_t)b2 * kHierStride; e[0] = 0; e[kHierQ] = 0; e[kHierM] = 0; }
}
__device__ __forceinline__ void fhier_publish(const FHier& f, int bid, int nb, uint64_t S, uint64_t Q, uint64_t mkey)
{
    const Hier& h = f.h;
    const_cast<uint64_t*>(h.lvl[0])[h.to_next + bid] = S;
    const_cast<uint64_t*>(f.q0)[h.to_next + bid] = Q;
    const_cast<uint64_t*>(f.m0)[h.to_next + bid] = mkey;
    fhier_forward(f, bid, nb, S, Q, mkey);
}
// One of `parts` workgroups that produce tile `bid` together: its share {S, Q, max} is added into the tile's entries (level 0 then
// carries an arrival count in its top byte, like the levels above: readers mask it), the last one to arrive forwards the tile's
// totals, part 0 clears the tile's entries of the third copy.  One thread.
__device__ __forceinline__ void fhier_publish_part(const FHier& f, int bid, int nb, int part, int parts, uint64_t S, uint64_t Q, uint64_t mkey)
{
    const Hier& h = f.h;
    using ull = unsigned long long;
    ull* e0 = reinterpret_cast<ull*>(const_cast<uint64_t*>(h.lvl[0]) + h.to_next + bid);
    ull* eq = reinterpret_cast<ull*>(const_cast<uint64_t*>(f.q0) + h.to_next + bid);
    ull* em = reinterpret_cast<ull*>(const_cast<uint64_t*>(f.m0) + h.to_next + bid);
    atomicAdd(eq, (ull)Q); atomicMax(em, (ull)mkey);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // performed before this part's arrival is counted (see fhier_forward)
    const ull old = atomicAdd(e0, (ull)(S + (1ull << 56)));
    if ((int)(old >> 56) == parts - 1) {
        const uint64_t totS = (old + S) & kMassMask;
        const ull totQ = atomicAdd(eq, (ull)0), totM = atomicMax(em, (ull)0);
        fhier_forward(f, bid, nb, totS, totQ, totM);
    }
    if (part == 0) {
        const_cast<uint64_t*>(h.lvl[0])[h.to_clear + bid] = 0; const_cast<uint64_t*>(f.q0)[h.to_clear + bid] = 0; const_cast<uint64_t*>(f.m0)[h.to_clear + bid] = 0;
    }
}

// Largest tile c in [0, nb) whose first owned output G(prefix(c)) is <= g (0 when there is none), with its exclusive prefix mass:
// top-down descent, one load + one scan per level (step_counts.hpp: hier_locate).
template <int RS = kFixSystematic>
__device__ __forceinline__ int fhier_locate(const HierTable* __restrict__ ht, int copy, const FixedCdf& fc, double g, uint64_t& P)
{
    const int lane = lane_id();
    int blk = 0;
    uint64_t p = 0;
    for (int l = ht->n_lev - 1; l >= 0; --l) {
        const int idx = (blk << 6) + lane;
        uint64_t w = 0;
        const bool in = idx < ht->n_ent[l];
        if (in) w = ht->lvl[copy][l][(int64_t)idx * (l == 0 ? 1 : kHierStride)] & kMassMask;
        const uint64_t incl = wave_incl_scan_u64(w);
        const uint64_t x = p + incl - w;                                                       // exclusive prefix at child `lane`
        const bool ok = in && fc.template first<RS>(x) <= g;
        const unsigned long long m = __ballot(ok);
        const int child = m ? (63 - __builtin_clzll(m)) : 0;                                     // (G is monotone: the set is a prefix)
        p = read_lane_u64(x, child);
        blk = (blk << 6) + child;
    }
    P = p;
    return blk;
}

struct FLocated { int c, c_last; uint64_t P; };

// The SEARCH (one wavefront): first source tile of the output tile that starts at global output gj_first, its exclusive prefix
// mass, and the last source tile (nb when the probe cannot tell).  As counts_locate.
template <int RS = kFixSystematic>
__device__ __forceinline__ FLocated fixed_locate(const FHier& f, const FixedCdf& fc, int nb, double gj_first, int n_out, int guess, const ProbeWords* first)
{
    const int lane = lane_id();
    const double gj_last = gj_first + (double)(n_out - 1);
    int c = 0, c_last = nb;
    uint64_t P = 0;
    auto probe = [&](int at, const ProbeWords& pw, double& d_out) -> bool {
        const int cs = at > 0 ? at - 1 : 0;
        const uint64_t Pc = fhier_prefix_sum(cs, pw.lvl);
        const uint64_t we = (lane < 4 && cs + lane < nb) ? (pw.we & kMassMask) : 0ull;
        const uint64_t incl = wave_incl_scan_u64(we);
        const uint64_t x = Pc + incl - we;                               // lanes 0..4: the prefix at cs + lane
        const double gt = fc.template first<RS>(x);
        const bool known = lane < 5 && cs + lane < nb;
        const unsigned long long m = __ballot(known && gt <= gj_first);
        const int i_lo = m ? (63 - __builtin_clzll(m)) : -1;
        d_out = gj_first - read_lane(gt, 0);
        if ((i_lo >= 0 || cs == 0) && i_lo < 4) {
            const int i = i_lo < 0 ? 0 : i_lo;
            c = cs + i;
            P = read_lane_u64(x, i);
            const unsigned long long mh = __ballot(known && gt <= gj_last);
            const int i_hi = mh ? (63 - __builtin_clzll(mh)) : i;
            c_last = (i_hi >= 4 && cs + 5 < nb) ? nb : cs + (i_hi > i ? i_hi : i);
            return true;
        }
        return false;
    };
    double d;
    bool hit;
    if (first) hit = probe(guess, *first, d);
    else { ProbeWords pw; probe_fetch(f.h, guess, nb, pw); hit = probe(guess, pw, d); }
    if (!hit) {
        // tile masses are comparable, so the miss distance in outputs approximates the miss distance in tiles (x 1024): aim again,
        // then descend from the top (weights so uneven that two local probes miss)
        const double aim = (double)(guess > 0 ? guess - 1 : 0) + floor(d * (1.0 / kTile));
        const int at = (int)fmin(fmax(aim, 0.0), (double)(nb - 1));
        ProbeWords pw;
        probe_fetch(f.h, at, nb, pw);
        if (!probe(at, pw, d)) { c = fhier_locate<RS>(f.h.table, f.h.copy, fc, gj_first, P); c_last = nb; }
    }
    return FLocated{c, c_last, P};
}


// A weight's square for the ESS, in units of 2^-32 of exp(2 R): floor(floor(q / 2^8)^2 / 2^16).  Below 2^32, so a tile's sum and the
// hierarchy's stay in 64 bits up to 2^28 particles.  What the two floors lose: at most 2^9 / q + 2^32 / q^2 of a term -- 1.2e-7 for a
// particle at the reference, 1.2e-4 for the heaviest particle of a generation that sits 6 nats below it (the gap at which a run is
// repeated), nothing that moves a decision.  (r03 squared q >> 16: 2^17 / q, i.e. 1.6 % at that gap, always downward.)
__host__ __device__ __forceinline__ uint32_t fix_square(uint32_t q)
{
    const uint32_t h = q >> 8;
    return (uint32_t)(((uint64_t)h * h) >> 16);
}

// What every rank derives from a generation's totals, identically: the decision, the comb, the next reference.
struct FixedDecision { double W, Qd, ess, inv; bool resample; };
__device__ __forceinline__ FixedDecision fixed_decide(uint64_t S, uint64_t Q, double n_pop, double ess_frac, bool may_resample)
{
    FixedDecision d;
    const double Sd = u64_to_double(S);
    d.W = Sd * kFixInv;                                      // sum of exp(lw - R)
    d.Qd = u64_to_double(Q) * kFixInv;                        // sum of exp(2 (lw - R)): fix_square(q) = e^2 2^32
    const double e = d.W * d.W / d.Qd;                        // thesis p.37
    d.ess = e > n_pop ? n_pop : e;                            // (fix_square under-counts: the estimate is kept in [.., N])
    d.resample = may_resample && d.ess < ess_frac * n_pop;
    d.inv = n_pop / Sd;
    return d;
}


// Reference of generation t from what is known before it exists.
__device__ __forceinline__ double fixed_reference(bool fresh, double m_prev, double bound) { return fresh ? bound : m_prev + bound; }

// ---- exact-maximum reference (callers without a host-known bound of the step's log-likelihood): maximum pass, then masses ----
// A block entry here: word 0 S | arrivals << 56, word kHierQ Q, word kHierM key(M), word kHierArrM arrivals of the maximum pass.
__device__ __forceinline__ void bbf_publish_max(const FHier& f, int bid, int nb, uint64_t mkey)
{
    const Hier& h = f.h;
    using ull = unsigned long long;
    uint64_t* l1 = const_cast<uint64_t*>(h.lvl[1]);
    uint64_t* l2 = const_cast<uint64_t*>(h.lvl[2]);
    const int b1 = bid >> 6, b2 = bid >> 12;
    const_cast<uint64_t*>(f.m0)[bid] = mkey;
    if (h.n_lev == 2) {
        atomicMax(reinterpret_cast<ull*>(l1 + (int64_t)b1 * kHierStride + kHierM), (ull)mkey);
    } else if (h.n_lev == 3) {
        ull* e = reinterpret_cast<ull*>(l1 + (int64_t)b1 * kHierStride);
        atomicMax(e + kHierM, (ull)mkey);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // performed before the arrival is counted (no cache write-back: step_fixed.hpp)
        const ull old = atomicAdd(e + kHierArrM, (ull)1);
        const int tiles_in_block = nb - (b1 << 6) < 64 ? nb - (b1 << 6) : 64;
        if ((int)old == tiles_in_block - 1) {
            const ull totM = atomicMax(e + kHierM, (ull)0);
            atomicMax(reinterpret_cast<ull*>(l2 + (int64_t)b2 * kHierStride + kHierM), totM);
        }
    }
    // the other copy's upper levels: clean for the next step
    if (h.n_lev >= 2 && (b1 << 6) == bid) { uint64_t* e = l1 + h.to_clear + (int64_t)b1 * kHierStride; e[0] = 0; e[kHierQ] = 0; e[kHierM] = 0; e[kHierArrM] = 0; }
    if (h.n_lev >= 3 && (b2 << 12) == bid) { uint64_t* e = l2 + h.to_clear + (int64_t)b2 * kHierStride; e[0] = 0; e[kHierQ] = 0; e[kHierM] = 0; e[kHierArrM] = 0; }
}

// the generation's largest log-weight from the top level's M words (wave-uniform)
__device__ __forceinline__ double bbf_top_max(const FHier& f)
{
    const int lane = lane_id();
    const int64_t i = (int64_t)(lane < f.h.top_n ? lane : 0) * f.h.top_stride;
    uint64_t m = f.h.n_lev == 1 ? f.m0[i] : f.h.top[i + kHierM];
    if (lane >= f.h.top_n) m = 0;
    return dkey_inv(wave_max_u64(m));
}

__device__ __forceinline__ void bbf_publish_mass(const FHier& f, int bid, int nb, uint64_t S, uint64_t Q)
{
    const Hier& h = f.h;
    using ull = unsigned long long;
    uint64_t* l0 = const_cast<uint64_t*>(h.lvl[0]);
    uint64_t* l1 = const_cast<uint64_t*>(h.lvl[1]);
    uint64_t* l2 = const_cast<uint64_t*>(h.lvl[2]);
    const int b1 = bid >> 6, b2 = bid >> 12;
    l0[bid] = S;
    const_cast<uint64_t*>(f.q0)[bid] = Q;
    if (h.n_lev == 2) {
        ull* e = reinterpret_cast<ull*>(l1 + (int64_t)b1 * kHierStride);
        atomicAdd(e, (ull)S); atomicAdd(e + kHierQ, (ull)Q);
    } else if (h.n_lev == 3) {
        ull* e = reinterpret_cast<ull*>(l1 + (int64_t)b1 * kHierStride);
        atomicAdd(e + kHierQ, (ull)Q);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const ull old = atomicAdd(e, (ull)(S + (1ull << 56)));
        const int tiles_in_block = nb - (b1 << 6) < 64 ? nb - (b1 << 6) : 64;
        if ((int)(old >> 56) == tiles_in_block - 1) {
            const ull totQ = atomicAdd(e + kHierQ, (ull)0);
            ull* e2 = reinterpret_cast<ull*>(l2 + (int64_t)b2 * kHierStride);
            atomicAdd(e2, (ull)((old + S) & kMassMask)); atomicAdd(e2 + kHierQ, totQ);
        }
    }
}

// Multinomial resampling, strata form: the strata w0 .. w1 of the outputs [s_first, s_last] -- the largest w with offs[w] <= s (offs:
// first output of every stratum, offs[K] = N).  One wavefront: a window of 64 offsets around the output tile's own place almost
// surely holds both (offs[w] wanders sqrt(N) outputs off w N / K); a binary search of the whole array otherwise.  w_near = the
// stratum the outputs are expected in: (tile of the POPULATION << k) / tiles of the population.
__device__ __forceinline__ int strata_near(int k, int64_t nb_pop, int64_t tile_pop) { return (int)((tile_pop << k) / (nb_pop > 0 ? nb_pop : 1)); }
__device__ __forceinline__ void strata_window(const uint32_t* __restrict__ offs, int k, int w_near, uint32_t s_first, uint32_t s_last, int& w0, int& w1)
{
    const int lane = lane_id();
    const int K = 1 << k;
    int w_at = w_near - 31;
    if (w_at > K + 1 - kWave) w_at = K + 1 - kWave;
    if (w_at < 0) w_at = 0;
    const int idx = w_at + lane;
    const bool valid = idx <= K;
    const uint32_t o = offs[valid ? idx : K];
    const unsigned long long m_lo = __ballot(valid && o <= s_first), m_hi = __ballot(valid && o <= s_last);
    const int top = K - w_at < kWave - 1 ? K - w_at : kWave - 1;          // the window's last valid lane
    const bool ok_lo = (m_lo & 1ull) && (63 - __builtin_clzll(m_lo | 1ull)) < top;
    const bool ok_hi = (m_hi & 1ull) && (63 - __builtin_clzll(m_hi | 1ull)) < top;
    auto search = [&](uint32_t s) -> int {
        int lo = 0, hi = K;                                           // offs[0] = 0 <= s < offs[K]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= s) lo = mid; else hi = mid; }
        return lo;
    };
    w0 = ok_lo ? w_at + (63 - __builtin_clzll(m_lo)) : search(s_first);
    w1 = ok_hi ? w_at + (63 - __builtin_clzll(m_hi)) : search(s_last);
}

// ---- the walk over the source tiles (the fused step kernels of the library and the unchanged-model quad step, cpprob/gpu.hpp) ----
template <int RS>
struct FixedLdsT {
    int32_t slot[RS == kFixMultinomial || RS == kFixMultinomialLiteral ? kPPT : kTile];          // scatter slots of the output tile
    uint64_t scan[2][kWaves];     // per-wave totals of the in-tile scan, double-buffered across source tiles
    int iscr[kWaves];
    uint32_t ustrat[RS == kFixStratified ? kTile : 1];   // stratified: the 32-bit uniforms of the output tile's outputs
    uint64_t mp[RS == kFixMultinomial ? kStrataTiles * kTile : 1];  // multinomial, strata form: the source tiles' prefix masses (per wavefront), side by side (or two in turn)
    uint64_t wtot[RS == kFixMultinomial ? kStrataTiles : 1][kWaves]; // ... and their wavefronts' totals
};
using FixedLds = FixedLdsT<kFixSystematic>;

// Stratified resampling: the uniforms of the kTile outputs that start at output id `uid_first` (= FixedCdf::uid0 + the tile's first
// output), four per lane -- one Philox block when the id is a multiple of four.  Visible behind the caller's barrier.
__device__ __forceinline__ void stratified_stage(FixedLdsT<kFixStratified>& L, uint64_t seed, uint64_t draw, uint64_t uid_first)
{
    uint32_t w[kPPT];
    draw_words4(seed, uid_first + (uint64_t)threadIdx.x * kPPT, draw, w);
    store4(L.ustrat, (int64_t)threadIdx.x * kPPT, w);
}

// The WALK (the whole workgroup): every source tile that owns outputs of this tile rebuilds its prefix masses (one scan), each
// source with a non-empty range writes its index into the slot of its FIRST output, one prefix-max hands every output its
// ancestor; -1 where the ancestor belongs to a shard that precedes this one.  q_m1 / q_0 / q_p1 = the weights of tiles guess-1,
// guess, guess+1 fetched by the caller (have = false: none were).  Slots must hold -1 and be visible on entry.
// Stratified: L.ustrat holds the outputs' uniforms (stratified_stage), and a boundary's place in the tile is F - gj_first + [u_F < H - F].
using U4 = unsigned int __attribute__((ext_vector_type(4)));
template <int RS = kFixSystematic>
__device__ __forceinline__ void fixed_walk(const FixedCdf& fc, const uint32_t* __restrict__ qprev, int64_t n, int nb, bool last_shard, double gj_first,
                                           int n_out, const FLocated& loc, int guess, bool have, U4 q_m1, U4 q_0, U4 q_p1, int32_t (&anc)[kPPT], FixedLdsT<RS>& L)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const double gj_last = gj_first + (double)(n_out - 1);
    int c = __builtin_amdgcn_readfirstlane(loc.c), c_last = __builtin_amdgcn_readfirstlane(loc.c_last);
    uint64_t P = loc.P;
    auto load_q = [&](int cc) -> U4 {
        if (have && cc == guess) return q_0;
        if (have && cc == guess - 1) return q_m1;
        if (have && cc == guess + 1) return q_p1;
        U4 z = {0u, 0u, 0u, 0u};
        return cc < nb ? *reinterpret_cast<const U4*>(qprev + (int64_t)cc * kTile + (int64_t)tid * kPPT) : z;
    };
    auto place = [&](double g) -> int { return (int)fmin(fmax(g - gj_first, 0.0), (double)kTile); };      // exact: integers
    // the place, in this output tile, of the first output the sources beyond inclusive mass C own
    auto place_of = [&](uint64_t C) -> int {
        if constexpr (RS == kFixStratified) {
            const double H = fc.h(C), F = floor(H), d = F - gj_first;
            if (!(d >= 0.0)) return 0;
            if (d >= (double)kTile) return kTile;
            const int i = (int)d;
            return i + (u01_32(L.ustrat[i]) < H - F ? 1 : 0);
        } else return place(fc.g(C));
    };
    U4 raw = load_q(c);
    int it = 0;
    while (c < nb && c <= c_last) {
        // (wave-uniform values -- the branches are made scalar so that the barrier inside the loop sits in uniform control flow)
        if (c_last >= nb && __builtin_amdgcn_readfirstlane(fc.template first<RS>(P) > gj_last ? 1 : 0)) break;      // the last tile is not known from the probe
        const U4 raw_next = c < c_last ? load_q(c + 1) : U4{0u, 0u, 0u, 0u};
        const bool edge = c == nb - 1;
        const int nvt = edge ? (int)(n - (int64_t)c * kTile) : kTile;   // valid particles of this tile (padding slots weigh 0)
        const int vb = tid * kPPT;
        uint64_t pre[kPPT];
        uint64_t run = 0;
#pragma unroll
        for (int k = 0; k < kPPT; ++k) { run += (uint64_t)raw[k]; pre[k] = run; }
        const uint64_t incl = wave_incl_scan_u34(run);                 // (four 32-bit weights: below 2^34)
        if (lane == kWave - 1) L.scan[it & 1][wv] = incl;
        __syncthreads();
        uint64_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { const uint64_t s = L.scan[it & 1][w]; if (w < wv) off += s; tot += s; }
        const uint64_t excl = P + off + incl - run;                     // mass before this lane's first particle
        const int src0 = c * kTile + vb;
        const int p_all = (edge && last_shard) ? place(fc.n_pop) : 0;
        int p_prev = place_of(excl);
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            int p = place_of(excl + pre[k]);
            if (edge && last_shard && vb + k + 1 == nvt) p = p_all;      // the population's last source owns the rest
            if (edge && vb + k + 1 > nvt) p = p_prev;                   // padding slots own nothing
            if (p > p_prev) { L.slot[p_prev] = src0 + k; p_prev = p; }
        }
        P += tot;
        ++it;
        raw = raw_next;
        ++c;
    }
    __syncthreads();
    // inclusive prefix-max over the slots
    int32_t v[kPPT];
    load4(L.slot, (int64_t)tid * kPPT, v);
    lane_prefix_max(v);
    int32_t incl = wave_incl_max_i32(v[kPPT - 1]);
    if (lane == kWave - 1) L.iscr[wv] = incl;
    int32_t excl = dpp_or_i32<0x138 /* wave_shr:1 */>(incl, -1);
    if (lane == 0) excl = -1;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
        if (w < wv) excl = max(excl, L.iscr[w]);
#pragma unroll
    for (int k = 0; k < kPPT; ++k) anc[k] = max(v[k], excl);
}

}  // namespace cph
