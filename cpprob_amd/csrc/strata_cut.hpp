// Multinomial resampling (strata form) of ONE population held by several ranks: what the ranks' boundaries cut.
//
// The thresholds are generated stratum by stratum (step_fixed.hpp): stratum w holds the outputs [offs[w], offs[w + 1]) and its
// thresholds lie in [B_w, B_w+1); rank r's sources hold the mass range [P_r, P_r+1) (all-gathered totals).  A stratum inside one
// rank's range hands ALL its outputs to that rank's sources -- one interval of outputs per rank, the shape the exchange plan has
// under systematic resampling -- and only the <= world - 1 strata a rank boundary CUTS look at their outputs one by one:
//     lo_b = min{w : B_w >= P_b},   hi_b = max{w : B_w <= P_b}   (table CDF: B_w < P_b -- a threshold there may round UP to B_w+1),
//     A_b = offs[lo_b],  Z_b = offs[hi_b]:   rank r's regular outputs = [A_r, Z_r+1),   boundary b's cut stratum = [Z_b, A_b).
// exchange_cut_kernel (exchange.hpp) writes, per boundary, {A_b, Z_b} and a table over the cut stratum's outputs:
//     src(s) = max{r : P_r <= tau_s}  |  cum(s) = #{s' < s in the stratum : src(s') = the rank whose shard holds s'}   ("home" outputs)
// -- a pure function of the all-gathered totals and Philox, so every rank holds the same tables without talking.
// An output that does not descend from its own shard's sources takes the next free annex column of its shard IN OUTPUT ORDER:
//     col(s) = (s - begin_d) - kept_before(d, s),    kept_before(d, s) = #{s' in [begin_d, s) : src(s') = d}
// = interval arithmetic on the regular part + two table look-ups (the cut strata at rank d's two boundaries).  The sending rank's
// packing launch and the receiving rank's next step evaluate the same expression.  Host statement: cpprob_amd/distributed.py
// (StrataCutPlan), checked against the definition on the CPU (tests/test_oracle.py).
#pragma once
#include "cpprob/detail/fixed_mass.hpp"

namespace cph {

constexpr int kCutCap = 8192;               // outputs of a cut stratum the table holds (a stratum's count is Binomial(N, 1/K), mean <= 256)
constexpr int kCutRow = kCutCap + 1;        // + the stratum's total
constexpr uint32_t kCutCumMask = 0xffffffu;
constexpr int kCutSlots = 64;               // boundaries 0 .. world (world <= 63)
struct CutHead { uint32_t A, Z, over, pad; };
struct CutView {
    const uint32_t* tab;                    // [kCutSlots][kCutRow]: src << 24 | cum, entry [len] = the stratum's home outputs
    const CutHead* head;                    // [kCutSlots]: boundary b = 0 .. world ({0, 0} and {N, N} at the ends)
    const uint32_t* srccnt;                 // [kCutSlots][kCutSlots]: outputs of boundary b's cut stratum by source rank
};

// the 53 bits of the uniforms of the four outputs uid .. uid + 3 (pair uid & 1 of Philox block uid >> 1, draw kResampleDrawBase2 + step)
__device__ __forceinline__ void strata_bits4(uint64_t seed, uint64_t draw, uint64_t uid, uint64_t (&v)[4])
{
    const u32x4 b0 = draw_block(seed, uid >> 1, draw), b1 = draw_block(seed, (uid >> 1) + 1, draw);
    if ((uid & 1) == 0) { v[0] = bits53(b0.x, b0.y); v[1] = bits53(b0.z, b0.w); v[2] = bits53(b1.x, b1.y); v[3] = bits53(b1.z, b1.w); }
    else {                                                              // (a shard that starts at an odd particle: kernel-uniform)
        const u32x4 b2 = draw_block(seed, (uid >> 1) + 2, draw);
        v[0] = bits53(b0.z, b0.w); v[1] = bits53(b1.x, b1.y); v[2] = bits53(b1.z, b1.w); v[3] = bits53(b2.x, b2.y);
    }
}

// What kept_before needs of rank d's two boundaries (wave-uniform when d is).
struct KeptCtx { uint32_t sb, Ad, Zd1; uint32_t st[2], en[2], base[2]; const uint32_t* tab[2]; };
__device__ __forceinline__ KeptCtx kept_ctx(const CutView& cv, int d, uint32_t sb)
{
    KeptCtx k;
    const CutHead h0 = cv.head[d], h1 = cv.head[d + 1];
    k.sb = sb; k.Ad = h0.A; k.Zd1 = h1.Z;
    const bool same = h0.A > h0.Z && h0.A == h1.A && h0.Z == h1.Z;     // both boundaries inside one stratum: counted once
    k.st[0] = h0.Z; k.en[0] = h0.A; k.tab[0] = cv.tab + (size_t)d * kCutRow;
    k.st[1] = h1.Z; k.en[1] = same ? h1.Z : h1.A; k.tab[1] = cv.tab + (size_t)(d + 1) * kCutRow;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t x0 = min(max(sb, k.st[i]), k.en[i]);
        const uint32_t at = x0 - k.st[i];
        k.base[i] = k.en[i] > k.st[i] ? (k.tab[i][at < (uint32_t)kCutCap ? at : (uint32_t)kCutCap] & kCutCumMask) : 0u;
    }
    return k;
}
// outputs of [begin_d, s) that descend from rank d's own sources; s in [begin_d, end_d]
__device__ __forceinline__ uint32_t kept_before(const KeptCtx& k, uint32_t s)
{
    const uint32_t lo = max(k.sb, k.Ad), hi = min(s, k.Zd1);
    uint32_t kept = hi > lo ? hi - lo : 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (k.en[i] > k.st[i]) {
            const uint32_t at = min(max(s, k.st[i]), k.en[i]) - k.st[i];
            kept += (k.tab[i][at < (uint32_t)kCutCap ? at : (uint32_t)kCutCap] & kCutCumMask) - k.base[i];
        }
    }
    return kept;
}

// The strata of a lane's four consecutive outputs s0 .. s0 + 3 among the strata w0 .. w1 of its output tile (strata_window): for every
// output the largest w with offs[w] <= s (an empty stratum is never the answer), or in = false where the output lies outside
// [offs[w0], offs[w1 + 1]).  The window's first outputs sit in ONE register, a stratum a lane (one load); every lane searches it for its
// own outputs through the crossbar (ds_bpermute: no memory, no loop over the strata -- with four to eight strata a tile that loop, run
// by every lane for every stratum, cost a 10^7-particle step 12 %).  Wavefront-uniform: offs, k, w0, w1.
__device__ __forceinline__ void lane_strata4(const uint32_t* __restrict__ offs, int k, int w0, int w1, uint32_t s0, int (&w)[4], bool (&in)[4])
{
    const int lane = lane_id();
    const int K1 = 1 << k;
    const int n_str = w1 - w0 + 1;
    if (n_str < kWave) {
        const uint32_t o_reg = offs[w0 + lane <= K1 ? w0 + lane : K1];
        const uint32_t o_first = (uint32_t)__builtin_amdgcn_readlane((int)o_reg, 0), o_end = (uint32_t)__builtin_amdgcn_readlane((int)o_reg, n_str);
        const int n_steps = n_str > 1 ? 32 - __builtin_clz((unsigned)(n_str - 1)) : 0;
        auto find = [&](uint32_t s) -> int {
            int lo = 0, hi = n_str - 1;
            for (int st = 0; st < n_steps; ++st) {
                const int mid = (lo + hi + 1) >> 1;
                const uint32_t val = (uint32_t)__builtin_amdgcn_ds_bpermute(mid << 2, (int)o_reg);
                const bool le = val <= s;
                lo = le ? mid : lo; hi = le ? hi : mid - 1;
            }
            return lo;
        };
        const int r0 = find(s0), r3 = find(s0 + 3u);
        int r1 = r0, r2 = r0;
        if (__any(r0 != r3)) { r1 = find(s0 + 1u); r2 = find(s0 + 2u); }       // (a stratum ends inside some lane's four outputs)
        w[0] = w0 + r0; w[1] = w0 + r1; w[2] = w0 + r2; w[3] = w0 + r3;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t s = s0 + (uint32_t)i; in[i] = s >= o_first && s < o_end; }
        return;
    }
    // (sixty-four strata or more for one tile of outputs -- strata of a handful of thresholds: the plain loop)
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = w0; in[i] = false; }
    uint32_t o_lo = offs[w0];
    for (int ww = w0; ww <= w1; ++ww) {
        const uint32_t o_hi = offs[ww + 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t s = s0 + (uint32_t)i; if (s >= o_lo && s < o_hi) { w[i] = ww; in[i] = true; } }
        o_lo = o_hi;
    }
}

}  // namespace cph
