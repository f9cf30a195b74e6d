// Counter-based random numbers and Boost-compatible variate generators for gfx950.
//
// Replaces get_rng() (reference src/cpprob/utils.cpp:16-20: one global, unseedable
// std::mt19937) and the Boost.Random 1.66 variate generators the models call
// (reference include/models/models.hpp:26,74,126,135).
//
// Generator: Philox4x32-10, key = run seed, counter = (draw index, group id) -- the stream
// rocRAND's device engine yields for rocrand_init(seed, subsequence = group, offset = 4*draw);
// rocrand4().  The draw index of a sample statement is its ordinal in the trace.  A 128-bit
// block is shared by neighbouring particles so that one lane, which owns 4 consecutive
// particles, needs ONE Philox evaluation per discrete statement and TWO per normal statement:
//   32-bit variates (uniform_smallint, discrete, stratified offsets):
//        word (pid & 3) of block(group = pid >> 2)
//   normal variates: rocRAND's box_muller_double(block(group = pid >> 1)); particle pid takes
//        component (pid & 1): x = s*sin(pi w), y = s*cos(pi w)  (= rocrand_normal_double2)
//   53-bit uniforms (uniform_real, multinomial positions): words (2(pid&1), 2(pid&1)+1) of
//        block(group = pid >> 1), rocRAND's uniform_distribution_double(v1, v2) bit layout
// Stateless: a variate is a pure function of (seed, global particle id, statement ordinal), so
// results do not depend on launch geometry, tile size or GPU count.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpprob/detail/fastmath.hpp"

namespace cph {

struct u32x4 { uint32_t x, y, z, w; };

// a ^ b ^ c as ONE instruction (gfx950's three-input bit operation, truth table 0x96): 1.7 ns where two v_xor_b32 take 2.2
// (tools/valu_rates.hip) -- forty of them a Philox block pair, in every kernel that draws
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ u32x4 draw_block(uint64_t seed, uint64_t group, uint64_t draw)
{
    return philox4x32_10((uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)group, (uint32_t)(group >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

constexpr double kTwoPowM53 = 1.1102230246251565e-16;
constexpr double kTwoPowM32 = 2.3283064365386963e-10;
constexpr uint64_t kResampleDrawBase = 1ull << 40;  // draw index of the resampling uniforms

__device__ __forceinline__ uint32_t word_of(const u32x4& b, uint32_t i) { return i == 0 ? b.x : (i == 1 ? b.y : (i == 2 ? b.z : b.w)); }

__device__ __forceinline__ uint64_t bits53(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)(hi >> 11) << 32); }
// [0, 1) from 53 bits
__device__ __forceinline__ double u01_53(uint32_t lo, uint32_t hi) { return (double)bits53(lo, hi) * kTwoPowM53; }
// [0, 1) from 32 bits
__device__ __forceinline__ double u01_32(uint32_t w) { return (double)w * kTwoPowM32; }

// rocRAND box_muller_double(uint4): both outputs
__device__ __forceinline__ void box_muller(const u32x4 r, double& x, double& y)
{
    const uint64_t v1 = (uint64_t)r.x ^ ((uint64_t)r.y << 21);
    const uint64_t v2 = (uint64_t)r.z ^ ((uint64_t)r.w << 21);
    const double u = kTwoPowM53 + (double)v1 * kTwoPowM53;
    const double w = (kTwoPowM53 * 2.0) + (double)v2 * (kTwoPowM53 * 2.0);
    // (range-specific forms of log / sincospi, faithfully rounded like the library's: cpprob/detail/fastmath.hpp)
    const double s = sqrt(-2.0 * log01(u));
    double sn, cs;
    sincospi02(w, sn, cs);
    x = s * sn;
    y = s * cs;
}

// ---- single-particle forms (building blocks, tails) -----------------------------------------
__device__ __forceinline__ uint32_t draw_word(uint64_t seed, uint64_t pid, uint64_t draw)
{
    return word_of(draw_block(seed, pid >> 2, draw), (uint32_t)(pid & 3));
}

__device__ __forceinline__ double draw_std_normal(uint64_t seed, uint64_t pid, uint64_t draw)
{
    double x, y;
    box_muller(draw_block(seed, pid >> 1, draw), x, y);
    return (pid & 1) ? y : x;
}

__device__ __forceinline__ double draw_u01_53(uint64_t seed, uint64_t pid, uint64_t draw)
{
    const u32x4 b = draw_block(seed, pid >> 1, draw);
    return (pid & 1) ? u01_53(b.z, b.w) : u01_53(b.x, b.y);
}

// boost::random::normal_distribution<>{mean, sigma}(rng)
__device__ __forceinline__ double draw_normal(uint64_t seed, uint64_t pid, uint64_t draw, double mean, double sigma)
{
    return mean + sigma * draw_std_normal(seed, pid, draw);
}

// boost::random::uniform_smallint<size_t>{a, b}(rng): a + floor(word * range / 2^32)
__device__ __forceinline__ uint64_t smallint_from_word(uint32_t w, uint64_t a, uint64_t b) { return a + (((uint64_t)w * (b - a + 1)) >> 32); }
__device__ __forceinline__ uint64_t draw_smallint(uint64_t seed, uint64_t pid, uint64_t draw, uint64_t a, uint64_t b)
{
    return smallint_from_word(draw_word(seed, pid, draw), a, b);
}

// boost::random::discrete_distribution<size_t>{w, w+k}(rng): inverse CDF on the normalised
// cumulative sums with u = word * 2^-32
__device__ __forceinline__ uint32_t discrete_from_u_dyn(double u, const double* w, int k)
{
    double tot = 0.0;
    for (int i = 0; i < k; ++i) tot += w[i];
    double acc = 0.0;
    uint32_t idx = 0;
    for (int i = 0; i < k - 1; ++i) {
        acc += w[i];
        if (u >= acc / tot) idx = (uint32_t)(i + 1);
    }
    return idx;
}

// boost::random::uniform_real_distribution<>{a, b}(rng)
__device__ __forceinline__ double draw_uniform_real(uint64_t seed, uint64_t pid, uint64_t draw, double a, double b)
{
    return a + (b - a) * draw_u01_53(seed, pid, draw);
}

// boost::random::poisson_distribution<>{mean}(rng): inversion by sequential search on one 53-bit uniform
// (k = 0; p = F = exp(-mean); while u > F: ++k, p *= mean / k, F += p).  Exact law; cost O(mean).
__device__ __forceinline__ int64_t poisson_from_u(double u, double mean)
{
    int64_t k = 0;
    double p = exp(-mean), F = p;
    while (u > F && k < 100000) { ++k; p *= mean / (double)k; F += p; }
    return k;
}
__device__ __forceinline__ int64_t draw_poisson(uint64_t seed, uint64_t pid, uint64_t draw, double mean)
{
    return poisson_from_u(draw_u01_53(seed, pid, draw), mean);
}

// ---- 4 consecutive particles per lane ---------------------------------------------------------
// 32-bit words of particles pid0 .. pid0+3 (one Philox evaluation when pid0 % 4 == 0)
__device__ __forceinline__ void draw_words4(uint64_t seed, uint64_t pid0, uint64_t draw, uint32_t (&w)[4])
{
    const u32x4 a = draw_block(seed, pid0 >> 2, draw);
    const uint32_t sh = (uint32_t)(pid0 & 3);
    if (sh == 0) {
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    } else {
        const u32x4 b = draw_block(seed, (pid0 >> 2) + 1, draw);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) w[k] = (sh + k) < 4 ? word_of(a, sh + k) : word_of(b, sh + k - 4);
    }
}

// standard normals of particles pid0 .. pid0+3 (two Philox + two Box-Muller when pid0 is even)
__device__ __forceinline__ void draw_std_normals4(uint64_t seed, uint64_t pid0, uint64_t draw, double (&z)[4])
{
    if ((pid0 & 1) == 0) {
        box_muller(draw_block(seed, pid0 >> 1, draw), z[0], z[1]);
        box_muller(draw_block(seed, (pid0 >> 1) + 1, draw), z[2], z[3]);
    } else {
        double d;
        box_muller(draw_block(seed, pid0 >> 1, draw), d, z[0]);
        box_muller(draw_block(seed, (pid0 >> 1) + 1, draw), z[1], z[2]);
        box_muller(draw_block(seed, (pid0 >> 1) + 2, draw), z[3], d);
    }
}

}  // namespace cph
