// Counter-based random numbers and Boost-compatible variate generators for gfx950.
//
// Replaces get_rng() (reference src/cpprob/utils.cpp:16-20: one global, unseedable
// std::mt19937) and the Boost.Random 1.66 variate generators the models call
// (reference include/models/models.hpp:26,74,126,135).  One Philox4x32-10 block per
// `sample` statement: key = run seed, counter = (draw index, global particle id), i.e.
// the stream rocRAND's device engine yields for rocrand_init(seed, subsequence = pid,
// offset = 4*draw); rocrand4().  Box-Muller follows rocRAND's box_muller_double(uint4)
// (first output).  Stateless: nothing but the particle id and the statement ordinal
// lives in registers, so results do not depend on launch geometry or GPU count.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cph {

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ u32x4 draw_block(uint64_t seed, uint64_t pid, uint64_t draw)
{
    return philox4x32_10((uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)pid, (uint32_t)(pid >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

constexpr double kTwoPowM53 = 1.1102230246251565e-16;
constexpr uint64_t kResampleDrawBase = 1ull << 40;  // draw index of the resampling uniforms

__device__ __forceinline__ uint64_t bits53(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)(hi >> 11) << 32); }
// (0, 1]
__device__ __forceinline__ double u01_open0(uint32_t lo, uint32_t hi) { return kTwoPowM53 + (double)bits53(lo, hi) * kTwoPowM53; }
// [0, 1)
__device__ __forceinline__ double u01_open1(uint32_t lo, uint32_t hi) { return (double)bits53(lo, hi) * kTwoPowM53; }

// Standard normal from one block (rocRAND box_muller_double(uint4).x)
__device__ __forceinline__ double std_normal(const u32x4 r)
{
    const uint64_t v1 = (uint64_t)r.x ^ ((uint64_t)r.y << 21);
    const uint64_t v2 = (uint64_t)r.z ^ ((uint64_t)r.w << 21);
    const double u = kTwoPowM53 + (double)v1 * kTwoPowM53;
    const double w = (kTwoPowM53 * 2.0) + (double)v2 * (kTwoPowM53 * 2.0);
    const double s = sqrt(-2.0 * log(u));
    return s * sinpi(w);
}

// boost::random::normal_distribution<>{mean, sigma}(rng)
__device__ __forceinline__ double draw_normal(uint64_t seed, uint64_t pid, uint64_t draw, double mean, double sigma)
{
    return mean + sigma * std_normal(draw_block(seed, pid, draw));
}

// boost::random::uniform_smallint<size_t>{a, b}(rng)
__device__ __forceinline__ uint64_t draw_smallint(uint64_t seed, uint64_t pid, uint64_t draw, uint64_t a, uint64_t b)
{
    const u32x4 r = draw_block(seed, pid, draw);
    return a + (((uint64_t)r.x * (b - a + 1)) >> 32);
}

// boost::random::discrete_distribution<size_t>{w, w+k}(rng): inverse CDF on normalised cumulative sums
template <int K>
__device__ __forceinline__ uint32_t discrete_from_u(double u, const double (&w)[K])
{
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) tot += w[i];
    double acc = 0.0;
    uint32_t idx = 0;
#pragma unroll
    for (int i = 0; i < K - 1; ++i) {
        acc += w[i];
        if (u >= acc / tot) idx = (uint32_t)(i + 1);
    }
    return idx;
}

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

__device__ __forceinline__ double draw_u01(uint64_t seed, uint64_t pid, uint64_t draw)
{
    const u32x4 r = draw_block(seed, pid, draw);
    return u01_open1(r.x, r.y);
}

// boost::random::uniform_real_distribution<>{a, b}(rng)
__device__ __forceinline__ double draw_uniform_real(uint64_t seed, uint64_t pid, uint64_t draw, double a, double b)
{
    return a + (b - a) * draw_u01(seed, pid, draw);
}

}  // namespace cph
