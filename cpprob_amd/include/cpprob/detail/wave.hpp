// Wavefront (64 lanes) and workgroup (256 threads = 4 waves) primitives in fp64:
// reductions and scans used by the weight normalisation (log-sum-exp, ESS) and the
// resampling CDF.  gfx950 only: the wave width is hard-coded to 64 and cross-lane traffic uses
// DPP (row shifts inside a 16-lane row, row_bcast:15 / row_bcast:31 across rows), not
// ds_bpermute shuffles: no LDS-pipe round trip per step.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace cph {

constexpr int kWave = 64;
constexpr int kThreads = 256;              // workgroup size of every particle kernel
constexpr int kWaves = kThreads / kWave;   // 4: one wave per SIMD
// Particles per lane: 4.  An earlier build of these kernels was generic over 4 / 8 / 16 and 4 measured fastest at every population
// size (profiles/r01_ppt_sweep.md: 8 and 16 amortise the per-lane fixed cost but lose more to register pressure and to having
// fewer workgroups); the code since then is maintained and tested for 4 only.
#ifndef CPPROB_PPT
#define CPPROB_PPT 4
#endif
constexpr int kPPT = CPPROB_PPT;           // consecutive particles per lane, moved as 32-B fp64 / 16-B int32 vectors
static_assert(kPPT == 4, "the kernels are maintained for 4 particles per lane");
constexpr int kTile = kThreads * kPPT;     // particles per workgroup (1024)

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// DPP controls
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// Value held by the DPP source lane; `fill` where the pattern has no source (row edge) or the
// row is masked out.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_or(double v, double fill)
{
    union { double d; int i[2]; } a, f, r;
    a.d = v; f.d = fill;
    r.i[0] = __builtin_amdgcn_update_dpp(f.i[0], a.i[0], CTRL, ROW_MASK, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(f.i[1], a.i[1], CTRL, ROW_MASK, 0xf, false);
    return r.d;
}

__device__ __forceinline__ double read_lane(double v, int lane)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
    r.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
    return r.d;
}

// Inclusive prefix sum across the 64 lanes (all lanes must be active).
__device__ __forceinline__ double wave_incl_scan(double v)
{
    v += dpp_or<kDppRowShr1>(v, 0.0);
    v += dpp_or<kDppRowShr2>(v, 0.0);
    v += dpp_or<kDppRowShr4>(v, 0.0);
    v += dpp_or<kDppRowShr8>(v, 0.0);
    v += dpp_or<kDppRowBcast15, 0xA>(v, 0.0);
    v += dpp_or<kDppRowBcast31, 0xC>(v, 0.0);
    return v;
}

// int32 forms (ancestor prefix-max)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int32_t dpp_or_i32(int32_t v, int32_t fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ int32_t wave_incl_max_i32(int32_t v)
{
    v = max(v, dpp_or_i32<kDppRowShr1>(v, INT32_MIN));
    v = max(v, dpp_or_i32<kDppRowShr2>(v, INT32_MIN));
    v = max(v, dpp_or_i32<kDppRowShr4>(v, INT32_MIN));
    v = max(v, dpp_or_i32<kDppRowShr8>(v, INT32_MIN));
    v = max(v, dpp_or_i32<kDppRowBcast15, 0xA>(v, INT32_MIN));
    v = max(v, dpp_or_i32<kDppRowBcast31, 0xC>(v, INT32_MIN));
    return v;
}

// Sum / max over the wave, result in every lane (scalar broadcast of lane 63).
__device__ __forceinline__ double wave_sum(double v) { return read_lane(wave_incl_scan(v), kWave - 1); }

__device__ __forceinline__ double wave_max(double v)
{
    const double ninf = -INFINITY;
    v = fmax(v, dpp_or<kDppRowShr1>(v, ninf));
    v = fmax(v, dpp_or<kDppRowShr2>(v, ninf));
    v = fmax(v, dpp_or<kDppRowShr4>(v, ninf));
    v = fmax(v, dpp_or<kDppRowShr8>(v, ninf));
    v = fmax(v, dpp_or<kDppRowBcast15, 0xA>(v, ninf));
    v = fmax(v, dpp_or<kDppRowBcast31, 0xC>(v, ninf));
    return read_lane(v, kWave - 1);
}

// Workgroup-wide combines.  `scratch` must hold NWAVES doubles and be a region no other
// in-flight combine of the same workgroup uses: each function has ONE barrier (after the
// per-wave slots are written); callers pass distinct regions to back-to-back calls instead of
// paying a second barrier.  Sums run in wave order: bitwise reproducible.
template <int NWAVES = kWaves>
__device__ __forceinline__ double block_max(double v, double* scratch)
{
    v = wave_max(v);
    if (lane_id() == 0) scratch[wave_id()] = v;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) r = fmax(r, scratch[w]);
    return r;
}

template <int NWAVES = kWaves>
__device__ __forceinline__ double block_sum(double v, double* scratch)
{
    v = wave_sum(v);
    if (lane_id() == 0) scratch[wave_id()] = v;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) r += scratch[w];
    return r;
}

// Two sums with one barrier; scratch holds 2*NWAVES doubles.
template <int NWAVES = kWaves>
__device__ __forceinline__ void block_sum2(double& a, double& b, double* scratch)
{
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane_id() == 0) { scratch[2 * wave_id()] = a; scratch[2 * wave_id() + 1] = b; }
    __syncthreads();
    double ra = scratch[0], rb = scratch[1];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) { ra += scratch[2 * w]; rb += scratch[2 * w + 1]; }
    a = ra; b = rb;
}

// Exclusive prefix (over threads, in thread order) of one double per thread; *total = sum over the
// workgroup.  One barrier; scratch holds NWAVES doubles.
template <int NWAVES = kWaves>
__device__ __forceinline__ double block_excl_scan(double v, double* scratch, double* total)
{
    const double incl = wave_incl_scan(v);
    if (lane_id() == kWave - 1) scratch[wave_id()] = incl;
    __syncthreads();
    double off = 0.0, tot = 0.0;
    const int wv = wave_id();
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
        const double s = scratch[w];
        if (w < wv) off += s;
        tot += s;
    }
    *total = tot;
    // exclusive value = inclusive value of the previous lane (0 for lane 0 of the wave)
    double excl = dpp_or<0x138 /* wave_shr:1 */>(incl, 0.0);
    if (lane_id() == 0) excl = 0.0;
    return off + excl;
}

// One workgroup sum and one exclusive scan behind a single barrier (same operation order as block_sum /
// block_excl_scan: bit-identical results).  scratch holds 2*NWAVES doubles.
template <int NWAVES = kWaves>
__device__ __forceinline__ double block_sum_and_excl_scan(double& sum_v, double scan_v, double* scratch, double* total)
{
    const double ws = wave_sum(sum_v);
    const double incl = wave_incl_scan(scan_v);
    if (lane_id() == 0) scratch[wave_id()] = ws;
    if (lane_id() == kWave - 1) scratch[NWAVES + wave_id()] = incl;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) r += scratch[w];
    sum_v = r;
    double off = 0.0, tot = 0.0;
    const int wv = wave_id();
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
        const double s = scratch[NWAVES + w];
        if (w < wv) off += s;
        tot += s;
    }
    *total = tot;
    double excl = dpp_or<0x138 /* wave_shr:1 */>(incl, 0.0);
    if (lane_id() == 0) excl = 0.0;
    return off + excl;
}

// ---------------------------------------------------------------------------------------------
// 4-wide accesses (arrays are padded to the tile: no tails)
// ---------------------------------------------------------------------------------------------
template <class T> struct Vec4;
template <> struct Vec4<double> { using type = double __attribute__((ext_vector_type(4))); };
template <> struct Vec4<int32_t> { using type = int __attribute__((ext_vector_type(4))); };
template <> struct Vec4<int8_t> { using type = signed char __attribute__((ext_vector_type(4))); };
template <> struct Vec4<uint32_t> { using type = unsigned int __attribute__((ext_vector_type(4))); };

template <class T>
__device__ __forceinline__ void load4(const T* __restrict__ p, int64_t i, T (&v)[kPPT])
{
#pragma unroll
    for (int q = 0; q < kPPT; q += 4) {
        const typename Vec4<T>::type x = *reinterpret_cast<const typename Vec4<T>::type*>(p + i + q);
        v[q] = x[0]; v[q + 1] = x[1]; v[q + 2] = x[2]; v[q + 3] = x[3];
    }
}

template <class T>
__device__ __forceinline__ void store4(T* __restrict__ p, int64_t i, const T (&v)[kPPT])
{
#pragma unroll
    for (int q = 0; q < kPPT; q += 4) {
        typename Vec4<T>::type x;
        x[0] = v[q]; x[1] = v[q + 1]; x[2] = v[q + 2]; x[3] = v[q + 3];
        *reinterpret_cast<typename Vec4<T>::type*>(p + i + q) = x;
    }
}

// in-lane helpers over the lane's kPPT values
__device__ __forceinline__ void lane_prefix_max(int32_t (&v)[kPPT])
{
#pragma unroll
    for (int k = 1; k < kPPT; ++k) v[k] = max(v[k], v[k - 1]);
}
template <class T> __device__ __forceinline__ void lane_fill(T (&v)[kPPT], T x)
{
#pragma unroll
    for (int k = 0; k < kPPT; ++k) v[k] = x;
}

}  // namespace cph
