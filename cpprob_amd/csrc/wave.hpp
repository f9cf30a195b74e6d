// Wavefront (64 lanes) and workgroup (256 threads = 4 waves) primitives in fp64:
// reductions and scans used by the weight normalisation (log-sum-exp, ESS) and the
// resampling CDF.  gfx950 only: the wave width is hard-coded to 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cph {

constexpr int kWave = 64;
constexpr int kThreads = 256;              // workgroup size of every particle kernel
constexpr int kWaves = kThreads / kWave;   // 4: one wave per SIMD
constexpr int kPPT = 4;                    // consecutive particles per lane (32-B fp64 / 16-B int32 accesses)
constexpr int kTile = kThreads * kPPT;     // 1024 particles per workgroup

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// DPP move of a double (two 32-bit halves).  Lanes whose source is out of range keep `v`
// (old = v, bound_ctrl = false).
template <int DPP_CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_mov(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(a.i[0], a.i[0], DPP_CTRL, ROW_MASK, BANK_MASK, false);
    r.i[1] = __builtin_amdgcn_update_dpp(a.i[1], a.i[1], DPP_CTRL, ROW_MASK, BANK_MASK, false);
    return r.d;
}

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, kWave); }
__device__ __forceinline__ double shfl_up_d(double v, int d) { return __shfl_up(v, d, kWave); }

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, shfl_xor_d(v, m));
    return v;
}

// Inclusive prefix sum across the 64 lanes.
__device__ __forceinline__ double wave_incl_scan(double v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const double o = shfl_up_d(v, d);
        if (l >= d) v += o;
    }
    return v;
}

// Workgroup-wide max / sum; result valid in every thread.  `scratch` holds >= 2*kWavesMax doubles
// and is reused; the functions synchronise before returning so back-to-back calls are safe.
template <int NWAVES = kWaves>
__device__ __forceinline__ double block_max(double v, double* scratch)
{
    v = wave_max(v);
    if (lane_id() == 0) scratch[wave_id()] = v;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) r = fmax(r, scratch[w]);
    __syncthreads();
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
    for (int w = 1; w < NWAVES; ++w) r += scratch[w];   // fixed order: bitwise reproducible
    __syncthreads();
    return r;
}

// Exclusive prefix (over threads, in thread order) of one double per thread; *total = sum over the
// workgroup.
template <int NWAVES = kWaves>
__device__ __forceinline__ double block_excl_scan(double v, double* scratch, double* total)
{
    const double incl = wave_incl_scan(v);
    if (lane_id() == kWave - 1) scratch[wave_id()] = incl;
    __syncthreads();
    double off = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
        const double s = scratch[w];
        if (w < wave_id()) off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    double excl = shfl_up_d(incl, 1);
    if (lane_id() == 0) excl = 0.0;
    return off + excl;
}

}  // namespace cph
