#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
// the non-resampling fixed-point step's memory pattern, nothing else: per particle read state (S bytes) + log-weight 8, write state S + q 4 + log-weight 8
template <class S>
__global__ __launch_bounds__(256) void pattern(const S* __restrict__ prev, const double* __restrict__ lw_in, S* __restrict__ next, uint32_t* __restrict__ q, double* __restrict__ lw_out, int64_t n, int work)
{
    const int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j0 >= n) return;
    S p[4]; double w[4];
    if (sizeof(S) == 1) { *reinterpret_cast<uint32_t*>(p) = *reinterpret_cast<const uint32_t*>(prev + j0); }
    else { for (int k = 0; k < 4; ++k) p[k] = prev[j0 + k]; }
    const double4 v = *reinterpret_cast<const double4*>(lw_in + j0);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    uint32_t qq[4];
    for (int k = 0; k < 4; ++k) {
        double a = w[k];
        for (int i = 0; i < work; ++i) a = fma(a, 0.999, 0.25);         // stand-in for the arithmetic (dependent chain)
        w[k] = a; qq[k] = (uint32_t)(a * 3.0); p[k] = (S)(p[k] + (S)1);
    }
    if (sizeof(S) == 1) *reinterpret_cast<uint32_t*>(next + j0) = *reinterpret_cast<uint32_t*>(p);
    else { for (int k = 0; k < 4; ++k) next[j0 + k] = p[k]; }
    *reinterpret_cast<uint4*>(q + j0) = uint4{qq[0], qq[1], qq[2], qq[3]};
    *reinterpret_cast<double4*>(lw_out + j0) = double4{w[0], w[1], w[2], w[3]};
}
template <class S> void run(const char* name, int64_t n)
{
    S *a, *b; double *l0, *l1; uint32_t* q;
    hipMalloc(&a, n * sizeof(S)); hipMalloc(&b, n * sizeof(S)); hipMalloc(&l0, n * 8); hipMalloc(&l1, n * 8); hipMalloc(&q, n * 4);
    hipMemset(a, 0, n * sizeof(S)); hipMemset(l0, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {0, 50, 100, 200}) {
        const int grid = (int)((n / 4 + 255) / 256);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pattern<S>, dim3(grid), dim3(256), 0, 0, a, l0, b, q, l1, n, work);
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(pattern<S>, dim3(grid), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? l1 : l0, (i & 1) ? a : b, q, (i & 1) ? l0 : l1, n, work); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, bytes = (double)n * (2 * sizeof(S) + 20);
        printf("%s n %lld work %d: %.1f us per launch, %.0f MB -> %.2f TB/s\n", name, (long long)n, work, us, bytes / 1e6, bytes / us / 1e6);
    }
    hipFree(a); hipFree(b); hipFree(l0); hipFree(l1); hipFree(q);
}
int main() { run<int8_t>("hmm(1B)", 12500000); run<double>("lgssm(8B)", 10000000); run<int8_t>("hmm(1B)", 100000000); return 0; }
