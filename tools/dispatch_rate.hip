#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// How fast does the chip start workgroups?  The same bytes (1-byte state + 8-byte log-weight in, state + 4 + 8 out), PPL particles a lane.
template <int PPL>
__global__ __launch_bounds__(256) void pattern(const uint8_t* __restrict__ prev, const double* __restrict__ lw_in, uint8_t* __restrict__ next, uint32_t* __restrict__ q, double* __restrict__ lw_out, int64_t n)
{
    const int64_t base = ((int64_t)blockIdx.x * 256) * PPL;
#pragma unroll
    for (int c = 0; c < PPL / 4; ++c) {
        const int64_t j0 = base + ((int64_t)c * 256 + threadIdx.x) * 4;
        if (j0 >= n) return;
        uint32_t p = *reinterpret_cast<const uint32_t*>(prev + j0);
        const double4 v = *reinterpret_cast<const double4*>(lw_in + j0);
        p += 0x01010101u;
        *reinterpret_cast<uint32_t*>(next + j0) = p;
        *reinterpret_cast<uint4*>(q + j0) = uint4{(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w};
        *reinterpret_cast<double4*>(lw_out + j0) = double4{v.x + 1.0, v.y + 1.0, v.z + 1.0, v.w + 1.0};
    }
}
__global__ __launch_bounds__(256) void empty_kernel(int* p) { if (p && threadIdx.x == 1024) *p = 1; }
template <int PPL> void run(int64_t n, uint8_t* a, uint8_t* b, double* l0, double* l1, uint32_t* q)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = (int)((n + 256 * PPL - 1) / (256 * PPL));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pattern<PPL>, dim3(grid), dim3(256), 0, 0, a, l0, b, q, l1, n);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(pattern<PPL>, dim3(grid), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? l1 : l0, (i & 1) ? a : b, q, (i & 1) ? l0 : l1, n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)n * 22;
    printf("n %lld, %d particles a lane, %d workgroups: %.1f us, %.2f TB/s, %.0f workgroups/us\n", (long long)n, PPL, grid, us, bytes / us / 1e6, grid / us);
}
int main()
{
    const int64_t n = 12500000;
    uint8_t *a, *b; double *l0, *l1; uint32_t* q;
    (void)hipMalloc(&a, n + 65536); (void)hipMalloc(&b, n + 65536); (void)hipMalloc(&l0, (n + 65536) * 8); (void)hipMalloc(&l1, (n + 65536) * 8); (void)hipMalloc(&q, (n + 65536) * 4);
    (void)hipMemset(a, 0, n); (void)hipMemset(l0, 0, n * 8);
    run<4>(n, a, b, l0, l1, q); run<8>(n, a, b, l0, l1, q); run<16>(n, a, b, l0, l1, q); run<32>(n, a, b, l0, l1, q);
    // empty workgroups: the dispatcher alone
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {1024, 4096, 12208, 48832, 97664}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, (int*)nullptr);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, (int*)nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel, %d workgroups of 256: %.2f us a launch, %.0f workgroups/us\n", grid, ms * 1e3 / 20, grid / (ms * 1e3 / 20));
    }
    return 0;
}
