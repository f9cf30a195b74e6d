// Issue cost of the vector instructions the step kernels are made of, measured on the chip: cycles a SIMD spends per wavefront
// instruction with every SIMD full of wavefronts that issue nothing else (eight independent chains a lane, so latency is hidden).
// Why: a "VALU floor" priced at 4 cycles an instruction undercounts kernels whose hot parts are 32x32->64 multiplies (Philox) and
// fp64 reciprocals; this prints the weights to price them with (profiles/r06_notes.md section 7).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rates tools/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int kIters = 2048;
constexpr int kChains = 8;

#define CHAIN8(stmt) { stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7) }

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t* out, uint32_t seed)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint64_t t_shader = __builtin_readcyclecounter(), t_wall = wall_clock64();      // s_memtime (shader clock), s_memrealtime (100 MHz)
    uint32_t a[kChains], b[kChains];
    uint64_t w[kChains];
    double d[kChains];
    float f[kChains];
    for (int k = 0; k < kChains; ++k) {
        a[k] = tid * 2654435761u + k + seed; b[k] = a[k] ^ 0x9e3779b9u; w[k] = ((uint64_t)a[k] << 32) | b[k];
        d[k] = 1.0 + (double)(a[k] & 1023) * 1e-6; f[k] = 1.0f + (float)(a[k] & 1023) * 1e-6f;
    }
    const uint32_t m0 = 0xD2511F53u + seed;
    const double c0 = 0.999999 + (double)seed * 1e-9, c1 = 1e-7;
    for (int i = 0; i < kIters; ++i) {
        if constexpr (OP == 0) {
#define S(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 1) {
#define S(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[k]) : "v"(a[k]), "v"(m0) : "vcc");
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 2) {
#define S(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m0));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 3) {
#define S(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m0));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 4) {
#define S(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(c0), "v"(c1));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 5) {
#define S(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(c1));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 6) {
#define S(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(c0));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 7) {
#define S(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 8) {
#define S(k) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[k]) : "v"(a[k])); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[k]) : "v"(d[k]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 9) {
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k]) : "v"(f[(k + 1) & 7]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 10) {
#define S(k) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 11) {
#define S(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 12) {
#define S(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[k]) : "v"(w[(k + 1) & 7]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 13) {
#define S(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[k]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 14) {
#define S(k) asm volatile("v_log_f32 %0, %0" : "+v"(f[k]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 15) {
#define S(k) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[k]), "+v"(b[k]) : "v"(m0), "v"(m0) : "vcc");
            // (operands deliberately crossed: two instructions, one carry chain)
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 16) {
#define S(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[k]) : "v"(c0));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 18) {
            // the Philox shape: product only (addend 0), the next round's input is the product's high word
#define S(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[k]) : "v"((uint32_t)(w[k] >> 32)), "v"(m0) : "vcc");
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 19) {
#define S(k) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[k]) : "v"(b[k]), "v"(m0));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 20) {
#define S(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            CHAIN8(S)
#undef S
        } else if constexpr (OP == 17) {
#define S(k) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[k]) : "v"(d[k]), "v"(c0), "v"(b[k]) : "vcc");
            CHAIN8(S)
#undef S
        }
    }
    uint64_t acc = 0;
    for (int k = 0; k < kChains; ++k) acc += a[k] + b[k] + w[k] + (uint64_t)d[k] + (uint64_t)f[k];
    if (acc == 0x1234567812345678ull) out[tid & 63] = acc;                    // keeps the chains alive
    if (tid == 0) { out[64] = __builtin_readcyclecounter() - t_shader; out[65] = wall_clock64() - t_wall; }
}

// The same question for DEPENDENT instructions: CHAINS independent chains a lane (1 = every instruction waits for the one before),
// WAVES wavefronts a SIMD.  What a wavefront's own instruction-level parallelism is worth when the SIMD has other wavefronts to issue from.
template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void chain_kernel(uint64_t* out, uint32_t seed)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    double d[CHAINS]; uint64_t w[CHAINS]; uint32_t a[CHAINS];
    for (int k = 0; k < CHAINS; ++k) { a[k] = tid * 2654435761u + k + seed; w[k] = ((uint64_t)a[k] << 32) | (a[k] ^ 0x9e3779b9u); d[k] = 1.0 + (double)(a[k] & 1023) * 1e-6; }
    const uint32_t m0 = 0xD2511F53u + seed;
    const double c0 = 0.999999 + (double)seed * 1e-9, c1 = 1e-7;
    for (int i = 0; i < kIters * 8 / CHAINS; ++i) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) {
            if constexpr (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m0));
            else if constexpr (OP == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[k]) : "v"((uint32_t)(w[k] >> 32)), "v"(m0) : "vcc");
            else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(c0), "v"(c1));
        }
    }
    uint64_t acc = 0;
    for (int k = 0; k < CHAINS; ++k) acc += a[k] + w[k] + (uint64_t)d[k];
    if (acc == 0x1234567812345678ull) out[tid & 63] = acc;
}
template <int OP, int CHAINS> void run_chain(const char* name, int waves_per_simd, uint64_t* out)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4;
    const int blocks = simds * waves_per_simd / 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chain_kernel<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((chain_kernel<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, 2u + i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)reps * waves_per_simd * kIters * 8);
    std::printf("%-14s %d chain(s) a lane, %d wavefronts a SIMD: %6.3f ns per wavefront instruction\n", name, CHAINS, waves_per_simd, ns);
}
template <int OP> void run_chains(const char* name, uint64_t* out)
{
    for (int w : {8, 4, 2, 1}) { run_chain<OP, 1>(name, w, out); run_chain<OP, 2>(name, w, out); run_chain<OP, 4>(name, w, out); run_chain<OP, 8>(name, w, out); }
}

template <int OP> double run(const char* name, int per_iter, uint64_t* out, double base)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4, waves_per_simd = 8;
    const int blocks = simds * waves_per_simd / 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u + i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)reps * waves_per_simd * kIters * kChains * per_iter;
    const double ns = ms * 1e6 / insts_per_simd;                              // per wavefront instruction on one SIMD
    uint64_t clk[2]; hipMemcpy(clk, out + 64, sizeof clk, hipMemcpyDeviceToHost);
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);             // shader cycles per ns while the kernel ran (wall clock: 100 MHz)
    std::printf("%-38s %6.3f ns per wavefront instruction = %5.2f x v_add_u32;  shader clock %.2f GHz -> %5.2f cycles\n", name, ns, base > 0 ? ns / base : 1.0, ghz, ns * ghz);
    return ns;
}

int main()
{
    uint64_t* out; hipMalloc(&out, 66 * 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    std::printf("%s, %d CUs, clock %d MHz; every SIMD holds 8 wavefronts that issue the one instruction, 8 independent chains a lane\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    const double base = run<0>("v_add_u32", 1, out, 0);
    run<1>("v_mad_u64_u32 (accumulating)", 1, out, base);
    run<18>("v_mad_u64_u32 (product only: Philox)", 1, out, base);
    run<2>("v_mul_lo_u32", 1, out, base);
    run<3>("v_mul_hi_u32", 1, out, base);
    run<11>("v_mul_u32_u24", 1, out, base);
    run<20>("v_xor_b32", 1, out, base);
    run<19>("v_bitop3_b32 (a ^ b ^ c)", 1, out, base);
    run<12>("v_lshl_add_u64", 1, out, base);
    run<15>("v_add_co_u32 + v_addc_co_u32", 2, out, base);
    run<4>("v_fma_f64", 1, out, base);
    run<5>("v_add_f64", 1, out, base);
    run<6>("v_mul_f64", 1, out, base);
    run<17>("v_cmp_lt_f64 + v_cndmask_b32", 2, out, base);
    run<7>("v_rcp_f64", 1, out, base);
    run<13>("v_rsq_f64", 1, out, base);
    run<8>("v_cvt_f64_u32 + v_cvt_u32_f64", 2, out, base);
    run<9>("v_fma_f32", 1, out, base);
    run<16>("v_pk_fma_f32", 1, out, base);
    run<10>("v_exp_f32", 1, out, base);
    run<14>("v_log_f32", 1, out, base);
    run_chains<0>("v_add_u32", out); run_chains<1>("v_mad_u64_u32", out); run_chains<2>("v_fma_f64", out);
    return 0;
}
