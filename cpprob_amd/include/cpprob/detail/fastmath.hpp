// Lean fp64 elementary functions for the particle kernels (device only).
//
// The Gaussian SIS kernel is bound by vector-ALU issue (profiles/r02_sis_sq_counters.md: VALU busy ~100 %, 1171 VALU instructions per
// wave of 256 particles), and more than a third of those instructions are the general-purpose log / sincospi / exp of the device
// math library, which pay for argument ranges, special values and (log, sincospi) double-double intermediates that the variate
// generators never exercise.  These forms serve exactly the ranges the callers have:
//   log01(u)       u in [2^-53, 1]            (Box-Muller radius; fdlibm's e_log.c reduction and minimax polynomial)
//   sincospi02(w)  w in (0, 2]                (Box-Muller angle;  quadrant reduction + minimax polynomials on [-1/4, 1/4])
//   exp_nonpos(x)  x in [-745, ~0]            (linear weights exp(logw - reference); Cody-Waite reduction + minimax polynomial)
// Error of each in ulps of the result (against 300-bit references, emulating the fp64 operations exactly; tools/fit_math.py regenerates
// the coefficients and the figures; tests/test_gpu_blocks.py::test_fastmath_* measures them on the device against 80-bit references):
// log01 <= 0.67, exp_nonpos <= 0.66 (faithful rounding, as the library's own), sincospi02: sin <= 0.99, cos <= 1.03 -- the cosine's
// worst case sits at |t| -> 1/4, where 1 + s Q(s) lands just above 1/2 with s Q = -0.29 (one ulp of the Box-Muller angle's cosine;
// a compensated last step would cost two more vector instructions per variate in a kernel bound by vector issue).  The parity tests
// compare draws with glibc-based values at 1e-12 relative (2 ulp = 4.4e-16).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace cph {

// One Horner step as ONE instruction.  hipcc turns fma(p, r, c) with a loop-invariant coefficient held in a VGPR into
// v_mov_b64 + v_fmac_f64 (the two-address form needs a scratch copy of c): an extra vector-issue slot per step in kernels whose
// bound IS vector issue.  The three-address VOP3 form takes c where it lies -- and it lies in a SCALAR register pair ("s": one
// scalar source per VOP3 instruction on this part): the coefficients of exp / log / sincos cost no vector registers and no
// v_mov to materialise (linear-Gaussian fixed-point step 97 -> 87 registers, four -> five wavefronts a SIMD; same bits).
__device__ __forceinline__ double horner(double p, double r, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "s"(c));
    return d;
}

__device__ __forceinline__ double log01(double u)
{
    int e;
    double m = frexp(u, &e);                                   // m in [0.5, 1)
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;                                       // m in [sqrt(1/2), sqrt(2))
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double k = (double)e;
    // s = f / (2 + f): reciprocal seed + two Newton steps + one residual correction
    const double d = 2.0 + f;
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    double s = f * y;
    s = fma(fma(-d, s, f), y, s);
    const double z = s * s, w = z * z;
    const double t1 = w * horner(horner(1.531383769920937332e-01, w, 2.222219843214978396e-01), w, 3.999999999940941908e-01);
    const double t2 = z * horner(horner(horner(1.479819860511658591e-01, w, 1.818357216161805012e-01), w, 2.857142874366239149e-01), w, 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return k * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + k * 1.90821492927058770002e-10)) - f);
}

__device__ __forceinline__ void sincospi02(double w, double& sn, double& cs)
{
    const double r = rint(w + w);                              // 0 .. 4
    const int i = (int)r;
    const double t = fma(-0.5, r, w);                          // [-1/4, 1/4], exact
    const double s = t * t;
    double c = -0x1.b2f3fd83ea607p-14;
    c = horner(c, s, 0x1.f9ce249429ee5p-10);
    c = horner(c, s, -0x1.a6d1eef4b82a8p-6);
    c = horner(c, s, 0x1.e1f5068689166p-3);
    c = horner(c, s, -0x1.55d3c7e3cb243p+0);
    c = horner(c, s, 0x1.03c1f081b5ac0p+2);
    c = horner(c, s, -0x1.3bd3cc9be45dep+2);
    c = horner(c, s, 1.0);
    double p = 0x1.e4a9f3937d930p-12;
    p = horner(p, s, -0x1.e3027e2cbc1fbp-8);
    p = horner(p, s, 0x1.50783208be31cp-4);
    p = horner(p, s, -0x1.32d2cce50061ep-1);
    p = horner(p, s, 0x1.466bc6775a478p+1);
    p = horner(p, s, -0x1.4abbce625be52p+2);
    p = p * (s * t);
    double q = fma(t, 3.14159265358979311600e+00, p);
    if (i & 2) { q = -q; c = -c; }
    if (i & 1) { const double tmp = -q; q = c; c = tmp; }
    sn = q; cs = c;
}

__device__ __forceinline__ double exp_nonpos(double x)
{
    const double k = rint(x * 0x1.71547652b82fep+0);
    double r = fma(-k, 0x1.62e42fefa3800p-1, x);
    r = fma(-k, 0x1.ef35793c76730p-45, r);
    double p = 0x1.1f8b4cd99e7aap-29;
    p = horner(p, r, 0x1.af4dea2bc3f25p-26);
    p = horner(p, r, 0x1.27e4cccda6fcfp-22);
    p = horner(p, r, 0x1.71de023137276p-19);
    p = horner(p, r, 0x1.a01a01acfae99p-16);
    p = horner(p, r, 0x1.a01a01abe8206p-13);
    p = horner(p, r, 0x1.6c16c16c151fcp-10);
    p = horner(p, r, 0x1.11111111100dbp-7);
    p = horner(p, r, 0x1.5555555555558p-5);
    p = horner(p, r, 0x1.5555555555557p-3);
    p = horner(p, r, 0.5);
    p = fma(p, r * r, r);
    return ldexp(p + 1.0, (int)k);
}

}  // namespace cph
