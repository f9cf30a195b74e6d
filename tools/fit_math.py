#!/usr/bin/env python3
"""Generator and error report of cpprob_amd/include/cpprob/detail/fastmath.hpp (log01, sincospi02, exp_nonpos).
`python tools/fit_math.py` prints the minimax coefficients (hex doubles, as the header spells them) and the worst error of each
algorithm in units of 2^-53 relative, against 300-bit mpmath references with every fp64 operation emulated exactly.  The same
functions are checked on the device by tests/test_gpu_blocks.py::test_fastmath_* through cpprob_hip_fastmath."""
# Derives polynomial coefficients for the lean fp64 device math (sinpi/cospi on [-1/4,1/4], exp on [-ln2/2, ln2/2]) and checks the
# algorithms' error against mpmath with an exact-rounding emulation of fp64 fma.
import mpmath as mp, numpy as np, random
mp.mp.prec = 300

def cheb_fit(f, a, b, deg):
    # interpolate at Chebyshev nodes of [a,b], return monomial coefficients (high precision)
    n = deg + 1
    xs = [ (a+b)/2 + (b-a)/2*mp.cos(mp.pi*(2*k+1)/(2*n)) for k in range(n)]
    A = mp.matrix(n, n); y = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n): A[i, j] = x**j
        y[i] = f(x)
    c = mp.lu_solve(A, y)
    return [c[j] for j in range(n)]

def remez_like(f, a, b, deg, iters=6, weight=lambda x: 1):
    # a few exchange steps starting from Chebyshev nodes (good enough: we check the final error explicitly)
    n = deg + 2
    xs = [ (a+b)/2 - (b-a)/2*mp.cos(mp.pi*k/(n-1)) for k in range(n)]
    for _ in range(iters):
        A = mp.matrix(n, n); y = mp.matrix(n, 1)
        for i, x in enumerate(xs):
            for j in range(deg+1): A[i, j] = x**j
            A[i, deg+1] = (-1)**i * weight(x)
            y[i] = f(x)
        sol = mp.lu_solve(A, y)
        c = [sol[j] for j in range(deg+1)]
        err = lambda x: (sum(c[j]*x**j for j in range(deg+1)) - f(x)) / weight(x)
        # new extrema: scan
        grid = [a + (b-a)*mp.mpf(k)/4000 for k in range(4001)]
        vals = [err(x) for x in grid]
        ext = []
        for k in range(4001):
            l = vals[k-1] if k > 0 else None; r = vals[k+1] if k < 4000 else None
            v = vals[k]
            if (l is None or abs(v) >= abs(l)) and (r is None or abs(v) >= abs(r)) and (not ext or mp.sign(v) != mp.sign(ext[-1][1]) or abs(v) > abs(ext[-1][1])):
                if ext and mp.sign(v) == mp.sign(ext[-1][1]): ext[-1] = (grid[k], v)
                else: ext.append((grid[k], v))
        if len(ext) < n: break
        # keep n largest alternating
        while len(ext) > n:
            if abs(ext[0][1]) < abs(ext[-1][1]): ext.pop(0)
            else: ext.pop()
        xs = [e[0] for e in ext]
    return c, max(abs(v) for v in vals)

def f64(x): return float(x)
def fma(a, b, c): return float(mp.mpf(a)*mp.mpf(b) + mp.mpf(c))

def hexd(x): return float(x).hex()

# ---- sin(pi t) = t*(pi + s*P(s)), cos(pi t) = 1 + s*Q(s), s = t^2 in [0, 1/16]
def Pf(s):
    t = mp.sqrt(s)
    return (mp.sin(mp.pi*t)/t - mp.pi)/s if s != 0 else -(mp.pi**3)/6
def Qf(s):
    t = mp.sqrt(s)
    return (mp.cos(mp.pi*t) - 1)/s if s != 0 else -(mp.pi**2)/2
P, eP = remez_like(Pf, mp.mpf(0), mp.mpf(1)/16, 5)
Q, eQ = remez_like(Qf, mp.mpf(0), mp.mpf(1)/16, 6)
print("sinpi P deg5 fit err", mp.nstr(eP, 3), " cospi Q deg6 fit err", mp.nstr(eQ, 3))
Pd = [f64(c) for c in P]; Qd = [f64(c) for c in Q]
print("P =", [hexd(c) for c in Pd]); print("P =", Pd)
print("Q =", [hexd(c) for c in Qd]); print("Q =", Qd)
PI = f64(mp.pi)

def sinpi_core(t):
    s = t*t
    r = Pd[5]
    for c in Pd[4::-1]: r = fma(r, s, c)
    st = s*t
    r = r*st
    return fma(t, PI, r)
def cospi_core(t):
    s = t*t
    r = Qd[6]
    for c in Qd[5::-1]: r = fma(r, s, c)
    return fma(r, s, 1.0)
import math
random.seed(1)
ms = mc = us = uc = 0
for i in range(60000):
    t = random.uniform(-0.25, 0.25)
    if i % 5 == 0: t = math.copysign(0.25 - random.random()*1e-3, t)     # the interval's ends, where cos(pi t) sits just above a binade boundary
    a = sinpi_core(t); b = cospi_core(t)
    ra = mp.sin(mp.pi*mp.mpf(t)); rb = mp.cos(mp.pi*mp.mpf(t))
    ulp_a = abs(mp.mpf(a)-ra)/abs(ra)*2**53 if ra != 0 else 0
    ulp_b = abs(mp.mpf(b)-rb)/abs(rb)*2**53
    ms = max(ms, ulp_a); mc = max(mc, ulp_b)
    if ra != 0: us = max(us, abs(mp.mpf(a)-ra)/mp.mpf(math.ulp(float(ra))))
    uc = max(uc, abs(mp.mpf(b)-rb)/mp.mpf(math.ulp(float(rb))))
print("sinpi max rel err (2^-53 units)", mp.nstr(ms, 4), " cospi", mp.nstr(mc, 4))
print("sinpi max err in ulps of the result", mp.nstr(us, 4), " cospi", mp.nstr(uc, 4), " (cospi's worst sits at |t| -> 1/4: 1 + s Q(s) with s Q = -0.29)")

# ---- exp(r) = 1 + r + r^2*E(r), |r| <= ln2/2
def Ef(r):
    return (mp.exp(r) - 1 - r)/(r*r) if r != 0 else mp.mpf(1)/2
h = mp.log(2)/2 * mp.mpf('1.0001')
E, eE = remez_like(Ef, -h, h, 10)
print("exp E deg10 fit err", mp.nstr(eE, 3))
Ed = [f64(c) for c in E]
print("E =", [hexd(c) for c in Ed]); print("E =", Ed)
LN2_HI = float.fromhex('0x1.62e42fefa39efp-1'); 
LN2_LO = f64(mp.log(2) - mp.mpf(LN2_HI))
# split so that k*LN2_HI is exact for |k| < 2^11: use hi with low 11 bits zero
hi_bits = int(mp.floor(mp.log(2) * 2**42))
LN2_HI = float(mp.mpf(hi_bits) / 2**42); LN2_LO = f64(mp.log(2) - mp.mpf(LN2_HI))
print("LN2_HI", hexd(LN2_HI), LN2_HI, "LN2_LO", hexd(LN2_LO), LN2_LO)
LOG2E = f64(1/mp.log(2))
def exp_fast(x):
    k = float(np.rint(x*LOG2E))
    r = fma(-k, LN2_HI, x); r = fma(-k, LN2_LO, r)
    p = Ed[10]
    for c in Ed[9::-1]: p = fma(p, r, c)
    r2 = r*r
    p = fma(p, r2, r)
    p = p + 1.0
    return float(mp.ldexp(mp.mpf(p), int(k)))
me = 0
for i in range(20000):
    x = -random.uniform(0, 700) if i % 2 else -random.uniform(0, 3)
    a = exp_fast(x); ra = mp.exp(mp.mpf(x))
    me = max(me, abs(mp.mpf(a)-ra)/ra*2**53)
print("exp max rel err (2^-53 units)", mp.nstr(me, 4))

# ---- log(u), u in [2^-53, 1]: fdlibm e_log.c form
Lg = [6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01]
ln2_hi = 6.93147180369123816490e-01; ln2_lo = 1.90821492927058770002e-10
import math
def log_fast(x):
    m, e = math.frexp(x)          # m in [0.5, 1)
    if m < 0.7071067811865476: m *= 2.0; e -= 1
    f = m - 1.0
    k = float(e)
    s = f/(2.0+f)
    z = s*s; w = z*z
    t1 = w*fma(w, fma(w, Lg[5], Lg[3]), Lg[1])
    t2 = z*fma(w, fma(w, fma(w, Lg[6], Lg[4]), Lg[2]), Lg[0])
    R = t2 + t1
    hfsq = 0.5*f*f
    return k*ln2_hi - ((hfsq - (s*(hfsq+R) + k*ln2_lo)) - f)
ml = 0
for i in range(30000):
    v1 = random.getrandbits(53)
    u = 2.0**-53 + v1*2.0**-53
    if i % 3 == 0: u = 2.0**-53 * (1 + random.getrandbits(random.randint(1, 30)))
    if u > 1: continue
    a = log_fast(u); ra = mp.log(mp.mpf(u))
    if ra != 0: ml = max(ml, abs(mp.mpf(a)-ra)/abs(ra)*2**53)
print("log max rel err (2^-53 units)", mp.nstr(ml, 4))
