"""Exact posteriors used as ground truth (test infrastructure only).

  * conjugate Gaussian: README.md:118 (2.32353, 1.05882); thesis p.85 N(7.25, 5/6)
  * HMM forward-backward for include/models/models.hpp:114-141
  * Kalman filter + RTS smoother for include/models/models.hpp:67-80
  * synthetic observation generators: simulate from the model itself.
"""
import numpy as np

HMM_MEAN = np.array([-1.0, 0.0, 1.0])
HMM_T = np.array([[0.1, 0.5, 0.4], [0.2, 0.2, 0.6], [0.15, 0.15, 0.7]])


def gaussian_posterior(mu0, sigma0, sigma, ys):
    prec = 1.0 / sigma0 ** 2 + len(ys) / sigma ** 2
    mean = (mu0 / sigma0 ** 2 + np.sum(ys) / sigma ** 2) / prec
    return mean, 1.0 / prec


def gaussian_log_evidence(mu0, sigma0, sigma, ys):
    """log p(y1..yn) for the conjugate model (marginal likelihood)."""
    ys = np.asarray(ys, float)
    n = len(ys)
    cov = sigma ** 2 * np.eye(n) + sigma0 ** 2 * np.ones((n, n))
    d = ys - mu0
    sign, logdet = np.linalg.slogdet(cov)
    return -0.5 * (d @ np.linalg.solve(cov, d) + logdet + n * np.log(2 * np.pi))


def normal_logpdf(x, mean, sigma):
    return -0.5 * (((x - mean) / sigma) ** 2 + np.log(2 * np.pi * sigma * sigma))


def hmm_forward_backward(obs):
    """Returns (smoothing[T,3], filtering[T,3], log_evidence)."""
    obs = np.asarray(obs, float)
    T = len(obs)
    lik = np.exp(normal_logpdf(obs[:, None], HMM_MEAN[None, :], 1.0))
    alpha = np.zeros((T, 3)); c = np.zeros(T)
    a = np.full(3, 1.0 / 3.0) * lik[0]
    c[0] = a.sum(); alpha[0] = a / c[0]
    for t in range(1, T):
        a = (alpha[t - 1] @ HMM_T) * lik[t]
        c[t] = a.sum(); alpha[t] = a / c[t]
    beta = np.ones((T, 3))
    for t in range(T - 2, -1, -1):
        beta[t] = (HMM_T @ (lik[t + 1] * beta[t + 1])) / c[t + 1]
    gamma = alpha * beta
    gamma /= gamma.sum(axis=1, keepdims=True)
    return gamma, alpha, np.log(c).sum()


def kalman_rts(obs):
    """x0=0, x_t~N(x_{t-1},1), y_t~N(x_t,1). Returns (smoothed mean[T], var[T], filt mean, filt var, log_evidence)."""
    obs = np.asarray(obs, float)
    T = len(obs)
    mf = np.zeros(T); pf = np.zeros(T); mp = np.zeros(T); pp = np.zeros(T)
    m, p, ll = 0.0, 0.0, 0.0
    for t in range(T):
        mp[t], pp[t] = m, p + 1.0
        s = pp[t] + 1.0
        ll += normal_logpdf(obs[t], mp[t], np.sqrt(s))
        k = pp[t] / s
        m = mp[t] + k * (obs[t] - mp[t]); p = (1 - k) * pp[t]
        mf[t], pf[t] = m, p
    ms = mf.copy(); ps = pf.copy()
    for t in range(T - 2, -1, -1):
        g = pf[t] / pp[t + 1]
        ms[t] = mf[t] + g * (ms[t + 1] - mp[t + 1])
        ps[t] = pf[t] + g * g * (ps[t + 1] - pp[t + 1])
    return ms, ps, mf, pf, ll


def simulate_hmm(T, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    s = rng.integers(0, 3)
    ys = np.zeros(T)
    for t in range(T):
        if t > 0:
            s = rng.choice(3, p=HMM_T[s])
        ys[t] = HMM_MEAN[s] + rng.standard_normal()
    return ys


def simulate_lgssm(T, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    x = 0.0
    ys = np.zeros(T)
    for t in range(T):
        x = x + rng.standard_normal()
        ys[t] = x + rng.standard_normal()
    return ys
