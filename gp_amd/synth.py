"""Deterministic synthetic SE-ARD data for the benchmark/parity configs.

PRNG: splitmix64 used counter-based (state_i = seed + (i+1)*GOLDEN), u = (z >> 11) * 2^-53, so the
same stream can be restated in C++ and is vectorisable here.  Inputs are unit-variance uniform
(same distribution as the reference's data/train_x, range about +-1.73); targets are a
random-Fourier-feature draw from an SE-ARD GP plus noise, so no O(n^3) sampling is needed.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_uniform(seed, start, count):
    """u_i in [0,1) for stream positions start .. start+count-1."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _normal(seed, start, count):
    """Box-Muller on the same stream (uses 2*count uniforms from position start)."""
    u = splitmix64_uniform(seed, start, 2 * count)
    u1 = np.maximum(u[0::2], 1e-300)
    u2 = u[1::2]
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def eval_hyps(d, log_sf=0.0, log_sn=np.log(0.1), mean=0.0):
    """Fixed evaluation hypers: log l_k = 0.5*log d + 0.1*sin(k), log sf = 0, log sn = log 0.1."""
    k = np.arange(d, dtype=np.float64)
    return np.concatenate([0.5 * np.log(d) + 0.1 * np.sin(k), [log_sf, log_sn, mean]])


def make_inputs(seed, n, d, start_point=0):
    """(n, d) array, row = point (C order == the reference's d x n column-major)."""
    u = splitmix64_uniform(seed, start_point * d, n * d)
    return (np.sqrt(3.0) * (2.0 * u - 1.0)).reshape(n, d)


def make_targets(seed, X, hyp, n_features=256, chunk=1 << 18, start_point=0):
    """y_a = mean + sf*sqrt(2/D)*sum_j cos(w_j.x_a + b_j) + sn*N(0,1)."""
    n, d = X.shape
    ell = np.exp(hyp[:d])
    sf, sn, mean = np.exp(hyp[d]), np.exp(hyp[d + 1]), hyp[d + 2]
    fseed = seed ^ 0x5EED
    w = _normal(fseed, 0, n_features * d).reshape(n_features, d) / ell
    b = 2.0 * np.pi * splitmix64_uniform(fseed, 2 * n_features * d, n_features)
    y = np.empty(n)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        y[s:e] = np.cos(X[s:e] @ w.T + b).sum(axis=1)
    y *= sf * np.sqrt(2.0 / n_features)
    y += mean + sn * _normal(seed ^ 0xA11CE, 2 * start_point, n)   # Box-Muller: two uniforms per point
    return y


def make_problem(config_index, n, d, t=0, n_features=256):
    """Returns X (n,d), y (n,), hyp (d+3,), Xs (t,d) for one of SURVEY §8(d)'s synthetic configs."""
    seed = 20240601 + config_index
    hyp = eval_hyps(d)
    X = make_inputs(seed, n, d)
    y = make_targets(seed, X, hyp, n_features)
    Xs = make_inputs(seed, t, d, start_point=n) if t else np.empty((0, d))
    return X, y, hyp, Xs
