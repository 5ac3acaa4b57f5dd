"""Is v_mfma_f64_16x16x4_f64 a k-ordered fp64 fma chain?  (developer probe; python tools/probe/run_mfma64.py on the GPU box)"""
import ctypes, os, math, numpy as np
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma64_probe.so"))
rng = np.random.default_rng(0)
P = ctypes.c_void_p


def fma(a, b, c):  # exact for fp32-valued a, b: the product is exact in fp64, one rounding in the sum
    return float(np.float64(a) * np.float64(b) + np.float64(c)) if False else math.fma(a, b, c) if hasattr(math, "fma") else None


for K in (4, 8, 64):
    for trial in range(4):
        A = (rng.standard_normal((16, K)) * 10.0 ** rng.integers(-6, 7, (16, K))).astype(np.float32).astype(np.float64)
        B = (rng.standard_normal((K, 16)) * 10.0 ** rng.integers(-6, 7, (K, 16))).astype(np.float32).astype(np.float64)
        C = rng.standard_normal((16, 16)) * 10.0 ** rng.integers(-6, 7, (16, 16)) if trial else np.zeros((16, 16))
        D = np.zeros((16, 16))
        rc = lib.probe64_run(P(A.ctypes.data), P(B.ctypes.data), P(C.ctypes.data), P(D.ctypes.data), K)
        chain = C.copy()
        for k in range(K):  # products exact in fp64 -> numpy's multiply-then-add has a single rounding, like an fma
            chain = chain + A[:, k:k + 1] * B[k:k + 1, :]
        quad = C.copy()     # hypothesis 2: four products summed pairwise first, then added
        for k0 in range(0, K, 4):
            p = [A[:, k:k + 1] * B[k:k + 1, :] for k in range(k0, k0 + 4)]
            quad = quad + ((p[0] + p[1]) + (p[2] + p[3]))
        print(f"K={K} trial {trial}: rc={rc} mismatches vs k-ordered chain {int((D != chain).sum())}/256, vs pairwise-quad {int((D != quad).sum())}/256, "
              f"max rel diff to chain {np.max(np.abs(D - chain) / np.maximum(np.abs(chain), 1e-300)):.2e}")
rates = np.zeros(2)
lib.probe64_rate(P(rates.ctypes.data))
print("s_memtime ticks (100 MHz) per f64 16x16x4 MFMA: dependent %.2f, 4-way independent %.2f" % tuple(rates))
