import ctypes, os, numpy as np
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtree_probe.so"))
rng = np.random.default_rng(0); n = 4096
x = (rng.standard_normal((n, 64)) * 10.0 ** rng.integers(-8, 8, (n, 1))).astype(np.float64)
out = np.zeros((n, 2)); assert lib.tree_run(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), n) == 0
print("DPP/permlane tree == shuffle tree (bitwise):", bool((out[:, 0].view(np.int64) == out[:, 1].view(np.int64)).all()), " max |diff|", np.abs(out[:, 0] - out[:, 1]).max())
