import ctypes, os, numpy as np
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libread_probe.so"))
cfg = [(d, l) for l in (1, 10, 18, 36) for d in (4, 8, 16, 32)]  # LDS per wave -> waves per CU: 1 KB: 32 (register/slot bound), 10 KB: 16, 18 KB: 8, 36 KB: 4
depth = (ctypes.c_int * len(cfg))(*[c[0] for c in cfg]); lds = (ctypes.c_int * len(cfg))(*[c[1] for c in cfg])
ms = (ctypes.c_double * len(cfg))()
assert lib.probe_read(ms, depth, lds, len(cfg)) == 0
for (d, l), t in zip(cfg, ms):
    print(f"LDS {l:2d} KB per wave (~{min(32, 160 // l)} waves/CU), {d:2d} KB in flight per wave: {t:.4f} ms = {617.349 / t / 1e3:.2f} TB/s")
