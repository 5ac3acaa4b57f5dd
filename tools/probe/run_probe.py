import ctypes, os, numpy as np
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma_probe.so"))
d = np.zeros((64, 4), np.float32); rates = np.zeros(4, np.float64)
lib.probe_run(d.ctypes.data_as(ctypes.c_void_p), rates.ctypes.data_as(ctypes.c_void_p))
print("cycles per MFMA (s_memtime ticks): 4x4x1 dependent %.1f, 4x4x1 4-way independent %.1f, 16x16x4 dependent %.1f, 16x16x4 4-way independent %.1f" % tuple(rates))
# decode D[lane][reg] = A[x] * B[y]
A = {float(l + 1): l for l in range(64)}; ok = True
for lane in (0, 1, 2, 3, 4, 5, 17, 63):
    row = []
    for reg in range(4):
        v = float(d[lane, reg]); found = None
        for x in range(64):
            for y in range(64):
                if (x + 1) * (67 + y) == v: found = (x, y)
        row.append(found)
    print("lane", lane, "regs -> (A lane, B lane):", row)
