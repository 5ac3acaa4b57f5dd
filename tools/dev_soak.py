"""Soak: the same encode call repeated N times, every result compared on the GPU with the first one (a race in an in-launch
hand-off of k_bcd_p, or in k_planes16_gram's staging, would show as a differing factor sooner or later).
usage: python tools/dev_soak.py <ranks> <images> <repeats>"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, lrf_amd
from lrf_amd import _lib
ranks = tuple(int(v) for v in sys.argv[1].split(","))
NB, N = int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device="cuda").manual_seed(5)
imgs = torch.randint(0, 256, (NB, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
U0, V0 = lrf_amd.qmf_factorize_batch(imgs, ranks)
U0, V0 = U0.clone(), V0.clone()
bad = 0
t0 = time.time()
for i in range(N):
    U, V = lrf_amd.qmf_factorize_batch(imgs, ranks)
    if not (torch.equal(U, U0) and torch.equal(V, V0)):
        bad += 1
        print("repeat", i, "differs:", int((U != U0).sum()), "U entries,", int((V != V0).sum()), "V entries", flush=True)
    if i % 200 == 199:
        print(f"{i + 1} repeats, {bad} differing, {time.time() - t0:.0f} s", flush=True)
_lib.context(0).synchronize()
print(f"soak ranks {ranks} x {NB}: {N} repeats, {bad} differing")
sys.exit(1 if bad else 0)
