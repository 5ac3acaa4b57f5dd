"""Diagnostic: tools/dev_rank_sweep.py's timing for one rank triple with a given build of the library (argv[1] = file under
lrf_amd/, argv[2] = ranks, argv[3] = images)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lrf_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
import torch, lrf_amd
ranks = tuple(int(v) for v in sys.argv[2].split(","))
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 256
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (NB, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
import hashlib
U, V = lrf_amd.qmf_factorize_batch(imgs, ranks)
torch.cuda.synchronize()
digest = hashlib.sha256(U.cpu().numpy().tobytes() + V.cpu().numpy().tobytes()).hexdigest()[:12]
del U, V
for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, ranks)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(4): lrf_amd.qmf_factorize_batch(imgs, ranks)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 4)
print(f"{sys.argv[1]} ranks {ranks} x {NB}: {min(ts)*1e3:.3f} ms (min of 5), median {sorted(ts)[2]*1e3:.3f}  factors {digest}")
