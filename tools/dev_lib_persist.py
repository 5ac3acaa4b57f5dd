"""Diagnostic: step time and the persistent kernel's time (k_bcd_p, HIP events) for one rank triple with a given build of the
library (argv[1] = file under lrf_amd/, argv[2] = ranks, argv[3] = images) — the A/B timer of ablation / variant builds of the
block bodies (`make -C lrf_amd/csrc variant VARIANT=... UNIT=lrf_bcd_persist DEFS=-DLRF_W16_NO_GS`: wrong factors, timing only)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lrf_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
import torch, lrf_amd
ranks = tuple(int(v) for v in sys.argv[2].split(","))
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 256
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (NB, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, ranks)
torch.cuda.synchronize()
ctx.profile_kernels([_lib.LRF_K_BCD, _lib.LRF_K_BCD_PERSIST]); ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(8): lrf_amd.qmf_factorize_batch(imgs, ranks)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 8
pm, pn = ctx.kernel_time(_lib.LRF_K_BCD_PERSIST)
bm, bn = ctx.kernel_time(_lib.LRF_K_BCD)
ctx.profile(False)
print(f"{sys.argv[1]} ranks {ranks} x {NB}: step {dt*1e3:.3f} ms, k_bcd_p {pm/max(pn,1):.4f} ms ({pm/max(pn,1)/9*1e3:.1f} us per iteration), first iteration {bm/max(bn,1):.4f} ms")
