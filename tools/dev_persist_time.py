"""ms per batch of qmf_factorize_batch at the headline workload and at 64 / 256 CLIC-sized images, for the library named by LRF_LIB
(a file name inside lrf_amd/; default liblrf_hip.so): the A/B timer for variants of the persistent kernel.  Run the variants
back to back on one box and alternate them: box-to-box and run-to-run differences are 2-3 %."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lrf_amd import _lib as _l0
if os.environ.get("LRF_LIB"): _l0.LIB_PATH = os.path.join(os.path.dirname(_l0.LIB_PATH), os.environ["LRF_LIB"])
import torch, lrf_amd
g = torch.Generator(device="cuda").manual_seed(0)
out = []
for B, (H, W), ranks in ((256, (512, 768), (7, 3, 3)), (64, (1365, 2048), (7, 3, 3)), (256, (1365, 2048), (7, 3, 3))):
    imgs = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda", generator=g)
    U, V = lrf_amd.qmf_factorize_batch(imgs, ranks)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        for _ in range(10): lrf_amd.qmf_factorize_batch(imgs, ranks, out=(U, V))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10)
    out.append(f"{B}x{H}x{W}: {min(ts)*1e3:.3f} min {sorted(ts)[4]*1e3:.3f} med")
print(os.environ.get("LRF_LIB", "liblrf_hip.so"), "persist=" + os.environ.get("LRF_PERSIST", "default"), " | ".join(out), flush=True)
