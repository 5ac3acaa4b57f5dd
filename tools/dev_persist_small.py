"""ms per batch for calls of MANY SMALL matrices (ranks (7,3,3)): run with LRF_PERSIST=0 and unset to see whether the persistent
kernel pays there (it is chosen by block count AND blocks per matrix: lrf_api.hip)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, lrf_amd
g = torch.Generator(device="cuda").manual_seed(0)
out = []
for B, (H, W) in ((1200, (173, 264)), (2400, (173, 264)), (1024, (256, 256)), (600, (352, 288))):
    imgs = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda", generator=g)
    U, V = lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(6): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3), out=(U, V))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 6)
    out.append(f"{B}x{H}x{W}: {min(ts)*1e3:.3f}")
print("persist=" + os.environ.get("LRF_PERSIST", "default"), " | ".join(out), flush=True)
