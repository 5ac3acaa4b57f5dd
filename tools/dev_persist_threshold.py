"""ms per batch of B x 512x768 images at ranks (7,3,3) for B = 48..192 (1152..4608 blocks): run with LRF_PERSIST=0 and =1 (k_bcd_p forced\nfrom 1024 blocks) to see where the persistent kernel starts to pay (the default threshold is LRF_PERSIST_MIN_BLOCKS)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, lrf_amd
g = torch.Generator(device="cuda").manual_seed(0)
out = []
for B in (48, 64, 96, 128, 160, 192):
    imgs = torch.randint(0, 256, (B, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
    U, V = lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        for _ in range(10): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3), out=(U, V))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10)
    out.append(f"{B}: {min(ts)*1e3:.3f}")
print("persist=" + os.environ.get("LRF_PERSIST", "default"), " | ".join(out), flush=True)
