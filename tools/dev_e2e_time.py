import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, lrf_amd
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
lrf_amd.qmf_encode_batch(imgs[:8], rank=7)
for w in (1, 16, 64):
    torch.cuda.synchronize(); t = time.perf_counter()
    s = lrf_amd.qmf_encode_batch(imgs, rank=7, pack_workers=w)
    dt = time.perf_counter() - t
    print(f"qmf_encode_batch 256 images, pack_workers={w}: {dt*1e3:.0f} ms end to end ({256*512*768/dt/1e6:.0f} Mpix/s), {sum(map(len,s))/256:.0f} B/image")
