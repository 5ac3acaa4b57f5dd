"""End to end time of the any-shape batch encode (host tensor in, byte streams out) against its GPU part (development aid)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(0)
base = torch.rand(B, 3, 64, 96, generator=g) * 255
imgs = (torch.nn.functional.interpolate(base, size=(512, 768), mode="bilinear") + torch.randn(B, 3, 512, 768, generator=g) * 4).clamp(0, 255).to(torch.uint8)
for kw in ({"patch_size": (16, 16)}, {"patch": False}, {"patch_size": (4, 4)}):
    lrf_amd.qmf_encode_batch(imgs[:2], quality=20, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    streams = lrf_amd.qmf_encode_batch(imgs, quality=20, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{kw}: {B} images in {dt*1e3:.1f} ms end to end ({dt*1e3/B:.2f} ms per image), {sum(len(s) for s in streams)/B:.0f} bytes per image", flush=True)
