"""End-to-end decode (byte streams on the host -> uint8 images on the host): one stream per call and a 256-stream batch,
with liblrf_pack.so's unpacker and with the Python container code (LRF_NO_NATIVE_UNPACK=1 in the environment of this tool)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
from lrf_amd import codec
if os.environ.get("LRF_NO_NATIVE_UNPACK") == "1":
    codec._factors_native = lambda streams: None
g = torch.Generator().manual_seed(0)
base = torch.rand(256, 3, 64, 96, generator=g) * 255
imgs = (torch.nn.functional.interpolate(base, size=(512, 768), mode="bilinear") + torch.randn(256, 3, 512, 768, generator=g) * 4).clamp(0, 255).to(torch.uint8)
streams = lrf_amd.qmf_encode_batch(imgs, quality=7)
for _ in range(3): lrf_amd.qmf_decode(streams[0])
t0 = time.perf_counter()
for i in range(200): lrf_amd.qmf_decode(streams[i % 256])
print(f"one stream per call: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
for _ in range(2): out = codec.qmf_decode_batch(streams).cpu()
t0 = time.perf_counter()
for _ in range(5): out = codec.qmf_decode_batch(streams).cpu()
dt = (time.perf_counter() - t0) / 5
print(f"256 streams per call: {dt * 1e3:.2f} ms ({256 * 512 * 768 / dt / 1e9:.2f} Gpix/s, device -> host copy of the images included)")
