import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch, lrf_amd
from lrf_amd import codec
g = torch.Generator().manual_seed(0)
base = torch.rand(256, 3, 64, 96, generator=g) * 255
imgs = (torch.nn.functional.interpolate(base, size=(512, 768), mode="bilinear") + torch.randn(256, 3, 512, 768, generator=g) * 4).clamp(0, 255).to(torch.uint8)
streams = lrf_amd.qmf_encode_batch(imgs, quality=7)
for _ in range(2): out = codec.qmf_decode_batch(streams).cpu()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): out = codec.qmf_decode_batch(streams).cpu()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
