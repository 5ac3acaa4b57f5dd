"""BASELINE config 3 on the GPU box: the R-D sweep quality 1..32 over the 24-image stand-in set (tests/conftest.py
config3_image) through lrf_amd.rd_sweep (the protocol of experiments/comparison/eval.py:83-110: one encode call and one
decode call per (image, quality), host tensor in, bytes out, wall-clock ms each).  Writes per-quality means to argv[1].
With "batched" as argv[2]: lrf_amd.rd_sweep_batched instead — the 24 images of a quality in one call (same streams, same
metrics; the times are the batch's divided by 24)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import lrf_amd  # noqa: E402
from conftest import config3_image  # noqa: E402

images = [config3_image(i) for i in range(24)]
qualities = list(range(1, 33))
lrf_amd.qmf_decode(lrf_amd.qmf_encode(images[0], quality=7))  # warm-up: context, workspace
batched = len(sys.argv) > 2 and sys.argv[2] == "batched"
if batched:
    lrf_amd.rd_sweep_batched(torch.stack(images), [7])  # warm-up: the pipelined encoder's slots
t0 = time.perf_counter()
records = lrf_amd.rd_sweep_batched(torch.stack(images), qualities) if batched else \
    lrf_amd.rd_sweep(images, qualities, lrf_amd.qmf_encode, lrf_amd.qmf_decode)
wall = time.perf_counter() - t0
table = []
for q in qualities:
    rs = [r for r in records if int(r["quality"]) == q]
    n = len(rs)
    ranks = lrf_amd.qmf_ranks((512, 768), quality=q)
    table.append({"quality": q, "ranks": ranks, "bpp": sum(r["bit rate (bpp)"] for r in rs) / n, "psnr_db": sum(r["PSNR (dB)"] for r in rs) / n,
                  "ssim_unpinned": sum(r["SSIM"] for r in rs) / n, "encode_ms": sum(r["encoding time (ms)"] for r in rs) / n,
                  "decode_ms": sum(r["decoding time (ms)"] for r in rs) / n})
out = {"what": "24 images 512x768 (20 smooth synthetic + 4 crops of the natural fixture), " +
               ("qmf_encode_batch/qmf_decode_batch, the 24 images of a quality per call, " if batched else "qmf_encode/qmf_decode one image per call, ") +
               "host tensor in, bytes out (zlib container included), means over the images per quality",
       "device": torch.cuda.get_device_name(0), "pairs": len(records), "wall_s_incl_metrics": round(wall, 2), "per_quality": table}
with open(sys.argv[1], "w") as f:
    json.dump(out, f, indent=1)
for t in table:
    print(f"q={t['quality']:2d} ranks={t['ranks']} bpp={t['bpp']:.4f} PSNR={t['psnr_db']:.3f} dB enc={t['encode_ms']:.3f} ms dec={t['decode_ms']:.3f} ms")
