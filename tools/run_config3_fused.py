"""BASELINE config 3 as ONE job (round 5): the 24 stand-in images (512x768) at qualities 1..32 — GPU time (device tensor in, int8
factors in HBM out; HIP events around the calls) of the fused sweep call lrf_qmf_encode_sweep_rgb_u8 against the sum of the per-quality
calls lrf_qmf_encode_rgb_u8 (what rd_sweep_batched did until round 4), and end to end (host tensor in, byte streams out).
usage: python tools/run_config3_fused.py [out.json]"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import lrf_amd
from lrf_amd import _lib
from lrf_amd.codec import qmf_ranks
from conftest import config3_image

imgs = torch.stack([config3_image(i) for i in range(24)])
dev = imgs.cuda()
H, W = imgs.shape[-2:]
qualities = list(range(1, 33))
triples = [tuple(qmf_ranks((H, W), None, q)) for q in qualities]
unique = sorted(set(triples))
ctx = _lib.context(0)


def gpu_ms(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


per_quality = {t: gpu_ms(lambda t=t: ctx.encode_rgb(dev, list(t), 10, -16, 15)) for t in unique}
sum_32 = sum(per_quality[t] for t in triples)          # one call per quality, duplicates included (what the sweep ran)
sum_unique = sum(per_quality.values())                 # ... if identical triples had been computed once
fused = gpu_ms(lambda: ctx.encode_sweep_rgb(dev, unique, 10, -16, 15))
ctx.synchronize()
t0 = time.perf_counter(); streams = lrf_amd.qmf_encode_sweep(imgs, qualities=qualities); e2e_fused = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
ref = {}
for q, t in zip(qualities, triples):
    if t not in ref: ref[t] = lrf_amd.qmf_encode_batch(imgs, quality=q)
e2e_unique = (time.perf_counter() - t0) * 1e3
assert all(s == ref[t] for s, t in zip(streams, triples)), "fused streams differ"
out = {"workload": "24 x 512x768 (BASELINE config 3 stand-ins), qualities 1..32, K = 10, bounds (-16,15)",
       "rank_triples": len(unique), "gpu_ms_sum_of_32_per_quality_calls": round(sum_32, 3),
       "gpu_ms_sum_of_unique_triples": round(sum_unique, 3), "gpu_ms_fused_sweep_call": round(fused, 3),
       "fused_over_sum_of_32": round(fused / sum_32, 4), "fused_over_sum_of_unique": round(fused / sum_unique, 4),
       "gpu_ms_per_triple": {str(t): round(v, 3) for t, v in per_quality.items()},
       "end_to_end_ms_fused_host_in_streams_out": round(e2e_fused, 1), "end_to_end_ms_per_unique_triple_calls": round(e2e_unique, 1),
       "streams_byte_identical": True,
       "pixels_per_s_fused_gpu": round(24 * len(qualities) * H * W / (fused * 1e-3) / 1e9, 2)}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
