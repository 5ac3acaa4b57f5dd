"""Experiment: svd_encode's batch (256 x 512x768, R = 5) cut into 2 / 4 sub-batches on contexts and
streams of their own, so that the one-workgroup-per-matrix latency chain of one sub-batch (k_any_tridiag_reg, k_any_eig) runs
beside the streaming kernels of another.  Prints ms per batch for 1, 2, 3, 4 lanes (results are checked equal)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, lrf_amd
from lrf_amd import _lib
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
R = 5
ctx0 = _lib.context(0)
ref = ctx0.svd_encode_rgb(imgs, R)
for lanes in (1, 2, 3, 4):
    ctxs = [ctx0] + [_lib.Context(0) for _ in range(lanes - 1)]
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    cuts = [round(i * 256 / lanes) for i in range(lanes + 1)]
    parts = [imgs[cuts[i]:cuts[i + 1]].contiguous() for i in range(lanes)]

    def run():
        cur = torch.cuda.current_stream()
        outs = []
        for i in range(lanes):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs.append(ctxs[i].svd_encode_rgb(parts[i], R))
        for i in range(lanes):
            cur.wait_stream(streams[i])
        return outs
    for _ in range(3): outs = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): outs = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    ok = all(torch.equal(torch.cat([o[k] for o in outs]), ref[k]) for k in range(3))
    print(f"{lanes} lane(s): {dt*1e3:.3f} ms per 256 images, equal to the single call: {ok}")
    for c in ctxs[1:]: c.close()
