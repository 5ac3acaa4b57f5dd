"""Experiment: the three planes of an any-shape encode on three contexts / streams against one after the other (the
initialisation is one workgroup of four waves per matrix: a latency chain that leaves most of a CU idle).
python tools/dev_any_streams.py [B] [quality]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from lrf_amd import _lib
from lrf_amd.codec import anyshape_ranks
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Q = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
H, W = 512, 768
g = torch.Generator(device="cuda").manual_seed(0)
base = torch.rand(B, 3, H // 8, W // 8, device="cuda", generator=g) * 255
imgs = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear") + torch.randn(B, 3, H, W, device="cuda", generator=g) * 4
        ).clamp(0, 255).to(torch.uint8)
ctxs = [_lib.Context(0) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]


def run(ps, ranks, parallel, init_only=False):
    out = [None] * 3
    cur = torch.cuda.current_stream()
    for c in range(3):
        if parallel:
            streams[c].wait_stream(cur)
            with torch.cuda.stream(streams[c]):
                X = ctxs[c].planes_any(imgs, ps, c)
                out[c] = ctxs[c].svd_init(X, ranks[c], None) if init_only else ctxs[c].decompose(X, ranks[c], 10, -16, 15)
        else:
            X = ctxs[0].planes_any(imgs, ps, c)
            out[c] = ctxs[0].svd_init(X, ranks[c], None) if init_only else ctxs[0].decompose(X, ranks[c], 10, -16, 15)
    if parallel:
        for s in streams:
            cur.wait_stream(s)
    return out


for ps in ((4, 4), (16, 16), (32, 32), None):
    ranks = anyshape_ranks((H, W), ps, None, Q)
    for init_only in (True, False):
        res = {}
        for parallel in (False, True):
            ref = run(ps, ranks, parallel, init_only)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                out = run(ps, ranks, parallel, init_only)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res[parallel] = (min(ts) * 1e3, out)
        same = all(torch.equal(a, b) for pa, pb in zip(res[False][1], res[True][1]) for a, b in zip(pa, pb))
        print(f"patch {ps} ranks {ranks} {'init only' if init_only else 'encode'}: one stream {res[False][0]:.1f} ms, three streams {res[True][0]:.1f} ms, "
              f"equal {same}", flush=True)
