"""Measurement for VERDICT r03 item 4 (CPU, oracle only): at which BCD iteration does a plane reach its fixed point
(an iteration that leaves its int8 U and V unchanged makes every later one the identity)?  24 config-3 images x qualities
{1,7,16,25,32} and uniform-noise images at rank 7 (the bench workload).  Writes profiles/r04_fixed_point.json."""
import json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import oracle
from conftest import config3_image
import lrf_amd.codec as codec  # qmf_ranks only (no GPU call)

K = 10
def planes_fixed_points(img_u8, ranks):
    X = oracle.rgb_to_planes(img_u8)
    out = []
    for c in range(3):
        u, v = oracle.svd_init(X[c], ranks[c])
        fixed, changed_rows = None, []
        for k in range(1, K + 1):
            un, vn = oracle.bcd(X[c], u, v, 1)
            if k > 1:
                du = int((un != u).any(axis=1).sum()); dv = int((vn != v).sum())
                changed_rows.append([du, dv])
                if du == 0 and dv == 0 and fixed is None:
                    fixed = k  # iteration k changed nothing: iterations k.. are the identity
            u, v = un, vn
        out.append({"plane": c, "R": int(ranks[c]), "M": int(X[c].shape[0]), "fixed_at": fixed, "changed_rows_and_v_entries": changed_rows})
    return out

def ranks_of(q=None, rank=None):
    dims = oracle.plane_dims(512, 768)
    if rank is not None:
        return (rank, max(rank // 2, 1), max(rank // 2, 1))
    qs = (q, q / 2, q / 2)
    return tuple(max(round(min(d[4], 64) * qq / 100), 1) for d, qq in zip(dims, qs))

res = {"K": K, "sets": {}}
cases = [("config3 q=%d" % q, [config3_image(i).numpy() for i in range(24)], ranks_of(q=q)) for q in (1, 7, 16, 25, 32)]
g = torch.Generator().manual_seed(1234)
noise = [torch.randint(0, 256, (3, 512, 768), dtype=torch.uint8, generator=g).numpy() for _ in range(4)]
cases.append(("uniform noise rank=7 (bench workload)", noise, ranks_of(rank=7)))
for name, imgs, ranks in cases:
    tot = skip = 0; fixed_hist = {}; rows_later = []
    for im in imgs:
        for p in planes_fixed_points(im, ranks):
            f = p["fixed_at"]
            fixed_hist[str(f)] = fixed_hist.get(str(f), 0) + 1
            w = p["M"]  # weight planes by their rows (luma 4x a chroma plane)
            tot += w * (K - 1)  # iterations 2..K are the ones an exit could skip
            if f is not None: skip += w * (K - f)  # iterations after the first unchanged one
            rows_later.append([c[0] / p["M"] for c in p["changed_rows_and_v_entries"]])
    mean_changed = np.mean(np.array(rows_later), axis=0).round(4).tolist()
    res["sets"][name] = {"ranks": list(ranks), "planes": sum(fixed_hist.values()), "fixed_at_histogram": fixed_hist,
                         "row_weighted_skippable_fraction_of_iterations_2_to_K": round(skip / tot, 4),
                         "mean_fraction_of_rows_changed_by_iteration_2_to_K": mean_changed}
    print(name, res["sets"][name], flush=True)
json.dump(res, open(os.path.join(ROOT, "profiles", "r04_fixed_point.json"), "w"), indent=1)
