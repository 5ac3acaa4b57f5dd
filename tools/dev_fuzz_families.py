"""Large calls with mixed rank families (the per-family launch plan, the family streams, k_bcd_w16 / k_bcd_mid / k_bcd_w side by
side) against the same call with every plane on one family and one stream: hashes of the int8 factors, random rank triples.
python tools/dev_fuzz_families.py [cases] [seed]   (parent: runs itself twice as a child, with and without the switches)"""
import hashlib, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, lrf_amd
    cases, seed = int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    B, H, W = 136, 512, 768
    base = torch.rand(B, 3, H // 8, W // 8, generator=g) * 255
    imgs = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
            + torch.randn(B, 3, H, W, generator=g) * 4).clamp(0, 255).to(torch.uint8).cuda()
    for case in range(cases):
        top = [8, 16, 32][int(rng.integers(0, 3))]
        ranks = (int(rng.integers(max(1, top // 2), top + 1)), int(rng.integers(1, top // 2 + 1)), int(rng.integers(1, top // 2 + 1)))
        K = int(rng.integers(1, 4))
        bounds = [(-16, 15), (-8, 7), (-128, 127)][int(rng.integers(0, 3))]
        U, V = lrf_amd.qmf_factorize_batch(imgs, ranks, num_iters=K, bounds=bounds)
        h = hashlib.sha256(U.cpu().numpy().tobytes() + V.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"{ranks} K={K} {bounds} {h}", flush=True)
    sys.exit(0)
cases = sys.argv[1] if len(sys.argv) > 1 else "12"
seed = sys.argv[2] if len(sys.argv) > 2 else "1"
outs = []
for extra in ({}, {"LRF_NO_FAMILY_STREAMS": "1"}, {"LRF_NO_FAMILY_SPLIT": "1"}):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", cases, seed], env=dict(os.environ, **extra), capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("(")]
    if r.returncode != 0 or not lines:
        print("child failed:", r.stderr[-1500:])
        sys.exit(2)
    outs.append(lines)
bad = 0
for a, b, c in zip(*outs):
    same = a == b == c
    bad += not same
    print(("ok   " if same else "DIFF ") + a + ("" if same else f" | no streams: {b.split()[-1]} | one family: {c.split()[-1]}"))
print(f"{len(outs[0])} cases, {bad} differing")
sys.exit(1 if bad else 0)
