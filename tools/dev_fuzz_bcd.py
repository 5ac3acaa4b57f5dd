"""Randomised HIP-vs-oracle parity sweep of the BCD from given initial factors (bit for bit): the 64-column path
(lrf_qmf_bcd_f32) and the RGB colour-space path (lrf_qmf_rgbspace_encode_u8 with init)."""
import os, sys, random
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch, lrf_amd
from lrf_amd import _lib
from oracle import oracle
oracle.build()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = _lib.context(0)
bad = 0
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    H = rnd.choice([16, 24, 64, 99, 173, 256]); W = rnd.choice([16, 56, 96, 130, 264, 384])
    K = rnd.choice([1, 2, 5, 10]); bounds = rnd.choice([(-16, 15), (-8, 7), (-128, 127)])
    g = torch.Generator().manual_seed(500 + i)
    base = torch.rand(1, 3, max(H // 8, 1), max(W // 8, 1), generator=g) * 255
    img = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)[0]
           + torch.randn(3, H, W, generator=g) * 6).clamp(0, 255).to(torch.uint8)
    # --- RGB colour-space path
    R = rnd.randint(1, 24)
    X = oracle.pad_patchify(img.numpy().astype(np.float32))
    u0, v0 = oracle.svd_topr(X, R)
    scale = rnd.choice([1.0, 0.5, 2.0])
    u0, v0 = (u0 * scale).astype(np.float32), (v0 / scale).astype(np.float32)
    uo, vo = oracle.bcd(X, u0, v0, K, bounds)
    U, V = ctx.qmf_rgbspace_encode(img.cuda().unsqueeze(0), R, K, bounds, init=(torch.from_numpy(u0)[None], torch.from_numpy(v0)[None]))
    ok1 = np.array_equal(U[0].cpu().numpy(), uo.astype(np.int8)) and np.array_equal(V[0].cpu().numpy(), vo.astype(np.int8))
    # --- 64-column path, one plane
    R2 = rnd.randint(1, 16)
    Xp = oracle.rgb_to_planes(img.numpy())[rnd.randint(0, 2)]
    a0, b0 = oracle.svd_init(Xp, R2)
    ao, bo = oracle.bcd(Xp, a0, b0, K, bounds)
    U2, V2 = ctx.bcd(torch.from_numpy(Xp).cuda()[None], torch.from_numpy(a0).cuda()[None], torch.from_numpy(b0).cuda()[None], K, bounds[0], bounds[1])
    ok2 = np.array_equal(U2[0].cpu().numpy(), ao.astype(np.int8)) and np.array_equal(V2[0].cpu().numpy(), bo.astype(np.int8))
    print(f"[{i}] {H}x{W} K={K} bounds={bounds}: rgbspace R={R} {'ok' if ok1 else 'FAIL'}; 64-col R={R2} {'ok' if ok2 else 'FAIL'}")
    bad += (not ok1) + (not ok2)
sys.exit(1 if bad else 0)
