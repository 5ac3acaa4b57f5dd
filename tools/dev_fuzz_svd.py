"""Fuzz of svd_encode (RGB branch, 8x8 patches, uint8 factors: the byte-matrix path for ranks <= 8, the fp32-matrix path above)
against the oracle run the way the library runs, byte for byte (development aid; the logic of
tests/test_svd_any.py::test_svd_encode_equals_oracle_bytes).  python tools/dev_fuzz_svd.py [cases] [seed]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch, lrf_amd
from lrf_amd.container import combine_bytes, dict_to_bytes, encode_tensor
from oracle import oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    H, W = int(rng.integers(8, 260)), int(rng.integers(8, 330))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        img = torch.from_numpy(rng.integers(0, 256, (3, H, W), dtype=np.uint8))
    elif kind == 1:
        base = torch.from_numpy(rng.random((1, 3, max(H // 8, 1), max(W // 8, 1))).astype(np.float32)) * 255
        img = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)[0]
               + torch.from_numpy(rng.normal(0, 3, (3, H, W)).astype(np.float32))).clamp(0, 255).to(torch.uint8)
    else:
        img = torch.full((3, H, W), int(rng.integers(0, 256)), dtype=torch.uint8)
    rank = int(rng.integers(1, 14))
    enc = lrf_amd.svd_encode(img, rank=rank)
    X = oracle.rgb_matrix_any(img.numpy(), (8, 8))
    M, N = X.shape
    u, v = oracle.svd_topr_u8(X, rank)
    Hp, Wp = H + (8 - H % 8) % 8, W + (8 - W % 8) % 8
    metadata = {"dtype": "uint8", "color space": "RGB", "patch": True, "patch size": (8, 8), "original size": [H, W], "padded size": [Hp, Wp]}
    qu, su, mu = oracle.quantize_u8(u)
    qv, sv, mv = oracle.quantize_u8(v)
    metadata["quantization"] = {"u": [su, mu], "v": [sv, mv]}
    want = combine_bytes([dict_to_bytes(metadata), combine_bytes([encode_tensor(np.ascontiguousarray(f)) for f in (qu, qv)])])
    if enc != want:
        bad += 1
        print(f"MISMATCH case {case}: {H}x{W} rank {rank} kind {kind} (M={M})", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
