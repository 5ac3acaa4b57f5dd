"""The persistent iteration kernel k_bcd_p against the launch-per-iteration path (LRF_PERSIST=0): factors' hashes (after one and
after 26 runs) and ms per batch for a few rank triples / batch sizes.  Run once with LRF_PERSIST=0 and once with LRF_PERSIST=1
(forced from 1024 blocks on) or unset (the default: from 3584 blocks; 2304 for one rank family) and compare the lines; LRF_SOAK=n repeats the first case n
times and checks every result against the first."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lrf_amd import _lib as _l0
if os.environ.get("LRF_LIB"): _l0.LIB_PATH = os.path.join(os.path.dirname(_l0.LIB_PATH), os.environ["LRF_LIB"])
import torch, lrf_amd
from lrf_amd import _lib
g = torch.Generator(device="cuda").manual_seed(0)
ctx = _lib.context(0)
for B, (H, W), ranks in ((256, (512, 768), (7, 3, 3)), (256, (512, 768), (4, 2, 2)), (128, (512, 768), (8, 8, 8)), (300, (173, 264), (7, 3, 1)), (64, (1365, 2048), (7, 3, 3))):
    imgs = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda", generator=g)
    U, V = lrf_amd.qmf_factorize_batch(imgs, ranks)
    torch.cuda.synchronize()
    ctx.synchronize()
    h = hashlib.sha256(U.cpu().numpy().tobytes() + V.cpu().numpy().tobytes()).hexdigest()[:16]
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(5): lrf_amd.qmf_factorize_batch(imgs, ranks, out=(U, V))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5)
    ctx.synchronize()
    h2 = hashlib.sha256(U.cpu().numpy().tobytes() + V.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"persist={os.environ.get('LRF_PERSIST', 'default')} {B} x {H}x{W} ranks {ranks}: sha {h} (after 26 runs {h2}) {min(ts)*1e3:.3f} ms min, {sorted(ts)[2]*1e3:.3f} median", flush=True)

n_soak = int(os.environ.get("LRF_SOAK", "0"))
if n_soak:
    imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
    U0, V0 = lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
    other = torch.empty((64, 1024, 1024), device="cuda")
    bad = 0
    for i in range(n_soak):
        if i % 3 == 1:
            other.normal_()  # uneven load next to the launch
        U, V = lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
        if i % 3 == 2:
            other.mul_(1.0001)
        bad += int(not (torch.equal(U, U0) and torch.equal(V, V0)))
    ctx.synchronize()
    print(f"soak: {n_soak} runs, {bad} differ from the first")
