"""Stage times of k_any_eig (the any-shape eigen-solver behind svd_encode and the RGB colour-space / no-patch branches) by
stopping it after stage n (LRF_DEBUG_INIT_SWEEPS, a developer switch of the library): child processes time svd_init on
B x M x N matrices at rank R, the differences are the stages.  python tools/dev_any_eig_stages.py [B M N R]"""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from lrf_amd import _lib as _l0
    if os.environ.get("LRF_LIB"): _l0.LIB_PATH = os.path.join(os.path.dirname(_l0.LIB_PATH), os.environ["LRF_LIB"])
    from lrf_amd import _lib
    B, M, N, R = (int(a) for a in sys.argv[2:6])
    X = torch.rand(B, M, N, device="cuda") * 255
    ctx = _lib.context(0)
    for _ in range(3): ctx.svd_init(X, R)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(5): ctx.svd_init(X, R)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5)
    print(min(ts) * 1e3)
    sys.exit(0)
args = sys.argv[1:5] if len(sys.argv) >= 5 else ["256", "6144", "192", "5"]
names = {1: "entry (d, e, tau to LDS)", 2: "+ hull, eigenvalues (multisection)", 3: "+ twisted factorisation", 4: "+ Gram-Schmidt", 0: "+ back-transformation, scaling, output (everything)"}
prev = None
for stop in (1, 2, 3, 4, 0):
    env = dict(os.environ, LRF_DEBUG_INIT_SWEEPS=str(stop))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + args, capture_output=True, text=True, env=env)
    ms = float(r.stdout.strip().splitlines()[-1])
    print(f"stop after {stop}: svd_init {ms:.3f} ms  {names[stop]}" + (f"  (stage: {1e3 * (ms - prev):.0f} us)" if prev is not None else ""), flush=True)
    prev = ms
