"""Developer aid: intermediate results of the any-shape eigen-solver (GPU: LRF_DEBUG_INIT_SWEEPS, oracle: LRF_ORACLE_ANY_*)
side by side: d / e / lambda (stage 9), the twisted vectors (13xx) and the orthonormalised ones (14xx) from vector xx on."""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    stage, M, N, R, seed = (int(v) for v in sys.argv[2:7])
    os.environ["LRF_DEBUG_INIT_SWEEPS"] = str(stage)
    if stage == 9: os.environ["LRF_ORACLE_ANY_DEBUG"] = "1"
    else: os.environ["LRF_ORACLE_ANY_STAGE"] = str(stage)
    import numpy as np, torch
    from lrf_amd import _lib
    from oracle import oracle
    oracle.build()
    ctx = _lib.context(0)
    rng = np.random.default_rng(abs(seed))
    k = max(1, min(M, N) // 3)
    X = (rng.integers(0, 16, (M, k)) @ rng.integers(0, 16, (k, N))).astype(np.float32)
    if seed < 0: X = np.full((M, N), 7, np.float32)  # constant matrix
    n = min(M, N); Rc = min(R, n)
    u, v = ctx.svd_init(torch.from_numpy(X[None]).cuda(), R)
    raw = (v if N <= M else u)[0].cpu().numpy().tobytes()
    g = np.frombuffer(raw[: len(raw) // 8 * 8], dtype=np.float64)
    uo, vo = oracle.svd_topr_any(X, R)
    raw = (vo if N <= M else uo).tobytes()
    o = np.frombuffer(raw[: len(raw) // 8 * 8], dtype=np.float64)
    if stage == 9:
        print("d/e/lam equal:", np.array_equal(g[:2 * n + Rc], o[:2 * n + Rc]))
        print("lam", g[2 * n: 2 * n + Rc])
    else:
        r0 = stage % 100
        for q in range(min(R // 2, Rc - r0)):
            a, b = g[q * n:(q + 1) * n], o[q * n:(q + 1) * n]
            same = np.array_equal(a.view(np.int64), b.view(np.int64))
            print(f"vector {r0 + q}: {'same' if same else 'DIFFERENT'}  norm gpu {np.linalg.norm(a):.6e} oracle {np.linalg.norm(b):.6e}" + ("" if same else f"  first diff at {int(np.argmax(a != b))}: {a[np.argmax(a != b)]!r} vs {b[np.argmax(a != b)]!r}"))
    sys.exit(0)
M, N, R, seed = (int(v) for v in sys.argv[1:5])
for stage in [9] + [int(v) for v in sys.argv[5:]]:
    print("== stage", stage, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(stage), str(M), str(N), str(R), str(seed)])
