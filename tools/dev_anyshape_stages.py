"""Stage times of k_any_eig (development aid): LRF_DEBUG_INIT_SWEEPS=s stops the kernel after stage s
(1 tridiagonalisation, 2 eigenvalues, 3 twisted factorisation, 4 Gram-Schmidt, 0 everything).  One process per setting."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from lrf_amd import _lib
    B, M, N, R = (int(a) for a in sys.argv[2:6])
    X = (torch.rand(B, M, N, device="cuda") * 255)
    ctx = _lib.context(0)
    ctx.svd_init(X, R)
    torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
    ctx.svd_init(X, R); torch.cuda.synchronize()
    print(f"{ctx.kernel_time(_lib.LRF_K_INIT)[0]:.2f}")
else:
    shapes = ((32, 512, 768, 102), (32, 1536, 256, 51), (32, 384, 1024, 77), (256, 512, 768, 102))
    if len(sys.argv) > 1:  # e.g. "256,6144,192,5"
        shapes = (tuple(int(v) for v in sys.argv[1].split(",")),)
    for shape in shapes:
        row = []
        for s in (1, 2, 3, 4, 0):
            env = dict(os.environ, LRF_DEBUG_INIT_SWEEPS=str(s))
            out = subprocess.run([sys.executable, __file__, "child"] + [str(v) for v in shape], env=env, capture_output=True, text=True)
            row.append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-200:])
        print(f"B,M,N,R={shape}: ms up to stage [tridiag, eigenvalues, twisted, gram-schmidt, all] = {row}", flush=True)
