import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lrf_amd import _lib
from lrf_amd.codec import anyshape_ranks
B=256; H,W=512,768
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0,256,(B,3,H,W),dtype=torch.uint8,device="cuda",generator=g)
ctx=_lib.context(0)
ranks = anyshape_ranks((H,W),(4,4),None,20.0)
for rep in range(2):
    for c in range(3):
        X = ctx.planes_any(imgs,(4,4),c); ctx.decompose(X, ranks[c], 10, -16, 15)
torch.cuda.synchronize()
