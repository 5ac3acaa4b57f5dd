"""One image through the patch=False branch (rocprofv3 --kernel-trace --stats target; development aid)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, lrf_amd
from conftest import config3_image
img = config3_image(3)
kw = {"patch": False} if len(sys.argv) < 2 or sys.argv[1] == "none" else {"patch_size": (int(sys.argv[1]), int(sys.argv[1]))}
for _ in range(6):
    s = lrf_amd.qmf_encode(img, quality=20, **kw)
torch.cuda.synchronize()
