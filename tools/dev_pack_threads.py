"""Diagnostic: liblrf_pack.so's container packing of 256 images' factors (ranks (7,3,3)) by thread count, and the host->bytes
encode through the pipelined encoder; prints the affinity count and the cgroup CPU quota of the box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, lrf_amd
from lrf_amd.codec import pack_streams_native
print("affinity cpus", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
U, V = lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
Un, Vn = U.cpu().numpy(), V.cpu().numpy()
host = imgs.cpu().pin_memory()
for th in (8, 16, 24, 32, 48, 64, 128, 0):
    pack_streams_native(Un, Vn, (512, 768), (7, 3, 3), (-16, 15), (8, 8), "uint8", threads=th)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); s = pack_streams_native(Un, Vn, (512, 768), (7, 3, 3), (-16, 15), (8, 8), "uint8", threads=th); ts.append(time.perf_counter() - t0)
    lrf_amd.qmf_encode_batch(host, rank=7, pack_workers=th)
    te = []
    for _ in range(5):
        t0 = time.perf_counter(); lrf_amd.qmf_encode_batch(host, rank=7, pack_workers=th); te.append(time.perf_counter() - t0)
    print(f"threads {th:3d}: pack alone min/median {min(ts)*1e3:.1f}/{sorted(ts)[2]*1e3:.1f} ms, host->bytes min/median {min(te)*1e3:.1f}/{sorted(te)[2]*1e3:.1f} ms")
