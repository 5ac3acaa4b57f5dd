import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_usable():
    """Is there an AMD GPU to run the gpu-marked tests on?  Asked WITHOUT any HIP / torch.cuda call: the two-rank tests
    start child processes from a parent that must not have initialised the GPU runtime, and torch.cuda.device_count()
    may fall back to hipGetDeviceCount (which does).  The kernel driver's device node and its topology say enough."""
    try:
        from lrf_amd import _lib
        if not os.path.exists(_lib.LIB_PATH) or not os.path.exists("/dev/kfd"):
            return False
        if os.environ.get("HIP_VISIBLE_DEVICES") == "" or os.environ.get("CUDA_VISIBLE_DEVICES") == "":
            return False
        if not os.access("/dev/kfd", os.R_OK | os.W_OK):
            return False
        nodes = "/sys/class/kfd/kfd/topology/nodes"
        if not os.path.isdir(nodes):
            return True  # a driver without the topology tree: the device node is all there is to ask
        for n in os.listdir(nodes):
            try:
                props = dict(ln.split(None, 1) for ln in open(os.path.join(nodes, n, "properties")).read().splitlines() if " " in ln)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:  # CPU nodes have simd_count 0
                return True
        return False
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU (or without the built library) skips the gpu-marked tests instead of
    failing them; `-m gpu` on the GPU box runs them (and the product path itself still fails loudly without a GPU)."""
    if _gpu_usable():
        return
    skip = pytest.mark.skip(reason="needs an MI355X and lrf_amd/liblrf_hip.so")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def make_image(spec):
    """Rebuilds a fixture's input from its recipe (tools/gen_golden.py make_image); the natural image is stored."""
    import torch
    kind = spec["kind"]
    if kind == "randint":
        g = torch.Generator().manual_seed(spec["seed"])
        return torch.randint(0, 256, (3, spec["H"], spec["W"]), dtype=torch.uint8, generator=g)
    if kind == "smooth":
        # torch.randn fills large tensors in per-thread chunks: the recipe is pinned at one thread (as the generator ran)
        nt = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            g = torch.Generator().manual_seed(spec["seed"])
            H, W = spec["H"], spec["W"]
            base = torch.rand(1, 3, H // 8, W // 8, generator=g) * 255
            sm = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)[0]
            return (sm + torch.randn(sm.shape, generator=g) * 4).clamp(0, 255).to(torch.uint8)
        finally:
            torch.set_num_threads(nt)
    if kind == "const":
        return torch.full((3, spec["H"], spec["W"]), spec["value"], dtype=torch.uint8)
    raise ValueError(kind)


def config3_image(idx):
    """Image idx (0..23) of the BASELINE config-3 stand-in set (tools/gen_golden.py config3_image): twenty smooth synthetic
    512x768 images and four 512x768 crops of the natural fixture image."""
    import torch
    if idx < 20:
        return make_image(dict(kind="smooth", seed=100 + idx, H=512, W=768))
    nat = torch.from_numpy(np.load(os.path.join(GOLDEN, "nat_q7.npz"))["image"])
    y0, x0 = ((0, 0), (150, 0), (0, 224), (150, 224))[idx - 20]
    return nat[:, y0:y0 + 512, x0:x0 + 768].contiguous()


class Case:
    def __init__(self, name):
        import hashlib

        import torch
        self.name = name
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        self.spec = json.loads(str(z["spec"]))
        self.kwargs = json.loads(str(z["kwargs"]))
        self.encoded = z["encoded"].tobytes()
        self.ranks = [int(r) for r in z["ranks"]] if "ranks" in z else None
        if "image" in z:
            self.image = torch.from_numpy(z["image"])
        elif self.spec["kind"] == "natural":
            self.image = torch.from_numpy(np.load(os.path.join(GOLDEN, "nat_q7.npz"))["image"])
        else:
            self.image = make_image(self.spec)
        assert hashlib.sha256(self.image.numpy().tobytes()).hexdigest() == str(z["image_sha256"]), \
            f"{name}: regenerated input differs from the fixture's"
        self.psnr = float(z["psnr"])
        self.bpp = float(z["bpp"])
        self.decoded_sha256 = str(z["decoded_sha256"])

    def signs(self):
        return [self.z[f"sign{c}"] for c in range(3)]

    def ref_factors(self):
        """the reference's int8 factors [u_y, v_y, u_cb, v_cb, u_cr, v_cr] parsed from its byte stream"""
        from lrf_amd.container import decode_tensor, separate_bytes
        _, fac = separate_bytes(self.encoded, 2)
        return [decode_tensor(f) for f in separate_bytes(fac, 6)]


QMF_CASES = ["tiny_q7", "tiny_r7", "tiny_q20", "tiny_rank2", "tiny_rank1", "tiny_it1", "tiny_it2", "odd_q7", "odd_r7",
             "smooth_q7", "smooth_r7", "zero_q7", "const_q7", "s1_q7", "s1_r7", "nat_q7", "nat_r7", "s2odd_q7"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
