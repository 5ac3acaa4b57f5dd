"""Load the reference's four hot-path modules WITHOUT executing its package __init__ files.

Only used in the build container (where /root/reference exists) to generate golden vectors
and to pin the oracle; nothing under tests/ -m gpu, bench.py or smoke() imports this.

`import lrf` as a whole fails here with ModuleNotFoundError (torchvision/skimage/... absent,
SURVEY.md §8c), so empty package objects are pre-registered with __path__ pointing at the
reference directories and the hot-path modules are imported one by one.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("LRF_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "lrf", "factorization"))


def load():
    """Returns a namespace with .fqmf (factorization.qmf), .cqmf, .cutils, .csvd modules."""
    if not available():
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    sys.dont_write_bytecode = True
    if "lrf.compression.qmf" not in sys.modules or not getattr(sys.modules.get("lrf"), "_graft_stub", False):
        for name in [m for m in sys.modules if m == "lrf" or m.startswith("lrf.")]:
            del sys.modules[name]
        for pkg, sub in (("lrf", "lrf"), ("lrf.factorization", "lrf/factorization"),
                         ("lrf.compression", "lrf/compression")):
            mod = types.ModuleType(pkg)
            mod.__path__ = [os.path.join(REF_ROOT, sub)]
            mod._graft_stub = True
            sys.modules[pkg] = mod
        importlib.import_module("lrf.factorization.utils")
        fq = importlib.import_module("lrf.factorization.qmf")
        sys.modules["lrf.factorization"].QMF = fq.QMF
        importlib.import_module("lrf.compression.utils")
        importlib.import_module("lrf.compression.qmf")
        importlib.import_module("lrf.compression.svd")
    ns = types.SimpleNamespace()
    ns.fqmf = sys.modules["lrf.factorization.qmf"]
    ns.futils = sys.modules["lrf.factorization.utils"]
    ns.cqmf = sys.modules["lrf.compression.qmf"]
    ns.cutils = sys.modules["lrf.compression.utils"]
    ns.csvd = sys.modules["lrf.compression.svd"]
    return ns
