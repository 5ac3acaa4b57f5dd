"""CPU placement of a rank next to its GPU (one process per GPU).

Standard library only and importable without the package (bench.py loads this file by path BEFORE `import torch`, so that
the threads torch, the HIP runtime and liblrf_pack.so start later — and the page-locked buffers the rank's thread touches
first — inherit the binding).  Nothing here touches the GPU: the GPUs are read from the KFD topology in sysfs, whose node
order is the order the HIP runtime enumerates devices in — restricted to the nodes whose render node this process may open
(a container that is handed some of the host's GPUs still sees every KFD node in sysfs); ROCR_VISIBLE_DEVICES /
HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES are honoured when they are plain index lists.  Because that mapping is an inference,
`confirm_binding` compares it with the PCI address the HIP runtime reports once torch is up and undoes a binding made for
the wrong GPU.  The reference has no counterpart (single process, experiments/comparison/eval.py:105-110)."""
import os
from typing import Dict, List, Optional


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _parse_cpulist(txt: str) -> List[int]:
    cpus: List[int] = []
    for part in txt.replace("\n", "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def _render_node_usable(minor: int, devfs: str) -> bool:
    """Whether /dev/dri/renderD<minor> can be opened by this process.  A host without /dev/dri at all (sysfs fixtures of the
    tests, unusual device managers) says nothing, and every node stays."""
    dri = os.path.join(devfs, "dri")
    if minor <= 0 or not os.path.isdir(dri):
        return True
    return os.access(os.path.join(dri, "renderD%d" % minor), os.R_OK | os.W_OK)


def gpu_table(sysfs: str = "/sys", devfs: str = "/dev") -> List[Dict]:
    """The GPUs of the KFD topology this process can open, in node order: [{"node", "bdf", "numa_node", "cpus"}].  numa_node
    is -1 when the kernel does not say (single-socket hosts, containers without the sysfs files); cpus then comes from the PCI
    device's local_cpulist when that exists, and is empty otherwise."""
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    gpus = []
    for n in nodes:
        props = _read(os.path.join(base, str(n), "properties"))
        if props is None:
            continue
        kv = {}
        for line in props.splitlines():
            parts = line.split()
            if len(parts) == 2 and parts[1].lstrip("-").isdigit():
                kv[parts[0]] = int(parts[1])
        if kv.get("simd_count", 0) <= 0:
            continue  # a CPU node
        if not _render_node_usable(kv.get("drm_render_minor", 0), devfs):
            continue  # a GPU of the host that this container was not given: HIP does not enumerate it
        loc, dom = kv.get("location_id", 0), kv.get("domain", 0)
        bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7)
        dev = os.path.join(sysfs, "bus", "pci", "devices", bdf)
        numa = _read(os.path.join(dev, "numa_node"))
        cpul = _read(os.path.join(dev, "local_cpulist"))
        try:
            numa_i = int(numa) if numa is not None else -1
        except ValueError:
            numa_i = -1
        cpus: List[int] = []
        node_list = _read(os.path.join(sysfs, "devices", "system", "node", "node%d" % numa_i, "cpulist")) if numa_i >= 0 else None
        txt = node_list if node_list else cpul
        try:
            cpus = _parse_cpulist(txt) if txt else []
        except ValueError:
            cpus = []
        gpus.append({"node": n, "bdf": bdf, "numa_node": numa_i, "cpus": cpus})
    return gpus


def _visible(gpus: List[Dict], env: Dict[str, str]) -> Optional[List[Dict]]:
    """Applies the visible-device lists (ROCR first, then HIP / CUDA on top of it); None when one is not a plain index list."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        val = env.get(var)
        if val is None and var == "HIP_VISIBLE_DEVICES":
            val = env.get("CUDA_VISIBLE_DEVICES")
        if val is None or val.strip() == "":
            continue
        try:
            idx = [int(v) for v in val.split(",")]
        except ValueError:
            return None
        if any(i < 0 or i >= len(gpus) for i in idx):
            return None
        gpus = [gpus[i] for i in idx]
    return gpus


def bind_to_gpu_numa(device_index: int, sysfs: str = "/sys", env: Optional[Dict[str, str]] = None, apply: bool = True,
                     devfs: str = "/dev") -> Dict:
    """Restricts the calling thread (and every thread it starts from now on) to the CPUs of the NUMA node HIP device
    `device_index` hangs off, intersected with the CPUs the process may use now.  Returns what was done:
    {"bound": bool, "numa_node", "bdf", "cpus": count after binding, "reason", "previous": the affinity before} — never
    raises: a host that does not say (no NUMA node and no local_cpulist, no KFD sysfs, exotic visible-device strings) leaves
    the affinity as it is."""
    env = dict(os.environ) if env is None else env
    try:
        have = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return {"bound": False, "reason": "no sched_getaffinity on this platform", "cpus": os.cpu_count() or 1}
    out = {"bound": False, "numa_node": -1, "bdf": None, "cpus": len(have), "reason": "", "previous": have}
    gpus = gpu_table(sysfs, devfs)
    if not gpus:
        out["reason"] = "no KFD topology in sysfs"
        return out
    vis = _visible(gpus, env)
    if vis is None:
        out["reason"] = "visible-device variables are not plain index lists"
        return out
    if device_index < 0 or device_index >= len(vis):
        out["reason"] = "device %d not among the %d visible GPUs of the KFD topology" % (device_index, len(vis))
        return out
    g = vis[device_index]
    out["numa_node"], out["bdf"] = g["numa_node"], g["bdf"]
    if not g["cpus"]:
        out["reason"] = "the kernel reports neither a NUMA node nor local CPUs for this GPU"
        return out
    want = sorted(set(g["cpus"]) & set(have))
    if not want:
        out["reason"] = "the GPU's NUMA node has no CPU this process may use"
        return out
    if len(want) == len(have):
        out["bound"], out["reason"] = True, "already inside the GPU's NUMA node"
        return out
    if apply:
        try:
            os.sched_setaffinity(0, want)
        except OSError as e:
            out["reason"] = "sched_setaffinity failed: %s" % e
            return out
    out["bound"], out["cpus"] = True, len(want)
    out["reason"] = "bound to NUMA node %d" % g["numa_node"] if g["numa_node"] >= 0 else "bound to the GPU's local_cpulist"
    return out


def _norm_bdf(bdf: str) -> str:
    bdf = bdf.strip().lower()
    return bdf if bdf.count(":") == 2 else "0000:" + bdf


def confirm_binding(info: Dict, pci_bus_id: Optional[str], apply: bool = True) -> Dict:
    """After the HIP runtime is up: `pci_bus_id` is what it reports for the rank's device ("0000:c5:00.0" or "c5:00.0";
    torch.cuda.get_device_properties(i).pci_bus_id gives the bus number on some builds only — pass None when unknown).  When
    it names another GPU than the one `bind_to_gpu_numa` inferred from sysfs, the binding was made for the wrong NUMA node:
    the affinity recorded in info["previous"] is restored and the returned copy says bound False with the reason."""
    out = dict(info)
    if not info.get("bound") or not info.get("bdf") or not pci_bus_id:
        return out
    if _norm_bdf(pci_bus_id) == _norm_bdf(info["bdf"]):
        out["confirmed"] = True
        return out
    out["bound"], out["confirmed"] = False, False
    out["reason"] = "sysfs said %s, the HIP runtime says %s: binding undone" % (info["bdf"], pci_bus_id)
    if apply and info.get("previous"):
        try:
            os.sched_setaffinity(0, info["previous"])
            out["cpus"] = len(info["previous"])
        except OSError as e:
            out["reason"] += " (restoring the affinity failed: %s)" % e
    return out
