"""lrf_amd/placement.py (CPU suite): the GPU -> NUMA node -> CPU list lookup on a fabricated sysfs tree, the visible-device
variables, and that nothing is bound when the host does not say."""
import importlib.util
import os

from conftest import ROOT


def _placement():
    spec = importlib.util.spec_from_file_location("lrf_placement_t", os.path.join(ROOT, "lrf_amd", "placement.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)  # no package import: bench.py loads it the same way, before torch
    return m


def _fake_sysfs(root, gpus, cpu_nodes=1, render_minor0=0):
    """gpus: list of (bus, numa_node or None).  KFD nodes 0 .. cpu_nodes-1 are CPU nodes (simd_count 0).  render_minor0 > 0
    gives GPU i the property drm_render_minor = render_minor0 + i."""
    base = root / "class" / "kfd" / "kfd" / "topology" / "nodes"
    for n in range(cpu_nodes):
        (base / str(n)).mkdir(parents=True)
        (base / str(n) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    have = sorted(os.sched_getaffinity(0))
    half = max(1, len(have) // 2)
    lists = {0: have[:half], 1: have[half:] or have[:half]}
    for i, (bus, numa) in enumerate(gpus):
        d = base / str(cpu_nodes + i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\nunique_id 12345\n"
                                        + (f"drm_render_minor {render_minor0 + i}\n" if render_minor0 else ""))
        dev = root / "bus" / "pci" / "devices" / ("0000:%02x:00.0" % bus)
        dev.mkdir(parents=True)
        if numa is not None:
            (dev / "numa_node").write_text(f"{numa}\n")
            (dev / "local_cpulist").write_text(",".join(str(c) for c in lists[numa]) + "\n")
            nd = root / "devices" / "system" / "node" / f"node{numa}"
            nd.mkdir(parents=True, exist_ok=True)
            (nd / "cpulist").write_text(",".join(str(c) for c in lists[numa]) + "\n")
    return lists


def test_gpu_table_and_binding_decision(tmp_path):
    m = _placement()
    lists = _fake_sysfs(tmp_path, [(0x05, 0), (0x85, 1), (0xc5, None)], cpu_nodes=2)
    table = m.gpu_table(str(tmp_path))
    assert [g["bdf"] for g in table] == ["0000:05:00.0", "0000:85:00.0", "0000:c5:00.0"]
    assert [g["numa_node"] for g in table] == [0, 1, -1] and table[0]["cpus"] == lists[0]
    r0 = m.bind_to_gpu_numa(0, sysfs=str(tmp_path), env={}, apply=False)
    assert r0["bound"] and r0["numa_node"] == 0 and r0["cpus"] == len(set(lists[0]))
    r1 = m.bind_to_gpu_numa(1, sysfs=str(tmp_path), env={}, apply=False)
    assert r1["bound"] and r1["numa_node"] == 1 and r1["bdf"] == "0000:85:00.0"
    r2 = m.bind_to_gpu_numa(2, sysfs=str(tmp_path), env={}, apply=False)
    assert not r2["bound"] and "neither a NUMA node nor local CPUs" in r2["reason"]
    r9 = m.bind_to_gpu_numa(9, sysfs=str(tmp_path), env={}, apply=False)
    assert not r9["bound"] and "not among" in r9["reason"]
    # visible-device lists re-index the table; strings that are not index lists switch the binding off
    rv = m.bind_to_gpu_numa(0, sysfs=str(tmp_path), env={"HIP_VISIBLE_DEVICES": "1,0"}, apply=False)
    assert rv["numa_node"] == 1
    rr = m.bind_to_gpu_numa(0, sysfs=str(tmp_path), env={"ROCR_VISIBLE_DEVICES": "1,2", "HIP_VISIBLE_DEVICES": "1"}, apply=False)
    assert rr["bdf"] == "0000:c5:00.0" and not rr["bound"]
    ru = m.bind_to_gpu_numa(0, sysfs=str(tmp_path), env={"ROCR_VISIBLE_DEVICES": "GPU-abcdef"}, apply=False)
    assert not ru["bound"] and "plain index" in ru["reason"]


def test_no_topology_means_no_binding(tmp_path):
    m = _placement()
    before = os.sched_getaffinity(0)
    r = m.bind_to_gpu_numa(0, sysfs=str(tmp_path))
    assert not r["bound"] and r["cpus"] == len(before) and os.sched_getaffinity(0) == before


def test_binding_applies_and_restores(tmp_path):
    m = _placement()
    before = os.sched_getaffinity(0)
    lists = _fake_sysfs(tmp_path, [(0x05, 0)])
    try:
        r = m.bind_to_gpu_numa(0, sysfs=str(tmp_path), env={})
        assert r["bound"] and os.sched_getaffinity(0) == set(lists[0])
    finally:
        os.sched_setaffinity(0, before)


def test_gpus_whose_render_node_cannot_be_opened_are_not_counted(tmp_path):
    """A container that was handed two of the host's three GPUs still sees three KFD nodes in sysfs; HIP enumerates the two
    it can open, and device index 1 is the THIRD node."""
    m = _placement()
    sysfs, devfs = tmp_path / "sys", tmp_path / "dev"
    _fake_sysfs(sysfs, [(0x05, 0), (0x85, 1), (0xc5, 1)], render_minor0=128)
    (devfs / "dri").mkdir(parents=True)
    for minor in (128, 130):
        (devfs / "dri" / f"renderD{minor}").write_text("")
    table = m.gpu_table(str(sysfs), str(devfs))
    assert [g["bdf"] for g in table] == ["0000:05:00.0", "0000:c5:00.0"]
    r1 = m.bind_to_gpu_numa(1, sysfs=str(sysfs), env={}, apply=False, devfs=str(devfs))
    assert r1["bdf"] == "0000:c5:00.0" and r1["numa_node"] == 1
    # no /dev/dri at all: nothing is known, every node stays
    assert len(m.gpu_table(str(sysfs), str(tmp_path / "nodev"))) == 3


def test_local_cpulist_is_used_when_there_is_no_numa_node(tmp_path):
    m = _placement()
    _fake_sysfs(tmp_path, [(0x05, None)])
    have = sorted(os.sched_getaffinity(0))
    (tmp_path / "bus" / "pci" / "devices" / "0000:05:00.0" / "numa_node").write_text("-1\n")
    (tmp_path / "bus" / "pci" / "devices" / "0000:05:00.0" / "local_cpulist").write_text(f"{have[0]}\n")
    r = m.bind_to_gpu_numa(0, sysfs=str(tmp_path), env={}, apply=False)
    assert r["numa_node"] == -1 and r["bound"] and (len(have) == 1 or (r["cpus"] == 1 and "local_cpulist" in r["reason"]))


def test_confirm_binding_undoes_a_binding_for_the_wrong_gpu(tmp_path):
    m = _placement()
    before = os.sched_getaffinity(0)
    lists = _fake_sysfs(tmp_path, [(0x05, 0)])
    try:
        r = m.bind_to_gpu_numa(0, sysfs=str(tmp_path), env={})
        same = m.confirm_binding(r, "05:00.0")
        assert same["bound"] and same.get("confirmed", True) and os.sched_getaffinity(0) == set(lists[0])
        unknown = m.confirm_binding(r, None)
        assert unknown["bound"] and "confirmed" not in unknown
        other = m.confirm_binding(r, "0000:85:00.0")
        assert not other["bound"] and other["confirmed"] is False and "binding undone" in other["reason"]
        assert os.sched_getaffinity(0) == before and other["cpus"] == len(before)
    finally:
        os.sched_setaffinity(0, before)
