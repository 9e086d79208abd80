"""Pin a rank to the host cores next to its GPU (one RGB-D stream per GPU).

A rank of the multi-GPU bench is one Python thread that enqueues ~15 launches per frame at ~6000 frames/s plus the HIP
runtime's helper threads; on an 8-GPU node eight of them run side by side, and a rank whose threads wander to the other
socket pays a cross-socket hop on every doorbell and every pinned-buffer copy.  `pin_to_gpu_numa_node` reads the NUMA node
of the rank's GPU from sysfs and restricts the process to that node's cores with `os.sched_setaffinity` - no `numactl`
wrapper (a launcher hop after the GPU was initialised is forbidden on this pool), no torch or HIP call (it runs before
either is imported, so every allocation and thread of the rank is made on the right node).

Device order: ROCm enumerates the GPUs in KFD topology order (/sys/class/kfd/kfd/topology/nodes/<n>, the nodes with
simd_count > 0), filtered by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when these hold plain
indices.  Anything unreadable means "not pinned" with the reason in the returned record; nothing here can fail a run.
"""
from __future__ import annotations

import glob
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in (text or "").replace("\n", "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def format_cpulist(cpus):
    cpus = sorted(set(cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def kfd_gpu_nodes(root="/sys/class/kfd/kfd/topology/nodes"):
    """[(domain, bus, device, function)] of the GPUs in KFD order (= the HIP device order without a visibility filter)"""
    gpus = []
    nodes = sorted((int(os.path.basename(p)), p) for p in glob.glob(os.path.join(root, "*")) if os.path.basename(p).isdigit())
    for _, path in nodes:
        props = {}
        for line in (_read(os.path.join(path, "properties")) or "").splitlines():
            kv = line.split()
            if len(kv) == 2:
                props[kv[0]] = kv[1]
        if int(props.get("simd_count", "0")) <= 0:
            continue                                         # a CPU node
        loc = int(props.get("location_id", "0"))
        gpus.append((int(props.get("domain", "0")), (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7))
    return gpus


def visible_indices(n, environ=None):
    """the KFD-order indices the runtime exposes as devices 0, 1, ... (None if a filter is set that is not a list of indices)"""
    environ = os.environ if environ is None else environ
    order = list(range(n))
    for name in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = environ.get(name)
        if v is None:
            continue
        try:
            picked = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return None                                      # UUIDs: not resolved here
        if any(i < 0 or i >= len(order) for i in picked):
            return None
        order = [order[i] for i in picked]
    return order


def gpu_numa_node(local_rank, environ=None, kfd_root="/sys/class/kfd/kfd/topology/nodes", pci_root="/sys/bus/pci/devices"):
    """(numa node or None, reason)"""
    gpus = kfd_gpu_nodes(kfd_root)
    if not gpus:
        return None, "no GPU node under %s" % kfd_root
    order = visible_indices(len(gpus), environ)
    if order is None:
        return None, "device visibility filter is not a list of indices"
    if local_rank >= len(order):
        return None, "local rank %d but %d visible GPU(s)" % (local_rank, len(order))
    dom, bus, dev, fn = gpus[order[local_rank]]
    bdf = "%04x:%02x:%02x.%x" % (dom, bus, dev, fn)
    node = _read(os.path.join(pci_root, bdf, "numa_node"))
    if node is None:
        return None, "no numa_node for %s" % bdf
    if int(node) < 0:
        return None, "%s reports numa_node %s (no NUMA information on this host)" % (bdf, node)
    return int(node), bdf


def pin_to_gpu_numa_node(local_rank, environ=None, node_root="/sys/devices/system/node", **roots):
    """Restrict this process to the cores of the NUMA node of GPU `local_rank`, within the cores it is already allowed.
    Returns {"pinned": bool, "cpus": "<list>", "numa_node": n or None, "why": "..."} - what bench.py prints."""
    rec = {"pinned": False, "cpus": None, "numa_node": None, "why": None}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        rec["cpus"] = format_cpulist(allowed)
        node, why = gpu_numa_node(local_rank, environ, **roots)
        rec["why"] = why
        if node is None:
            return rec
        rec["numa_node"] = node
        cpus = parse_cpulist(_read(os.path.join(node_root, "node%d" % node, "cpulist")))
        want = sorted(set(cpus) & set(allowed))
        if not want:
            rec["why"] = "NUMA node %d has no core this process may use" % node
            return rec
        if want == allowed:
            rec["why"] = "already confined to NUMA node %d (GPU %s)" % (node, why)
            rec["pinned"] = True
            return rec
        os.sched_setaffinity(0, want)
        rec.update(pinned=True, cpus=format_cpulist(want), why="GPU %s" % why)
    except Exception as e:       # noqa: BLE001  (never fail a run over placement)
        rec["why"] = "not pinned: %r" % (e,)
    return rec
