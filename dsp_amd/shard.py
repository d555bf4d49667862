"""Work partitioning for several GPUs: files shard independently, nothing is exchanged.

The reference treats one file as the unit of work, transaction and recovery
(tools/xmltodb.cpp:163-218, :306) and walks a sorted file list (tools/xmltodb.cpp:147).  With G
ranks (one process per GPU) rank r takes files r, r+G, ... of that list.  One huge file is cut
at record terminators, so every part is a complete input of its own: the terminator is a '\\n'
preceded by endbz (cifile::readline, public/_public.cpp:866-892), the first part keeps whatever
precedes the first record and the bytes behind the last terminator stay with the last part
(where the parser drops them, as the reference does).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def assign_files(files: Sequence[str], rank: int, world: int) -> List[str]:
    """Files of this rank: round-robin over the sorted list."""
    if not 0 <= rank < world:
        raise ValueError("rank outside the world")
    return sorted(files)[rank::world]


def assign_by_size(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Size-balanced alternative: largest file first onto the least loaded rank.  Returns, per
    rank, the indices into `sizes` in ascending order."""
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(sizes)), key=lambda k: (-sizes[k], k)):
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += sizes[i]
    return [sorted(x) for x in out]


def _is_terminator(buf: np.ndarray, nl: int, mark: bytes) -> bool:
    m = len(mark)
    if nl < m:
        return m == 0
    return m == 0 or buf[nl - m:nl].tobytes() == mark


def split_points(buf, parts: int, endbz: str | bytes = "<endl/>") -> List[Tuple[int, int]]:
    """Cuts `buf` into at most `parts` byte ranges [lo, hi) that end behind a record terminator
    (the last one ends at len(buf)).  Ranges are contiguous and cover the buffer; fewer ranges
    come back when there are not enough terminators."""
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    mark = endbz.encode() if isinstance(endbz, str) else bytes(endbz)
    n = int(a.size)
    if parts < 1:
        raise ValueError("parts must be positive")
    cuts = [0]
    for k in range(1, parts):
        want = max(n * k // parts, cuts[-1])
        pos = want
        cut = None
        while pos < n:
            # next '\n' at or behind `pos`, looked for in bounded windows
            win = a[pos:pos + (1 << 16)]
            hits = np.flatnonzero(win == 10)
            found = False
            for h in hits:
                nl = pos + int(h)
                if _is_terminator(a, nl, mark):
                    cut = nl + 1
                    found = True
                    break
            if found:
                break
            pos += win.size
        if cut is None or cut >= n:
            break
        if cut > cuts[-1]:
            cuts.append(cut)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]


# ---------------------------------------------------------------------------------------------
# One process per GPU: which GPU a rank sees and which host cores its threads run on.
#
# The reference scales by processes (one xmltodb per directory, started by tools/procctl.cpp:55-67);
# here every rank is given ONE GPU (through ROCR_VISIBLE_DEVICES, set before the process touches the
# runtime, so the rank's device 0 is that GPU and no rank can reach another's) and a CPU set of its
# own: the cores of the GPU's NUMA node, divided among the ranks whose GPUs sit on that node, or an
# even share of the cores the process may use where the topology cannot be read.  The reading and
# uploading threads a rank starts later inherit its set.

def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def format_cpulist(cpus: Sequence[int]) -> str:
    cpus = sorted(set(cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def gpu_numa_nodes(sysfs: str = "/sys") -> List[int] | None:
    """NUMA node of every GPU in the order the runtime numbers them (the KFD topology's nodes that
    have compute units, in node order), or None where the topology is not readable."""
    import os
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    try:
        ids = sorted(int(d) for d in os.listdir(base) if d.isdigit())
    except OSError:
        return None
    out: List[int] = []
    for i in ids:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(base, str(i), "properties")) if " " in line)
            if int(props.get("simd_count", "0")) == 0:
                continue  # a CPU node
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            addr = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
            out.append(int(open(os.path.join(sysfs, "bus/pci/devices", addr, "numa_node")).read()))
        except (OSError, ValueError):
            return None
    return out or None


def rank_cpu_sets(world: int, avail: Sequence[int], gpu_nodes: Sequence[int] | None = None,
                  node_cpus: dict | None = None, gpus: Sequence[int] | None = None) -> List[List[int]]:
    """Disjoint CPU sets, one per rank.  `gpus[r]` is rank r's GPU (default r), `gpu_nodes[g]` the NUMA
    node of GPU g and `node_cpus[n]` the cores of node n.  Ranks whose GPUs share a node divide that
    node's cores (those of them in `avail`) evenly; without a usable topology, or where a node's share
    would be empty, all ranks divide `avail` evenly."""
    avail = sorted(set(avail))
    gpus = list(gpus) if gpus is not None else list(range(world))

    def even() -> List[List[int]]:
        k = max(1, len(avail) // world)
        return [avail[r * k:(r + 1) * k] if (r + 1) * k <= len(avail) else avail[(r * k) % max(1, len(avail)):][:k] or avail[:1]
                for r in range(world)]

    if not gpu_nodes or not node_cpus or any(g >= len(gpu_nodes) or gpu_nodes[g] < 0 for g in gpus):
        return even()
    out: List[List[int]] = [[] for _ in range(world)]
    for node in sorted(set(gpu_nodes[g] for g in gpus)):
        ranks = [r for r in range(world) if gpu_nodes[gpus[r]] == node]
        cpus = [c for c in node_cpus.get(node, []) if c in set(avail)]
        k = len(cpus) // len(ranks)
        if k == 0:
            return even()
        for j, r in enumerate(ranks):
            out[r] = cpus[j * k:(j + 1) * k]
    return out


def count_gpus(sysfs: str = "/sys", dev: str = "/dev") -> int | None:
    """GPUs of this machine without touching the runtime: the KFD topology's compute nodes, else the
    render nodes under /dev/dri; None where neither can be read."""
    import glob
    import os
    nodes = gpu_numa_nodes(sysfs)
    if nodes:
        return len(nodes)
    n = len(glob.glob(os.path.join(dev, "dri", "renderD*")))
    return n or None


def isolate_rank(local_rank: int, world: int, ngpus: int | None = None, sysfs: str = "/sys") -> dict:
    """Call before the process touches the GPU runtime.  Narrows ROCR_VISIBLE_DEVICES to this rank's GPU
    (the rank then uses device 0; with fewer GPUs than ranks several ranks share one) and the process's
    CPU affinity to this rank's share.  A list a launcher has narrowed already (ROCR_VISIBLE_DEVICES or
    HIP_VISIBLE_DEVICES set) is respected: this rank takes its entry of the former, the latter is left
    alone.  Returns what was set: {"gpu": the runtime's name of the GPU or None, "cpus": cpulist}."""
    import os
    nodes = gpu_numa_nodes(sysfs)
    if ngpus is None:
        ngpus = count_gpus(sysfs) or world
    seen = [v for v in os.environ.get("ROCR_VISIBLE_DEVICES", "").split(",") if v != ""]
    ids = seen if seen else [str(g) for g in range(max(1, ngpus))]
    mine = None
    if not os.environ.get("HIP_VISIBLE_DEVICES"):
        mine = ids[local_rank % len(ids)]
        os.environ["ROCR_VISIBLE_DEVICES"] = mine
    node_cpus = None
    if nodes and all(v.isdigit() for v in ids):
        node_cpus = {}
        for n in set(nodes):
            try:
                node_cpus[n] = parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % n)).read())
            except OSError:
                node_cpus = None
                break
    gpus = [int(ids[r % len(ids)]) for r in range(world)] if node_cpus else None
    cpus = rank_cpu_sets(world, sorted(os.sched_getaffinity(0)), nodes if node_cpus else None, node_cpus, gpus)[local_rank]
    if cpus:
        os.sched_setaffinity(0, cpus)
    return {"gpu": mine, "cpus": format_cpulist(cpus)}
