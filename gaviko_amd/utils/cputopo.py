"""CPU sets for the ranks of one host (one process per GPU): whole physical cores, on the NUMA node of the rank's GPU.

A rank issues ~360 launches per step from its plan-replay thread while the autograd worker and the HIP runtime's helper threads run
beside it (2.3 ms of host issue per 5.4 ms step).  Cutting the sorted LOGICAL cpu ids into equal runs (round 4) is
disjoint by id, not by core: on the usual two-socket SMT numbering (0-63 socket 0, 64-127 socket 1, 128-255 their hyper-threads) an
eight-way cut hands rank r and rank r + 4 the two hardware threads of the same cores and puts ranks 2, 3 on the socket their GPUs do
not hang off.  Everything here is read from sysfs -- no HIP call, so it runs before GPU_MAX_HW_QUEUES has to be in place:

  /sys/devices/system/cpu/cpuN/topology/thread_siblings_list      hardware threads of cpuN's core
  /sys/devices/system/node/nodeK/cpulist                          NUMA node of every cpu
  /sys/class/kfd/kfd/topology/nodes/*/properties                  the GPUs in KFD order (simd_count > 0): `domain`, `location_id` = PCI address
  /sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node              NUMA node of that GPU

`plan(...)` is a pure function of those tables: every rank computes the whole table and keeps its own row.  When sysfs has nothing (a
container without the KFD topology, one thread per core, one node) it degrades step by step to the contiguous cut of physical cores.
"""
from __future__ import annotations

import glob
import os
import re


def parse_cpulist(text: str):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def core_of(cpus, sys_root="/sys"):
    """{cpu: core key} -- the key is the smallest hardware-thread id of the cpu's physical core (the cpu itself when sysfs says nothing)."""
    out = {}
    for c in cpus:
        t = _read(f"{sys_root}/devices/system/cpu/cpu{c}/topology/thread_siblings_list")
        try:
            out[c] = min(parse_cpulist(t)) if t else c
        except ValueError:
            out[c] = c
    return out


def node_of(cpus, sys_root="/sys"):
    """{cpu: NUMA node} (-1 where unknown)"""
    out = {c: -1 for c in cpus}
    for d in glob.glob(f"{sys_root}/devices/system/node/node[0-9]*"):
        t = _read(os.path.join(d, "cpulist"))
        if not t:
            continue
        n = int(re.search(r"node(\d+)$", d).group(1))
        try:
            for c in parse_cpulist(t):
                if c in out:
                    out[c] = n
        except ValueError:
            pass
    return out


def gpu_nodes(sys_root="/sys"):
    """NUMA node of every GPU in KFD enumeration order (what HIP's device index follows when no *_VISIBLE_DEVICES variable reorders it);
    -1 where the PCI device does not say.  [] when the KFD topology is not visible."""
    nodes = []
    dirs = glob.glob(f"{sys_root}/class/kfd/kfd/topology/nodes/[0-9]*")
    for d in sorted(dirs, key=lambda p: int(os.path.basename(p))):
        t = _read(os.path.join(d, "properties"))
        if not t:
            continue
        prop = {}
        for line in t.splitlines():
            kv = line.split()
            if len(kv) == 2 and re.fullmatch(r"-?\d+", kv[1]):
                prop[kv[0]] = int(kv[1])
        if prop.get("simd_count", 0) <= 0:
            continue                                              # a CPU node
        loc, dom = prop.get("location_id", 0), prop.get("domain", 0)
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        n = _read(f"{sys_root}/bus/pci/devices/{bdf}/numa_node")
        try:
            nodes.append(int(n.strip()) if n else -1)
        except ValueError:
            nodes.append(-1)
    return nodes


def visible_gpu_order(n_gpus: int, env=None):
    """KFD indices of HIP devices 0.. as the *_VISIBLE_DEVICES variables leave them (integers only; UUID forms -> no reordering known)."""
    env = os.environ if env is None else env
    order = list(range(n_gpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if not v:
            continue
        try:
            pick = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return None
        if any(i < 0 or i >= len(order) for i in pick):
            return None
        order = [order[i] for i in pick]
    return order


def plan(local_world: int, cpus, core, node, gpu_node):
    """-> (table, how): table[r] = sorted cpu list of local rank r.

    cpus: allowed logical cpus; core / node: {cpu: key}; gpu_node: NUMA node of local rank r's GPU, or [] / -1 entries when unknown.
    Rules, in order:  (1) a physical core (all its allowed hardware threads) goes to ONE rank;  (2) a rank's cores lie on its GPU's NUMA
    node, the node's cores shared evenly by the ranks whose GPUs hang off it;  (3) when (2) cannot be met for every rank (unknown nodes,
    a node with fewer allowed cores than ranks) the cores, ordered by (node, id), are cut into equal runs;  (4) fewer cores than ranks:
    round-robin over logical cpus, as before."""
    cpus = sorted(cpus)
    local_world = max(1, local_world)
    cores = {}
    for c in cpus:
        cores.setdefault(core.get(c, c), []).append(c)
    keys = sorted(cores, key=lambda k: (node.get(cores[k][0], -1), k))
    if len(keys) < local_world:
        k = max(1, len(cpus) // local_world)
        return [cpus[(r * k) % len(cpus): (r * k) % len(cpus) + k] or cpus[:1] for r in range(local_world)], "logical cpus (fewer physical cores than ranks)"

    def flat(ks):
        return sorted(c for k in ks for c in cores[k])

    by_node = {}
    for k in keys:
        by_node.setdefault(node.get(cores[k][0], -1), []).append(k)
    ok = len(gpu_node) >= local_world and all(n >= 0 and n in by_node for n in gpu_node[:local_world])
    if ok:
        ranks_on = {}
        for r in range(local_world):
            ranks_on.setdefault(gpu_node[r], []).append(r)
        ok = all(len(by_node[n]) >= len(rs) for n, rs in ranks_on.items())
    if ok:
        table = [None] * local_world
        for n, rs in ranks_on.items():
            share = len(by_node[n]) // len(rs)
            for i, r in enumerate(rs):
                table[r] = flat(by_node[n][i * share: (i + 1) * share])
        return table, "physical cores on the NUMA node of each rank's GPU"
    share = len(keys) // local_world
    return [flat(keys[r * share: (r + 1) * share]) for r in range(local_world)], "physical cores, equal runs (GPU NUMA nodes unknown or too few cores on one)"


def rank_cpu_table(local_world: int, cpus=None, sys_root="/sys", env=None):
    """The table of `plan` from this host's sysfs."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    gn = gpu_nodes(sys_root)
    order = visible_gpu_order(len(gn), env) if gn else None
    gpu_node = [gn[i] for i in order] if order is not None else []
    return plan(local_world, cpus, core_of(cpus, sys_root), node_of(cpus, sys_root), gpu_node)
