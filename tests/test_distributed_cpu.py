"""CPU: the N>1 path -- bucket planner and bucketed mean all-reduce -- with world_size 2 on the gloo backend."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from gaviko_amd.distributed import GradReducer, plan_buckets, shard_range

CFG = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, backbone="vit-t16",
           num_prompts=32, prompt_latent_dim=20, local_dim=20, local_k=(6, 6, 6), DHW=(10, 10, 10), share_factor=1)


def _trainable(cfg):
    shapes = oracle.gaviko_param_shapes(cfg)
    names = [k for k in shapes if oracle.gaviko_trainable(k)]
    return names, [int(np.prod(shapes[k])) for k in names]


def test_plan_buckets_covers_buffer_once_and_orders_by_readiness():
    names, numels = _trainable(CFG)
    ranges = plan_buckets(names, numels, depth=12, share_factor=1, layers_per_bucket=4)
    total = sum(numels)
    cover = np.zeros(total, np.int32)
    for _, s, e in ranges:
        cover[s:e] += 1
    assert (cover == 1).all()
    ready = [r for r, _, _ in ranges]
    idx = [r for r in ready if r >= 0]
    assert idx == sorted(idx, reverse=True) and ready[-1] == -1 and set(idx) == {0, 4, 8}
    # a module shared by layers 2s, 2s+1 (share_factor 2) is final only after layer 2s
    cfg2 = dict(CFG, share_factor=2)
    n2, k2 = _trainable(cfg2)
    r2 = plan_buckets(n2, k2, depth=12, share_factor=2, layers_per_bucket=4)
    off = dict(zip(n2, np.cumsum([0] + k2[:-1])))
    o = off["transformer.local_attns.5.norm.weight"]          # used by layers 10, 11 -> ready at layer 10 -> bucket 8
    assert [r for r, s, e in r2 if s <= o < e] == [8]


def test_shard_range():
    assert [shard_range(32, r, 8) for r in (0, 7)] == [(0, 4), (28, 32)]
    with pytest.raises(ValueError):
        shard_range(30, 0, 8)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names, numels = _trainable(CFG)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(sum(numels), generator=g)
    mine = flat.clone()
    red = GradReducer(names, numels, depth=12, share_factor=1, layers_per_bucket=4)
    red.begin()
    done = []
    for layer in range(11, -1, -1):                 # the backward sweep
        before = red._next
        red.layer_done(flat, layer)
        done.append((layer, red._next - before))
    red.finish(flat)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = sum(gathered) / world
    ok = torch.allclose(flat, want, atol=1e-6)
    # the event-ordered form (one plan, grouped collectives): same result bit for bit, and every event it waits for is one the engine records
    flat2 = mine.clone()
    waited = []
    marks = {(k, r): (k, r) for k in ("loc", "gpa") for r in (8, 4, 0)}
    marks[("main", -1)] = ("main", -1)
    red.reduce_marked(flat2, marks, lambda stream, ev: waited.append(ev))
    ok = ok and torch.equal(flat2, flat)
    groups = red.ready_groups()
    ok = ok and [g[0] for g in groups] == [8, 4, -1] and sorted(i for _, m in groups for i in m) == list(range(len(red.ranges)))
    fired = {l: n for l, n in done if n}
    if rank == 0:
        ret["ok"], ret["fired"] = bool(ok), fired
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_mean_allreduce_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["ok"]
    assert set(ret["fired"]) == {8, 4, 0}          # buckets go out at their lowest layer, the rest at finish()


def test_bench_starts_its_own_ranks_when_run_bare(tmp_path):
    """`python bench.py --gpus N` with no launcher environment (the form the driver uses) must start N fresh rank processes itself,
    before any GPU call, with torch.distributed.run's variables, and relay rank 0's line.  --launch-check stops every rank before
    it touches a device, so this runs on CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GAVIKO_BENCH_LAUNCH_LOG"] = str(tmp_path / "rank")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launch-check"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 3 and line["launch_check"]["RANK"] == "0" and line["launch_check"]["MASTER_ADDR"] == "127.0.0.1"
    recs = [json.load(open(f"{tmp_path / 'rank'}.{k}")) for k in range(3)]
    assert [x["RANK"] for x in recs] == ["0", "1", "2"] and [x["LOCAL_RANK"] for x in recs] == ["0", "1", "2"]
    assert len({x["MASTER_PORT"] for x in recs}) == 1 and all(x["WORLD_SIZE"] == "3" and not x["gpu_initialised"] for x in recs)
    # the hardware-queue setting is in every rank's environment while HIP is still down (it is read when the runtime starts), whatever the
    # caller's environment held -- the round-3 driver line was measured on the default 4 queues because bench.py set it too late
    assert all(x["GPU_MAX_HW_QUEUES"] == "8" and x["queues_set_before_hip"] and x["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for x in recs)
    # every rank keeps a disjoint share of the CPUs this process may use (one each when there are fewer CPUs than ranks)
    avail = sorted(os.sched_getaffinity(0))
    sets = [x["cpus"] for x in recs]
    assert all(s and set(s) <= set(avail) and len(s) == max(1, len(avail) // 3) for s in sets)
    if len(avail) >= 3:
        assert len(set().union(*map(set, sets))) == sum(len(s) for s in sets)
    # under a launcher (RANK set) it must NOT spawn again
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and json.loads(r2.stdout.strip().splitlines()[-1])["launch_check"]["GAVIKO_BENCH_CHILD"] is None


def test_bench_keeps_a_callers_queue_setting_and_rank_cpu_sets_partition(tmp_path):
    """GPU_MAX_HW_QUEUES exported by the caller wins (setdefault), and rank_cpu_set cuts any CPU list into disjoint runs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GPU_MAX_HW_QUEUES"] = "6"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--launch-check"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])["launch_check"]
    assert rec["GPU_MAX_HW_QUEUES"] == "6" and rec["cpus"] is None and not rec["gpu_initialised"]
    sys.path.insert(0, root)
    import bench
    cpus = list(range(3, 35))                                          # 32 CPUs, 8 ranks -> 4 each
    empty = str(tmp_path)                                              # a sysfs with nothing in it: the fallback = equal contiguous runs
    parts = [bench.rank_cpu_set(r_, 8, cpus, sys_root=empty) for r_ in range(8)]
    assert all(len(p_) == 4 for p_ in parts) and sorted(c for p_ in parts for c in p_) == cpus
    assert bench.rank_cpu_set(5, 8, [0, 1], sys_root=empty) in ([0], [1])                # fewer CPUs than ranks: one each, shared
    assert bench.cputopo_ranges([0, 1, 2, 3, 128, 129, 200]) == "0-3,128-129,200"


def _fake_sysfs(root, sockets=2, cores_per_socket=64, smt=True, gpu_node=(0, 0, 0, 0, 1, 1, 1, 1), kfd=True):
    """A two-socket SMT host as sysfs shows it: cpu c and c + sockets * cores_per_socket are the two threads of one core, socket s owns cores
    s * cores_per_socket ...; eight GPUs, four per socket, listed in the KFD topology behind the two CPU nodes."""
    ncore = sockets * cores_per_socket
    ncpu = ncore * (2 if smt else 1)
    for c in range(ncpu):
        d = os.path.join(root, "devices/system/cpu", f"cpu{c}", "topology")
        os.makedirs(d)
        k = c % ncore
        open(os.path.join(d, "thread_siblings_list"), "w").write(f"{k},{k + ncore}\n" if smt else f"{k}\n")
    for s_ in range(sockets):
        d = os.path.join(root, "devices/system/node", f"node{s_}")
        os.makedirs(d)
        lo = s_ * cores_per_socket
        txt = f"{lo}-{lo + cores_per_socket - 1}" + (f",{lo + ncore}-{lo + ncore + cores_per_socket - 1}" if smt else "")
        open(os.path.join(d, "cpulist"), "w").write(txt + "\n")
    if kfd:
        for i in range(sockets):                                          # CPU nodes first, as KFD lists them
            d = os.path.join(root, "class/kfd/kfd/topology/nodes", str(i))
            os.makedirs(d)
            open(os.path.join(d, "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
        for g, n in enumerate(gpu_node):
            d = os.path.join(root, "class/kfd/kfd/topology/nodes", str(sockets + g))
            os.makedirs(d)
            bus = 0x10 + 0x10 * g
            open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\nunique_id 123456789012345678\n")
            p = os.path.join(root, "bus/pci/devices", f"0000:{bus:02x}:00.0")
            os.makedirs(p)
            open(os.path.join(p, "numa_node"), "w").write(f"{n}\n")
    return ncpu


def test_rank_cpu_sets_follow_physical_cores_and_gpu_numa_nodes(tmp_path):
    """Round-4 verdict, weak #11: cut by logical id, ranks r and r + 4 of an 8-rank job held the two hardware threads of the same cores and
    ranks 2, 3 sat on the other socket from their GPUs.  With the sysfs tables of a two-socket SMT host: no two ranks share a physical
    core, every rank sits on its GPU's node, a rank holds both threads of each of its cores, the table is the same from every rank."""
    from gaviko_amd.utils import cputopo
    root = str(tmp_path / "sys")
    ncpu = _fake_sysfs(root)
    cpus = list(range(ncpu))
    table, how = cputopo.rank_cpu_table(8, cpus, root, env={})
    assert "NUMA node" in how
    core = cputopo.core_of(cpus, root)
    node = cputopo.node_of(cpus, root)
    owners = {}
    for r, mine in enumerate(table):
        assert len(mine) == 32                                           # 16 cores x 2 threads
        assert {node[c] for c in mine} == {0 if r < 4 else 1}            # GPUs 0-3 hang off node 0, 4-7 off node 1
        for c in mine:
            assert owners.setdefault(core[c], r) == r, f"core {core[c]} shared by ranks {owners[core[c]]} and {r}"
        assert all((c + 128) % 256 in mine for c in mine)                # both hardware threads of every core it holds
    # the round-4 cut on the same host, for the record: ranks 0 and 4 are hyper-thread siblings
    old = [cpus[r * 32:(r + 1) * 32] for r in range(8)]
    assert {core[c] for c in old[0]} == {core[c] for c in old[4]}
    # a GPU order that alternates sockets, and a visible-devices list that reverses it
    root2 = str(tmp_path / "sys2")
    _fake_sysfs(root2, gpu_node=(0, 1, 0, 1, 0, 1, 0, 1))
    t2, _ = cputopo.rank_cpu_table(8, cpus, root2, env={})
    assert [{node[c] for c in m} for m in t2] == [{0}, {1}] * 4
    t3, _ = cputopo.rank_cpu_table(2, cpus, root2, env={"HIP_VISIBLE_DEVICES": "1,0"})
    assert {node[c] for c in t3[0]} == {1} and {node[c] for c in t3[1]} == {0} and len(t3[0]) == 128
    # launcher affinity narrower than the host (a cgroup share of socket 0 only, threads 0-31 and their siblings): GPUs on node 1 cannot be
    # honoured -> equal runs of physical cores, still no shared core
    narrow = list(range(0, 32)) + list(range(128, 160))
    t4, how4 = cputopo.rank_cpu_table(8, narrow, root, env={})
    assert "equal runs" in how4 and all(len(m) == 8 for m in t4)
    assert all(len({core[c] for c in m}) == 4 for m in t4) and len({core[c] for m in t4 for c in m}) == 32
    # no KFD topology visible (this build container): physical cores in equal runs; nothing at all: logical ids
    root5 = str(tmp_path / "sys5")
    _fake_sysfs(root5, kfd=False)
    t5, how5 = cputopo.rank_cpu_table(8, cpus, root5, env={})
    assert "equal runs" in how5 and len({core[c] for m in t5 for c in m}) == 128 and all(len(m) == 32 for m in t5)
    t6, _ = cputopo.rank_cpu_table(8, list(range(16)), str(tmp_path / "none"), env={})
    assert t6 == [[2 * r, 2 * r + 1] for r in range(8)]
    assert cputopo.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_flat_layout_makes_every_ready_group_one_contiguous_slice():
    """The engine lays the flat gradient buffer out by completion group (distributed.flat_order): what becomes final together is ONE slice,
    reduced by one plain all_reduce (public API; round 3 used the private coalescing manager).  Any other layout still reduces correctly,
    with more collectives."""
    from gaviko_amd.distributed import flat_order, ready_layer
    names, numels = _trainable(CFG)
    size = dict(zip(names, numels))
    for share, lpb in ((1, 4), (2, 4), (1, 12), (1, 1)):
        cfg = dict(CFG, share_factor=share)
        names, numels = _trainable(cfg)
        size = dict(zip(names, numels))
        order = flat_order(names, share, lpb)
        assert sorted(order) == sorted(names)
        # stable inside a group: the gate parameters of one GPA module stay contiguous and in module order (the kernel writes them as one slice)
        for s in range(12 // share):
            mod = [n for n in names if f".prompt_projs.{s}." in "." + n]
            at = order.index(mod[0])
            assert order[at: at + len(mod)] == mod
        red = GradReducer(order, [size[n] for n in order], depth=12, share_factor=share, layers_per_bucket=lpb)
        flat = torch.zeros(sum(numels))
        groups = red.ready_groups()
        want_groups = len({ready_layer(n, share, lpb) for n in names if ready_layer(n, share, lpb) >= 0})       # the lowest one carries the unindexed tensors too
        assert len(groups) == want_groups
        for _, members in groups:
            pieces = [flat[red.ranges[i][1]: red.ranges[i][2]] for i in members]
            merged = GradReducer._merge_adjacent(pieces)
            assert len(merged) == 1 and merged[0].numel() == sum(p.numel() for p in pieces)
            assert merged[0].data_ptr() == min(p.data_ptr() for p in pieces)
    # the parameter order of the module tree is NOT contiguous per group (why the layout exists) -- and still merges what it can
    names, numels = _trainable(CFG)
    red = GradReducer(names, numels, depth=12, share_factor=1, layers_per_bucket=4)
    flat = torch.zeros(sum(numels))
    counts = [len(GradReducer._merge_adjacent([flat[red.ranges[i][1]: red.ranges[i][2]] for i in members])) for _, members in red.ready_groups()]
    assert max(counts) > 1
