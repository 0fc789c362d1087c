#!/usr/bin/env python3
"""Headline benchmark: MRI volumes/sec (fwd+bwd), ViT-B/16 --method gaviko, synthetic 120x160x160 volumes.

  python bench.py --gpus N --steps K --warmup W            (any N: for N > 1 without a launcher's RANK / WORLD_SIZE in the environment
                                                            this process starts the N ranks itself -- see spawn_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch: forward, loss seed, backward of every trainable tensor (frozen ViT,
trainable prompts + MWSA + GPA + head), and for N>1 the mean all-reduce of the flat gradient buffer.  The optimizer step
is outside the metric (BASELINE.json: fwd+bwd).  Inputs are resident in HBM before the timed region starts.
Weak scaling: 4 volumes per GPU (BASELINE.json configs[1]).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel (the bf16 MFMA GEMM class with the largest total time), timed per launch by HIP event
                  pairs on its launch stream inside a second, instrumented copy of the step's launch plans (same
                  three-stream schedule as the timed region).  `achieved` uses the RAW event-pair time (it agrees with the
                  rocprofv3 kernel duration of the same launches) and the ALGORITHMIC flops (padding columns of the
                  K-concatenated operand excluded); the cost of an empty event pair is reported as `event_pair_us`, not subtracted
  cpu_baseline -- the oracle (CPU restatement of the reference, fp32 torch) timed on this host's cores on one batch.
"""
from __future__ import annotations

import argparse
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The HIP runtime reads GPU_MAX_HW_QUEUES when it starts (first HIP call of the process: torch.cuda.set_device, init_process_group).  A step
# runs on three streams, plus the collective stream and RCCL's own at N > 1: with the default of 4 hardware queues two of them alias and
# serialise (5.9 -> 6.9-7.7 ms).  So the variable is set HERE, before torch is even imported, and main() refuses to
# run if HIP was somehow up before this line took effect (_queues_set_before_hip).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver

import numpy as np  # noqa: E402
import torch  # noqa: E402

_queues_set_before_hip = not torch.cuda.is_initialized()       # torch imported just now: False only if something initialised HIP at import

_COMMON = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, pool="cls", dim_head=64,
               dropout=0.1, emb_dropout=0.1, freeze_vit=True, fp16=False)
# the `model:` blocks of the reference's shipped configs (src/configs/*.yaml), one per BASELINE.json configuration
METHODS = {
    "gaviko": dict(_COMMON, method="gaviko", num_prompts=32, prompt_latent_dim=20, local_dim=20, local_k=(6, 6, 6), DHW=(10, 10, 10),
                   attn_drop=0.2, proj_drop=0.2, share_factor=1),                                  # gaviko.yaml:13-39   (cfg2, cfg5)
    "deep_vpt": dict(_COMMON, method="deep_vpt", num_prompts=8, prompt_dropout=0.1, prompt_dim=64, deep_prompt=True),   # vpt.yaml:13-34      (cfg3)
    "adaptformer": dict(_COMMON, method="adaptformer"),                                            # adaptformer.yaml    (cfg4)
    "melo": dict(_COMMON, method="melo", r=4, alpha=4, lora_layer=None),                           # melo.yaml:13-34     (cfg4)
}
MODEL = METHODS["gaviko"]
DEFAULT_BATCH = {"gaviko": 4, "deep_vpt": 4, "adaptformer": 8, "melo": 8}          # volumes per GPU (BASELINE.json configs)
PEAK_TFLOPS = {"bf16": 2516.0, "fp32": 157.3}   # dense MFMA peaks of MI355X: 256 CU x 2.4 GHz x 4096 (bf16) / 256 (fp32 in, fp32 acc) flop/clk/CU
PEAK_BF16_TFLOPS = PEAK_TFLOPS["bf16"]
# fwd + bwd GF per volume (2.M.N.K over GEMM-like ops, frozen linears dgrad x1, attention x2, trainable x2)
GF_PER_VOLUME = {("gaviko", "vit-b16"): 482.44, ("gaviko", "vit-l16"): 1590.06, ("deep_vpt", "vit-b16"): 460.13,
                 ("adaptformer", "vit-b16"): 462.70, ("melo", "vit-b16"): 456.50}


def build(backbone, device, method="gaviko", precision="bf16"):
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    m = build_model(dict(METHODS[method], backbone=backbone, precision=precision))
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    m.to(device)
    m.train()
    return m


def host_cores():
    """Cores this process may actually use: the scheduler affinity mask, cut down to the cgroup CPU quota when there is one (a GPU box
    hands a one-GPU job a share of the host; more threads than that quota only oversubscribe it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    how = f"affinity {n}"
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / 100000.0)):
        try:
            with open(path) as f:
                q = parse(f.read())
            if q is not None and q >= 1 and q < n:
                n = int(q + 0.5)
                how += f", cgroup quota {q:.1f}"
            break
        except (OSError, ValueError, IndexError):
            continue
    if os.environ.get("GAVIKO_BENCH_CPU_THREADS"):
        n = int(os.environ["GAVIKO_BENCH_CPU_THREADS"])
        how += ", GAVIKO_BENCH_CPU_THREADS"
    return max(1, n), how


def cpu_baseline(backbone, batch, method="gaviko"):
    """Oracle on the host cores: fwd + CE + bwd steps of the same batch, every core this process may use, one
    warm-up step, best of three timed ones (dropouts off: the oracle has none)."""
    import oracle
    from gaviko_amd.utils import synth
    avail, how = host_cores()
    torch.set_num_threads(avail)
    cfg = dict(METHODS[method], backbone=backbone)
    sd = {k: torch.from_numpy(v).requires_grad_(oracle.trainable(method, k, cfg)) for k, v in synth.fill_state_dict(oracle.SHAPES[method](cfg)).items()}
    x = torch.from_numpy(synth.volumes(0, batch))
    y = torch.from_numpy(synth.labels(0, batch))
    times = []
    for it in range(4):                                      # 0 = warm-up (allocator, thread pool), then best of 3
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        loss = torch.nn.functional.cross_entropy(oracle.FORWARD[method](sd, x, cfg), y)
        loss.backward()
        times.append(time.perf_counter() - t0)
        print(f"[cpu_baseline] step {it}: {times[-1]:.1f} s on {avail} threads", file=sys.stderr, flush=True)
        if sum(times) > 90.0:                                # keep the default run within minutes whatever the host is
            break
    timed = times[1:] or times
    dt = min(timed)
    return {"value": batch / dt, "unit": "volumes/s", "cores": avail, "kind": "port",
            "sample": f"fwd + CE + bwd of the same {batch}-volume batch, fp32 torch oracle on {avail} threads ({how}): warm-up step "
                      f"{times[0]:.1f} s, best of {len(times) - 1} timed step(s) ({', '.join(f'{t:.1f}' for t in times[1:])} s)"}


# newest committed PMC summary first; each records the hash of the GEMM sources AS THE PRODUCT BUILD COMPILES THEM (comments and
# `#ifdef GVK_DIAG` text excluded: gaviko_amd/utils/srchash.py) and is ignored when that differs from this tree's
PMC_TRAFFIC_FILE = next((f for f in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f))),
                        "r05_pmc_traffic.json")


def gemm_source_hash():
    from gaviko_amd.utils.srchash import gemm_source_hash as h
    return h()


def gemm_alg_bytes(epi: int, shape):
    """Algorithmic HBM bytes of one GEMM launch: bf16 operands in + (aux in, out) per epilogue id -- 0 bf16 out; 2 bias+gelu ->
    pre-act + act bf16 out; 4 pre-act in, bf16 out; 1 fp32 residual read-modify-write; 5 fp32 out; 6 residual + bf16 copy."""
    M, N, K = shape
    return 2 * (M * K + N * K) + M * N * {0: 2, 1: 8, 2: 4, 4: 4, 5: 4, 6: 10}.get(epi, 4)


def gemm_kernel_epilogue(kernel_name: str):
    """Epilogue id of a GEMM kernel instantiation as rocprofv3 prints it, or None for other kernels."""
    m = re.match(r"gemm_nt_kernel<\d+, \d+, (\d+)(, (\d+|true|false))*>", kernel_name) or re.match(r"gemm8p_kernel<(\d+), \d+>", kernel_name)
    return int(m.group(1)) if m else None


def pmc_traffic(name, stats, path=None):
    """HBM-side bytes per launch of the dominant GEMM from the committed PMC passes (profiles/r0N_pmc_traffic.json, produced
    by tools/pmc_traffic.py from two `rocprofv3 --pmc` runs of this same command: FETCH_SIZE and WRITE_SIZE, KiB units, reads
    x2 on gfx950).  PMC counters cannot be read from inside the process, so the figure is the last profiled one.  The profiler sees
    kernel instantiations, not shapes, so tools/pmc_traffic.py records with every traffic cluster the [M, N, K] it was measured on
    (from `bench.py --dump-gemm-shapes` of the profiled command); a figure is attached ONLY when exactly one cluster of this
    kernel's instantiation carries this run's shape -- a PMC file taken at another workload (other batch / backbone), or one
    without recorded shapes, yields `traffic_note` and no number."""
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", PMC_TRAFFIC_FILE)
    m = re.match(r"gemm_nt_bf16\[(\w+)\]", name)
    if not (m and os.path.exists(path)):
        return {}
    from gaviko_amd.engine_common import _EPI_NAMES
    epi = {v: k for k, v in _EPI_NAMES.items()}[m.group(1)]
    with open(path) as f:
        doc = json.load(f)
    base = os.path.basename(path)
    if doc.get("gemm_source_sha") != gemm_source_hash():     # measured on other code: no figure rather than a stale one
        return {"traffic_note": f"{base} was measured on GEMM sources {doc.get('gemm_source_sha')}, this build is {gemm_source_hash()}: traffic withheld"}
    shape = list(stats[name]["shape"])
    cl = [c for k, v in doc["kernels"].items() if gemm_kernel_epilogue(k) == epi for c in (v.get("clusters") or [v])]
    if not any("shape" in c for c in cl):
        return {"traffic_note": f"{base} records no [M, N, K] with its traffic clusters (taken before round 5): traffic withheld"}
    hits = [c for c in cl if c.get("shape") == shape]
    if len(hits) != 1:
        return {"traffic_note": f"{base} holds {len(hits)} cluster(s) measured on M, N, K = {shape} for epilogue '{m.group(1)}' "
                                f"(its workload: {doc.get('workload')}): traffic withheld"}
    return {"traffic": hits[0]["total_bytes"], "traffic_source": f"profiles/{base} (2*FETCH_SIZE + WRITE_SIZE, KiB; GEMM sources {doc.get('gemm_source_sha')}; "
                                                                 f"cluster measured on M, N, K = {shape})",
            "algorithmic_bytes": gemm_alg_bytes(epi, shape)}


def rank_cpu_set(local_rank: int, local_world: int, cpus=None, sys_root="/sys"):
    """The share of this process's allowed CPUs that rank `local_rank` of `local_world` keeps (gaviko_amd/utils/cputopo.py): whole
    PHYSICAL cores -- a core's hardware threads never go to two ranks -- on the NUMA node of the rank's GPU, the node's cores split
    evenly among the ranks whose GPUs hang off it; read from sysfs (thread_siblings_list, node*/cpulist, the KFD topology and the
    GPU's PCI numa_node), all before any HIP call.  Where sysfs has nothing it falls back to equal contiguous runs.  N ranks issuing
    ~360 launches per step from one host would otherwise migrate over each other's cores -- or, cut by logical id, share them as
    hyper-thread siblings."""
    from gaviko_amd.utils import cputopo
    table, _ = cputopo.rank_cpu_table(local_world, cpus, sys_root)
    return table[local_rank % len(table)]


def pin_rank_cpus(local_rank: int, local_world: int):
    """Called before any GPU call of a rank (the runtime's helper threads inherit the mask).  Returns the CPU list, or None when the
    platform has no affinity call or there is a single rank (nothing to separate)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity") or os.environ.get("GAVIKO_BENCH_NO_PIN") == "1":
        return None
    from gaviko_amd.utils import cputopo
    try:
        table, how = cputopo.rank_cpu_table(local_world)
        mine = table[local_rank % len(table)]
        os.sched_setaffinity(0, mine)
    except Exception as e:                                   # an unreadable / unexpected sysfs must never cost the run: equal runs of logical ids
        cpus = sorted(os.sched_getaffinity(0))
        k = max(1, len(cpus) // local_world)
        mine, how = cpus[(local_rank * k) % len(cpus):][:k] or cpus[:1], f"logical ids (topology lookup failed: {type(e).__name__}: {e})"
        os.sched_setaffinity(0, mine)
    print(f"bench.py: rank {local_rank}/{local_world} pinned to {len(mine)} CPU(s) {cputopo_ranges(mine)} [{how}]", file=sys.stderr, flush=True)
    return mine


def cputopo_ranges(cpus):
    """[0, 1, 2, 3, 128, 129] -> '0-3,128-129'"""
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no launcher: start the N ranks as FRESH child processes (never a re-exec) -- this parent has not
    touched the GPU (importing torch does not initialise HIP) and never will.  Every child gets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT like torch.distributed.run would set them and runs this same file; rank 0's stdout goes to a temporary
    file (no pipe that could fill up and block it) and is relayed at the end, stderr of every rank passes through.  A wall-clock limit
    (GAVIKO_BENCH_TIMEOUT seconds, default 1500) ends the exact children started here if a rank hangs in a collective.  Returns the
    exit code (first failing rank's, 124 on the limit, else 0)."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    limit = float(os.environ.get("GAVIKO_BENCH_TIMEOUT", "1500"))
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GAVIKO_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc, t0 = 0, time.monotonic()
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in sorted(pending):
                        procs[q].terminate()                      # the exact child processes started above
            if pending and time.monotonic() - t0 > limit:
                print(f"bench.py: ranks {sorted(pending)} still running after {limit:.0f} s; stopping them", file=sys.stderr, flush=True)
                rc = rc or 124
                for q in sorted(pending):
                    procs[q].terminate()
                limit = float("inf")
            if pending:
                time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    out0.close()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="volumes per GPU (default: the BASELINE configuration's: 4, or 8 for adaptformer / melo)")
    ap.add_argument("--backbone", default="vit-b16")
    ap.add_argument("--method", default="gaviko", choices=sorted(METHODS),
                    help="which --method of train.py:111-153 to time: gaviko = the headline (cfg2 / cfg5), deep_vpt = cfg3, adaptformer / melo = cfg4")
    ap.add_argument("--precision", default=None, choices=["bf16", "fp32"],
                    help="operand precision (default bf16; adaptformer / melo default to fp32, BASELINE cfg4's precision)")
    ap.add_argument("--loss", default="ce", choices=["ce", "focal", "ce-torch"])   # ce-torch: torch's own op, for A/B only
    ap.add_argument("--zero-grad", default="none", choices=["none", "flat"],
                    help="none (default): optimizer.zero_grad() as train.py:296 calls it (set_to_none=True); flat: zero_grad(set_to_none=False), one "
                         "memset of the flat gradient buffer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-gemm-shapes", default=None, metavar="FILE",
                    help="write {workload, classes: {GEMM class: [M, N, K]}} of this run's instrumented pass to FILE (tools/pmc_traffic.py attaches "
                         "the shapes to the traffic clusters of a PMC pass of the same command)")
    ap.add_argument("--launch-check", action="store_true", help="every rank reports its launcher environment and exits before any GPU call (tests)")
    ap.add_argument("--allow-diag", "--allow-ablate", dest="allow_diag", action="store_true",
                    help="diagnostics only: run on the measurement build (GAVIKO_HIP_DIAG=1: A/B switches and timing ablations live there); the "
                         "output line is marked INVALID")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = DEFAULT_BATCH[args.method]
    if args.precision is None:
        args.precision = "fp32" if args.method in ("adaptformer", "melo") else "bf16"
    if os.environ.get("GAVIKO_HIP_DIAG", "0") == "1" and not args.allow_diag:
        raise SystemExit("bench.py: GAVIKO_HIP_DIAG=1 selects the measurement build (A/B switches, timing ablations that compute WRONG results): "
                         "not a benchmark; unset it (or pass --allow-diag: the line is then marked INVALID)")

    if args.gpus > 1 and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        raise SystemExit(spawn_ranks(args.gpus))              # bare `python bench.py --gpus N`: the ranks are started here, before any GPU call
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one disjoint CPU set per rank, before any GPU call (LOCAL_WORLD_SIZE is set by torch.distributed.run and by spawn_ranks)
    cpus = pin_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if args.launch_check:
        rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GAVIKO_BENCH_CHILD",
                                              "GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY")}
        rec["gpu_initialised"] = bool(torch.cuda.is_initialized())
        rec["queues_set_before_hip"] = bool(_queues_set_before_hip)
        rec["cpus"] = cpus
        rec["cpu_ranges"] = cputopo_ranges(cpus) if cpus else None
        path = os.environ.get("GAVIKO_BENCH_LAUNCH_LOG")
        if path:
            with open(f"{path}.{rank}", "w") as f:
                json.dump(rec, f)
        if rank == 0:
            print(json.dumps({"launch_check": rec, "n_gpus": world}))
        return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    # Rehearsal switches (not used by the driver): GAVIKO_BENCH_REHEARSAL=1 runs every rank on cuda:0 over gloo, so that the
    # whole multi-rank flow (bucketed backward, all-reduce, instrumented pass) can be exercised on a one-GPU box.
    rehearsal = os.environ.get("GAVIKO_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if not _queues_set_before_hip or torch.cuda.is_initialized():
        raise SystemExit("bench.py: the HIP runtime was initialised before GPU_MAX_HW_QUEUES could take effect (something imported ahead of "
                         "bench.py touched the GPU): the step's streams would share hardware queues and the line would not be the product's")
    from gaviko_amd import lib as L                    # (sets nothing any more: the variable is already in the environment)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from gaviko_amd import engine as eng_mod
    from gaviko_amd.utils import synth
    model = build(args.backbone, dev, args.method, args.precision)
    if world > 1:
        model.make_reducer()
    B = args.batch
    lo = rank * B                                    # shard of the global batch owned by this rank
    eng = model._engine()
    x = eng.input_buffer(B, dev, train=True)          # inputs resident in HBM, in the engine's own input slot: no per-step copy-in launch
    x.copy_(torch.from_numpy(synth.volumes(lo, B)))
    y = torch.from_numpy(synth.labels(lo, B)).to(dev)
    eng.static_io = True                              # the loss kernel reads the logits in place (no clone launch inside the step)
    # the loss seed is part of the step: one fused launch (loss + dlogits + the running loss / accuracy sums of train.py:327-328)
    from gaviko_amd.losses import CrossEntropyLoss, FocalLoss, StepMeter
    criterion = (FocalLoss(gamma=1.2) if args.loss == "focal" else CrossEntropyLoss()).attach_meter(StepMeter(dev))
    if args.loss == "ce-torch":
        criterion = torch.nn.functional.cross_entropy

    # The step opens the way train.py:296 does: `optimizer.zero_grad()` of the optimizer the loop would own (gaviko_amd.optim.FusedAdamOneCycle,
    # never stepped here: the optimizer step is outside the fwd+bwd metric).  Its default is torch's (set_to_none=True: every .grad dropped,
    # the backward re-attaches ~300 views, ~0.6 ms of host time); `--zero-grad flat` times zero_grad(set_to_none=False) instead, which is ONE
    # memset of the flat gradient buffer and lets the backward skip that bookkeeping.  DESIGN.md 7e quotes both.
    from gaviko_amd.optim import FusedAdamOneCycle
    optimizer = FusedAdamOneCycle(model, lr=3e-4, eps=1e-8)
    to_none = args.zero_grad == "none"

    def step():
        optimizer.zero_grad(set_to_none=to_none)
        loss = criterion(model(x), y)
        loss.backward()
        return loss

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    vps = world * B * args.steps / dt

    bbname = {"vit-b16": "ViT-B/16", "vit-l16": "ViT-L/16", "vit-t16": "ViT-T/16"}.get(args.backbone, args.backbone)
    eng0 = model._engine()
    what = {"gaviko": "frozen ViT; prompts+MWSA+GPA+head train; attn_drop=proj_drop=0.2 live",
            "deep_vpt": "frozen ViT, its transformer and dropouts in eval (vpt.py:106-115); deep prompts + prompt_proj + head train; prompt_dropout 0.1 live",
            "adaptformer": "frozen ViT; adapters + head train",
            "melo": "frozen ViT; LoRA (r=4) on q and v + head train; backbone dropout 0.1 live (no train() override)"}[args.method]
    arith = "bf16 MFMA operands / fp32 accumulate" if args.precision == "bf16" else "exact fp32 (f32-input MFMA, fp32 flash attention)"
    out = {"metric": f"MRI volumes/sec (fwd+bwd) {bbname} {args.method}, 120x160x160",
           "value": round(vps, 3), "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
           "config": {"workload": f"{args.backbone} --method {args.method} {arith}, batch={B}/GPU, fwd + "
                                  f"{'CrossEntropy' if args.loss != 'focal' else 'Focal(1.2)'} + bwd ({what}), "
                                  f"grads all-reduced over {world} rank(s)",
                      "global_batch": world * B, "tokens": eng0.T, "parallelism": f"dp{world}",
                      "zero_grad": "optimizer.zero_grad() [set_to_none=True]" if to_none else "optimizer.zero_grad(set_to_none=False) [one memset]"}}
    if os.environ.get("GAVIKO_HIP_DIAG", "0") == "1":
        out["INVALID_measurement_build"] = {k: v for k, v in os.environ.items() if k.startswith("GAVIKO_HIP_")}
    peak = PEAK_TFLOPS[args.precision]
    gf = GF_PER_VOLUME.get((args.method, args.backbone))
    if gf:
        out["mfma_roofline_frac_whole_step"] = round(vps / world * gf / 1e3 / peak, 4)
    if eng0.prune_dead_rows:
        # Rows nobody reads are not computed (engine.py: prune_dead_rows; results bit-identical, tests/test_model_gpu.py::test_pruned_rows_are_dead):
        # the last layer's MLP forward + backward on the rows the head pools, the first layer's qkv dgrad + LayerNorm-1 backward on the
        # prompt rows.  The flops those rows would have cost are part of BASELINE's algorithmic count but are NOT executed: the fraction of
        # the roofline by executed flops is reported beside the algorithmic one, and the same K steps are timed once more with every row
        # computed (`value_all_rows`) so that both rates come from this run.
        Tt, Cc, mlp_, dep = eng0.T, eng0.C, eng0.mlp, eng0.depth
        skipped = 2.0 * (Tt - 64) * Cc * (4 * mlp_ + 3 * Cc) / 1e9      # fc1 + fc2 fwd, fc2 + fc1 dgrad (last layer); qkv dgrad (first layer), 64-row panels kept
        out["dead_row_pruning"] = {"on": True, "gf_not_executed_per_volume": round(skipped, 2)}
        if gf:
            out["dead_row_pruning"]["mfma_roofline_frac_executed_flops"] = round(vps / world * (gf - skipped) / 1e3 / peak, 4)
        eng0.set_prune(False)
        for _ in range(max(3, min(args.warmup, 6))):
            step()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt1], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt1 = t.item()
        out["dead_row_pruning"]["value_all_rows"] = round(world * B * args.steps / dt1, 3)
        out["dead_row_pruning"]["ms_per_step_all_rows"] = round(1e3 * dt1 / args.steps, 4)
        eng0.set_prune(True)
        for _ in range(3):
            step()
        sync()

    if not args.no_roofline:
        # (every rank runs this pass -- its steps contain the gradient all-reduce -- but only rank 0 reports)
        # Second, instrumented pass: the step is re-recorded as launch plans in which every GEMM launch (and the patch-embed
        # stage) is bracketed by timestamped HIP events on its launch stream, then replayed like the timed region -- same
        # three-stream schedule, same neighbours on the other queues.  An empty event pair per plan calibrates the bracket's
        # own cost, which is subtracted.
        L.load().gvk_plan_set_timing(1)
        eng_mod.GEMM_MARKS = {}
        eng._graphs.clear()
        step()                                              # records (and runs) the instrumented plans
        torch.cuda.synchronize(dev)
        acc = {}
        for _ in range(min(args.steps, 10)):
            step()
            torch.cuda.synchronize(dev)
            eng.collect_gemm_marks(acc)
        eng_mod.GEMM_MARKS = None
        L.load().gvk_plan_set_timing(0)
        eng._graphs.clear()
        stats = {k: {"avg_ms": sum(v["ms"]) / len(v["ms"]), "total_ms": sum(v["ms"]), "n": len(v["ms"]), "flops_per_launch": v["flops"],
                     "shape": v["shape"], "bytes": v["bytes"], "overhead_ms": v["overhead_ms"]} for k, v in acc.items()}
        pe = stats.pop("__patch_embed__", None)
        if rank != 0:
            pe, stats = None, {}
        if pe:
            # HBM-bound stage on the bf16 path: algorithmic bytes = fp32 volume in + fp32 token rows out (global + local stream), per launch.
            # On the exact-fp32 path the same stage is bound by the fp32 matrix rate (1/16 of bf16): labelled and priced against that peak.
            hbm = {"achieved": round(pe["bytes"] / (pe["avg_ms"] * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                   "frac": round(pe["bytes"] / (pe["avg_ms"] * 1e-3) / 8e12, 4)}
            if args.precision == "fp32":
                pflop = 2.0 * B * 1000 * 3072 * eng0.C
                ptf = pflop / (pe["avg_ms"] * 1e-3) / 1e12
                out["patch_embed"] = {"bound": "mfma_f32", "achieved": round(ptf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ptf / peak, 4),
                                      "avg_us": round(pe["avg_ms"] * 1e3, 1), "flop_per_launch": pflop, "algorithmic_bytes": pe["bytes"], "hbm_side": hbm,
                                      "note": "im2col(fp32) + f32-input MFMA GEMM with fused bias/pos/scatter epilogue"}
            else:
                out["patch_embed"] = {"bound": "hbm", **hbm, "avg_us": round(pe["avg_ms"] * 1e3, 1),
                                      "algorithmic_bytes": pe["bytes"], "note": "im2col(bf16) + MFMA GEMM with fused bias/pos/scatter epilogue"}
        if stats:
            name, s = max(stats.items(), key=lambda kv: kv[1]["total_ms"])
            raw_us, pair_us = s["avg_ms"] * 1e3, s["overhead_ms"] * 1e3
            tf = s["flops_per_launch"] / (s["avg_ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(tf / peak, 4), "traffic": None, "avg_launch_us": round(raw_us, 2), "launches": s["n"],
                               "flop_per_launch": s["flops_per_launch"], "shape": s["shape"], "event_pair_us": round(pair_us, 2),
                               "avg_launch_us_minus_event_pair": round(max(raw_us - pair_us, 0.0), 2),
                               "timing": "RAW HIP event pairs on the launch stream inside the replayed plan (nothing subtracted; an empty "
                                         "pair costs event_pair_us); flop_per_launch = 2*M*N*K_algorithmic (padding columns of the "
                                         "K-concatenated fc2 operand excluded); cross-check: profiles/r05_kernel_stats_by_shape.csv"}
            if args.method == "gaviko" and args.precision == "bf16":
                out["roofline"].update(pmc_traffic(name, stats))
            if args.dump_gemm_shapes:
                with open(args.dump_gemm_shapes, "w") as f:
                    json.dump({"workload": {"backbone": args.backbone, "method": args.method, "batch": B, "precision": args.precision},
                               "classes": {k: v["shape"] for k, v in stats.items()}}, f, indent=1)
            out["gemm_classes"] = {k: {"avg_us": round(v["avg_ms"] * 1e3, 2), "n": v["n"], "tflops": round(v["flops_per_launch"] / (v["avg_ms"] * 1e-3) / 1e12, 1)}
                                   for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.backbone, B, args.method)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
