"""GPU: the N > 1 path as far as one GPU allows.  Two FRESH processes (gloo, both on cuda:0) each run one data-parallel step of a small
GAViKO model on their shard with the event-ordered bucket reducer (distributed.GradReducer mode 'events': one backward plan, every bucket
all-reduced behind the event of the stream that finalises it).  The reduced flat gradient must equal the mean of the two shards'
single-process gradients BIT FOR BIT -- eagerly (torch events) and from the replayed launch plan (plan events)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent('''
    import os, sys
    import torch, torch.distributed as dist
    sys.path.insert(0, os.environ["GVK_ROOT"])
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, pool="cls", dim_head=64,
               dropout=0.0, emb_dropout=0.0, backbone="vit-t16", method="gaviko", num_prompts=8, prompt_latent_dim=20, local_dim=20,
               local_k=(3, 6, 6), DHW=(10, 10, 10), attn_drop=0.0, proj_drop=0.0, freeze_vit=True, share_factor=int(os.environ["GVK_SHARE"]), fp16=False)
    m = build_model(cfg)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    m.to(dev).train()
    B = 2

    def grads(first):
        for p in m.parameters():
            p.grad = None
        x = torch.from_numpy(synth.volumes(first, B)).to(dev); y = torch.from_numpy(synth.labels(first, B)).to(dev)
        torch.nn.functional.cross_entropy(m(x), y).backward()
        torch.cuda.synchronize()
        return m._engine().flat_grad.clone()

    # single-process gradients of BOTH shards (no reducer attached), several times: eager, eager, recorded, replayed
    single = [[grads(B * r) for _ in range(4)][-1] for r in range(world)]
    want = (single[0] + single[1]) * 0.5
    red = m.make_reducer(layers_per_bucket=4)                       # mode 'events'
    assert red.mode == "events" and red.world == 2
    kinds = {k for k in red.kinds}
    assert kinds == {"loc", "gpa", "main"}, kinds
    for it in range(5):                                              # eager x2, record, replay x2
        got = grads(B * rank)
        how = m._engine()._last_run[0]
        assert torch.equal(got, want), (it, how, (got - want).abs().max().item())
    assert how == "replayed", how
    dist.barrier()
    print("DP-OK", rank, how, flush=True)
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("share", [1, 2])
def test_two_process_event_ordered_reduction_bitwise(dev, tmp_path, share):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GVK_ROOT=ROOT, GVK_SHARE=str(share),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"DP-OK {rank} replayed" in out, out[-3000:]



RCCL_CHILD = textwrap.dedent('''
    import os, sys
    import torch, torch.distributed as dist
    sys.path.insert(0, os.environ["GVK_ROOT"])
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1)           # backend "nccl" IS RCCL on ROCm
        probe = torch.ones(4, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as e:                                                # no RCCL on this box: nothing to test
        print("RCCL-UNAVAILABLE", repr(e)[:200], flush=True)
        sys.exit(0)
    cfg = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, pool="cls", dim_head=64,
               dropout=0.0, emb_dropout=0.0, backbone="vit-t16", method="gaviko", num_prompts=8, prompt_latent_dim=20, local_dim=20,
               local_k=(3, 6, 6), DHW=(10, 10, 10), attn_drop=0.0, proj_drop=0.0, freeze_vit=True, share_factor=1, fp16=False)
    m = build_model(cfg)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    m.to(dev).train()
    x = torch.from_numpy(synth.volumes(0, 2)).to(dev); y = torch.from_numpy(synth.labels(0, 2)).to(dev)

    def grads():
        for p in m.parameters():
            p.grad = None
        torch.nn.functional.cross_entropy(m(x), y).backward()
        torch.cuda.synchronize()
        return m._engine().flat_grad.clone()

    want = [grads() for _ in range(4)][-1]
    red = m.make_reducer(layers_per_bucket=4)
    assert red.mode == "events" and red.world == 1 and red.active
    for it in range(5):                                                   # eager x2, record, replay x2: RCCL kernels behind the plan's events
        got = grads()
        assert torch.equal(got, want), (it, (got - want).abs().max().item())
    assert m._engine()._last_run[0] == "replayed"
    print("RCCL-OK", flush=True)
    dist.destroy_process_group()
''')


def test_rccl_collectives_behind_plan_events_on_one_rank(dev, tmp_path):
    """RCCL itself (backend "nccl") on the one GPU this box has: a single-rank process group with the collectives forced on
    (GAVIKO_DP_FORCE_COLLECTIVES=1: an all-reduce over one rank is the identity).  Every bucket's all-reduce is issued on the collective
    stream behind the event of the stream that finalises it, eagerly and from the replayed plan; the gradients must not change by a bit."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GVK_ROOT=ROOT, GAVIKO_DP_FORCE_COLLECTIVES="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", RCCL_CHILD], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=420)
    if "RCCL-UNAVAILABLE" in p.stdout:
        pytest.skip("RCCL could not be initialised on this box: " + p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0 and "RCCL-OK" in p.stdout, p.stdout[-3000:]


def test_bench_bare_two_ranks_rehearsal(dev, tmp_path):
    """The driver's multi-GPU command form, `python bench.py --gpus 2 ...` with no launcher: bench.py starts both ranks itself.  On a
    one-GPU box the ranks share cuda:0 over gloo (GAVIKO_BENCH_REHEARSAL=1); the line must report two ranks and a sane rate."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GAVIKO_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["value"] > 0 and line["scaling"] == "weak"
