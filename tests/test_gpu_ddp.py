"""-m gpu: the data-parallel path ON A HIP DEVICE (TaskPrompter/main.py:92-94, InvPT/main.py:87-89).

  * two processes share the one GPU of the box, process group = gloo on DEVICE tensors (RCCL refuses two ranks on one device), the
    model goes through the REAL `nn.SyncBatchNorm.convert_sync_batchnorm` and `DistributedDataParallel(device_ids=[0])`, each rank
    runs the HIP kernels on its (UNEQUAL) share of the images; the gradients must equal those of a single process running the joint
    batch through the same kernels, and after a FusedClipAdam step the replicas must be bitwise identical;
  * `torchrun --nproc-per-node 1` of bench.py with the nccl backend: RCCL communicator initialisation + DDP's bucketed all-reduce over
    RCCL on a HIP device (world size 1 — the box has one GPU; the 8-GPU curve is the driver's)."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHARES = [[0, 1, 2], [3]]            # 3 + 1 images: unequal per-rank batches


def _loss(out, rows, n_total, world, seed=7):
    g = torch.Generator().manual_seed(seed)
    tot = 0.0
    for k in sorted(out):
        r = torch.randn((n_total,) + tuple(out[k].shape[1:]), generator=g)[rows].to(out[k].device)
        tot = tot + (out[k] * r).sum() / (n_total * out[k][0].numel()) * world
    return tot


def _build(device):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    from oracle import configs, weights
    cfg = configs.taskprompter("mini_ctr")
    model = conftest.build_product_model(cfg, "x3")
    contract = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(weights.synth_state_dict(contract, 0), strict=True)
    n_total = sum(len(s) for s in SHARES)
    x = weights.synth_images(n_total, cfg["img_size"], 2)
    return model.to(device).train(), x.to(device), n_total


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        model, x, n_total = _build(dev)
        import mtt_amd
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)                        # TaskPrompter/main.py:92
        n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules())
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=False)   # main.py:94
        rows = SHARES[rank]
        out = ddp(x[rows])
        _loss(out, rows, n_total, world).backward()
        torch.cuda.synchronize()
        grads = {k: (v.grad.detach().cpu().clone() if v.grad is not None else None) for k, v in model.named_parameters()}
        opt = mtt_amd.optim.FusedClipAdam([p_ for p_ in model.parameters() if p_.grad is not None], lr=1e-3, weight_decay=1e-6, max_norm=0.5)
        norm = opt.step()
        torch.cuda.synchronize()
        chk = torch.cat([p_.detach().double().flatten() for p_ in model.parameters()] + [norm.double().flatten()]).cpu()
        seen = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(seen, chk)                                                           # host tensors
        same = all(bool(torch.equal(seen[0], s_)) for s_ in seen)
        if rank == 0:
            q.put(("ok", {k: (v.numpy() if v is not None else None) for k, v in grads.items()}, n_sync, same))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put(("error", f"rank {rank}: {e!r}\n{traceback.format_exc()}", 0, False))
        raise


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_ddp_syncbn_two_ranks_on_one_gpu_match_single_process():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = len(SHARES)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    status, got, n_sync, same = q.get(timeout=600)
    assert status == "ok", got
    for p_ in procs:
        p_.join(timeout=300)
        assert p_.exitcode == 0
    assert n_sync > 0, "convert_sync_batchnorm found no BatchNorm holders"
    assert same, "replicas diverged after FusedClipAdam.step"
    # single process, joint batch, same kernels
    model, x, n_total = _build(torch.device("cuda", 0))
    out = model(x)
    _loss(out, list(range(n_total)), n_total, 1).backward()
    torch.cuda.synchronize()
    worst, n = 0.0, 0
    for k, v in model.named_parameters():
        if v.grad is None or float(v.grad.norm()) < 1e-6:
            continue
        assert got[k] is not None, k
        e = float((torch.from_numpy(got[k]).double() - v.grad.double().cpu()).norm() / v.grad.double().norm().cpu())
        worst, n = max(worst, e), n + 1
    assert n > 50 and worst < 2e-4, (n, worst)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_under_torchrun_rccl_single_rank():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-ref-batch", "--no-parity", "--no-roofline", "--no-torch-baseline", "--no-fast-mode"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    rec = json.loads(lines[-1])
    assert rec["config"]["rccl_ranks"] == 1 and rec["value"] > 0, rec


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_ddp_buckets_are_handed_to_rccl_while_backward_is_still_running():
    """TaskPrompter/main.py:94 relies on DDP overlapping the gradient all-reduce with backward.  The hand-written backward must therefore
    deliver parameter gradients PROGRESSIVELY (heads -> decoder -> blocks 23..0), not in one piece at the end: with an RCCL process group
    (world size 1) and a comm hook that timestamps every bucket on the backward's stream (tools/ddp_overlap.py), all buckets but the
    last ones are ready — i.e. their all-reduce is launched on ProcessGroupNCCL's stream — well before the backward's GPU work ends."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MASTER_PORT=str(29300 + (os.getpid() % 90)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ddp_overlap.py"), "--batch", "4", "--bucket-mb", "100", "--json"],
                       capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    s = json.loads(lines[-1])
    n = s["n_buckets"]
    assert n >= 8, s                                               # 1.6 GB of fp32 gradients in 100 MB buckets
    assert s["buckets_ready_before_97pct_of_backward"] >= n - 2, s
    assert s["first_bucket_ready_frac"] < 0.35, s                  # the heads' / decoder's gradients leave long before the encoder's backward ends
    fr = [b["ready_gpu_frac"] for b in s["buckets"]]
    assert fr == sorted(fr), fr                                    # buckets arrive in order along the backward
