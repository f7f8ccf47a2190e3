"""CPU, world_size 2, gloo: the data-parallel path (DistributedDataParallel gradient all-reduce + cross-rank
BatchNorm statistics through SyncBatchNorm holders) on the ABI emulator.  Two ranks with one image each must
produce the gradients of a single process running both images (same global batch), as the reference's
DDP + SyncBatchNorm training does (TaskPrompter/main.py:92-94)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import conftest
    import mtt_amd
    from oracle import abi_emul, configs, weights
    from tests.golden.make_golden import loss_of
    mtt_amd.ops.call = abi_emul.call
    cfg = configs.taskprompter("mini_ctr")
    meta, _ = conftest.load_golden("mini_ctr")
    model = conftest.build_product_model(cfg, "x3")
    model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=True)
    model.train()
    for m in model.modules():          # DDP refuses nn.SyncBatchNorm on CPU modules: flag the holders instead (same code path)
        if isinstance(m, torch.nn.BatchNorm2d):
            m._mtt_sync = True
    ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=False)
    x = weights.synth_images(2, cfg["img_size"], 2)[rank:rank + 1]
    out = ddp(x)
    # per-rank loss on its image with the global-batch random weights -> DDP averages gradients over ranks
    g = torch.Generator().manual_seed(7)
    tot = 0.0
    for k in sorted(out):
        r = torch.randn((2,) + tuple(out[k].shape[1:]), generator=g)[rank:rank + 1]
        tot = tot + (out[k] * r).sum() / (2 * out[k][0].numel()) * world      # mean over the global batch, undo DDP's 1/world
    tot.backward()
    grads = {k: v.grad.clone() for k, v in model.named_parameters()}
    # the fused clip + Adam step consumes DDP's (bucket-view) gradients: parameters must stay identical on every rank
    opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-6, max_norm=0.5)
    norm = opt.step()
    chk = torch.stack([p_.detach().double().sum() for p_ in model.parameters()]).sum() + norm.double().sum()
    seen = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(seen, chk)
    assert all(bool(torch.equal(seen[0], s_)) for s_ in seen), seen
    if rank == 0:
        q.put({k: v.numpy() for k, v in grads.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_match_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import train_check
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    got = q.get(timeout=240)
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    # single-process oracle gradients on the 2-image batch (train_check uses the same inputs / loss)
    import conftest
    from oracle import configs, taskprompter_oracle as tpo, weights
    from tests.golden.make_golden import loss_of
    cfg = configs.taskprompter("mini_ctr")
    meta, _ = conftest.load_golden("mini_ctr")
    sd = weights.synth_state_dict(meta["contract"], 0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    x = weights.synth_images(2, cfg["img_size"], 2)
    loss_of(tpo.forward(dict(sd, **params), cfg, x, training=True)).backward()
    worst = 0.0
    for k, ref in params.items():
        if ref.grad is None or float(ref.grad.norm()) < 1e-6:
            continue
        e = float((torch.from_numpy(got[k]) - ref.grad).norm() / ref.grad.norm())
        worst = max(worst, e)
    assert worst < 1e-3, worst
