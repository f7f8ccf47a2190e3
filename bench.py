"""Headline benchmark: training images/sec of the TaskPrompter ViT-L hot path, 512x512, 6 tasks (BASELINE.json),
on N MI355X of one node (one process per GPU; RCCL all-reduce of gradients over xGMI via DistributedDataParallel).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = forward + MultiTaskLoss + backward (+ gradient all-reduce) + clip_grad_norm_ + Adam step on one synthetic batch
that is already resident in HBM (TaskPrompter/utils/train_utils.py:32-51).  Weak scaling: the per-GPU batch is fixed.
Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — the dominant kernel (`gemm_fast256_kernel`, the 256x256x64 bf16 MFMA GEMM): algorithmic FLOPs / HIP-event time,
                 measured in one extra instrumented step right after the timed region (keeps event overhead out of `value`)
  cpu_baseline — the CPU oracle (restatement of the reference, `kind: "port"`) timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GFLOP_FWD_PER_IMG = 1046.8          # NS-6 algorithmic forward GFLOP / image (SURVEY.md §8d closed form)
MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=63,
                    help="per-GPU batch (weak scaling).  63*1030 token rows = 254 row tiles of 256, so the N=1024/3072/4096 encoder GEMMs launch "
                         "1016/3048/4064 workgroups = 3.97/11.9/15.9 full rounds of the 256 CUs (batch 40 left the last round half empty); 93 GB of HBM")
    ap.add_argument("--prec", default="bf16", choices=["bf16", "x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the cpu_baseline leg (256 threads thrash on this workload)")
    return ap.parse_args()


class GemmTimer:
    """Wraps the C-ABI call hook: HIP events (on the launch stream = torch's current stream) around every mtt_gemm of the
    dominant variant, with its algorithmic FLOPs."""

    def __init__(self, lib, variant_of):
        self.lib, self.orig, self.rec, self.variant_of = lib, lib.call, [], variant_of

    def __enter__(self):
        def hooked(name, **kw):
            if name == "gemm" and self.variant_of(**kw) == 3:          # the dominant kernel: gemm_fast256_kernel
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.orig(name, **kw)
                e1.record()
                self.rec.append((2.0 * kw["M"] * kw["N"] * kw["K"] * max(1, kw.get("batch", 1)), e0, e1))
            else:
                self.orig(name, **kw)
        self.lib.call = hooked
        return self

    def __exit__(self, *a):
        self.lib.call = self.orig

    def result(self):
        torch.cuda.synchronize()
        flops = sum(r[0] for r in self.rec)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.rec)
        return flops, ms, len(self.rec)


def _cpu_baseline_worker(batch, threads, q):
    """Runs in a subprocess: oracle (CPU restatement of the reference) forward + loss + backward on a bounded sample."""
    import torch as T
    T.set_num_threads(threads)
    from oracle import configs, taskprompter_oracle as tpo, weights
    import mtt_amd
    cfg = dict(configs.taskprompter("ns6"))
    p = mtt_amd.factory.make_p(mtt_amd.factory.TASK_ORDER, (512, 512), backbone="TaskPrompter_vitL", head="conv")
    model = mtt_amd.factory.get_model(p)                         # only for the state-dict contract (names, shapes)
    contract = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    del model
    sd = weights.synth_state_dict(contract, 0)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    x = weights.synth_images(batch, cfg["img_size"], 1)
    crit = mtt_amd.losses.MultiTaskLoss(p, p.TASKS.NAMES)
    gt = mtt_amd.losses.synthetic_targets(p, batch, 512, 512, "cpu")
    t0 = time.time()
    out = tpo.forward(dict(sd, **params), cfg, x, training=True)
    crit(out, gt)["total"].backward()
    q.put(time.time() - t0)


def cpu_baseline(batch, threads=16, limit_s=150):
    """images/s of one oracle training step on `threads` host cores; bounded by a subprocess timeout."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_cpu_baseline_worker, args=(batch, threads, q))
    pr.start()
    pr.join(limit_s)
    if pr.is_alive():
        pr.terminate()
        pr.join()
        return dict(value=None, unit="images/s", cores=threads, kind="port",
                    sample=f"1 oracle training step at batch {batch} did not finish within {limit_s} s on {threads} threads")
    dt = q.get(timeout=5)
    return dict(value=batch / dt, unit="images/s", cores=threads, kind="port",
                sample=f"1 training step (fwd+loss+bwd, no optimizer) of the same config at batch {batch} on the CPU oracle, {dt:.1f} s")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
    import mtt_amd

    p = mtt_amd.factory.make_p(mtt_amd.factory.TASK_ORDER, (512, 512), backbone="TaskPrompter_vitL", head="conv",
                               embed_dim=300, final_embed_dim=350, chan_nheads=1, use_ctr=True, prec=a.prec)
    torch.manual_seed(0)
    model = mtt_amd.factory.get_model(p)
    if world > 1:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)          # TaskPrompter/main.py:92
    model = model.to(dev).train()
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=False,
                                                        gradient_as_bucket_view=True, bucket_cap_mb=100)
    crit = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES).to(dev)      # HIP loss kernels (the CPU baseline uses the torch restatement)
    # pascal_vitLp16_taskprompter.yml:19-24: Adam(lr 2e-5, wd 1e-6) + clip_grad_norm_(10), fused into two multi-tensor HIP launches
    opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=2e-5, weight_decay=1e-6, max_norm=10.0)
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.randn(a.batch, 3, 512, 512, generator=g).to(dev)
    gt = mtt_amd.losses.synthetic_targets(p, a.batch, 512, 512, dev, seed=rank)

    def step():
        out = net(x)
        loss = crit(out, gt)["total"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()                                        # global-norm clip (yml:24) + Adam
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_step = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt

    # forward-only latency (the metric's second half), same process
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
    fwd_ms_img = (time.perf_counter() - t1) / 3 / a.batch * 1e3
    model.train()

    roof = None
    if not a.no_roofline and rank == 0:
        with GemmTimer(mtt_amd.ops, mtt_amd._lib.gemm_variant) as gt_:
            step()
            flops, ms, n = gt_.result()
        tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        roof = dict(bound="mfma", achieved=round(tf, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                    traffic=None, kernel="gemm_fast256_kernel (256x256x64 bf16 MFMA, LDS-DMA)", launches=n, kernel_ms_per_step=round(ms, 3))
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(a.cpu_sample_batch, a.cpu_threads)
        except Exception as e:  # noqa: BLE001
            cpu = dict(value=None, unit="images/s", cores=os.cpu_count(), kind="port", sample=f"failed: {e!r}")
    if rank == 0:
        train_tflops = 3 * GFLOP_FWD_PER_IMG * value / 1e3
        line = dict(metric="training images/sec (512x512, 6 tasks)", value=round(value, 3), unit="images/s", n_gpus=world,
                    steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="bf16" if a.prec == "bf16" else "f32(bf16x3)", data="synthetic",
                    config=dict(workload="TaskPrompter ViT-L/16 (taskprompter_vit_large_patch16_384), PASCAL-Context 5 tasks + depth = 6 tasks, "
                                         "512x512, ConvHead, embed 300/350, ctr; random-init weights",
                                per_gpu_batch=a.batch, global_batch=a.batch * world, parallelism=f"dp{world}",
                                optimizer="clip_grad_norm 10 + Adam (mtt_grad_sqnorm / mtt_adam_step)", loss=float(loss.detach())),
                    fwd_ms_per_img=round(fwd_ms_img, 3), peak_hbm_gb=round(torch.cuda.max_memory_allocated() / 2**30, 1),
                    model_tflops=dict(train=round(train_tflops, 1), frac_of_bf16_peak=round(train_tflops / world / MFMA_BF16_PEAK_TFLOPS, 4),
                                      fwd=round(GFLOP_FWD_PER_IMG / fwd_ms_img, 1)),
                    roofline=roof, cpu_baseline=cpu)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
