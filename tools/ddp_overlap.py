"""When do DistributedDataParallel's gradient buckets become ready during the hand-written backward?  (SURVEY.md §8 a19 / e;
TaskPrompter/main.py:94: the reference relies on DDP's bucketed all-reduce overlapping with backward.)

One process, RCCL process group of world size 1 on the box's GPU (the 8-GPU curve is the driver's): the NS-6 model goes through
`DistributedDataParallel(gradient_as_bucket_view=True, bucket_cap_mb=...)` exactly as bench.py builds it, with a communication hook that, for
every bucket, records a HIP event on the backward's stream at the moment DDP hands the bucket to the communicator (= all of its gradients
are written) and then runs the stock all-reduce.  ProcessGroupNCCL launches the collective on ITS OWN stream behind that point, so
everything the backward enqueues after the event overlaps the bucket's all-reduce on the xGMI links.  Output: one row per bucket
(bytes, position in the backward's GPU timeline and in the host's enqueue timeline at which it became ready) + a JSON summary.

    python tools/ddp_overlap.py [--batch 16] [--prec x3f] [--bucket-mb 100] [--json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--prec", default="x3f")
    ap.add_argument("--bucket-mb", type=int, default=100)
    ap.add_argument("--config", default="ns6")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
    import bench
    import mtt_amd
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    _, _, (H, W), _, _ = bench.CONFIGS[a.config]
    torch.manual_seed(0)
    p, model = bench.build(a.config, a.prec, mtt_amd)
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model).to(dev).train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=a.config == "cfg4",
                                                    gradient_as_bucket_view=True, bucket_cap_mb=a.bucket_mb)
    rec = []

    def hook(state, bucket):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()                                              # on the backward's stream: every gradient of this bucket is written before it
        rec.append((bucket.index(), bucket.buffer().numel() * bucket.buffer().element_size(), ev, time.perf_counter(), bucket.is_last()))
        return default_hooks.allreduce_hook(state, bucket)
    net.register_comm_hook(None, hook)
    crit = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES).to(dev)
    opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=2e-5, weight_decay=1e-6, max_norm=10.0)
    x = torch.randn(a.batch, 3, H, W, device=dev)
    gt = mtt_amd.losses.synthetic_targets(p, a.batch, H, W, dev, seed=0)
    for it in range(3):                                          # DDP rebuilds its buckets in gradient-arrival order after the first iteration
        rec.clear()
        loss = crit(net(x), gt)["total"]
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter()
        e0.record()
        loss.backward()
        e1.record()
        h1 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
    bwd_ms, host_ms = e0.elapsed_time(e1), (h1 - h0) * 1e3
    rows = []
    for idx, nbytes, ev, ht, last in rec:
        rows.append(dict(bucket=idx, mbytes=round(nbytes / 2**20, 1), ready_gpu_ms=round(e0.elapsed_time(ev), 2),
                         ready_gpu_frac=round(e0.elapsed_time(ev) / bwd_ms, 4), ready_host_frac=round((ht - h0) * 1e3 / host_ms, 4), last=bool(last)))
    total_mb = sum(r["mbytes"] for r in rows)
    before_end = [r for r in rows if r["ready_gpu_frac"] < 0.97]
    summary = dict(config=a.config, prec=a.prec, batch=a.batch, bucket_cap_mb=a.bucket_mb, n_buckets=len(rows), gradient_mbytes=round(total_mb, 1),
                   backward_gpu_ms=round(bwd_ms, 2), backward_host_enqueue_ms=round(host_ms, 2),
                   buckets_ready_before_97pct_of_backward=len(before_end),
                   mbytes_ready_before_97pct_of_backward=round(sum(r["mbytes"] for r in before_end), 1),
                   first_bucket_ready_frac=rows[0]["ready_gpu_frac"] if rows else None,
                   exposed_tail_mbytes=round(total_mb - sum(r["mbytes"] for r in before_end), 1), buckets=rows)
    if a.json:
        print(json.dumps(summary))
    else:
        print(f"# {a.config} {a.prec} per-GPU batch {a.batch}, bucket cap {a.bucket_mb} MB: {len(rows)} buckets, {total_mb:.0f} MB of gradients; "
              f"backward = {bwd_ms:.1f} ms of GPU time, enqueued by the host in {host_ms:.1f} ms")
        print("# bucket   MB    ready at (ms of backward GPU time)   fraction of the backward   fraction of the host's enqueue time")
        for r in rows:
            print(f"  {r['bucket']:4d}  {r['mbytes']:6.1f}   {r['ready_gpu_ms']:10.2f}                          {r['ready_gpu_frac']:6.3f}                    {r['ready_host_frac']:6.3f}"
                  + ("   (last)" if r["last"] else ""))
        print(f"# {len(before_end)} of {len(rows)} buckets ({summary['mbytes_ready_before_97pct_of_backward']:.0f} of {total_mb:.0f} MB) are handed to RCCL before 97 % of the "
              f"backward's GPU work has run: their all-reduce (ProcessGroupNCCL's own stream) overlaps the rest of the backward; exposed tail = "
              f"{summary['exposed_tail_mbytes']:.0f} MB")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
