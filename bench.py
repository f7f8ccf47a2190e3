"""Headline benchmark: training images/sec of the TaskPrompter ViT-L hot path, 512x512, 6 tasks (BASELINE.json),
on N MI355X of one node (one process per GPU; RCCL all-reduce of gradients over xGMI via DistributedDataParallel).

    python bench.py --gpus 1 --steps K --warmup W [--config ns6|cfg2|cfg3|cfg4|cfg5] [--prec x3f|bf16|x3] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = forward + MultiTaskLoss + backward (+ gradient all-reduce) + clip_grad_norm_ + Adam step on one synthetic batch
that is already resident in HBM (TaskPrompter/utils/train_utils.py:32-51), including the re-packing of every weight the optimizer
changed.  Weak scaling: the per-GPU batch is fixed.  Prints ONE JSON line on rank 0 (contract in the task statement).

The HEADLINE (`value`, `ms_per_step`, `fwd_ms_per_img`, `roofline`, `parity`) is the tolerance-compliant arithmetic mode `x3f`
(--prec x3f, the default): the reference computes in fp32 (taskprompter.py:195-214), north_star asks for 1e-3 per task head, bf16 misses
it (1.5e-2) and x3f meets it (2.6e-5) — forward = 3 bf16 MFMAs per product on hi / lo split operands, backward bf16 on the hi planes.
  roofline     — the headline mode's dominant kernel (gemm_ring3_kernel: the 256-row LDS-DMA MFMA GEMM on split planes): bf16 MFMA work of
                 its algorithm / HIP-event time, measured in one extra instrumented step right after the timed region (keeps event overhead
                 out of `value`); `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
                 tools/pmc_traffic.py), null when that record was measured on other kernel sources than this tree's
  roofline_bwd_gemm — the bf16 kernel that runs the headline mode's input-gradient GEMMs
  parity       — the headline mode's worst per-head relative error against the CPU ORACLE's eval forward on this model's weights and 2 of
                 these images (the oracle runs inside the cpu_baseline subprocess)
  fast_mode    — the bf16 mode (BASELINE.json's configs and north_star's 40 % target are stated on it), measured in the same process with
                 the same steps / warm-up: images/s, fwd ms/img, its own roofline (gemm_dma_kernel<1>) and its own parity (which fails 1e-3)
  full_fp32_mode — the fully fp32-class step (x3 forward AND backward: gradients match the oracle's autograd to 6e-5), 10 steps
  torch_rocm_baseline — stock PyTorch-ROCm (the reference's op graph through hipBLASLt / MIOpen / ATen) on the same GPU, fp32 and bf16 autocast
  ref_batch    — the headline step at the reference's own per-GPU batch (trBatch: 2), eager and replayed from one hipGraph
  cpu_baseline — the CPU oracle (restatement of the reference, `kind: "port"`) timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N = 1 only): median of 3 training steps after one warm-up

`--config` selects the other BASELINE.json configurations (cfg2..cfg5) for their own img/s + roofline lines; the driver's default
(no flag) is the metric's configuration, NS-6.
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

MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md)

PASCAL5 = ["semseg", "human_parts", "sal", "normals", "edge"]
PASCAL6 = ["semseg", "depth", "human_parts", "sal", "normals", "edge"]      # TaskPrompter/utils/config.py:30-87 order
# name -> (description, make_p kwargs, image size, default per-GPU batch, algorithmic forward GFLOP / image (SURVEY.md §8d closed form))
CONFIGS = {
    # test-only miniature (oracle/configs.py "mini_ctr": same code path, 64x96, ViT-tiny): the launcher / DDP tests on CPU
    "mini": ("TaskPrompter ViT-tiny miniature, 6 tasks, 64x96 (launcher / host-logic tests only)",
             dict(tasks=PASCAL6, backbone=(128, 4, 2, (1, 2, 3)), head="conv", embed_dim=44, final_embed_dim=52, chan_nheads=1, use_ctr=True,
                  drop_path_rate=0.0),
             (64, 96), 2, 0.217),
    "ns6": ("TaskPrompter ViT-L/16 (taskprompter_vit_large_patch16_384), PASCAL-Context 5 tasks + depth = 6 tasks, 512x512, ConvHead, "
            "embed 300/350, ctr; random-init weights",
            dict(tasks=PASCAL6, backbone="TaskPrompter_vitL", head="conv", embed_dim=300, final_embed_dim=350, chan_nheads=1, use_ctr=True),
            (512, 512), 126, 1046.8),
    "cfg2": ("TaskPrompter ViT-B/16, PASCAL-Context 5 tasks, 512x512, ConvHead, embed 780/1024, 4x4 channel windows, ctr (pascal_vitBp16_taskprompter.yml)",
             dict(tasks=PASCAL5, backbone="TaskPrompter_vitB", head="conv", embed_dim=780, final_embed_dim=1024, chan_nheads=16, use_ctr=True),
             (512, 512), 96, 2306.7),
    "cfg3": ("TaskPrompter ViT-L/16, NYUD-v2 4 tasks, 448x576, ConvHead, embed 768/768, 4x4 channel windows, no ctr (nyud_vitLp16_taskprompter.yml)",
             dict(tasks=["semseg", "depth", "normals", "edge"], backbone="TaskPrompter_vitL", head="conv", embed_dim=768, final_embed_dim=768,
                  chan_nheads=16, use_ctr=False, num_output=dict(semseg=40)),
             (448, 576), 96, 1679.3),
    "cfg4": ("InvPT ViT-L/16 (vit_large_patch16_384 + TransformerDecoder + MLPHead), PASCAL-Context 5 tasks + depth = 6 tasks, 512x512, "
             "embed 512 + 64, intermediate supervision (pascal_vitLp16.yml)",
             dict(tasks=PASCAL6, model="TransformerNet", backbone="vitL", head="mlp", embed_dim=512, PRED_OUT_NUM_CONSTANT=64,
                  mtt_resolution_downsample_rate=2, intermediate_supervision=True),
             (512, 512), 64, 1465.4),
    "cfg5": ("TaskPrompter ViT-L/16, Cityscapes semseg(19) + depth, 1024x2048 (N = 8194 tokens), DEConvHead, embed 300/350, ctr",
             dict(tasks=["semseg", "depth"], backbone="TaskPrompter_vitL", head="deconv", embed_dim=300, final_embed_dim=350, chan_nheads=1,
                  use_ctr=True, num_output=dict(semseg=19)),
             (1024, 2048), 16, 12543.0),
    "swinb": ("TaskPrompter Swin-B (taskprompter_swin_base_patch4_window12_384), Cityscapes semseg(19) + depth (cs_swinB_taskprompter.yml without "
              "the 3ddet task), 1024x2048 x 0.75, window 12, level 256 / final 450, DEConvHead; SURVEY.md §8f rank 3",
              dict(tasks=["semseg", "depth"], backbone="TaskPrompter_swinB", head="deconv", final_embed_dim=450, chan_nheads=1, img_ds_ratio=0.75,
                   level_embed_dim=256, chan_embed_dim=256, prompt_len=1, num_output=dict(semseg=19)),
              (1024, 2048), 16, 3678.8),
}


X3_SUB_BATCH = {"ns6": 63}      # full_fp32_mode (fp32 storage of every activation) runs on this part of the headline batch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree-sha", action="store_true", help="print the source-tree hash the bench line reports as git.tree_sha and exit")
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks of ONE node, one process per GPU.  N > 1 without RANK in the environment: this process becomes the launcher "
                         "(python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 <this script> <same flags>), "
                         "as TaskPrompter/run_taskprompter_*.sh:1 launches main.py; under an external torch.distributed.run the flag must "
                         "equal WORLD_SIZE")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu = host-logic runs of the launcher / DDP path (gloo); the C-ABI library has NO CPU implementation — a cpu run "
                         "only works under tests/bench_emulated.py, which routes ops.call to the test emulator")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="ns6", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0,
                    help="per-GPU batch (weak scaling); 0 = the config's default.  ns6: 126*1030 token rows = 507 row tiles of 256, so the "
                         "N=1024/3072/4096 encoder GEMMs launch 7.9/23.8/31.7 rounds of the 256 CUs; 196 GB of the 288 GB of HBM (round 6: "
                         "the per-step constants — Adam, weight re-packing, ~2 000 small launches: ~10 ms — amortise over twice the images of "
                         "round 5's 63: +2.8 % images/s, profiles/r06_bench_p_batch_sweep.log)")
    ap.add_argument("--prec", default="x3f", choices=["bf16", "x3", "x3f"],
                    help="arithmetic mode of the HEADLINE: x3f (default) meets north_star's 1e-3 per-head tolerance against the fp32 reference; "
                         "bf16 does not (1.5e-2) and is reported as the `fast_mode` sub-record of the default run")
    ap.add_argument("--bucket-mb", type=int, default=100, help="DDP gradient bucket size (MB)")
    ap.add_argument("--grad-comm", default="fp32", choices=["fp32", "bf16"],
                    help="gradient all-reduce payload: fp32 (reference semantics) or bf16-compressed (halves the xGMI bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-ref-batch", action="store_true")
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip the stock PyTorch-ROCm leg (the oracle's torch ops on the GPU)")
    ap.add_argument("--no-fwd", action="store_true", help="skip the forward-only latency leg (profiling runs: every launch then belongs to a training step)")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the bf16 sub-record (fast_mode) of a tolerance-compliant headline run")
    ap.add_argument("--no-x3-mode", action="store_true", help="skip the fully fp32-class sub-record (full_fp32_mode: x3 forward and backward, up to 10 steps)")
    ap.add_argument("--torch-baseline-worker", default=None, help="(internal) subprocess leg of torch_rocm_baseline: 'fp32' or 'bf16'")
    ap.add_argument("--no-fuse-upsample", action="store_true",
                    help="A/B: materialise the x4-upsampled task features and run ConvHead's 3x3 conv on them (the reference's operation order)")
    ap.add_argument("--measure-no-repack", action="store_true",
                    help="MEASUREMENT ONLY (invalid training: the forward keeps using the step-0 weight packs): what the per-step re-packing costs")
    ap.add_argument("--gemm-variant", type=int, default=None,
                    help="A/B: mtt_gemm_desc.variant for every GEMM left at AUTO (1 = general kernel, 3 = LDS-DMA kernel, 11 = general epilogue everywhere)")
    ap.add_argument("--cpu-roofline", action="store_true",
                    help="(host-logic tests, --device cpu only) keep the instrumented roofline step — wall-clock stamps instead of HIP events — so "
                         "that a gloo run exercises what every rank does around it under DDP")
    ap.add_argument("--no-head-prologue", action="store_true",
                    help="A/B: ConvHead's BatchNorm + GELU as a pass of their own writing the fp32 activated map (round 5) instead of riding on the "
                         "prediction GEMM's operand load (round 6)")
    ap.add_argument("--no-gelu-daux", action="store_true",
                    help="A/B: GELU'(z) evaluated in the fc2 input-gradient epilogue (round 5) instead of stored by the fc1 epilogue (round 6)")
    ap.add_argument("--pitch32-from", type=int, default=None,
                    help="A/B: ops.PITCH32_FROM — channel counts from this value on take a channel pitch that is a multiple of 32 (a huge value = the "
                         "multiple-of-8 pitch of rounds 1-5 everywhere)")
    ap.add_argument("--graphed-worker", action="store_true", help="(internal) the subprocess leg of ref_batch.graphed")
    ap.add_argument("--cpu-sample-batch", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the cpu_baseline leg (256 threads thrash on this workload)")
    return ap.parse_args()


class _WallEvent:
    """stand-in for torch.cuda.Event on a box without a GPU (host-logic runs of the instrumented step: tests/bench_emulated.py --cpu-roofline)"""

    def __init__(self, enable_timing=True):
        self.t = None

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class GemmTimer:
    """Wraps the C-ABI call hook: HIP events (on the launch stream = torch's current stream) around every mtt_gemm that the library
    dispatches to one of `variants` (mtt_gemm_variant codes: 3 = the 256 x 256 LDS-DMA bf16 kernel, 8 = the same tile on MTT_SPLIT
    planes, three MFMA products per K step), with its algorithmic FLOPs and bytes."""

    def __init__(self, lib, variant_of, variants=(3,)):
        self.lib, self.orig, self.rec, self.variant_of, self.variants = lib, lib.call, {v: [] for v in variants}, variant_of, variants

    def __enter__(self):
        def hooked(name, **kw):
            if name == "gemm" and getattr(self.lib, "GEMM_VARIANT", None) is not None and not kw.get("variant"):
                kw["variant"] = self.lib.GEMM_VARIANT
            v = self.variant_of(**kw) if name == "gemm" else -1
            v = 8 if v == 9 else v                                     # 9 = the same kernel's implicit-GEMM 3x3 form (gemm_ring3_kernel<true>)
            if v in self.rec:
                Ev = torch.cuda.Event if torch.cuda.is_available() else _WallEvent
                e0, e1 = Ev(enable_timing=True), Ev(enable_timing=True)
                e0.record()
                self.orig(name, **kw)
                e1.record()
                z = max(1, kw.get("batch", 1))
                planes = 2 if v == 8 else 1                            # MTT_SPLIT operands: a hi and a lo bf16 plane each
                out_b = {0: 4, 1: 2, 2: 4}[kw.get("d_dtype", 1)]       # fp32 / bf16 / split (two bf16 planes)
                byts = z * ((kw["M"] + kw["N"]) * kw["K"] * 2 * planes
                            + kw["M"] * kw["N"] * (out_b + (4 if kw.get("resid") is not None else 0)
                                                   + (2 if (kw.get("aux_out") is not None or kw.get("aux_in") is not None) else 0)))
                self.rec[v].append((2.0 * kw["M"] * kw["N"] * kw["K"] * z, e0, e1, byts, (kw["M"], kw["N"], kw["K"], z)))
            else:
                self.orig(name, **kw)
        self.lib.call = hooked
        return self

    def __exit__(self, *a):
        self.lib.call = self.orig

    def result(self, v):
        """(product FLOPs = 2 M N K summed, kernel ms, launches, algorithmic bytes) of variant v"""
        torch.cuda.synchronize()
        rec = self.rec[v]
        return (sum(r[0] for r in rec), sum(r[1].elapsed_time(r[2]) for r in rec), len(rec), sum(r[3] for r in rec))

    def by_shape(self, v, mfma_per_product, top=8):
        """the launches of variant v grouped by (M, N, K, batch): launches, summed ms, TFLOP/s of MFMA work — the kernel's live per-shape table"""
        agg = {}
        for fl, e0, e1, _, shp in self.rec[v]:
            a = agg.setdefault(shp, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += fl
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]
        return [dict(M=k[0], N=k[1], K=k[2], batch=k[3], launches=a[0], ms=round(a[1], 3),
                     tflops=round(a[2] / (a[1] * 1e-3) / 1e12, 1) if a[1] > 0 else None,
                     tflops_mfma_issued=round(mfma_per_product * a[2] / (a[1] * 1e-3) / 1e12, 1) if a[1] > 0 else None) for k, a in rows]


KERNELS = {
    3: ("gemm_dma_kernel<1>", "gemm_dma_kernel<1> (256x256x64 tile, 8 waves x (8x4) v_mfma_f32_16x16x32_bf16, operands streamed HBM -> LDS by "
        "global_load_lds_dwordx4 with wave-uniform base + 32-bit lane offset addressing, staggered read / MFMA phases, specialised "
        "interior-tile epilogue; <0> = the same kernel with general addressing for calls with a K tail): every encoder Linear forward "
        "(bf16 mode) and input gradient of the step"),
    8: ("gemm_ring3_kernel", "gemm_ring3_kernel (the same tile on MTT_SPLIT operands: hi / lo bf16 planes of both operands staged once per 32-deep K step "
        "into a part-recycled LDS ring with counted vmcnt, three MFMA products Ah Bh + Ah Bl + Al Bh per step, fp32 accumulate; <true> = its "
        "implicit-GEMM 3x3 form): every encoder Linear, fea_decode, fea_fuse[0], the fea_fuse 3x3 convs and the taps-first head GEMM of the "
        "fp32-class forward"),
}


def roofline_of(gt_, v, pmc_ok=True):
    """roofline record of GEMM variant v from the instrumented step: achieved = MFMA work the kernel's algorithm issues / HIP-event time."""
    flops, ms, n, byts = gt_.result(v)
    if n == 0 or ms <= 0:
        return None
    mfma_per_product = 3 if v == 8 else 1
    tf = flops / (ms * 1e-3) / 1e12                      # ALGORITHMIC rate: 2 M N K per product whatever the kernel issues for it
    tf_issued = mfma_per_product * tf
    kname, kdesc = KERNELS[v]
    traffic = _pmc_traffic(kname) if pmc_ok else dict(note="the committed PMC passes were taken on the default workload (ns6, default per-GPU batch)")
    rec = dict(bound="mfma", achieved=round(tf, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
               achieved_mfma_issued=round(tf_issued, 2), frac_mfma_issued=round(tf_issued / MFMA_BF16_PEAK_TFLOPS, 4),
               traffic=traffic.get("hbm_bytes_per_launch"), traffic_source=traffic.get("source"), traffic_commit=traffic.get("commit"),
               traffic_csrc_sha=traffic.get("csrc_sha"),
               traffic_note=traffic.get("note"), algorithmic_bytes_per_launch=int(byts / n), kernel=kdesc, launches=n,
               kernel_ms_per_step=round(ms, 3), avg_launch_us=round(ms / n * 1e3, 1), algorithmic_tflop_per_step=round(flops / 1e12, 2),
               mfma_issued_tflop_per_step=round(mfma_per_product * flops / 1e12, 2),
               flop_convention=("achieved / frac = ALGORITHMIC FLOPs (2 M N K per GEMM, SURVEY.md 8d) / HIP-event time / the dense bf16 MFMA peak; "
                                "achieved_mfma_issued / frac_mfma_issued = the bf16 MFMA work the kernel's algorithm issues for them (%d MFMA "
                                "product%s per algorithmic product)" % (mfma_per_product, "s" if mfma_per_product > 1 else "")),
               by_shape=gt_.by_shape(v, mfma_per_product))
    if v == 8:
        rec["fp32_class_vs_fp32_matrix_peak_157"] = round(tf / 157.3, 2)
    return rec


def build(cfg_name, prec, mtt_amd):
    desc, kw, size, _, _ = CONFIGS[cfg_name]
    kw = dict(kw)
    p = mtt_amd.factory.make_p(kw.pop("tasks"), size, prec=prec, **kw)
    return p, mtt_amd.factory.get_model(p)


def _cpu_baseline_worker(cfg_name, batch, threads, q, ref_in=None, ref_out=None, timed_steps=4):
    """Runs in a subprocess: oracle (CPU restatement of the reference) forward + loss + backward on a bounded sample.
    ref_in / ref_out: additionally run the oracle's EVAL forward on the state dict + images saved in `ref_in` (the bench model's own
    weights) and save its per-task outputs to `ref_out` — the reference the bench line's parity records are measured against."""
    import torch as T
    T.set_num_threads(threads)
    from oracle import configs, losses_oracle, weights
    import mtt_amd
    okey = {"ns6": "ns6", "cfg2": "cfg2", "cfg3": "cfg3", "cfg4": "cfg4_6", "cfg5": "cfg5", "swinb": "cs_swinB"}[cfg_name]
    invpt, swin = cfg_name == "cfg4", cfg_name == "swinb"
    cfg = dict(configs.invpt(okey) if invpt else (configs.swin(okey) if swin else configs.taskprompter(okey)))
    p, model = build(cfg_name, "x3", mtt_amd)                     # only for the state-dict contract (names, shapes)
    contract = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    del model
    sd = weights.synth_state_dict(contract, 0)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    x = weights.synth_images(batch, cfg["img_size"], 1)
    crit = losses_oracle.MultiTaskLoss(p, p.TASKS.NAMES)
    gt = mtt_amd.losses.synthetic_targets(p, batch, cfg["img_size"][0], cfg["img_size"][1], "cpu")
    if invpt:
        from oracle import invpt_oracle as orc
    elif swin:
        from oracle import swin_oracle as orc
    else:
        from oracle import taskprompter_oracle as orc
    times = []
    for _ in range(timed_steps):                                  # one warm-up (allocator, thread pool), three timed
        for v in params.values():
            v.grad = None
        t0 = time.time()
        out = orc.forward(dict(sd, **params), cfg, x, training=True)
        crit(out, gt)["total"].backward()
        times.append(time.time() - t0)
    if ref_in is not None:
        blob = T.load(ref_in, map_location="cpu")
        t0 = time.time()
        with T.no_grad():
            ref = orc.forward(blob["state_dict"], cfg, blob["images"])
        T.save({k: v for k, v in ref.items() if T.is_tensor(v)}, ref_out)
        times.append(time.time() - t0)
    q.put(times)


def cpu_baseline(cfg_name, batch, threads=16, limit_s=300, ref_in=None, ref_out=None, parity_only=False):
    """images/s of one oracle training step on `threads` host cores; bounded by a subprocess timeout.  parity_only: no timed steps, only the
    oracle's eval forward on `ref_in` (the parity reference of a --no-cpu-baseline run)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_cpu_baseline_worker, args=(cfg_name, batch, threads, q, ref_in, ref_out, 0 if parity_only else 4))
    pr.start()
    pr.join(limit_s)
    if parity_only:
        if pr.is_alive():
            pr.terminate()
            pr.join()
        return None
    host = dict(host_cores=os.cpu_count(), host_cpu=_cpu_model())
    if pr.is_alive():
        pr.terminate()
        pr.join()
        return dict(value=None, unit="images/s", cores=threads, kind="port", **host,
                    sample=f"4 oracle training steps at batch {batch} did not finish within {limit_s} s on {threads} threads")
    ts = q.get(timeout=5)
    warm, timed = ts[0], sorted(ts[1:4])
    dt = timed[1]                                                 # median of the three timed steps
    return dict(value=batch / dt, unit="images/s", cores=threads, kind="port", **host, step_s=[round(t, 2) for t in ts[1:4]],
                sample=f"3 training steps (fwd+loss+bwd, no optimizer) of the same config at batch {batch} on the CPU oracle after one warm-up "
                       f"step ({warm:.1f} s): median {dt:.1f} s (min {timed[0]:.1f}, max {timed[2]:.1f}) on {threads} of {os.cpu_count()} host threads")


def _torch_baseline_worker(cfg_name, mode):
    """(subprocess) STOCK PyTorch-ROCm on this GPU: the oracle's torch ops (the reference's op graph: hipBLASLt / rocBLAS GEMMs, MIOpen
    convs, ATen softmax / LayerNorm / GELU kernels) run on cuda:0, fp32 like the reference or under bf16 autocast — forward + criterion +
    backward + torch.optim.Adam + clip_grad_norm_.  BASELINE ONLY: the number a user gets for free on this chip, next to the hand-written path."""
    import torch as T
    from oracle import configs, losses_oracle, weights
    import mtt_amd
    okey = {"ns6": "ns6", "cfg2": "cfg2", "cfg3": "cfg3", "cfg4": "cfg4_6", "cfg5": "cfg5", "swinb": "cs_swinB"}[cfg_name]
    invpt, swin = cfg_name == "cfg4", cfg_name == "swinb"
    cfg = dict(configs.invpt(okey) if invpt else (configs.swin(okey) if swin else configs.taskprompter(okey)))
    p, model = build(cfg_name, "x3", mtt_amd)
    contract = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    del model
    dev = T.device("cuda", 0)
    sd = {k: v.to(dev) for k, v in weights.synth_state_dict(contract, 0).items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    if invpt:
        from oracle import invpt_oracle as orc
    elif swin:
        from oracle import swin_oracle as orc
    else:
        from oracle import taskprompter_oracle as orc
    crit = losses_oracle.MultiTaskLoss(p, p.TASKS.NAMES)
    opt = T.optim.Adam(list(params.values()), lr=2e-5, weight_decay=1e-6)
    H, W = cfg["img_size"]
    res = {}
    for batch in (2, 16):
        try:
            x = weights.synth_images(batch, cfg["img_size"], 1).to(dev)
            gt = mtt_amd.losses.synthetic_targets(p, batch, H, W, dev)

            def step():
                with T.autocast("cuda", dtype=T.bfloat16, enabled=(mode == "bf16")):
                    out = orc.forward(dict(sd, **params), cfg, x, training=True)
                out = {k: (v.float() if T.is_tensor(v) else {kk: vv.float() for kk, vv in v.items()}) for k, v in out.items()}
                loss = crit(out, gt)["total"]
                opt.zero_grad(set_to_none=True)
                loss.backward()
                T.nn.utils.clip_grad_norm_(list(params.values()), 10.0)
                opt.step()
                return loss
            step()
            step()                                   # two untimed steps: MIOpen / hipBLASLt pick their kernels on first use
            T.cuda.synchronize()
            n = 3 if batch <= 8 else 2
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            T.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            res[str(batch)] = dict(images_per_s=round(batch * 1e3 / ms, 2), ms_per_step=round(ms, 1),
                                   peak_hbm_gb=round(T.cuda.max_memory_allocated() / 2**30, 1))
            print(json.dumps(dict(partial=res)), flush=True)
        except Exception as e:  # noqa: BLE001  (out of memory at the larger batches ends the sweep)
            res[str(batch)] = dict(error=repr(e)[:160])
            break
        finally:
            x = gt = None
            T.cuda.empty_cache()
    print(json.dumps(dict(result=res)), flush=True)


def torch_rocm_baseline(cfg_name, limit_s=150):
    """Both legs (fp32 = the reference's arithmetic; bf16 autocast) in subprocesses, bounded; the last complete / partial result is kept."""
    import subprocess
    out = dict(what="stock PyTorch-ROCm (torch %s) on the same GPU: the CPU oracle's torch op graph = the reference's (hipBLASLt / MIOpen / ATen "
                    "kernels), training step = fwd + criterion + bwd + clip_grad_norm_ + torch.optim.Adam, synthetic batch resident in HBM; "
                    "per-GPU batch -> images/s.  Baseline only." % torch.__version__)
    for mode in ("fp32", "bf16"):
        stdout, err = "", "timeout"
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--torch-baseline-worker", mode, "--config", cfg_name],
                               capture_output=True, text=True, timeout=limit_s)
            stdout, err = r.stdout, (r.stderr.strip().splitlines() or ["?"])[-1][:200]
        except subprocess.TimeoutExpired as e:
            stdout = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:200]
        recs = [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]
        fin = [x["result"] for x in recs if "result" in x] or [x["partial"] for x in recs if "partial" in x]
        out[mode] = fin[-1] if fin else dict(error="no result: " + err)
    return out


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _git(*args):
    import subprocess
    try:
        return subprocess.run(["git", "-C", ROOT] + list(args), capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:  # noqa: BLE001
        return ""


def source_tree_sha():
    """sha256 over the product's sources (package .py / .hip / .h, include/, bench.py), path-sorted: identifies the tree on a box that has
    no .git (gpurun / driver snapshots).  `python bench.py --tree-sha` prints it for any checkout."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "bench.py")]
    for top in ("multi-task-transformer_amd", "include"):
        for d, _, fs in os.walk(os.path.join(ROOT, top)):
            files += [os.path.join(d, f) for f in fs if f.endswith((".py", ".hip", ".h"))]
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _source_id():
    """git.head: the commit when .git travels with the tree, else MTT_COMMIT, else `tree-<source_tree_sha>` (never null: the driver's
    box is a snapshot without .git); tree_sha is always present and can be recomputed on any checkout with `bench.py --tree-sha`."""
    tree = source_tree_sha()
    head = _git("rev-parse", "--short", "HEAD")
    return dict(head=head or os.environ.get("MTT_COMMIT") or f"tree-{tree}", tree_sha=tree,
                dirty=bool(_git("status", "--porcelain", "--untracked-files=no")) if head else None)


def _pmc_traffic(kernel_name):
    """HBM bytes per launch of a kernel from the committed PMC passes (profiles/pmc_traffic.json, written by tools/pmc_traffic.py):
    {kernel name: {hbm_bytes_per_launch, source, commit, csrc_sha}}.  The record names the commit and the hash of the kernel sources it
    was measured on; if csrc/gemm.hip has changed since (hash mismatch) the number is STALE and `traffic` is reported as null."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path)).get(kernel_name)
    except (OSError, ValueError, AttributeError):
        rec = None
    if not rec:
        return dict(note="no PMC record for this kernel in profiles/pmc_traffic.json")
    try:
        sha = hashlib.sha256(open(os.path.join(ROOT, "multi-task-transformer_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]
    except OSError:
        sha = None
    if rec.get("csrc_sha") != sha:
        return dict(source=rec.get("source"), commit=rec.get("commit"),
                    note=f"STALE: measured on gemm.hip {rec.get('csrc_sha')}, this tree has {sha}; re-run tools/pmc_traffic.py")
    return rec


def _graphed_worker(cfg_name, prec):
    """(subprocess) the reference-batch iteration recorded once in a hipGraph and replayed (graphs.GraphedTrainStep); prints one JSON object."""
    import mtt_amd
    dev = torch.device("cuda", 0)
    _, _, (H, W), _, _ = CONFIGS[cfg_name]
    torch.manual_seed(0)
    p, model = build(cfg_name, prec, mtt_amd)
    model = model.to(dev).train()
    crit = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES).to(dev)
    opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=2e-5, weight_decay=1e-6, max_norm=10.0, capturable=True)
    x = torch.randn(2, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    gt = mtt_amd.losses.synthetic_targets(p, 2, H, W, dev, seed=0)
    step = mtt_amd.graphs.GraphedTrainStep(model, crit, opt, x, gt, warmup=2)
    for _ in range(3):
        step(x, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        loss = step(x, gt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps(dict(images_per_s=round(2e3 / ms, 2), ms_per_step=round(ms, 2), loss=float(loss.detach()), steps=10,
                          what="forward + criterion + backward + clip + Adam + weight re-packing replayed from ONE hipGraph "
                               "(graphs.GraphedTrainStep), measured in its own process")), flush=True)


def graphed_ref_batch(cfg_name, prec, limit_s=240):
    """Runs _graphed_worker in a subprocess (a failed stream capture can leave the HIP runtime unusable; the bench line must survive it)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--graphed-worker", "--config", cfg_name, "--prec", prec],
                           capture_output=True, text=True, timeout=limit_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return dict(error=f"worker exit {r.returncode}: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200])
    except Exception as e:  # noqa: BLE001
        return dict(error=repr(e)[:200])


MODE_DTYPE = {
    "bf16": "bf16",
    "x3f": "f32(bf16x3)",
    "x3": "f32(bf16x3)",
}
MODE_TEXT = {
    "bf16": "bf16 operands / bf16 MFMA, fp32 accumulate, fp32 residual stream / statistics / parameter gradients / optimizer; forward AND backward",
    "x3f": "forward: fp32-class (every product = 3 bf16 MFMAs on hi / lo split operands, fp32 accumulate; encoder Linears on the LDS-DMA kernel "
           "over pre-split planes); backward: bf16 on the hi planes; fp32 residual stream, statistics, parameter gradients and optimizer in both",
    "x3": "fp32-class forward AND backward (every product = 3 bf16 MFMAs on hi / lo split operands, fp32 accumulate)",
}


def launch_ranks(n, argv=None, script=None):
    """`bench.py --gpus N` started as ONE process: become the launcher of N ranks on this node, one process per GPU — what
    TaskPrompter/run_taskprompter_*.sh:1 does for main.py with torch.distributed.launch:
        python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node N <script> <same flags>
    `--standalone` lets the elastic agent's own store pick the rendezvous port (an endpoint of port 0 that is bound ONCE, by the process
    that keeps it) — the earlier "bind a socket, read its port, close it, pass --master-port" left a window in which a concurrent launch
    on the box could take the port; `--local-addr 127.0.0.1` because the container's hostname may not resolve.
    Rank 0's JSON line passes through on stdout; returns the launcher's exit code."""
    import subprocess
    script = script or os.path.abspath(sys.argv[0])
    argv = list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 1) // max(1, n)))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           script] + argv
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.tree_sha:
        return print(source_tree_sha())
    if a.graphed_worker:
        return _graphed_worker(a.config, a.prec)
    if a.torch_baseline_worker:
        return _torch_baseline_worker(a.config, a.torch_baseline_worker)
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if "RANK" in os.environ and a.gpus != world and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size is what runs and what n_gpus reports", file=sys.stderr)
    on_gpu = a.device == "cuda"
    if on_gpu:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:                                 # host-logic runs only (tests/bench_emulated.py): everything GPU-specific is skipped
        dev = torch.device("cpu")
        a.no_ref_batch = a.no_torch_baseline = a.no_cpu_baseline = a.no_fast_mode = a.no_x3_mode = a.no_parity = True
        a.no_roofline = a.no_roofline or not a.cpu_roofline
        torch.cuda.synchronize = lambda *x, **k: None
        torch.cuda.reset_peak_memory_stats = lambda *x, **k: None
        torch.cuda.max_memory_allocated = lambda *x, **k: 0
        torch.cuda.empty_cache = lambda *x, **k: None
    # launched by torch.distributed.run (RANK / MASTER_ADDR in the env): RCCL process group + DDP even at world size 1, so that a
    # one-GPU box exercises the same communicator / bucketed all-reduce code path the 8-GPU run takes
    ddp_mode = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if ddp_mode:
        if on_gpu:
            dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group(backend="gloo", init_method="env://")
    import mtt_amd

    desc, _, (H, W), dflt_batch, gflop_fwd = CONFIGS[a.config]
    batch = a.batch or dflt_batch
    if a.no_fuse_upsample:
        mtt_amd.taskprompter.TaskPrompterWrapper.fuse_upsample = False
    if a.gemm_variant is not None:
        mtt_amd.ops.GEMM_VARIANT = a.gemm_variant
    if a.no_gelu_daux:
        mtt_amd.autograd_path.GELU_DAUX = False
    if a.pitch32_from is not None:
        mtt_amd.ops.PITCH32_FROM = a.pitch32_from
    if os.environ.get("MTT_MLP_SPLIT_RULE64") is not None:   # A/B: MlpHalfFn's split-plane condition of rounds 3-5 (channel counts % 64)
        mtt_amd.autograd_path.MLP_SPLIT_RULE64 = os.environ["MTT_MLP_SPLIT_RULE64"] == "1"
    if os.environ.get("MTT_WINATTN_BIAST") is not None:      # A/B: transposed bias table for the window-attention backward's key-owner pass
        import importlib
        importlib.import_module(mtt_amd.__name__ + ".swin_autograd").WINATTN_BIAST = os.environ["MTT_WINATTN_BIAST"] == "1"
    if os.environ.get("MTT_CHAN_KV_FN") is not None:         # A/B: Swin chan_kv as the split-K node (1, default) or through BLinearFn on a transposed copy (0)
        import importlib
        importlib.import_module(mtt_amd.__name__ + ".swin_autograd").CHAN_KV_FN = os.environ["MTT_CHAN_KV_FN"] == "1"
    if os.environ.get("MTT_DECONV_SPLIT") is not None:       # A/B: ConvTranspose2d as split-plane GEMM + mtt_pixshuf2 (1, default) or the general kernel's pixel-shuffle store (0)
        mtt_amd.ops.DECONV_SPLIT = os.environ["MTT_DECONV_SPLIT"] == "1"
    if a.no_head_prologue:
        mtt_amd.autograd_path.HEAD_PROLOGUE = False
    if a.measure_no_repack:
        mtt_amd.ops.bump_param_epoch = lambda *a, **k: None
        torch.autograd.graph.increment_version = lambda *x, **k: None
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.randn(batch, 3, H, W, generator=g).to(dev)
    solo = rank == 0 and world == 1                       # legs that only make sense for the single-GPU line

    import tempfile
    tmpd = tempfile.mkdtemp(prefix="mtt_bench_") if solo else None
    ref_paths = (os.path.join(tmpd, "in.pt"), os.path.join(tmpd, "ref.pt")) if solo else (None, None)
    outs, saved = {}, {}

    def run_mode(prec, headline, nb=None):
        """One arithmetic mode, measured like a headline: W warm-up + K timed steps of the full training iteration, the forward-only latency,
        an instrumented step for the roofline of ITS dominant GEMM kernel, and its eval outputs on 2 images for the parity record."""
        nonlocal outs
        bsz = min(nb, batch) if nb else batch              # (the fully fp32-class sub-record runs on a part of the batch: fp32 storage of everything)
        xb_ = x[:bsz]
        torch.manual_seed(0)
        p, model = build(a.config, prec, mtt_amd)
        if ddp_mode and headline and on_gpu:
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)          # TaskPrompter/main.py:92
        elif ddp_mode and headline:         # DDP refuses nn.SyncBatchNorm on CPU modules: plain holders flagged for the same cross-rank code path
            for m in model.modules():
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                    m._mtt_sync = True
        model = model.to(dev)
        if not headline and "sd" in saved:                 # the second mode starts from the weights the parity reference was taken on
            model.load_state_dict(saved["sd"])
            if solo and not a.no_parity:
                model.eval()
                with torch.no_grad():
                    outs[prec] = {t: v.float().cpu() for t, v in model(x[:2]).items() if torch.is_tensor(v)}
        model.train()
        net = model
        if ddp_mode and headline:
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local] if on_gpu else None, find_unused_parameters=a.config == "cfg4",
                                                            gradient_as_bucket_view=True, bucket_cap_mb=a.bucket_mb)
            if a.grad_comm == "bf16":
                from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
                net.register_comm_hook(None, default_hooks.bf16_compress_hook)
        crit = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES).to(dev)      # HIP loss kernels (the CPU baseline uses the torch restatement)
        # pascal_vitLp16_taskprompter.yml:19-24: Adam(lr 2e-5, wd 1e-6) + clip_grad_norm_(10), fused into two multi-tensor HIP launches
        opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=2e-5, weight_decay=1e-6, max_norm=10.0)
        gt = mtt_amd.losses.synthetic_targets(p, bsz, H, W, dev, seed=rank)

        def make_step(xb, gtb):
            def step():
                out = net(xb)
                loss = crit(out, gtb)["total"]
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()                                        # global-norm clip (yml:24) + Adam
                return loss
            return step

        step = make_step(xb_, gt)
        torch.cuda.reset_peak_memory_stats()
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1 and headline:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step()
        torch.cuda.synchronize()
        if world > 1 and headline:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1 and headline:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        rec = dict(mode=prec, dtype=MODE_DTYPE[prec], arithmetic=MODE_TEXT[prec], ms_per_step=dt / a.steps * 1e3,
                   images_per_s=bsz * (world if headline else 1) * a.steps / dt, steps=a.steps, warmup=a.warmup, per_gpu_batch=bsz,
                   loss=float(loss.detach()), peak_hbm_gb=round(torch.cuda.max_memory_allocated() / 2**30, 1), tasks=list(p.TASKS.NAMES))

        def host_share(fn):
            """host time to ENQUEUE one step (python + ctypes + launches, no sync) against the step's wall time: far below 1 = GPU-bound."""
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            fn()
            h1 = time.perf_counter()
            torch.cuda.synchronize()
            h2 = time.perf_counter()
            return dict(enqueue_ms=round((h1 - h0) * 1e3, 2), step_ms=round((h2 - h0) * 1e3, 2))

        rec["host"] = host_share(step)

        # forward-only latency (the metric's second half), same process
        rec["fwd_ms_per_img"] = None
        if not a.no_fwd:
            model.eval()
            with torch.no_grad():
                for _ in range(2):
                    model(xb_)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    model(xb_)
                torch.cuda.synchronize()
            rec["fwd_ms_per_img"] = (time.perf_counter() - t1) / 5 / bsz * 1e3
            model.train()

        # roofline of this mode's dominant kernel: one extra instrumented step (keeps the event overhead out of the timed region)
        rec["roofline"] = rec["roofline_bwd_gemm"] = None
        if not a.no_roofline and rank != 0 and headline:
            step()                               # every rank takes the instrumented step: under DDP its gradient all-reduce is a collective
        if not a.no_roofline and rank == 0:
            with GemmTimer(mtt_amd.ops, mtt_amd._lib.gemm_variant, variants=(3, 8)) as gt_:
                step()
                pmc_ok = a.config == "ns6" and batch == dflt_batch and prec == a.prec
                rec["roofline"] = roofline_of(gt_, 8 if prec == "x3f" else 3, pmc_ok)
                if prec == "x3f":                # its bf16 backward's input-gradient GEMMs run on the bf16 kernel
                    rec["roofline_bwd_gemm"] = roofline_of(gt_, 3, pmc_ok)

        # the reference's own per-GPU batch (trBatch: 2, yml:8), same step, same process
        rec["ref_batch"] = None
        if headline and not a.no_ref_batch and solo and batch > 2:
            s2 = make_step(x[:2].contiguous(), {k: v[:2].contiguous() for k, v in gt.items()})
            for _ in range(2):
                s2()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(5):
                s2()
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t2) / 5 * 1e3
            rec["ref_batch"] = dict(per_gpu_batch=2, images_per_s=round(2e3 / ms2, 2), ms_per_step=round(ms2, 2), host=host_share(s2))
            del s2

        # parity inputs: the headline model's weights (after its timed steps) + 2 of the bench images go to the CPU oracle (which runs in
        # the cpu_baseline subprocess — the only leg that may touch oracle/); every mode's eval outputs on them are compared with its answer
        if headline and solo and not (a.no_parity and a.no_fast_mode):
            saved["sd"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
            if not a.no_parity:
                torch.save(dict(state_dict={k: v.cpu() for k, v in saved["sd"].items()}, images=x[:2].cpu()), ref_paths[0])
                model.eval()
                with torch.no_grad():
                    outs[prec] = {t: v.float().cpu() for t, v in model(x[:2]).items() if torch.is_tensor(v)}
        fused_heads = (not a.no_fuse_upsample) and any(type(hd).__name__ == "ConvHead" for hd in getattr(model, "heads", {}).values())
        rec["_heads"] = (p.final_embed_dim, sum(1 for hd in model.heads.values() if type(hd).__name__ == "ConvHead")) if fused_heads else None
        del step, net, opt, crit, model, loss
        torch.cuda.empty_cache()
        return rec

    head = run_mode(a.prec, True)
    fast = None
    if solo and not a.no_fast_mode and a.prec != "bf16":
        try:
            fast = run_mode("bf16", False)
        except Exception as e:  # noqa: BLE001
            fast = dict(mode="bf16", error=repr(e)[:300])
    # the fully fp32-class step (x3: 3-MFMA products forward AND backward, fp32 storage) as a third record: the reference trains in fp32
    # (no AMP), x3 is the mode whose GRADIENTS match the oracle's autograd to 6e-5 (tests/test_gpu_fullsize.py)
    full = None
    if solo and not a.no_x3_mode and a.prec != "x3":
        keep = (a.steps, a.warmup, a.no_fwd, a.no_roofline)
        try:
            a.steps, a.warmup, a.no_fwd, a.no_roofline = min(10, a.steps), 1, True, True      # 10 steps of ~1.3 s (3 were too few)
            r3 = run_mode("x3", False, nb=X3_SUB_BATCH.get(a.config))
            full = dict(mode="x3", dtype=MODE_DTYPE["x3"], arithmetic=MODE_TEXT["x3"], images_per_s=round(r3["images_per_s"], 3),
                        ms_per_step=round(r3["ms_per_step"], 3), steps=r3["steps"], warmup=r3["warmup"], per_gpu_batch=r3["per_gpu_batch"], loss=r3["loss"],
                        peak_hbm_gb=r3["peak_hbm_gb"])
        except Exception as e:  # noqa: BLE001
            full = dict(mode="x3", error=repr(e)[:300])
        a.steps, a.warmup, a.no_fwd, a.no_roofline = keep
    saved.clear()
    torch.cuda.empty_cache()
    if head.get("ref_batch") is not None:
        head["ref_batch"]["graphed"] = graphed_ref_batch(a.config, a.prec)

    torch_base = None
    if solo and not a.no_torch_baseline:
        torch_base = torch_rocm_baseline(a.config)

    cpu = None
    if solo and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(a.config, a.cpu_sample_batch, a.cpu_threads, ref_in=ref_paths[0] if outs else None, ref_out=ref_paths[1])
        except Exception as e:  # noqa: BLE001
            cpu = dict(value=None, unit="images/s", cores=a.cpu_threads, kind="port", host_cores=os.cpu_count(), sample=f"failed: {e!r}")
    elif solo and outs:                           # --no-cpu-baseline: the parity reference alone (the oracle's eval forward, untimed)
        cpu_baseline(a.config, a.cpu_sample_batch, a.cpu_threads, ref_in=ref_paths[0], ref_out=ref_paths[1], parity_only=True)
    parity = {}
    if rank == 0 and outs:
        ref_txt = ("the CPU oracle's eval forward (fp32, %d host threads) on this model's weights and 2 of the bench images; per-head relative "
                   "L2 error, north_star's tolerance = 1e-3" % a.cpu_threads)
        try:
            ref = torch.load(ref_paths[1], map_location="cpu")
            for mode, o in outs.items():
                errs = {t: float((o[t].double() - ref[t].double()).norm() / ref[t].double().norm()) for t in head["tasks"]}
                parity[mode] = dict(mode=mode, worst_head_rel_err=max(errs.values()), per_head=errs, reference=ref_txt,
                                    meets_1e_3=max(errs.values()) <= 1e-3)
        except Exception as e:  # noqa: BLE001  (no oracle reference: cpu_baseline skipped or failed)
            for mode in outs:
                parity[mode] = dict(mode=mode, error="no oracle reference: " + repr(e)[:200])
    if full is not None and "error" not in full:
        full["parity"] = parity.get("x3")
    if tmpd:
        import shutil
        shutil.rmtree(tmpd, ignore_errors=True)

    if rank == 0:
        value = head["images_per_s"]

        def flops_block(rec):
            """model FLOPs follow the REFERENCE's operation order.  ConvHeads run "taps first" (3x3 conv commuted with the x4
            bilinear resize: the channel mixing happens on the h x w map), which executes 15/16 of the head conv's MACs less."""
            ips, fwd = rec["images_per_s"], rec["fwd_ms_per_img"]
            gflop_exec = gflop_fwd
            if rec.get("_heads"):
                F_, n_conv = rec["_heads"]
                gflop_exec = gflop_fwd - (15.0 / 16.0) * 2.0 * (H // 4) * (W // 4) * F_ * F_ * 9 * n_conv / 1e9
            train_tflops = 3 * gflop_fwd * ips / 1e3
            return dict(train=round(train_tflops, 1), frac_of_bf16_peak=round(train_tflops / world / MFMA_BF16_PEAK_TFLOPS, 4),
                        fwd=None if fwd is None else round(gflop_fwd / fwd, 1),
                        fwd_frac_of_bf16_peak=None if fwd is None else round(gflop_fwd / fwd / MFMA_BF16_PEAK_TFLOPS, 4),
                        # both conventions side by side: the reference-order FLOPs the metric is defined on, and the
                        # FLOPs this implementation executes for them
                        fwd_frac_reference_order=None if fwd is None else round(gflop_fwd / fwd / MFMA_BF16_PEAK_TFLOPS, 4),
                        fwd_frac_executed=None if fwd is None else round(gflop_exec / fwd / MFMA_BF16_PEAK_TFLOPS, 4),
                        gflop_fwd_per_img=gflop_fwd, gflop_fwd_executed_per_img=round(gflop_exec, 1),
                        convention="model FLOPs of the reference's operation order (one multiply-add = 2 FLOPs whatever the arithmetic mode "
                                   "issues for it); 'executed' subtracts what the taps-first ConvHead (upsample x4 + 3x3 conv commuted) does not compute")

        fast_rec = None
        if fast is not None:
            if "error" in fast:
                fast_rec = fast
            else:
                fast_rec = dict(mode="bf16", dtype="bf16", arithmetic=fast["arithmetic"], images_per_s=round(fast["images_per_s"], 3),
                                ms_per_step=round(fast["ms_per_step"], 3), steps=fast["steps"], warmup=fast["warmup"], per_gpu_batch=fast["per_gpu_batch"],
                                fwd_ms_per_img=None if fast["fwd_ms_per_img"] is None else round(fast["fwd_ms_per_img"], 3),
                                loss=fast["loss"], peak_hbm_gb=fast["peak_hbm_gb"], host=fast["host"], model_tflops=flops_block(fast),
                                roofline=fast["roofline"], parity=parity.get("bf16"),
                                note="BASELINE.json's configs and north_star's 40 % MFMA target are stated on bf16: this sub-record is that mode, "
                                     "measured in the same process with the same steps / warm-up.  Its outputs miss north_star's 1e-3 per-head "
                                     "tolerance (see its parity), so it is NOT the headline.")
        metric = "training images/sec (512x512, 6 tasks)" if a.config == "ns6" else f"training images/sec ({H}x{W}, {len(head['tasks'])} tasks)"
        line = dict(metric=metric, value=round(value, 3), unit="images/s", n_gpus=world,
                    steps=a.steps, warmup=a.warmup, ms_per_step=round(head["ms_per_step"], 3), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype=MODE_DTYPE[a.prec], data="synthetic",
                    config=dict(workload=desc, name=a.config, per_gpu_batch=batch, global_batch=batch * world, parallelism=f"dp{world}",
                                mode=a.prec, arithmetic=MODE_TEXT[a.prec],
                                optimizer="clip_grad_norm 10 + Adam (mtt_grad_sqnorm / mtt_adam_step)", loss=head["loss"],
                                grad_comm=a.grad_comm if ddp_mode else None, bucket_mb=a.bucket_mb if ddp_mode else None,
                                rccl_ranks=world if (ddp_mode and on_gpu) else None, gloo_ranks=None if on_gpu or not ddp_mode else world,
                                device=a.device),
                    fwd_ms_per_img=None if head["fwd_ms_per_img"] is None else round(head["fwd_ms_per_img"], 3),
                    peak_hbm_gb=head["peak_hbm_gb"], host=head["host"], model_tflops=flops_block(head),
                    roofline=head["roofline"], roofline_bwd_gemm=head["roofline_bwd_gemm"], parity=parity.get(a.prec),
                    fast_mode=fast_rec, full_fp32_mode=full, ref_batch=head["ref_batch"], torch_rocm_baseline=torch_base, cpu_baseline=cpu,
                    git=_source_id())
        print(json.dumps(line), flush=True)
    if ddp_mode:
        dist.barrier()                            # rank 0's extra legs (second mode, JSON line) end before any rank tears the group down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
