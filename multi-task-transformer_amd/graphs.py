"""Whole-step hipGraph capture of the reference training iteration (TaskPrompter/utils/train_utils.py:32-51: forward, criterion, backward,
clip + Adam) for the launch-bound regime.  One iteration is ~4 300 kernel launches; at the reference's own per-GPU batch of 2 the Python /
ctypes cost of issuing them (~50 ms) exceeds the GPU time several times over, so the step is recorded ONCE into a graph and replayed.
At the throughput-optimal batch the GPU is the bottleneck and eager launches are already hidden; the graph buys nothing there.

What makes the step capturable: every kernel of the C ABI is launched on torch's current stream, no entry point synchronises or reads
device memory from the host, the criterion normalises on the device, the weight re-packing that follows an optimizer step is part of
the recorded forward, and FusedClipAdam(capturable=True) reads its step-dependent scalars from device memory
(`mtt_adam_desc.hyper`).  Single process only: DistributedDataParallel's bucketed all-reduce is not captured here.
"""
import torch

from . import ops


class GraphedTrainStep:
    """step = GraphedTrainStep(model, criterion, optimizer, x, targets); loss = step(x, targets) replays the recorded iteration on new
    inputs of the same shapes.  `loss` is a device scalar that the next replay overwrites."""

    def __init__(self, model, criterion, optimizer, x, targets, warmup=2, loss_key="total"):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs FusedClipAdam(..., capturable=True)")
        if not x.is_cuda:
            raise ValueError("GraphedTrainStep records a HIP graph: inputs must live on the GPU")
        if isinstance(model, torch.nn.parallel.DistributedDataParallel):
            raise ValueError("GraphedTrainStep is single-process: DistributedDataParallel's bucketed all-reduce is not captured")
        self.model, self.criterion, self.optimizer, self.loss_key = model, criterion, optimizer, loss_key
        self.x = x.clone()
        self.targets = {k: v.clone() for k, v in targets.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                         # eager iterations: lazy initialisation (LDS opt-ins, optimizer state, workspaces)
            for _ in range(max(1, warmup)):
                self._iteration()
        torch.cuda.current_stream().wait_stream(side)
        optimizer.zero_grad(set_to_none=True)                 # gradients are re-created inside the graph's memory pool
        ops.finalize_packs(x.device)                          # the recorded forward's weight refresh is then one bare launch
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._iteration()
        self.replays = 0

    def _iteration(self):
        out = self.model(self.x)
        loss = self.criterion(out, self.targets)[self.loss_key]
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return loss

    def __call__(self, x=None, targets=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if targets is not None:
            for k, v in targets.items():
                self.targets[k].copy_(v, non_blocking=True)
        self.optimizer.prepare_replay()
        self.graph.replay()
        self.optimizer.after_replay()
        self.replays += 1
        return self.loss


def clear():
    """Drop the cached packs / gradient-scatter tables (call after deleting a GraphedTrainStep: lazily cached one-off packs built under
    its capture live in the released graph's memory pool)."""
    ops.clear_pack_cache()
