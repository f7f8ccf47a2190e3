"""Optimizer step of the reference training loop (TaskPrompter/utils/train_utils.py:47-51 and :139-150) on the HIP kernels:
`clip_grad_norm_(params, max_norm)` + `torch.optim.Adam.step()` fused into two multi-tensor launches (mtt_grad_sqnorm,
mtt_adam_step), and the reference's polynomial learning-rate schedule.

`FusedClipAdam` keeps torch.optim.Adam's state layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so a reference
checkpoint's optimizer state loads with `load_state_dict` and vice versa.
"""
import math

import numpy as np
import torch

from . import ops


class FusedClipAdam(torch.optim.Optimizer):
    """Adam (torch.optim.Adam semantics: L2 weight decay added to the gradient, bias-corrected moments, no amsgrad) preceded by
    global-norm gradient clipping.  `step()` returns the total gradient norm before clipping (like clip_grad_norm_)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.0, capturable=False):
        """capturable=True: the step-dependent scalars (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)) are read by the kernel from a device
        pair (`mtt_adam_desc.hyper`) instead of being launch constants, so that a step captured in a hipGraph (graphs.GraphedTrainStep) can be
        replayed with the next step count and learning rate; `prepare_replay()` advances them."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = float(max_norm)
        self.capturable = bool(capturable)
        self._hyper = {}           # (group, part) -> device fp32 [2]
        self._hyper_pinned = {}    # (group, part, slot) -> pinned fp32 [2]
        self._captured = None      # work list of the captured step: [(group index, part, params)]
        self._tables = {}
        self._pinned = {}          # (group, slot) -> pinned int64 staging buffer of gradient pointers (rotated: copies are asynchronous)
        self._staged = {}          # pinned-buffer key -> event recorded after the stream-ordered copy that READS the buffer
        self._slot = 0

    def _reuse(self, key):
        """A pinned staging buffer is read by an ASYNCHRONOUS copy: before the host rewrites slot `key`, wait until the copy enqueued the
        last time this slot was used has executed (the ring is 4 deep, the host can run further ahead than that — it queues whole
        steps in about a millisecond and never synchronises)."""
        ev = self._staged.get(key)
        if ev is not None:
            ev.synchronize()

    def _staged_copy(self, key, buf, dst=None, dev=None):
        out = buf.to(dev, non_blocking=True) if dst is None else dst.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._staged[key] = ev
        return out

    def _grad_ptrs(self, gi, grads, dev):
        """Device array of the gradients' addresses.  The table is staged through a ring of pinned host buffers and copied with
        non_blocking=True: a pageable copy would block the host until the stream drains and cost the CPU its run-ahead."""
        n = len(grads)
        if dev.type != "cuda":
            return torch.from_numpy(np.array([g.data_ptr() for g in grads], dtype=np.int64))
        key = (gi, "captured" if self._capturing(dev) else self._slot % 4)     # a captured copy re-reads ITS host buffer on every replay
        buf = self._pinned.get(key)
        if buf is None or buf.numel() != n:
            if self._capturing(dev):
                raise RuntimeError("FusedClipAdam: run at least one eager step() with the same gradients before capturing (pinned staging "
                                   "buffers cannot be allocated while a stream is capturing)")
            buf = torch.empty(n, dtype=torch.int64).pin_memory()
            self._pinned[key] = buf
        if self.capturable and (gi, "captured") not in self._pinned:
            self._pinned[(gi, "captured")] = torch.empty(n, dtype=torch.int64).pin_memory()      # host allocations are illegal while capturing
        if self._capturing(dev):
            buf.numpy()[:] = [g.data_ptr() for g in grads]
            return buf.to(dev, non_blocking=True)
        self._reuse(("ptr",) + key)
        buf.numpy()[:] = [g.data_ptr() for g in grads]
        return self._staged_copy(("ptr",) + key, buf, dev=dev)

    def _group_tables(self, gi, part, plist):
        """Device pointer / chunk tables of one launch group.  The key covers the parameter AND the moment tensors' addresses:
        `load_state_dict`, `add_param_group` or a state reset replace the moment tensors, and a table built before that would
        make the kernel update freed memory.  The table holds references to the tensors whose addresses it caches."""
        st = [self.state[p] for p in plist]
        key = (tuple(p.data_ptr() for p in plist), tuple(s["exp_avg"].data_ptr() for s in st),
               tuple(s["exp_avg_sq"].data_ptr() for s in st))
        tab = self._tables.get((gi, part))
        if tab is not None and tab["key"] == key:
            return tab
        dev = plist[0].device
        chunk = ops.adam_chunk()
        numel = [p.numel() for p in plist]
        ct, co = [], []
        for t, n in enumerate(numel):
            for off in range(0, n, chunk):
                ct.append(t)
                co.append(off)

        def ptrs(ts):
            return torch.from_numpy(np.array([t.data_ptr() for t in ts], dtype=np.int64)).to(dev)
        keep = [s["exp_avg"] for s in st] + [s["exp_avg_sq"] for s in st]
        tab = dict(key=key, keep=keep, params=ptrs(plist), exp_avg=ptrs([s["exp_avg"] for s in st]),
                   exp_avg_sq=ptrs([s["exp_avg_sq"] for s in st]),
                   numel=torch.tensor(numel, dtype=torch.int64, device=dev), chunk_tensor=torch.tensor(ct, dtype=torch.int32, device=dev),
                   chunk_off=torch.tensor(co, dtype=torch.int64, device=dev), n_chunks=len(ct))
        self._tables[(gi, part)] = tab
        return tab

    def load_state_dict(self, state_dict):
        """torch's loader REPLACES the moment tensors.  A captured step (graphs.GraphedTrainStep) holds their addresses, so when one exists
        the loaded values are copied INTO the tensors the graph updates and those stay in place; otherwise only the pointer tables are dropped."""
        if self._captured is None:
            super().load_state_dict(state_dict)
            self._tables.clear()
            return
        pinned = {p: (self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for _, _, plist in self._captured for p in plist}
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for p, (m, v) in pinned.items():
                st = self.state[p]
                if "exp_avg" in st:                         # a fresh / partial optimizer state has no entry for p: zero moments, step 0
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                else:
                    m.zero_()
                    v.zero_()
                    st["step"] = torch.zeros((), dtype=torch.float32)
                st["exp_avg"], st["exp_avg_sq"] = m, v
        # the captured launches take ONE (step-dependent) scalar pair per partition: its parameters must share a step count
        for _, _, plist in self._captured:
            if len({float(self.state[p]["step"]) for p in plist}) > 1:
                raise RuntimeError("FusedClipAdam.load_state_dict: parameters of one captured launch group carry different step counts; "
                                   "re-capture the step (GraphedTrainStep) after loading this state")

    @staticmethod
    def _capturing(dev):
        return dev.type == "cuda" and torch.cuda.is_current_stream_capturing()

    def _set_hyper(self, key, dev, group, t):
        """-> device pair {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} of one launch group, refreshed by a stream-ordered copy from a ring of
        pinned host pairs (never while a stream is capturing: a captured copy would re-read the same host pair on every replay)."""
        hy = self._hyper.get(key)
        if hy is None:
            hy = self._hyper[key] = torch.zeros(2, dtype=torch.float32, device=dev)
        if self._capturing(dev):
            return hy
        b1, b2 = group["betas"]
        vals = (group["lr"] / (1.0 - b1 ** t), 1.0 / math.sqrt(1.0 - b2 ** t))
        if dev.type != "cuda":
            hy.copy_(torch.tensor(vals, dtype=torch.float32))
            return hy
        pk = key + (self._slot % 4,)
        buf = self._hyper_pinned.get(pk)
        if buf is None:
            buf = self._hyper_pinned[pk] = torch.empty(2, dtype=torch.float32).pin_memory()
        self._reuse(("hyper",) + pk)
        buf[0], buf[1] = vals
        self._staged_copy(("hyper",) + pk, buf, dst=hy)
        return hy

    def prepare_replay(self):
        """Before replaying a captured step: advance every captured parameter's step count and enqueue the new step-dependent scalars
        (and whatever the learning-rate scheduler set in `param_groups`) on the current stream."""
        if self._captured is None:
            raise RuntimeError("FusedClipAdam.prepare_replay: no captured step (call step() under stream capture with capturable=True first)")
        self._slot += 1
        for gi, part, plist in self._captured:
            steps = [self.state[p]["step"] for p in plist]
            torch._foreach_add_(steps, 1)
            self._set_hyper((gi, part), plist[0].device, self.param_groups[gi], float(steps[0]))

    def after_replay(self):
        """After a replay: the graph wrote the parameters behind torch's back — same bookkeeping as the end of step()."""
        for _, _, plist in self._captured:
            torch.autograd.graph.increment_version(plist)
        ops.bump_param_epoch([p for _, _, plist in self._captured for p in plist])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []
        for gi, group in enumerate(self.param_groups):
            parts = {}                                    # step count -> parameters (torch.optim.Adam keeps one `step` per parameter)
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedClipAdam: fp32 contiguous parameters and gradients only")
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                parts.setdefault(float(st["step"]), []).append(p)
            for part, (t0, plist) in enumerate(sorted(parts.items())):
                tab = self._group_tables(gi, part, plist)
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in plist]
                gp = self._grad_ptrs((gi, part), grads, plist[0].device)
                work.append((group, plist, tab, grads, gp, t0 + 1.0, (gi, part)))
        if not work:
            return loss
        self._slot += 1
        dev = work[0][1][0].device
        capturing = self._capturing(dev)
        if capturing:
            if not self.capturable:
                raise RuntimeError("FusedClipAdam: construct with capturable=True to capture step() in a graph")
            self._captured = [(key[0], key[1], plist) for _, plist, _, _, _, _, key in work]
        total_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        base = dict(max_norm=0.0, step_size=0.0, beta1=0.0, beta2=0.0, eps=0.0, weight_decay=0.0, inv_sqrt_bc2=0.0)
        ws = ops.workspace(max(w[2]["n_chunks"] for w in work), dev)      # per-chunk partial sums (deterministic clip norm)
        for group, plist, tab, grads, gp, t, key in work:
            ops.call("grad_sqnorm", ws=ws, grads=gp, params=tab["params"], exp_avg=tab["exp_avg"], exp_avg_sq=tab["exp_avg_sq"], numel=tab["numel"],
                     chunk_tensor=tab["chunk_tensor"], chunk_off=tab["chunk_off"], n_chunks=tab["n_chunks"], xargs=[total_sq], **base)
        for group, plist, tab, grads, gp, t, key in work:
            if not capturing:                                     # a capture records the launch; nothing is executed, no step is taken
                torch._foreach_add_([self.state[p]["step"] for p in plist], 1)
            b1, b2 = group["betas"]
            hyper = self._set_hyper(key, dev, group, t) if self.capturable else None
            ops.call("adam_step", hyper=hyper, grads=gp, params=tab["params"], exp_avg=tab["exp_avg"], exp_avg_sq=tab["exp_avg_sq"], numel=tab["numel"],
                     chunk_tensor=tab["chunk_tensor"], chunk_off=tab["chunk_off"], n_chunks=tab["n_chunks"], max_norm=self.max_norm,
                     step_size=group["lr"] / (1.0 - b1 ** t), beta1=b1, beta2=b2, eps=group["eps"], weight_decay=group["weight_decay"],
                     inv_sqrt_bc2=1.0 / math.sqrt(1.0 - b2 ** t), xargs=[total_sq if self.max_norm > 0 else None])
            # the kernel wrote the parameters through raw pointers: tell torch (autograd's saved-tensor checks, any cache keyed
            # on `_version`) and the pack cache of ops.py that they changed
            if not capturing:
                torch.autograd.graph.increment_version(plist)
        ops.bump_param_epoch([p for g_ in self.param_groups for p in g_["params"]])
        self.last_grad_norm = total_sq.sqrt()
        return loss if loss is not None else self.last_grad_norm


class PolynomialLR(torch.optim.lr_scheduler._LRScheduler):
    """The reference's per-iteration schedule (TaskPrompter/utils/train_utils.py:139-150): the distance of every group's learning rate
    above `min_lr` decays as (1 - iteration / max_iterations) ** gamma.  `scheduler.step()` is called once per training iteration."""

    def __init__(self, optimizer, max_iterations, gamma=0.9, min_lr=0.0, last_epoch=-1):
        self.max_iterations, self.gamma, self.min_lr = int(max_iterations), float(gamma), float(min_lr)
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        progress = min(self.last_epoch, self.max_iterations) / float(self.max_iterations)     # clamped: never a negative base
        decay = (1.0 - progress) ** self.gamma
        return [self.min_lr + (base - self.min_lr) * decay for base in self.base_lrs]
