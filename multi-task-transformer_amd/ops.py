"""Forward/backward building blocks of the hot path on top of the C ABI (`_lib.call`).

Data layout: every activation is a token-major (NHWC) 2-D or 3-D torch tensor `[..., rows, ld]` whose
last dim is the channel pitch `ld = pitch(C)`; channels >= C are kept at zero.  `Prec` selects the
arithmetic: BF16 = bf16 storage + bf16 MFMA (throughput path), X3 = fp32 storage + split-bf16 x3 MFMA
(fp32-class accuracy: the 1e-3 parity gate).  The residual stream, logits side channels, statistics and
all gradients of parameters are fp32 in both modes.

Nothing here falls back to torch math: every function launches a HIP kernel through `_lib.call`.
torch is used for allocation, views and the (tiny) parameter re-packing.
"""
import weakref

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_RELU, BF16, F32, OP_CONV_K, OP_CONV_R, OP_K, OP_R, PREC_BF16, PREC_X3, SPLIT)


PITCH32_FROM = 160         # channel counts >= this take a pitch that is a multiple of 32 (see pitch); 1 << 30 = the multiple-of-8 pitch everywhere.
                           # (160: below it the rounding costs > 10 % of a map, and InvPT's 144-channel stage keeps its width as its pitch)


def pitch(n):
    """The channel pitch of a map with n channels (`pad8` until round 6, when every pitch was the next multiple of 8): the next multiple of 8 — of 32 from PITCH32_FROM channels on, so that the K loop of a GEMM or
    3x3 implicit GEMM reading the map is whole 32-deep LDS-DMA steps (176 -> 192, 232 -> 256, 456 -> 480, 784 -> 800: the split-plane / ring
    kernels instead of the register-staged ones).  Every kernel writes the channels C..ld-1 of its output maps as zeros."""
    return (n + 31) // 32 * 32 if n >= PITCH32_FROM else (n + 7) // 8 * 8


pad8 = pitch               # historical name (tools / tests)


DEFAULT_PREC = "x3f"        # arithmetic mode of a model whose config does not name one (`p.mtt_prec`): the tolerance-compliant mode


class Prec:
    """Arithmetic mode: name in {'bf16', 'x3', 'x3f'}.

    bf16 — bf16 storage + bf16 MFMA (throughput).  x3 — fp32 storage, every product as 3 bf16 MFMAs on split operands, forward AND
    backward (fp32-class everywhere; register-staged kernels).  x3f — the x3 FORWARD arithmetic (same 3-MFMA products, 1e-3 per-head
    parity with the reference's fp32 forward) with the encoder's GEMM operands stored as pre-split hi / lo bf16 planes (`Split`) so
    that they stream through the LDS-DMA GEMM kernel, and a bf16 BACKWARD on the hi planes (`.bwd`): mixed-precision training whose
    forward outputs match the reference."""

    def __init__(self, name):
        assert name in ("bf16", "x3", "x3f")
        self.name = name
        self.code = PREC_BF16 if name == "bf16" else PREC_X3
        self.adt = torch.bfloat16 if name == "bf16" else torch.float32
        self.split = name == "x3f"                       # encoder operands as Split planes (LDS-DMA x3 GEMM)
        self.bwd = Prec("bf16") if name == "x3f" else self   # arithmetic of the backward pass

    def __repr__(self):
        return f"Prec({self.name})"


class Split:
    """An fp32-class activation / weight as two bf16 planes x = hi + lo (MTT_SPLIT) of identical shape and strides: hi = bf16(x),
    lo = bf16(x - hi).  The hi plane alone is an ordinary bf16 tensor — what the bf16 backward of the x3f mode reads; the planes are
    separate allocations so that saving `hi` for the backward does not keep `lo` alive."""

    def __init__(self, hi, lo):
        assert hi.dtype == torch.bfloat16 and lo.dtype == torch.bfloat16 and hi.shape == lo.shape and hi.stride() == lo.stride()
        self.hi, self.lo = hi, lo

    @property
    def shape(self):
        return self.hi.shape

    @property
    def device(self):
        return self.hi.device

    @staticmethod
    def empty(shape, device):
        return Split(torch.empty(tuple(shape), dtype=torch.bfloat16, device=device), torch.empty(tuple(shape), dtype=torch.bfloat16, device=device))

    def __getitem__(self, idx):
        return Split(self.hi[idx], self.lo[idx])

    def float(self):
        return self.hi.float() + self.lo.float()


def dtype_code(t):
    return SPLIT if isinstance(t, Split) else _lib.dtype_code(t)


def _hi(t):
    return t.hi if isinstance(t, Split) else t


def split_cast(x2d, cols=None):
    """fp32 [rows, ld] -> Split [rows, pitch(cols)] (mtt_split_cast)."""
    rows = x2d.shape[0]
    cols = cols or x2d.shape[1]
    out = Split.empty((rows, pitch(cols)), x2d.device)
    call("split_cast", args=[x2d, out.hi, out.lo, rows, cols, x2d.stride(0), pitch(cols)])
    return out


GEMM_VARIANT = None            # A/B hook (bench.py --gemm-variant): mtt_gemm_desc.variant for every call that leaves it at AUTO


def call(name, **kw):          # single indirection point (tests monkeypatch this with the CPU emulator)
    if GEMM_VARIANT is not None and name == "gemm" and not kw.get("variant"):
        kw["variant"] = GEMM_VARIANT
    return _lib.call(name, **kw)


def adam_chunk():
    return int(_lib.load().mtt_adam_chunk())


# ---------------------------------------------------------------------------------------------
# parameter packing (cached per parameter version)
# ---------------------------------------------------------------------------------------------
_pack_cache = {}
_param_epoch = 0


_epoch_origin = {}           # epoch -> data_ptrs of the parameters that bump named, or None = unknown writer (assume any parameter)


def bump_param_epoch(params=None):
    """Invalidate every cached pack: called by optimizers that write parameters through raw device pointers (optim.FusedClipAdam),
    where torch's own `_version` counter cannot see the update.  (FusedClipAdam also bumps `_version`; the epoch is the second,
    independent line of defence so that a stale pack can never be served after a step.)  `params`: the tensors that were written, when
    the caller knows them — the next refresh then bumps the autograd versions of exactly the packs built from them (plus those whose
    `_version` moved); without it the origin of the update is unknown and EVERY refreshed pack is treated as modified (an
    epoch-only bump by a raw-pointer writer must still make a pending backward that saved the old pack fail loudly)."""
    global _param_epoch
    _param_epoch += 1
    _epoch_origin[_param_epoch] = None if params is None else frozenset(p.data_ptr() for p in params)
    for k in [k for k in _epoch_origin if k < _param_epoch - 256]:
        del _epoch_origin[k]


def _epoch_touched(old_epoch, ptrs):
    """did a bump since `old_epoch` (possibly) write one of the parameters at `ptrs`?"""
    for ep in range(old_epoch + 1, _param_epoch + 1):
        org = _epoch_origin.get(ep, None)
        if org is None or any(q in org for q in ptrs):
            return True
    return False


def _cached(key, params, build):
    """Pack cache keyed by `key`, valid while every source tensor is the same object with the same storage and version."""
    ver = (_param_epoch, tuple((p.data_ptr(), p._version) for p in params))
    hit = _pack_cache.get(key)
    if hit is not None and hit[0] == ver and all(r() is p for r, p in zip(hit[2], params)):
        return hit[1]
    val = build()
    if len(_pack_cache) > 4096:                       # drop packs of parameters that no longer exist (models built and discarded)
        for k in [k for k, v in _pack_cache.items() if any(r() is None for r in v[2])]:
            del _pack_cache[k]
    _pack_cache[key] = (ver, val, [weakref.ref(p) for p in params])
    return val


def cast2d(src, rows, cols, lds, dst_dtype, ldd=None, zero_pad=True):
    ldd = ldd or pitch(cols)
    dst = torch.empty(rows, ldd, dtype=dst_dtype, device=src.device)
    call("cast2d", args=[src, dst, rows, cols, lds, ldd, dtype_code(src), dtype_code(dst), 1 if zero_pad else 0])
    return dst


def cast_rows(src2d, dst_dtype):
    """[rows, ld] -> a copy in `dst_dtype` with the same pitch (all ld columns, padding included): the 8-elements-per-lane cast kernel
    (mtt_rowscale_cast without scales) when the pitch allows it — the generic cast2d kernel works element by element."""
    rows, ld = src2d.shape
    if ld % 8 or src2d.stride(0) % 8 or src2d.stride(1) != 1:
        return cast2d(src2d, rows, ld, src2d.stride(0), dst_dtype, ldd=ld)
    dst = torch.empty(rows, ld, dtype=dst_dtype, device=src2d.device)
    call("rowscale_cast", args=[src2d, dst, rows, ld, src2d.stride(0), ld, dtype_code(src2d), dtype_code(dst), None, 0, 0])
    return dst


def pack_matrix(w2d, prec):
    """[N, K] fp32 -> [N, pitch(K)] in the activation dtype (zero padded); the lazily cached one-off packs (InvPT, deconv heads)."""
    N, K = w2d.shape
    if prec.adt == torch.float32 and K == pitch(K) and w2d.is_contiguous():
        return w2d
    return cast2d(w2d.contiguous(), N, K, K, prec.adt)


# ---------------------------------------------------------------------------------------------
# persistent packs, refreshed by ONE mtt_segcopy launch per parameter update
# ---------------------------------------------------------------------------------------------
_ITEM = {torch.float32: 4, torch.bfloat16: 2}


def segment(src, src_off, dst, dst_off, n, s, d, dst_lo=None):
    """One mtt_segcopy segment: the box n = (n0, n1, n2) read from `src` (tensor, element offset src_off, element strides s) and written to
    `dst` (element offset dst_off, strides d; `dst_lo` = the lo plane of a Split destination).  Strides of singleton dimensions are ignored."""
    n = tuple(int(v) for v in n)
    s = tuple(int(v) if k > 1 else 0 for v, k in zip(s, n))
    d = tuple(int(v) if k > 1 else 0 for v, k in zip(d, n))
    sa = src.data_ptr() + src_off * _ITEM[src.dtype]
    da = dst.data_ptr() + dst_off * 2 if dst.dtype == torch.bfloat16 else dst.data_ptr() + dst_off * 4
    la = dst_lo.data_ptr() + dst_off * 2 if dst_lo is not None else 0
    ddt = SPLIT if dst_lo is not None else _lib.dtype_code(dst)
    vec = int(n[2] % 4 == 0 and (s[2] == 1 or n[2] == 1) and (d[2] == 1 or n[2] == 1) and n[2] > 1 and all(v % 4 == 0 for v in s[:2] + d[:2])
              and sa % 16 == 0 and da % 16 == 0 and la % 16 == 0)
    if n[1] >= 16 and n[2] >= 16 and s[1] == 1 and d[2] == 1 and s[2] != 1:      # a transposition: tiled through LDS by the kernel
        tiles = n[0] * ((n[1] + 63) // 64) * ((n[2] + 63) // 64)
        return [sa, da, la, tiles * 4096, n[1], n[2], s[0], 1, s[2], d[0], d[1], 1, _lib.dtype_code(src), ddt, 0, 1]
    return [sa, da, la, n[0] * n[1] * n[2], n[1], n[2], s[0], s[1], s[2], d[0], d[1], d[2], _lib.dtype_code(src), ddt, vec, 0]


def _seg_elems(r):
    """logical element count of a segment() row (its field 3 is the chunked WORK size: padded 64 x 64 tiles for a transposing segment)."""
    if r[15]:
        n1, n2 = r[4], r[5]
        return r[3] // 4096 // (((n1 + 63) // 64) * ((n2 + 63) // 64)) * n1 * n2
    return r[3]


class SegProgram:
    """Device tables of an mtt_segcopy launch over a list of `segment(...)` rows (absolute addresses: the tensors must outlive it)."""

    def __init__(self, rows, device):
        import numpy as np
        chunk = int(_lib.load().mtt_segcopy_chunk())
        rows = [r for r in rows if r[3] > 0]
        seg, off = [], []
        for i, r in enumerate(rows):
            for o in range(0, r[3], chunk):
                seg.append(i)
                off.append(o)
        self.n_chunks = len(seg)
        self.nbytes = sum(r[3] for r in rows)
        if self.n_chunks:
            self.table = torch.from_numpy(np.asarray(rows, dtype=np.int64).reshape(-1, 16)).to(device)
            self.chunk_seg = torch.from_numpy(np.asarray(seg, dtype=np.int32)).to(device)
            self.chunk_off = torch.from_numpy(np.asarray(off, dtype=np.int64)).to(device)

    def run(self, src_base=0, dst_base=0):
        if self.n_chunks:
            call("segcopy", table=self.table, chunk_seg=self.chunk_seg, chunk_off=self.chunk_off, n_chunks=self.n_chunks,
                 src_base=src_base, dst_base=dst_base)


class _Pack:
    __slots__ = ("value", "params", "ver", "rows")


_packs = {}                  # key -> _Pack: persistent operand layouts of parameters
_packs_gen = 0               # bumped whenever the set of packs changes
_packs_prog = {}             # device -> (generation, SegProgram over the rows of every pack on that device)
pack_refreshes = 0           # number of whole-registry refresh launches (tests / launch accounting)


def _capturing(device):
    return device.type == "cuda" and torch.cuda.is_current_stream_capturing()


def _live_packs(device):
    """[(pack, parameters)] of the registered packs on `device`; entries whose parameters died or were re-allocated are dropped (their
    next lookup rebuilds them)."""
    global _packs_gen
    live = []
    for k in list(_packs):
        e = _packs[k]
        ps = [r() for r in e.params]
        if any(p is None for p in ps) or tuple(p.data_ptr() for p in ps) != tuple(v[0] for v in e.ver[1]):
            del _packs[k]
            _packs_gen += 1
        elif ps[0].device == device:
            live.append((e, ps))
    return live


def _packs_program(device, live):
    hit = _packs_prog.get(device)
    if hit is None or hit[0] != _packs_gen:
        if _capturing(device):
            raise RuntimeError("the set of weight packs changed under stream capture (run a warm-up step before capturing)")
        hit = _packs_prog[device] = (_packs_gen, SegProgram([r for e, _ in live for r in e.rows], device))
    return hit[1]


def finalize_packs(device):
    """Build the whole-registry refresh program now (host -> device table copy) so that the next refresh is a bare kernel launch:
    called before a stream capture whose recorded forward must contain that launch (graphs.GraphedTrainStep)."""
    _packs_program(device, _live_packs(device))


def _refresh_packs(device):
    """Parameters moved on (optimizer step): re-copy EVERY registered pack of this device with one launch."""
    global pack_refreshes
    live = _live_packs(device)
    _packs_program(device, live).run()
    pack_refreshes += 1
    bumped = []
    for e, ps in live:
        new = _pver(ps)
        # this pack's own parameters moved on: their `_version` changed, or a raw-pointer writer named them in bump_param_epoch, or a
        # writer of unknown origin bumped the epoch.  (The epoch also changes when ANOTHER model in the process stepped through
        # FusedClipAdam, which names its parameters: that pack's bytes are rewritten identically, its pending backward stays valid.)
        if e.ver[1] != new[1] or (e.ver[0] != new[0] and _epoch_touched(e.ver[0], [v[0] for v in new[1]])):
            v = e.value
            bumped += [v.hi, v.lo] if isinstance(v, Split) else [v]
        e.ver = new
    # the packs were rewritten IN PLACE by a raw launch: tell autograd, so that a backward still holding one of them from an earlier
    # forward (ctx.save_for_backward) raises "modified by an inplace operation" instead of silently using the new weights
    if bumped and not _capturing(device):
        torch.autograd.graph.increment_version(bumped)


def _pver(params):
    return (_param_epoch, tuple((p.data_ptr(), p._version) for p in params))


def seg_pack(key, params, alloc, rows_of, check=None):
    """Persistent pack `key` of the parameters `params`: `alloc()` -> zero-initialised destination (tensor or Split), `rows_of(value)` ->
    its segcopy rows.  Served from the registry while the parameters are unchanged; after a parameter update the FIRST stale lookup
    refreshes every registered pack in one launch (mtt_segcopy), so a training step re-packs all weights with one kernel instead of one
    cast / copy per weight.  Padding (outside the boxes) is written once, by `alloc`."""
    global _packs_gen
    ver = _pver(params)
    e = _packs.get(key)
    if e is not None and len(e.params) == len(params) and all(r() is p for r, p in zip(e.params, params)) \
            and tuple(v[0] for v in e.ver[1]) == tuple(v[0] for v in ver[1]):
        if e.ver != ver:
            _refresh_packs(params[0].device)
            e = _packs.get(key)
        if e is not None and e.ver == ver:
            return e.value
    if _capturing(params[0].device):
        raise RuntimeError("a weight pack would be built under stream capture (run a warm-up step before capturing)")
    if check is not None:
        check()                                     # layout / dtype of the sources: verified when the entry is (re)built, not per lookup
    with torch.no_grad():
        e = _Pack()
        e.value = alloc()
        e.rows = rows_of(e.value)
        e.params = [weakref.ref(p) for p in params]
        e.ver = ver
        SegProgram(e.rows, params[0].device).run()
    _packs[key] = e
    _packs_gen += 1
    return e.value


_unpack_progs = {}


def unpack_grads(src, key, shapes, rows_of, partial=False):
    """Packed fp32 weight gradients `src` -> one contiguous fp32 gradient per parameter shape in `shapes` (views of ONE flat buffer),
    written by ONE mtt_segcopy launch: the inverse of the pack layouts above (column padding dropped, tap-major -> [Co, Ci, 3, 3], padded
    concatenations split).  `rows_of(src, flat, offsets)` -> segcopy rows built with `segment(src, ..., flat, offsets[z] + ...)`; they are
    stored relative to the two buffers, so the cached program serves every later step's fresh buffers.  Unless `partial`, every
    destination element must be covered by a segment (the flat buffer is then not zeroed)."""
    import math
    sizes = [math.prod(sh) for sh in shapes]
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + (n + 3) // 4 * 4)                      # 16-byte aligned starts (vector path)
    flat = (torch.zeros if partial else torch.empty)(offs[-1], dtype=torch.float32, device=src.device)
    k = (key, tuple(src.shape), tuple(src.stride()), tuple(tuple(sh) for sh in shapes), src.device)
    prog = _unpack_progs.get(k)
    if prog is None:
        if _capturing(src.device):
            raise RuntimeError("a gradient-scatter table would be built under stream capture (run a warm-up step before capturing)")
        rows = rows_of(src, flat, offs)
        assert src.data_ptr() % 16 == 0 and flat.data_ptr() % 16 == 0
        for r in rows:
            r[0] -= src.data_ptr()
            r[1] -= flat.data_ptr()
        assert partial or sum(_seg_elems(r) for r in rows) == sum(sizes), "gradient scatter must cover every parameter element exactly once"
        prog = _unpack_progs[k] = SegProgram(rows, src.device)
    assert src.data_ptr() % 16 == 0
    prog.run(src_base=src.data_ptr(), dst_base=flat.data_ptr())
    return [flat[o:o + n].view(sh) for o, n, sh in zip(offs, sizes, shapes)]


def clear_pack_cache():
    global _packs_gen
    _pack_cache.clear()
    _packs.clear()
    _unpack_progs.clear()
    _packs_prog.clear()
    _packs_gen += 1


def _w2d(w):
    return w.shape[0], w.numel() // w.shape[0]


def _check_sources(weights, ok):
    """Pack sources are read by address (mtt_segcopy): they must be contiguous fp32 tensors of one shape."""
    for w in weights:
        if not (w.is_contiguous() and w.dtype == torch.float32 and ok(w)):
            raise ValueError(f"weight pack: parameter of shape {tuple(w.shape)}, strides {tuple(w.stride())}, {w.dtype} is not a contiguous fp32 "
                             "tensor of the layer's shape (channels_last / sliced / half-precision parameters are not supported: call "
                             ".contiguous().float() on the module's parameters)")


def pack_linear(weights, prec, tag):
    """List of Z parameters [N, K] (or 1x1 conv [N, K, 1, 1]) -> one [Z, N, Kp] buffer in the activation dtype (zero padded)."""
    N, K = _w2d(weights[0])
    Kp, Z = pitch(K), len(weights)
    if prec.adt == torch.float32 and Z == 1 and K == Kp and weights[0].is_contiguous():
        return weights[0].detach().reshape(1, N, K)               # fp32 storage: the parameter itself is the operand
    return seg_pack((tag, prec.name, tuple(id(w) for w in weights)), list(weights),
                    lambda: torch.zeros(Z, N, Kp, dtype=prec.adt, device=weights[0].device),
                    lambda buf: [segment(w, 0, buf, z * N * Kp, (1, N, K), (0, K, 1), (0, Kp, 1)) for z, w in enumerate(weights)],
                    check=lambda: _check_sources(weights, lambda w: _w2d(w) == (N, K)))


def pack_linear_T(weight, dtype, tag):
    """Parameter [N, K] -> its transpose [K, N] in `dtype`: the reduction-contiguous operand of the input-gradient GEMM dx = dy @ W."""
    N, K = _w2d(weight)
    return seg_pack((tag, 'wT', dtype, id(weight)), [weight], lambda: torch.zeros(K, N, dtype=dtype, device=weight.device),
                    lambda buf: [segment(weight, 0, buf, 0, (1, N, K), (0, K, 1), (0, 1, N))] if min(N, K) < 16 else
                                [segment(weight, 0, buf, 0, (1, K, N), (0, 1, K), (0, N, 1))],
                    check=lambda: _check_sources([weight], lambda w: True))


def pack_linear_split(weights, tag):
    """List of Z parameters [N, K] -> Split [Z, N, pitch(K)]: pre-split weight planes for the LDS-DMA x3 GEMM (x3f mode)."""
    N, K = _w2d(weights[0])
    Kp, Z = pitch(K), len(weights)
    dev = weights[0].device
    return seg_pack((tag, 'split', tuple(id(w) for w in weights)), list(weights),
                    lambda: Split(torch.zeros(Z, N, Kp, dtype=torch.bfloat16, device=dev), torch.zeros(Z, N, Kp, dtype=torch.bfloat16, device=dev)),
                    lambda sp: [segment(w, 0, sp.hi, z * N * Kp, (1, N, K), (0, K, 1), (0, Kp, 1), dst_lo=sp.lo) for z, w in enumerate(weights)],
                    check=lambda: _check_sources(weights, lambda w: _w2d(w) == (N, K)))


def pack_conv3(weights, prec, tag, transpose=False):
    """List of Z conv weights [Co, Ci, 3, 3] -> [Z, Co, 9*Cip] with k = tap*Cip + ci (taps row-major).
    transpose=True packs the dgrad operand [Z, Ci, 9*Cop] with k = tap*Cop + co."""
    Co, Ci = weights[0].shape[:2]
    if prec.split and not transpose and split_conv_ok(Ci, Co):
        # x3f forward: pre-split planes -> conv3x3 runs the split-plane implicit-GEMM kernel (an fp32 input is split by one pass first)
        return pack_conv3_split(weights, tag)
    R, Cin = (Ci, Co) if transpose else (Co, Ci)
    Cp, Z = pitch(Cin), len(weights)
    # logical box (r, tap, c): source W[co, ci, tap] has strides (Ci*9, 9, 1) over (co, ci, tap)
    s = (9, 1, Ci * 9) if transpose else (Ci * 9, 1, 9)
    return seg_pack((tag, prec.name, transpose, tuple(id(w) for w in weights)), list(weights),
                    lambda: torch.zeros(Z, R, 9 * Cp, dtype=prec.adt, device=weights[0].device),
                    lambda buf: [segment(w, 0, buf, z * R * 9 * Cp, (R, 9, Cin), s, (9 * Cp, Cp, 1)) for z, w in enumerate(weights)],
                    check=lambda: _check_sources(weights, lambda w: tuple(w.shape) == (Co, Ci, 3, 3)))


def pack_conv3_split(weights, tag):
    """pack_conv3 as pre-split planes (Split [Z, Co, 9*pitch(Ci)], k = tap*Cip + ci) for the split-plane implicit-GEMM conv."""
    Co, Ci = weights[0].shape[:2]
    Cp, Z = pitch(Ci), len(weights)
    dev = weights[0].device
    return seg_pack((tag, 'split', 'conv3', tuple(id(w) for w in weights)), list(weights),
                    lambda: Split(torch.zeros(Z, Co, 9 * Cp, dtype=torch.bfloat16, device=dev), torch.zeros(Z, Co, 9 * Cp, dtype=torch.bfloat16, device=dev)),
                    lambda sp: [segment(w, 0, sp.hi, z * Co * 9 * Cp, (Co, 9, Ci), (Ci * 9, 1, 9), (9 * Cp, Cp, 1), dst_lo=sp.lo)
                                for z, w in enumerate(weights)],
                    check=lambda: _check_sources(weights, lambda w: tuple(w.shape) == (Co, Ci, 3, 3)))


def split_conv_ok(Ci, Co=None):
    """the split-plane implicit-GEMM conv (mtt_gemm variant 9, gemm_variant_for in csrc/gemm.hip) needs a channel pitch that is a multiple
    of 32 (a K step inside one tap), at most 4096 channels, and a weight operand within 32-bit element offsets; the pixel-row limit
    (M * pitch < 2^31) is met by conv3x3 through batch chunks.  Outside these the caller keeps the non-split pack and the register-staged
    x3 kernel."""
    Cp = pitch(Ci)
    return Cp % 32 == 0 and Cp <= 4096 and (Co is None or Co * 9 * Cp < 2 ** 31)


def pack_upconv9(weights, prec, tag):
    """List of Z conv weights [Co, Ci, 3, 3] -> [Z, 9*pitch(Co), pitch(Ci)]: row (ky*3+kx)*pitch(Co) + co holds W[co, :, ky, kx] — the nine
    tap matrices of the "taps first" form of upsample x4 + 3x3 conv (mtt_upconv_desc) stacked as ONE linear layer; rows of the channel
    padding are zero, so its output planes carry zero padding channels."""
    Co, Ci = weights[0].shape[:2]
    Cop, Kp, Z = pitch(Co), pitch(Ci), len(weights)
    return seg_pack((tag, prec.name, 'up9', tuple(id(w) for w in weights)), list(weights),
                    lambda: torch.zeros(Z, 9 * Cop, Kp, dtype=prec.adt, device=weights[0].device),
                    lambda buf: [segment(w, 0, buf, z * 9 * Cop * Kp, (9, Co, Ci), (1, Ci * 9, 9), (Cop * Kp, Kp, 1)) for z, w in enumerate(weights)],
                    check=lambda: _check_sources(weights, lambda w: tuple(w.shape) == (Co, Ci, 3, 3)))


def pack_upconv9_split(weights, tag):
    """pack_upconv9 as pre-split planes (Split [Z, 9*pitch(Co), pitch(Ci)]) for the split-plane GEMM."""
    Co, Ci = weights[0].shape[:2]
    Cop, Kp, Z = pitch(Co), pitch(Ci), len(weights)
    dev = weights[0].device
    return seg_pack((tag, 'split', 'up9', tuple(id(w) for w in weights)), list(weights),
                    lambda: Split(torch.zeros(Z, 9 * Cop, Kp, dtype=torch.bfloat16, device=dev), torch.zeros(Z, 9 * Cop, Kp, dtype=torch.bfloat16, device=dev)),
                    lambda sp: [segment(w, 0, sp.hi, z * 9 * Cop * Kp, (9, Co, Ci), (1, Ci * 9, 9), (Cop * Kp, Kp, 1), dst_lo=sp.lo)
                                for z, w in enumerate(weights)],
                    check=lambda: _check_sources(weights, lambda w: tuple(w.shape) == (Co, Ci, 3, 3)))


def pack_kmap(weights, N, Kp, kmap, prec, tag):
    """List of Z parameters [N, K...] -> [Z, N, Kp] with the column ranges (dst0, src0, len) of `kmap` copied (inputs that are padded
    concatenations: taskprompter.py:471 torch.cat([spa, chan], 1) feeding fea_fuse[0])."""
    Z = len(weights)
    K = weights[0].numel() // N
    return seg_pack((tag, prec.name, 'kmap', tuple(id(w) for w in weights)), list(weights),
                    lambda: torch.zeros(Z, N, Kp, dtype=prec.adt, device=weights[0].device),
                    lambda buf: [segment(w, s0, buf, z * N * Kp + d0, (1, N, ln), (0, K, 1), (0, Kp, 1))
                                 for z, w in enumerate(weights) for (d0, s0, ln) in kmap],
                    check=lambda: _check_sources(weights, lambda w: w.numel() == N * K))


def pack_kmap_split(weights, N, Kp, kmap, tag):
    """pack_kmap as pre-split planes (Split [Z, N, Kp]) for the split-plane GEMM."""
    Z = len(weights)
    K = weights[0].numel() // N
    dev = weights[0].device
    return seg_pack((tag, 'split', 'kmap', tuple(id(w) for w in weights)), list(weights),
                    lambda: Split(torch.zeros(Z, N, Kp, dtype=torch.bfloat16, device=dev), torch.zeros(Z, N, Kp, dtype=torch.bfloat16, device=dev)),
                    lambda sp: [segment(w, s0, sp.hi, z * N * Kp + d0, (1, N, ln), (0, K, 1), (0, Kp, 1), dst_lo=sp.lo)
                                for z, w in enumerate(weights) for (d0, s0, ln) in kmap],
                    check=lambda: _check_sources(weights, lambda w: w.numel() == N * K))


def stack_vec(vs, tag):
    """List of Z fp32 vectors (biases) -> [Z, n] fp32."""
    n, Z = vs[0].numel(), len(vs)
    if any(v.dtype != torch.float32 or not v.is_contiguous() or v.numel() != n for v in vs):
        def build():
            with torch.no_grad():
                return torch.stack([v.detach().float().reshape(-1) for v in vs], 0).contiguous()
        return _cached((tag, tuple(id(v) for v in vs)), vs, build)
    return seg_pack((tag, 'vec', tuple(id(v) for v in vs)), list(vs),
                    lambda: torch.zeros(Z, n, dtype=torch.float32, device=vs[0].device),
                    lambda buf: [segment(v, 0, buf, z * n, (1, 1, n), (0, 0, 1), (0, 0, 1)) for z, v in enumerate(vs)])



# ---------------------------------------------------------------------------------------------
# GEMM-family forward ops
# ---------------------------------------------------------------------------------------------
def linear(x, wpack, N, prec, *, bias=None, act=ACT_NONE, out=None, out_dtype=None, M=None,
           a_rows=None, d_rows=None, resid=None, r_rows=None, rowscale=None, n_prompt=0, aux_out=None, aux_in=None,
           colscale=None, alpha=1.0, batch_inner=1, d_z=None, ldd=None, n_store=None, a_affine=None):
    """out[z] = epi(x[z] @ wpack[z]^T)  (mtt_gemm, both operands reduction-contiguous).

    x: [Z, M, lda] / [M, lda] (broadcast over Z) — or, with a_rows=(mb, bs, ld), any view whose first
    element is row 0 (then pass M).  wpack [Z, N, Kp]; bias / colscale fp32 [Z, N] (or [1, N] broadcast).
    out: None -> new [Z, M, pitch(N)] in `out_dtype` (default activation dtype); else a tensor / base view
    addressed by d_rows=(mb, bs, ld) or, for a 3-D `out`, its own strides; d_z=(zo, zi) overrides the
    per-batch offsets of D (z = zo*batch_inner + zi).  resid fp32 is added in the epilogue (r_rows mapping,
    default = D's); may alias `out`."""
    Z, Nw, Kp = wpack.shape
    assert N <= Nw
    x_split = isinstance(x, Split)
    assert x_split == isinstance(wpack, Split), "split planes: both operands or neither"
    if a_rows is not None:
        a_mb, a_bs, lda = a_rows
        a_z = 0
    else:
        xh = _hi(x)
        xv = xh if xh.dim() == 3 else xh[None]
        if M is None:
            M = xv.shape[1]
        a_mb, a_bs, lda = 0, 0, xv.shape[2]
        a_z = xv.stride(0) if xv.shape[0] > 1 else 0
        assert xv.shape[0] in (1, Z) and lda >= Kp
    Np = ldd if (out is None and ldd is not None) else pitch(N)      # a new output takes the channel pitch of N — or the caller's (composite widths: 9 tap planes)
    if out is None:
        out = Split.empty((Z, M, Np), x.device) if out_dtype == "split" else torch.empty(Z, M, Np, dtype=out_dtype or prec.adt, device=x.device)
    oh = _hi(out)
    if d_rows is not None:
        d_mb, d_bs, ldd_ = d_rows
        dz = 0
        nst = N
    else:
        d_mb, d_bs = 0, 0
        ldd_ = ldd if ldd is not None else oh.shape[-1]
        dz = oh.stride(0) if oh.dim() == 3 else 0
        nst = min(Np, ldd_)
    if n_store is not None:
        nst = n_store
    wh = _hi(wpack)
    kw = dict(A=_hi(x), B=wh, D=oh, M=M, N=N, K=Kp, a_op=OP_K, b_op=OP_K,
              a_dtype=dtype_code(x), b_dtype=dtype_code(wpack), d_dtype=dtype_code(out), prec=prec.code,
              lda=lda, ldb=Kp, ldd=ldd_, a_mb=a_mb, a_bs=a_bs, d_mb=d_mb, d_bs=d_bs,
              batch=Z, batch_inner=batch_inner, alpha=alpha, act=act, n_store=nst)
    if x_split:
        kw.update(A_lo=x.lo, B_lo=wpack.lo)
    if isinstance(out, Split):
        kw.update(D_lo=out.lo)
    # batch offsets: linear in z for A / B / bias; D may be two-level (d_z)
    kw.update(a_zo=a_z * batch_inner, a_zi=a_z if batch_inner > 1 else 0,
              b_zo=wh.stride(0) * batch_inner, b_zi=wh.stride(0) if batch_inner > 1 else 0)
    if d_z is not None:
        kw.update(d_zo=d_z[0], d_zi=d_z[1])
    else:
        kw.update(d_zo=dz * batch_inner, d_zi=dz if batch_inner > 1 else 0)
    if bias is not None or colscale is not None:
        ref = bias if bias is not None else colscale
        cz = ref.stride(0) if (ref.dim() == 2 and ref.shape[0] > 1) else 0
        kw.update(colshift=bias, colscale=colscale, col_zo=cz * batch_inner, col_zi=cz if batch_inner > 1 else 0)
    if resid is not None:
        rr = r_rows if r_rows is not None else (d_mb, d_bs, ldd_)
        kw.update(resid=resid, r_mb=rr[0], r_bs=rr[1], ldr=rr[2])
    if rowscale is not None:
        kw.update(rowscale=rowscale, n_prompt=n_prompt)
    if aux_out is not None or aux_in is not None:
        aux = aux_out if aux_out is not None else aux_in
        az = aux.stride(0) if aux.dim() == 3 else 0
        kw.update(aux_out=aux_out, aux_in=aux_in, aux_dtype=dtype_code(aux), ldaux=aux.shape[-1],
                  aux_zo=az * batch_inner, aux_zi=az if batch_inner > 1 else 0)
    if a_affine is not None:          # (scale [K], shift [K], act, bf16 side copy or None): the A prologue of the exact-fp32 tall GEMM (head_prologue_ok)
        sc, sh, a_act, a16 = a_affine
        kw.update(a_scale=sc, a_shift=sh, a_act=a_act, a_aux16=a16, ld_a16=a16.stride(-2) if a16 is not None else 0)
    call("gemm", **kw)
    return out


def head_prologue_ok(rows, K, n, lda):
    """can mtt_gemm apply BatchNorm + activation while loading the A operand (mtt_gemm_desc.a_scale)?  Only its exact-fp32 tall kernel does
    (gemm_f32n_kernel: fp32 operands, at most 32 outputs, K <= 1024 in whole 8-chunks; with a prologue it takes any row count)."""
    return n <= 32 and K % 8 == 0 and K <= 1024 and rows >= 1 and lda % 4 == 0


SPLIT_CONV_MAX_ELEMS = 2 ** 31 - 1          # gemm_variant_for: (int64) M * lda < 2^31 for variant 9 (tests lower it to force chunks)


def conv3x3(x, wpack, Co, Ci, B, H, W, prec, *, bias=None, colscale=None, act=ACT_NONE, dil=1, flip=0, out_dtype=None):
    """x [Z, B*H*W, Cp] -> [Z, B*H*W, pitch(Co)]; wpack [Z, Co, 9*Cp]; bias/colscale [Z, Co] fp32.
    Implicit GEMM (no im2col buffer): A rows are gathered per 16-byte channel chunk."""
    Z, rows, Cp = x.shape
    assert rows == B * H * W and wpack.shape[-1] == 9 * Cp
    if isinstance(wpack, Split) and not isinstance(x, Split):
        # x3f: the weights are pre-split planes (pack_conv3 under a split Prec); an fp32 input is split by one pass (8 bytes per element,
        # against a kernel that runs 2-3x the register-staged x3 conv's rate)
        x = x.float().contiguous()
        sp = split_cast(x.view(Z * rows, Cp))
        x = Split(sp.hi.view(Z, rows, Cp), sp.lo.view(Z, rows, Cp))
    assert isinstance(x, Split) == isinstance(wpack, Split), "split planes: both operands or neither"
    Cop = pitch(Co)
    if out_dtype == "split":                     # the output as hi / lo planes (the split-plane kernel's epilogue kinds 5 / 6): feeds a split-plane GEMM
        assert isinstance(x, Split), "a split conv output needs the split-plane conv kernel"
        out = Split.empty((Z, rows, Cop), x.device)
    else:
        out = torch.empty(Z, rows, Cop, dtype=out_dtype or prec.adt, device=x.device)
    oh = _hi(out)
    # the split-plane kernel addresses a batch member's pixel rows with 32-bit element offsets (M * pitch < 2^31): larger batches go
    # in chunks of whole images (a conv never mixes images); the general kernel has no such limit
    per = max(1, SPLIT_CONV_MAX_ELEMS // (H * W * Cp)) if isinstance(x, Split) else B
    for b0 in range(0, B, per):
        nb = min(per, B - b0)
        r0, r1 = b0 * H * W, (b0 + nb) * H * W
        xa, xl = _hi(x)[:, r0:r1], (x.lo[:, r0:r1] if isinstance(x, Split) else None)
        kw = dict(A=xa, B=_hi(wpack), D=oh[:, r0:r1], M=r1 - r0, N=Co, K=9 * Cp, a_op=OP_CONV_K, b_op=OP_K,
                  a_dtype=dtype_code(x), b_dtype=dtype_code(wpack), d_dtype=dtype_code(out), prec=prec.code,
                  lda=Cp, ldb=9 * Cp, ldd=Cop, batch=Z, batch_inner=1, a_zo=_hi(x).stride(0), b_zo=_hi(wpack).stride(0), d_zo=oh.stride(0),
                  conv=dict(H=H, W=W, C=Ci, Cp=Cp, dil=dil, flip=flip), alpha=1.0, act=act, n_store=Cop)
        if isinstance(x, Split):                     # implicit-GEMM form of the split-plane kernel (mtt_gemm variant 9): Cp % 32 == 0
            kw.update(A_lo=xl, B_lo=wpack.lo)
        if isinstance(out, Split):
            kw.update(D_lo=out.lo[:, r0:r1])
        if bias is not None:
            kw.update(colshift=bias, col_zo=bias.stride(0))
        if colscale is not None:
            kw.update(colscale=colscale)
        call("gemm", **kw)
    return out


def upconv4_expand(z, C, B, h, w, *, bias=None, colscale=None, act=ACT_NONE):
    """z [Z, B*h*w, 9*Cp] (tap planes from linear(x, pack_upconv9(...))) -> [Z, B*4h*4w, Cp] = act(conv3x3(up4(x)) * colscale + bias)."""
    Z, rows, n9 = z.shape
    Cp = n9 // 9
    assert rows == B * h * w and n9 == 9 * Cp and z.is_contiguous()
    y = torch.empty(Z, B * 16 * h * w, Cp, dtype=z.dtype, device=z.device)
    call("upconv4_expand", z=z, y=y, bias=bias, colscale=colscale, Z=Z, B=B, h=h, w=w, C=C, Cp=Cp,
         z_dtype=dtype_code(z), y_dtype=dtype_code(y), act=act)
    return y


def upconv4_gather(dy, C, B, h, w):
    """adjoint of upconv4_expand (without act / scale): dy [Z, B*4h*4w, Cp] -> dz [Z, B*h*w, 9*Cp]."""
    Z, rows, Cp = dy.shape
    assert rows == B * 16 * h * w and dy.is_contiguous()
    dz = torch.empty(Z, B * h * w, 9 * Cp, dtype=dy.dtype, device=dy.device)
    call("upconv4_gather", z=dz, y=dy, bias=None, colscale=None, Z=Z, B=B, h=h, w=w, C=C, Cp=Cp,
         z_dtype=dtype_code(dz), y_dtype=dtype_code(dy), act=ACT_NONE)
    return dz


def upconv3x3(x, w9, Co, B, h, w, prec, *, bias=None, colscale=None, act=ACT_NONE):
    """F.interpolate(x, scale_factor=4, 'bilinear') -> Conv2d(3x3, padding 1) on the LOW-resolution task stack x [Z, B*h*w, Cip]
    (taskprompter.py:420 -> :692) in its taps-first form: one GEMM with the nine stacked tap matrices, then the expansion kernel."""
    if isinstance(w9, Split) and not isinstance(x, Split):
        # x3f: the nine-tap GEMM (N = 9 * pitch(Co): whole 256-wide tiles) on the split-plane LDS-DMA kernel; the fp32 task features are
        # split by one pass over the LOW-resolution stack (1 / 16 of the head maps)
        Z, M, Kp = x.shape
        x = split_cast(x.reshape(Z * M, Kp))
        x = Split(x.hi.view(Z, M, Kp), x.lo.view(Z, M, Kp))
    z = linear(x, w9, w9.shape[1], prec, out_dtype=torch.float32 if isinstance(w9, Split) else None, ldd=w9.shape[1])   # nine planes of pitch Cop each
    return upconv4_expand(z, Co, B, h, w, bias=bias, colscale=colscale, act=act)


DECONV_SPLIT = True        # A/B switch: False = ConvTranspose2d through the general kernel's pixel-shuffle store in every mode


def deconv2x2(x, wpack, Co, Ci, B, H, W, prec, *, bias4=None, out_dtype=None):
    """ConvTranspose2d(k=2, s=2) as a GEMM with a pixel-shuffle store (taskprompter.py:705).
    x [B*H*W, Cip]; wpack [1, 4*Co, Cip] with n = (dy*2+dx)*Co + co; bias4 [4*Co].  -> [B*2H*2W, pitch(Co)]"""
    Cop = pitch(Co)
    if (DECONV_SPLIT and prec.split and torch.is_tensor(x) and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
            and wpack.dtype == torch.float32 and x.shape[1] == wpack.shape[-1] and split_gemm_ok(x.shape[1]) and B * H * W >= 2048):
        # x3f: the product as a PLAIN GEMM on the split-plane LDS-DMA kernel (the pixel-shuffle store exists in the general kernel's epilogue
        # only, i.e. register-staged x3: 6.4 ms per task at the Swin-B shape), then one shuffle pass (mtt_pixshuf2)
        wsp = getattr(wpack, "_mtt_split", None)          # lives and dies with the pack (rebuilt after every parameter update)
        if wsp is None:
            wsp = wpack._mtt_split = split_cast(wpack[0].contiguous())
        z = linear(split_cast(x), Split(wsp.hi[None], wsp.lo[None]), 4 * Co, prec, bias=None if bias4 is None else bias4[None], out_dtype=torch.float32)
        out = torch.empty(B * 4 * H * W, Cop, dtype=out_dtype or prec.adt, device=x.device)
        call("pixshuf2", args=[z[0], out, B, H, W, Co, z.shape[-1], Cop, dtype_code(z), dtype_code(out)])
        return out
    out = torch.zeros(B * 4 * H * W, Cop, dtype=out_dtype or prec.adt, device=x.device)
    kw = dict(A=x, B=wpack, D=out, M=B * H * W, N=4 * Co, K=wpack.shape[-1], a_op=OP_K, b_op=OP_K,
              a_dtype=dtype_code(x), b_dtype=dtype_code(wpack), d_dtype=dtype_code(out), prec=prec.code,
              lda=x.shape[-1], ldb=wpack.shape[-1], ldd=Cop, batch=1, batch_inner=1, alpha=1.0, b_zo=0,
              store_mode=_lib.STORE_PIXSHUF2, ps_H=H, ps_W=W, ps_Co=Co)
    if bias4 is not None:
        kw.update(colshift=bias4)
    call("gemm", **kw)
    return out


# ---------------------------------------------------------------------------------------------
# row ops
# ---------------------------------------------------------------------------------------------
def layernorm(x, gamma, beta, eps, prec, save_stats=False, out_dtype=None, want32=False):
    """x fp32 [rows, C] -> y [rows, C] (activation dtype unless out_dtype; "split": a Split, and with want32 also the fp32 rows —
    then y is the pair (Split, fp32 tensor)).  Returns (y, mean, rstd)."""
    rows, C = x.shape
    split = out_dtype == "split"
    y = Split.empty((rows, C), x.device) if split else torch.empty(rows, C, dtype=out_dtype or prec.adt, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    kw = {}
    y32 = None
    if split:
        kw.update(y_lo=y.lo)
        if want32:
            y32 = torch.empty(rows, C, dtype=torch.float32, device=x.device)
            kw.update(y32=y32, ldy32=C)
    call("layernorm_fwd", x=x, y=_hi(y), gamma=gamma, beta=beta, mean=mean, rstd=rstd, rows=rows, C=C,
         ldx=x.stride(0), ldy=C, y_dtype=dtype_code(y), eps=eps, **kw)
    return ((y, y32) if want32 else y), mean, rstd


def attention(qkv, B, N, nH, T, prec, want_lse=False):
    """qkv [B*N, 3*nH*64] -> (out [B*N, nH*64], rawlog fp32 [B, nH, T, N] or None, lse or None)."""
    C = nH * 64
    split = isinstance(qkv, Split)
    out = Split.empty((B * N, C), qkv.device) if split else torch.empty(B * N, C, dtype=qkv.dtype, device=qkv.device)
    rawlog = torch.empty(B, nH, T, N, dtype=torch.float32, device=qkv.device) if T > 0 else None
    lse = torch.empty(B, nH, N, dtype=torch.float32, device=qkv.device) if want_lse else None
    kw = dict(qkv_lo=qkv.lo, out_lo=out.lo) if split else {}
    call("attn_fwd", qkv=_hi(qkv), out=_hi(out), rawlog=rawlog, lse=lse, B=B, N=N, nH=nH, T=T,
         dtype=dtype_code(qkv), prec=prec.code, scale=64 ** -0.5, **kw)
    return out, rawlog, lse


def patchify(img, prec):
    B, _, H, W = img.shape
    cols = torch.empty(B * (H // 16) * (W // 16), 768, dtype=prec.adt, device=img.device)
    call("patchify16", args=[img.contiguous(), cols, B, H, W, dtype_code(cols)])
    return cols


def chan_logits(cq, xn, B, T, N, C, grid, nwin_hw):
    """cq [B*T, ldq] (token_trans output), xn [B*N, C] (norm1 output; a Split: its hi / lo planes are read and summed, cq is then fp32)
    -> rawchan fp32 [B, T, nwin, C]."""
    nh, nw = nwin_hw
    rawchan = torch.empty(B, T, nh * nw, C, dtype=torch.float32, device=xn.device)
    kw = {}
    if isinstance(xn, Split):
        assert cq.dtype == torch.float32
        kw["xn_lo"] = xn.lo
    call("chan_logits", q=cq, xn=_hi(xn), rawchan=rawchan, B=B, T=T, N=N, C=C, h=grid[0], w=grid[1], nh=nh, nw=nw,
         dtype=dtype_code(xn), ldq=cq.shape[-1], ws=ws_for("chan_logits", xn.device, B=B, T=T, C=C, h=grid[0], w=grid[1], nh=nh, nw=nw), **kw)
    return rawchan


def prompt_rows32(xs, B, N, T, C):
    """The T prompt rows of every image of a Split token matrix [B*N, C] as fp32 [B*T, C] (hi + lo): the A operand of the prompt-row
    Linears of the channel attention (B*T rows: a few hundred KB — torch ops, not worth a kernel)."""
    return (xs.hi.view(B, N, C)[:, :T].float() + xs.lo.view(B, N, C)[:, :T].float()).reshape(B * T, C)


def modulate(x, x_ld, x_bs, rawlog, rawchan, B, T, N, C, grid, nwin_hw, prec, hg=0, split=False):
    """x: fp32 base view of the patch rows ([B, hw, C] with row pitch x_ld, batch stride x_bs).
    -> [2T, B*hw, C] activation dtype (spatially- then channel-modulated copy per task); split=True: a Split (hi / lo bf16 planes
    written by the kernel) for the fea_decode GEMMs of the fp32-class forward on the split-plane LDS-DMA kernel."""
    hw = grid[0] * grid[1]
    out = Split.empty((2 * T, B * hw, C), x.device) if split else torch.empty(2 * T, B * hw, C, dtype=prec.adt, device=x.device)
    call("modulate", x=x, x_ld=x_ld, x_bs=x_bs, rawlog=rawlog, rawchan=rawchan, out=_hi(out), out_lo=out.lo if split else None,
         B=B, T=T, N=N, C=C, h=grid[0], w=grid[1], nh=nwin_hw[0], nw=nwin_hw[1], out_dtype=dtype_code(out), hg=hg)
    return out


def split_gemm_ok(K):
    """mtt_gemm takes MTT_SPLIT operands on ONE kernel (gemm_ring3_kernel): whole 32-deep K steps, at least two."""
    return K % 32 == 0 and K >= 64


def ctr_mix(fea, wmix, B, C, acc=None, out_dtype=torch.float32):
    """fea [T, rows, ld]; wmix fp32 [B, T, T] -> acc (+)= mix.  Returns [T, rows, ld] in fp32 (or `out_dtype` bf16 when there is no `acc`)."""
    T, rows, ld = fea.shape
    out = acc if acc is not None else torch.empty(T, rows, ld, dtype=out_dtype, device=fea.device)
    # the mix is linear and the padding channels of `fea` are zero: run it over the whole pitch, so that `out`'s padding is written (as zeros) too
    call("ctr_mix", fea=fea, out=out, wmix=wmix, T=T, B=B, rows_per_b=rows // B, ld=ld, C=ld,
         fea_dtype=dtype_code(fea), accumulate=1 if acc is not None else 0, out_dtype=dtype_code(out))
    return out


def ctr_weights(rawlog, w0, b0, w2, b2, B, T):
    """rawlog fp32 [B, nH, T, N]; w0 [T, nH*nH], b0 [T, nH], w2 [T, nH], b2 [T, 1] fp32 packs -> wmix fp32 [B, T, T] (mtt_ctr_weights:
    the per-task MLP over the head dimension of the prompt<->prompt raw logits, taskprompter.py:482-484)."""
    nH, N = rawlog.shape[1], rawlog.shape[3]
    wmix = torch.empty(B, T, T, dtype=torch.float32, device=rawlog.device)
    call("ctr_weights", rawlog=rawlog, w0=w0, b0=b0, w2=w2, b2=b2, wmix=wmix, B=B, T=T, nH=nH, N=N)
    return wmix


def bilinear(x, B, C, Hin, Win, Hout, Wout, out_dtype, nchw=False):
    """x [Z, B*Hin*Win, ld] -> [Z, B*Hout*Wout, ld] (NHWC) or fp32 NCHW [B, C, Hout, Wout] (Z must be 1)."""
    Z, _, ld = x.shape
    if nchw:
        out = torch.empty(B, C, Hout, Wout, dtype=torch.float32, device=x.device)
        call("bilinear_fwd", **{"in": x}, out=out, B=B, C=C, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, ld_in=ld, ld_out=0,
             in_dtype=dtype_code(x), out_dtype=F32, out_nchw=1, accumulate=0)
        return out
    out = torch.empty(Z, B * Hout * Wout, ld, dtype=out_dtype, device=x.device)
    # Z maps of the same geometry: fold Z into the batch dimension
    call("bilinear_fwd", **{"in": x}, out=out, B=Z * B, C=ld, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, ld_in=ld, ld_out=ld,
         in_dtype=dtype_code(x), out_dtype=dtype_code(out), out_nchw=0, accumulate=0)
    return out


_ws_cache = {}
_ws_retired = []         # outgrown workspaces are kept alive: a captured hipGraph (graphs.GraphedTrainStep) has their raw addresses baked in


def workspace(nfloats, device):
    """Grow-only fp32 scratch buffer per device for the kernels' workspaces (launches are stream-ordered, so one buffer serves all).
    An outgrown buffer is never freed (a recorded graph may still write its partial sums there), and growing while a stream is
    capturing is an error (the allocation would land in the graph's private pool and die with it)."""
    key = (device.type, device.index)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nfloats:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ops.workspace: the kernel workspace must not grow during stream capture; run one eager iteration of the same "
                               "shapes first (GraphedTrainStep's warm-up does)")
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(max(int(nfloats), 1 << 18), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def ws_for(entry, device, **fields):
    """the caller-owned reduction workspace of entry point `entry` (mtt_<entry>_ws_floats(desc) floats; the kernels that used fp32 atomics
    until round 3 now write per-workgroup partials there and sum them in a fixed order: deterministic training steps)."""
    return workspace(_lib.ws_floats(entry, **fields), device)


def ln_bwd_ws(rows, C, device):
    return workspace(_lib.load().mtt_layernorm_bwd_ws_floats(rows, C), device)


def colsum(x2d, cols):
    """fp32 [cols] column sums of x2d [rows, ld] (bias gradients): deterministic two-stage reduction (mtt_colsum)."""
    out = torch.empty(cols, dtype=torch.float32, device=x2d.device)
    ws = workspace(_lib.load().mtt_colsum_ws_floats(x2d.shape[0], cols), x2d.device)
    call("colsum", args=[x2d, out, x2d.shape[0], cols, x2d.stride(0), dtype_code(x2d), ws])
    return out


def colsum_batched(x3d, cols, Z, src_zs, col_off=0):
    """fp32 [Z, cols] column sums of Z maps [rows, ld] of one tensor (map z starts col_off + z * src_zs elements into x3d's storage view
    `x3d.reshape(-1)`): the bias gradients of a task-batched layer in ONE launch pair."""
    rows, ld = x3d.shape[-2], x3d.stride(-2)
    out = torch.empty(Z, cols, dtype=torch.float32, device=x3d.device)
    ws = workspace(Z * _lib.load().mtt_colsum_ws_floats(rows, cols), x3d.device)
    src = x3d.reshape(-1)[col_off:] if col_off else x3d
    call("colsum_batched", args=[src, out, rows, cols, ld, dtype_code(x3d), Z, src_zs, cols, ws])
    return out


def _bn_ws(rows, C, Z, device):
    return workspace(_lib.load().mtt_bn_reduce_ws_floats(rows, C, Z), device)


def _as3(x):
    return x if x.dim() == 3 else x[None]


def bn_stats(x, C):
    """x [Z, rows, ld] (or [rows, ld]) -> (mean [Z, C], M2 [Z, C] = sum_r (x - mean)^2): centred, deterministic (mtt_bn_stats)."""
    x = _as3(x)
    Z, rows, ld = x.shape
    out = torch.empty(2, Z, C, dtype=torch.float32, device=x.device)
    call("bn_stats", x=x, mean_out=out[0], m2_out=out[1], rows=rows, C=C, ld=ld, dtype=dtype_code(x), Z=Z, x_zs=x.stride(0), p_zs=C,
         xargs=[_bn_ws(rows, C, Z, x.device)])
    return out[0], out[1]


def bn_apply(x, C, mean, rstd, gamma, beta, act, out=None):
    """y[z] = act((x[z] - mean[z]) * rstd[z] * gamma[z] + beta[z]); x [Z, rows, ld] or [rows, ld]; vectors [Z, C] or [C]."""
    x3 = _as3(x)
    Z, rows, ld = x3.shape
    y = torch.empty_like(x) if out is None else out
    vec = [v.reshape(Z, C) for v in (mean, rstd, gamma, beta)]
    vec = [v if v.is_contiguous() else v.contiguous() for v in vec]
    call("bn_apply", x=x3, y=y, mean=vec[0], rstd=vec[1], gamma=vec[2], beta=vec[3], rows=rows, C=C, ld=ld,
         dtype=dtype_code(x3), act=act, Z=Z, x_zs=x3.stride(0), p_zs=C)
    return y


def bn_bwd_reduce(x, dy, C, mean, rstd, gamma, beta, act):
    """-> s [2, Z, C]: s[0] = sum_r du, s[1] = sum_r du * xhat (du = dy * act'(u)); deterministic (mtt_bn_bwd_reduce)."""
    x, dy = _as3(x), _as3(dy)
    Z, rows, ld = x.shape
    s = torch.empty(2, Z, C, dtype=torch.float32, device=x.device)
    assert dy.shape == x.shape and dy.stride() == x.stride()
    call("bn_bwd_reduce", x=x, dy=dy, mean=mean, rstd=rstd, gamma=gamma, beta=beta, dsum=s[0], dsumxh=s[1], rows=rows, C=C, ld=ld,
         dtype=dtype_code(x), g_dtype=dtype_code(dy) + 1, act=act, Z=Z, x_zs=x.stride(0), p_zs=C, xargs=[_bn_ws(rows, C, Z, x.device)])
    return s


def bn_bwd_apply(x, dy, C, mean, rstd, gamma, beta, act, red):
    """dx from the (rank-summed, pre-scaled to the local row count) sums `red` [2, Z, C]; dx is stored like dy (which may be bf16 next
    to an fp32-stored x: the gradient maps of a bf16-arithmetic backward, mtt_bn_desc.g_dtype)."""
    x3, dy3 = _as3(x), _as3(dy)
    Z, rows, ld = x3.shape
    assert dy3.shape == x3.shape and dy3.stride() == x3.stride()
    dx = torch.empty_like(dy)
    call("bn_bwd_apply", x=x3, dy=dy3, dx=dx, mean=mean, rstd=rstd, gamma=gamma, beta=beta, dsum=red[0], dsumxh=red[1], rows=rows, C=C,
         ld=ld, dtype=dtype_code(x3), g_dtype=dtype_code(dy3) + 1, act=act, Z=Z, x_zs=x3.stride(0), p_zs=C)
    return dx
