"""ctypes binding of libmtt_hip.so (include/mtt_hip.h).

`call(name, **fields)` fills the descriptor struct of entry point `mtt_<name>` from keyword
arguments — torch tensors (or views: only the base address is used) for pointer fields, ints /
floats for the rest — launches it on torch's current HIP stream and raises on a non-zero status.

There is NO fallback: if the shared library is missing or a tensor is not on a HIP device the
call raises.  (tests/ may monkeypatch `call` with the CPU emulator in oracle/abi_emul.py to
exercise the host-side wiring without a GPU; the product never does.)
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmtt_hip.so")

ABI_VERSION = 13
F32, BF16, SPLIT = 0, 1, 2
PREC_BF16, PREC_X3 = 0, 1
OP_K, OP_R, OP_CONV_K, OP_CONV_R = 0, 1, 2, 3
ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_BWD, ACT_RELU_BWD, ACT_GELU_DAUX, ACT_MUL_AUX = 0, 1, 2, 3, 4, 5, 6
STORE_ROWS, STORE_PIXSHUF2 = 0, 1
GEMM_AUTO, GEMM_GENERAL, GEMM_DMA256, GEMM_DMA128, GEMM_GENERAL_EPILOGUE = 0, 1, 3, 4, 11
ATTN_AUTO, ATTN_PLAIN = 0, 1

i32, i64, f32, ptr = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvGeom(C.Structure):
    _fields_ = [("H", i32), ("W", i32), ("C", i32), ("Cp", i32), ("dil", i32), ("flip", i32)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", ptr), ("B", ptr), ("D", ptr),
        ("M", i32), ("N", i32), ("K", i32),
        ("a_op", i32), ("b_op", i32),
        ("a_dtype", i32), ("b_dtype", i32), ("d_dtype", i32),
        ("prec", i32),
        ("lda", i64), ("ldb", i64), ("ldd", i64),
        ("a_mb", i32), ("a_bs", i64),
        ("d_mb", i32), ("d_bs", i64),
        ("batch", i32), ("batch_inner", i32),
        ("a_zo", i64), ("a_zi", i64), ("b_zo", i64), ("b_zi", i64), ("d_zo", i64), ("d_zi", i64),
        ("conv", ConvGeom),
        ("alpha", f32),
        ("colscale", ptr), ("colshift", ptr), ("col_zo", i64), ("col_zi", i64),
        ("act", i32),
        ("aux_in", ptr), ("aux_out", ptr), ("aux_dtype", i32), ("ldaux", i64), ("aux_zo", i64), ("aux_zi", i64),
        ("rowscale", ptr), ("n_prompt", i32),
        ("resid", ptr), ("ldr", i64), ("r_mb", i32), ("r_bs", i64), ("r_zo", i64), ("r_zi", i64),
        ("n_store", i32), ("store_mode", i32),
        ("ps_H", i32), ("ps_W", i32), ("ps_Co", i32),
        ("variant", i32),
        ("A_lo", ptr), ("B_lo", ptr), ("D_lo", ptr),
        ("colsum_out", ptr), ("colsum_ws", ptr),
        ("a_scale", ptr), ("a_shift", ptr), ("a_act", i32), ("a_aux16", ptr), ("ld_a16", i64),
    ]


class AttnDesc(C.Structure):
    _fields_ = [("qkv", ptr), ("out", ptr), ("rawlog", ptr), ("lse", ptr),
                ("B", i32), ("N", i32), ("nH", i32), ("T", i32), ("dtype", i32), ("prec", i32), ("scale", f32),
                ("variant", i32), ("qkv_lo", ptr), ("out_lo", ptr)]


class SoftmaxDesc(C.Structure):
    _fields_ = [("S", ptr), ("P", ptr), ("dP", ptr), ("dS", ptr), ("extra", ptr),
                ("rows", i64), ("cols", i64), ("ld", i64), ("s_dtype", i32), ("p_dtype", i32), ("scale", f32),
                ("rows_per_mat", i64), ("extra_rows", i32), ("extra_ld", i64)]


class LnDesc(C.Structure):
    _fields_ = [("x", ptr), ("y", ptr), ("gamma", ptr), ("beta", ptr), ("mean", ptr), ("rstd", ptr),
                ("dy", ptr), ("dx", ptr), ("dgamma", ptr), ("dbeta", ptr),
                ("rows", i64), ("C", i32), ("ldx", i64), ("ldy", i64), ("y_dtype", i32), ("eps", f32), ("dx_in", ptr), ("ws", ptr),
                ("y_lo", ptr), ("y32", ptr), ("ldy32", i64)]


class ChanLogitDesc(C.Structure):
    _fields_ = [("q", ptr), ("xn", ptr), ("rawchan", ptr),
                ("B", i32), ("T", i32), ("N", i32), ("C", i32), ("h", i32), ("w", i32), ("nh", i32), ("nw", i32),
                ("dtype", i32), ("ldq", i64), ("ws", ptr), ("xn_lo", ptr)]


class ModulateDesc(C.Structure):
    _fields_ = [("x", ptr), ("x_ld", i64), ("x_bs", i64), ("rawlog", ptr), ("rawchan", ptr), ("out", ptr),
                ("B", i32), ("T", i32), ("N", i32), ("C", i32), ("h", i32), ("w", i32), ("nh", i32), ("nw", i32),
                ("out_dtype", i32), ("hg", i32), ("out_lo", ptr)]


class CtrDesc(C.Structure):
    _fields_ = [("fea", ptr), ("out", ptr), ("wmix", ptr), ("T", i32), ("B", i32), ("rows_per_b", i64), ("ld", i64),
                ("C", i32), ("fea_dtype", i32), ("accumulate", i32), ("out_dtype", i32)]


class ResizeDesc(C.Structure):
    _fields_ = [("in_", ptr), ("out", ptr), ("B", i32), ("C", i32), ("Hin", i32), ("Win", i32), ("Hout", i32), ("Wout", i32),
                ("ld_in", i64), ("ld_out", i64), ("in_dtype", i32), ("out_dtype", i32), ("out_nchw", i32), ("accumulate", i32)]


class BnDesc(C.Structure):
    _fields_ = [("x", ptr), ("y", ptr), ("dy", ptr), ("dx", ptr),
                ("mean_out", ptr), ("m2_out", ptr), ("mean", ptr), ("rstd", ptr), ("gamma", ptr), ("beta", ptr),
                ("dsum", ptr), ("dsumxh", ptr),
                ("rows", i64), ("C", i32), ("ld", i64), ("dtype", i32), ("act", i32),
                ("Z", i32), ("x_zs", i64), ("p_zs", i64), ("g_dtype", i32)]


class DwconvDesc(C.Structure):
    _fields_ = [("x", ptr), ("w", ptr), ("y", ptr), ("scale", ptr), ("shift", ptr),
                ("Z", i32), ("B", i32), ("H", i32), ("W", i32), ("ld", i64), ("dtype", i32)]


class PoolDesc(C.Structure):
    _fields_ = [("x", ptr), ("y", ptr), ("B", i32), ("H", i32), ("W", i32), ("k", i32), ("ld", i64), ("dtype", i32)]


class LnMtDesc(C.Structure):
    _fields_ = [("x", ptr), ("y", ptr), ("gamma", ptr), ("beta", ptr), ("rows", i64), ("T", i32), ("D", i32),
                ("ldx", i64), ("ldy", i64), ("y_dtype", i32), ("eps", f32)]


class AttnMsgDesc(C.Structure):
    _fields_ = [("cur", ptr), ("prev", ptr), ("out", ptr), ("w", ptr), ("bias", ptr),
                ("B", i32), ("heads", i32), ("T", i32), ("qh", i32), ("qw", i32), ("K", i32), ("ldk", i64), ("ldkp", i64)]


class ConvtDesc(C.Structure):
    _fields_ = [("yall", ptr), ("out", ptr), ("bias", ptr), ("B", i32), ("H", i32), ("W", i32), ("Cop", i32),
                ("dtype", i32), ("out_dtype", i32)]


class UpconvDesc(C.Structure):
    _fields_ = [("z", ptr), ("y", ptr), ("bias", ptr), ("colscale", ptr),
                ("Z", i32), ("B", i32), ("h", i32), ("w", i32), ("C", i32), ("Cp", i32),
                ("z_dtype", i32), ("y_dtype", i32), ("act", i32)]


class AdamDesc(C.Structure):
    _fields_ = [("grads", ptr), ("params", ptr), ("exp_avg", ptr), ("exp_avg_sq", ptr), ("numel", ptr),
                ("chunk_tensor", ptr), ("chunk_off", ptr), ("n_chunks", i32),
                ("max_norm", f32), ("step_size", f32), ("beta1", f32), ("beta2", f32), ("eps", f32), ("weight_decay", f32),
                ("inv_sqrt_bc2", f32), ("hyper", ptr), ("ws", ptr)]


class LossDesc(C.Structure):
    _fields_ = [("pred", ptr), ("label", ptr), ("dpred", ptr), ("loss", ptr), ("stats", ptr), ("B", i64), ("HW", i64),
                ("C", i32), ("Cl", i32), ("kind", i32), ("ignore", f32), ("pos_weight", f32), ("ws", ptr)]


class CtrwDesc(C.Structure):
    _fields_ = [("rawlog", ptr), ("w0", ptr), ("b0", ptr), ("w2", ptr), ("b2", ptr), ("wmix", ptr), ("B", i32), ("T", i32), ("nH", i32), ("N", i64)]


class DetLossDesc(C.Structure):
    _fields_ = [("pred", ptr), ("target", ptr), ("weight", ptr), ("out", ptr), ("sum", ptr), ("ws", ptr), ("N", i64), ("C", i32), ("kind", i32),
                ("wmode", i32), ("gamma", f32), ("alpha", f32), ("beta", f32)]


class GatherDesc(C.Structure):
    _fields_ = [("src", ptr), ("dst", ptr), ("idx", ptr), ("rows", i64), ("C", i32), ("ld_src", i64), ("ld_dst", i64),
                ("src_dtype", i32), ("dst_dtype", i32), ("B", i32), ("src_bs", i64), ("dst_bs", i64), ("idx_bs", i64), ("skip_neg", i32)]


class WinAttnDesc(C.Structure):
    _fields_ = [("qkv", ptr), ("out", ptr), ("rawmap", ptr), ("bias", ptr), ("mask", ptr), ("pix", ptr),
                ("nwin", i32), ("nW", i32), ("nH", i32), ("T", i32), ("ws2", i32), ("dtype", i32), ("scale", f32),
                ("map_ld", i64), ("map_off", i64), ("mfma", i32), ("biasT", ptr)]


class ChanAttnDesc(C.Structure):
    _fields_ = [("q", ptr), ("kvT", ptr), ("rawchan", ptr), ("cx", ptr),
                ("B", i32), ("T", i32), ("C", i32), ("ce", i32), ("nh", i32), ("nw", i32), ("kv_dtype", i32), ("ldk", i64), ("scale", f32), ("kvbias", ptr)]


class Conv3s2Desc(C.Structure):
    _fields_ = [("x", ptr), ("w", ptr), ("bias", ptr), ("y", ptr),
                ("B", i32), ("Ci", i32), ("Co", i32), ("H", i32), ("W", i32),
                ("x_bs", i64), ("x_cs", i64), ("x_off", i64), ("y_bs", i64), ("y_cs", i64), ("y_off", i64)]


class SegcopyDesc(C.Structure):
    _fields_ = [("table", ptr), ("chunk_seg", ptr), ("chunk_off", ptr), ("n_chunks", i32), ("src_base", i64), ("dst_base", i64)]


# entry point -> (descriptor struct, size index in mtt_desc_size) ; None = positional-argument entry
DESCS = {
    "loss_fwd": LossDesc,
    "gemm": GemmDesc, "attn_fwd": AttnDesc, "softmax_fwd": SoftmaxDesc, "softmax_bwd": SoftmaxDesc,
    "layernorm_fwd": LnDesc, "layernorm_bwd": LnDesc, "chan_logits": ChanLogitDesc, "modulate": ModulateDesc,
    "ctr_mix": CtrDesc, "bilinear_fwd": ResizeDesc, "bilinear_bwd": ResizeDesc,
    "bn_apply": BnDesc, "bn_bwd_apply": BnDesc,
    "dwconv3x3s2": DwconvDesc, "avgpool_ceil": PoolDesc, "layernorm_mt": LnMtDesc, "attn_msg": AttnMsgDesc,
    "convt3x3s2_gather": ConvtDesc,
    "upconv4_expand": UpconvDesc, "upconv4_gather": UpconvDesc,
    "gather_rows": GatherDesc, "winattn_fwd": WinAttnDesc, "chanattn_fwd": ChanAttnDesc, "conv3s2_nchw": Conv3s2Desc,
    "segcopy": SegcopyDesc,
    "ctr_weights": CtrwDesc, "detloss_fwd": DetLossDesc,
}
_SIZE_INDEX = [GemmDesc, AttnDesc, SoftmaxDesc, LnDesc, ChanLogitDesc, ModulateDesc, CtrDesc, ResizeDesc, BnDesc, ConvGeom,
               DwconvDesc, PoolDesc, LnMtDesc, AttnMsgDesc, ConvtDesc, AdamDesc, LossDesc, UpconvDesc,
               GatherDesc, WinAttnDesc, ChanAttnDesc, Conv3s2Desc, SegcopyDesc, CtrwDesc, DetLossDesc]
POSITIONAL = {
    "patchify16": [ptr, ptr, C.c_int, C.c_int, C.c_int, C.c_int, ptr],
    "patchify": [ptr, ptr, C.c_int, C.c_int, C.c_int, C.c_int, i64, C.c_int, ptr],
    "resize_nchw": [ptr, ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ptr],
    "boxes_overlap_bev": [ptr, C.c_int, ptr, C.c_int, ptr, C.c_int, ptr],
    "nms_bev": [ptr, C.c_int, f32, C.c_int, ptr, ptr, ptr, ptr],
    "cast2d": [ptr, ptr, i64, i64, i64, i64, C.c_int, C.c_int, C.c_int, ptr],
    "split_cast": [ptr, ptr, ptr, i64, i64, i64, i64, ptr],
    "pixshuf2": [ptr, ptr, i32, i32, i32, i32, i64, i64, C.c_int, C.c_int, ptr],
    "colsum": [ptr, ptr, i64, i32, i64, C.c_int, ptr, ptr],
    "colsum_batched": [ptr, ptr, i64, i32, i64, C.c_int, i32, i64, i64, ptr, ptr],
    "add_rows": [ptr, ptr, i64, i32, i64, i64, C.c_int, f32, ptr],
    "rowscale_cast": [ptr, ptr, i64, i32, i64, i64, C.c_int, C.c_int, ptr, i32, i32, ptr],
    "rowscale_cast_colsum": [ptr, ptr, i64, i32, i64, i64, C.c_int, C.c_int, ptr, i32, i32, ptr, ptr, ptr],
}

# descriptor + extra positional arguments: mtt_<name>(const desc*, extras..., stream)
DESC_EXTRA = {
    "bn_stats": (BnDesc, [ptr]), "bn_bwd_reduce": (BnDesc, [ptr]),
    "attn_msg_bwd": (AttnMsgDesc, [ptr, ptr, ptr, ptr, ptr, ptr]),
    "modulate_bwd": (ModulateDesc, [ptr, ptr, ptr, ptr, ptr]),
    "chan_logits_bwd": (ChanLogitDesc, [ptr, ptr, C.c_int, ptr]),
    "ctr_dw": (CtrDesc, [ptr, ptr, ptr]),
    "ctr_weights_bwd": (CtrwDesc, [ptr, ptr, ptr, ptr, ptr, ptr]),
    "detloss_bwd": (DetLossDesc, [ptr, ptr, f32, ptr]),
    "attn_bwd": (AttnDesc, [ptr, ptr, ptr, ptr]),
    "winattn_bwd": (WinAttnDesc, [ptr, ptr, ptr, ptr]),
    "chanattn_bwd": (ChanAttnDesc, [ptr, ptr, ptr, ptr, i64, ptr]),
    "conv3s2_nchw_bwd": (Conv3s2Desc, [ptr, ptr, ptr, ptr]),
    "grad_sqnorm": (AdamDesc, [ptr]),
    "adam_step": (AdamDesc, [ptr]),
    "loss_label_stats": (LossDesc, [ptr]),
    "loss_bwd": (LossDesc, [ptr]),
    "dwconv3x3s2_bwd": (DwconvDesc, [ptr, ptr, ptr]),
    "avgpool_ceil_bwd": (PoolDesc, [ptr, ptr]),
    "convt3x3s2_gather_bwd": (ConvtDesc, [ptr, ptr]),
}

# workspace-size queries mtt_<entry>_ws_floats(const desc*) of the entry points whose cross-workgroup reductions go through caller-owned partials
WS_QUERIES = {"gemm_colsum": GemmDesc, "chan_logits": ChanLogitDesc, "modulate_bwd": ModulateDesc, "ctr_dw": CtrDesc, "attn_msg_bwd": AttnMsgDesc, "loss": LossDesc,
              "chanattn_bwd": ChanAttnDesc, "detloss": DetLossDesc}
EXPORTS = ["mtt_abi_version", "mtt_desc_size", "mtt_gemm_variant", "mtt_adam_chunk", "mtt_segcopy_chunk", "mtt_bn_reduce_ws_floats", "mtt_colsum_ws_floats", "mtt_rowscale_cast_colsum_ws_floats", "mtt_layernorm_bwd_ws_floats", "mtt_nms_ws_bytes"] + ["mtt_%s_ws_floats" % n for n in WS_QUERIES] + ["mtt_" + n for n in list(DESCS) + list(POSITIONAL) + list(DESC_EXTRA)]

_lib = None


def load():
    """Load the shared library (once) and verify ABI version + descriptor sizes.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU/eager fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    lib.mtt_abi_version.restype = C.c_int
    lib.mtt_desc_size.restype = C.c_size_t
    lib.mtt_desc_size.argtypes = [C.c_int]
    lib.mtt_bn_reduce_ws_floats.restype = C.c_size_t
    lib.mtt_bn_reduce_ws_floats.argtypes = [i64, i32, i32]
    lib.mtt_layernorm_bwd_ws_floats.restype = C.c_size_t
    lib.mtt_layernorm_bwd_ws_floats.argtypes = [i64, i32]
    lib.mtt_colsum_ws_floats.restype = C.c_size_t
    lib.mtt_colsum_ws_floats.argtypes = [i64, i32]
    if lib.mtt_abi_version() != ABI_VERSION:
        raise RuntimeError("libmtt_hip.so ABI version mismatch")
    for idx, st in enumerate(_SIZE_INDEX):
        if lib.mtt_desc_size(idx) != C.sizeof(st):
            raise RuntimeError(f"descriptor layout mismatch for {st.__name__}: C {lib.mtt_desc_size(idx)} vs ctypes {C.sizeof(st)}")
    for name, st in DESCS.items():
        fn = getattr(lib, "mtt_" + name)
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(st), ptr]
    for name, at in POSITIONAL.items():
        fn = getattr(lib, "mtt_" + name)
        fn.restype = C.c_int
        fn.argtypes = at
    for name, (st, extra) in DESC_EXTRA.items():
        fn = getattr(lib, "mtt_" + name)
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(st)] + extra + [ptr]
    _lib = lib
    return lib


def ws_floats(entry, **fields):
    """mtt_<entry>_ws_floats(desc) for the geometry in `fields` (ints only; pointer fields stay NULL)."""
    lib = load()
    desc = WS_QUERIES[entry]()
    for k, v in fields.items():
        setattr(desc, k, v)
    fn = getattr(lib, "mtt_%s_ws_floats" % entry)
    fn.restype = C.c_size_t
    fn.argtypes = [C.POINTER(WS_QUERIES[entry])]
    return int(fn(C.byref(desc)))


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported tensor dtype {t.dtype}")


def _addr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libmtt_hip.so operates on HIP device memory only (tensor is on %s)" % t.device)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def gemm_variant(**kw):
    """Kernel mtt_gemm would dispatch these descriptor fields to (0 general 128x128, 3 LDS-DMA 256x256, 6 token-major weight gradient,
    8 LDS-DMA 256x256 on split operands; < 0: no kernel takes them)."""
    lib = load()
    desc = GemmDesc()
    for k, v in kw.items():
        if k == "conv":
            for ck, cv in v.items():
                setattr(desc.conv, ck, int(cv))
        elif isinstance(v, torch.Tensor):
            setattr(desc, k, v.data_ptr())
        elif v is not None:
            setattr(desc, k, v)
    lib.mtt_gemm_variant.restype = C.c_int
    lib.mtt_gemm_variant.argtypes = [C.POINTER(GemmDesc)]
    return lib.mtt_gemm_variant(C.byref(desc))


def call(name, **kw):
    """Launch `mtt_<name>`; tensors in `kw` become device pointers.  For descriptor entry points
    nested dict `conv={...}` fills mtt_conv_geom.  Positional entry points take `args=[...]`."""
    lib = load()
    fn = getattr(lib, "mtt_" + name)
    if name in POSITIONAL:
        args = [(_addr(a) if isinstance(a, torch.Tensor) else a) for a in kw["args"]]
        rc = fn(*args, _stream())
    else:
        extra = None
        if name in DESC_EXTRA:
            st, _ = DESC_EXTRA[name]
            desc = st()
            extra = [(_addr(a) if isinstance(a, torch.Tensor) else a) for a in kw["xargs"]]
        else:
            desc = DESCS[name]()
        for k, v in kw.items():
            if k == "xargs":
                continue
            if k == "in":
                k = "in_"
            if k == "conv":
                for ck, cv in v.items():
                    setattr(desc.conv, ck, int(cv))
            elif isinstance(v, torch.Tensor):
                setattr(desc, k, _addr(v))
            elif v is None:
                setattr(desc, k, None)
            else:
                setattr(desc, k, v)
        rc = fn(C.byref(desc), *extra, _stream()) if extra is not None else fn(C.byref(desc), _stream())
    if rc != 0:
        raise RuntimeError(f"mtt_{name} failed with status {rc}" + (" (argument error)" if rc < 0 else " (hipError_t)"))
