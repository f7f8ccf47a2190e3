"""GPU parity cases: every C-ABI entry point of libmtt_hip.so against the CPU emulator
(oracle/abi_emul.py) on identical seeded buffers.  Each case compares EVERY buffer's whole storage, so
out-of-bounds / stray writes show up as well as wrong values.  Used by tests/test_gpu_ops.py (pytest,
`-m gpu`) and tools/gpu_diag.py (runs all cases without stopping and writes a JSON report).
"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import abi_emul  # noqa: E402

F32, BF16, SPLIT = 0, 1, 2
OP_K, OP_R, OP_CONV_K, OP_CONV_R = 0, 1, 2, 3
DT = {0: torch.float32, 1: torch.bfloat16}


def pkg():
    return importlib.import_module("multi-task-transformer_amd")


def rnd(g, *shape, dtype=torch.float32, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def scratch(n):
    """A kernel workspace: contents are unspecified after the call, so its storage is excluded from the comparison."""
    t = torch.zeros(n)
    t._mtt_scratch = True
    return t


def run_case(name, kw, outputs, tol):
    """Run entry `name` with CPU emulator and on the GPU; compare the `outputs` tensors' whole storages.
    Returns dict(ok, errs={key: rel_err})."""
    lib = pkg()._lib
    tensors = {k: v for k, v in kw.items() if isinstance(v, torch.Tensor)}
    # GPU copies sharing storage structure
    gpu_store, gpu_kw = {}, dict(kw)

    skip = set()

    def to_gpu(t):
        key = t.untyped_storage().data_ptr()
        if getattr(t, "_mtt_scratch", False):
            skip.add(key)
        if key not in gpu_store:
            flat, _ = abi_emul.flat(t)
            gpu_store[key] = (flat.clone().cuda(), flat)
        return gpu_store[key][0].as_strided(t.size(), t.stride(), t.storage_offset())

    for lk in ("args", "xargs"):
        if lk in kw:
            gpu_kw[lk] = [to_gpu(a) if isinstance(a, torch.Tensor) else a for a in kw[lk]]
    for k, t in tensors.items():
        gpu_kw[k] = to_gpu(t)
    lib.call(name, **gpu_kw)
    torch.cuda.synchronize()
    abi_emul.call(name, **kw)
    errs, ok = {}, True
    # MTT_SPLIT outputs (hi / lo bf16 planes): the lo plane alone is ill-conditioned (a 1-ulp flip of hi moves lo by a whole ulp), so
    # the pair is judged by its SUM (fp32-class: tol["split"]); the hi plane is still compared on its own as a bf16 storage
    pairs = [(kw[h], kw[l], gpu_kw[h], gpu_kw[l], f"{h}+{l}") for h, l in tol.get("split_pairs", ())]
    if "split_args" in tol:
        ih, il = tol["split_args"]
        pairs.append((kw["args"][ih], kw["args"][il], gpu_kw["args"][ih], gpu_kw["args"][il], "args"))
    for ch, cl, gh, gl, hk in pairs:
        lk = ""
        skip.add(cl.untyped_storage().data_ptr())
        a = gh.float().cpu().double() + gl.float().cpu().double()
        b = ch.double() + cl.double()
        e = float((a - b).norm()) / (float(b.norm()) + 1e-30)
        errs[f"split({hk}{lk})"] = (e, float((a - b).abs().max()))
        if not (e <= tol["split"]):
            ok = False
    for key, (gflat, cflat) in gpu_store.items():
        if key in skip:
            continue
        a, b = gflat.cpu().double(), cflat.double()
        if not torch.isfinite(a).all():
            errs[f"storage{len(errs)}"] = float("nan")
            ok = False
            continue
        denom = float(b.norm()) + 1e-30
        e = float((a - b).norm()) / denom
        mx = float((a - b).abs().max())
        errs[f"storage{len(errs)}(n={a.numel()},{cflat.dtype})"] = (e, mx)
        t = tol["bf16"] if cflat.dtype == torch.bfloat16 else tol["f32"]
        if e > t:
            ok = False
    return dict(ok=ok, errs=errs)


# tolerances (norm-wise relative error over a whole storage)
TOL_X3 = dict(f32=2e-5, bf16=5e-3)
TOL_BF = dict(f32=2e-5, bf16=5e-3)        # emulator rounds operands to bf16 too -> only accumulation order differs
TOL_ROW = dict(f32=1e-5, bf16=5e-3)
TOL_SPLIT_D = dict(f32=2e-5, bf16=5e-3, split=2e-5, split_pairs=[("D", "D_lo")])      # a split-plane GEMM output: hi as bf16, hi + lo as fp32-class


def gemm_cases():
    cases = []
    g = torch.Generator().manual_seed(11)

    def base(M, N, K, adt, bdt, ddt, prec, **extra):
        ldd = (N + 7) // 8 * 8 + 8
        kw = dict(A=rnd(g, M, K + 8, dtype=DT[adt]), B=rnd(g, N, K + 16, dtype=DT[bdt]),
                  D=torch.full((M, ldd), 7.0, dtype=DT[ddt]), M=M, N=N, K=K, a_op=OP_K, b_op=OP_K,
                  a_dtype=adt, b_dtype=bdt, d_dtype=ddt, prec=prec, lda=K + 8, ldb=K + 16, ldd=ldd,
                  batch=1, batch_inner=1, alpha=1.0)
        kw.update(extra)
        return kw

    # 1. plain NT, several tiles, ragged M/N/K tails
    for prec, adt, bdt, ddt, tag in ((0, BF16, BF16, BF16, "bf16"), (0, F32, BF16, F32, "bf16-f32A"), (1, F32, F32, F32, "x3")):
        cases.append((f"gemm_plain_{tag}", "gemm", base(200, 150, 136, adt, bdt, ddt, prec), TOL_X3 if prec else TOL_BF))
    # f32 A operand (converted while staging) in the transposed layouts used by dgrad / wgrad of the fp32 residual-stream gradient
    kw = dict(A=rnd(g, 203, 72), B=rnd(g, 203, 40, dtype=torch.bfloat16), D=torch.zeros(72, 40), M=72, N=40, K=203, a_op=OP_R, b_op=OP_R,
              a_dtype=F32, b_dtype=BF16, d_dtype=F32, prec=0, lda=72, ldb=40, ldd=40, batch=1, batch_inner=1, alpha=1.0)
    cases.append(("gemm_wgrad_f32A_bf16", "gemm", kw, TOL_BF))
    kw = dict(A=rnd(g, 90, 72), B=rnd(g, 72, 136, dtype=torch.bfloat16), D=torch.zeros(90, 136, dtype=torch.bfloat16), M=90, N=136, K=72,
              a_op=OP_K, b_op=OP_R, a_dtype=F32, b_dtype=BF16, d_dtype=BF16, prec=0, lda=72, ldb=136, ldd=136, batch=1, batch_inner=1, alpha=1.0)
    cases.append(("gemm_dgrad_f32A_bf16", "gemm", kw, TOL_BF))
    # 1b. direct-to-LDS fast path (bf16, K % 64 == 0): many K tiles (ring wrap-around), ragged M / N, row groups, two-level batch
    for (M, N, K) in ((200, 150, 256), (128, 128, 64), (300, 260, 704), (48, 1024, 1024)):
        cases.append((f"gemm_fast_{M}x{N}x{K}", "gemm", base(M, N, K, BF16, BF16, BF16, 0), TOL_BF))
    Bn, Mb, K = 3, 37, 128
    XA = rnd(g, Bn, Mb + 5, K + 8, dtype=torch.bfloat16)
    kw = dict(A=XA[:, 5:], B=rnd(g, 70, K, dtype=torch.bfloat16), D=torch.zeros(Bn * Mb, 72), M=Bn * Mb, N=70, K=K, a_op=OP_K, b_op=OP_K,
              a_dtype=BF16, b_dtype=BF16, d_dtype=F32, prec=0, lda=K + 8, ldb=K, ldd=72, a_mb=Mb, a_bs=(Mb + 5) * (K + 8),
              batch=1, batch_inner=1, alpha=1.0, colshift=rnd(g, 70), act=1, n_store=72)
    cases.append(("gemm_fast_rowgroups", "gemm", kw, TOL_BF))
    Z, M, N, K = 4, 150, 40, 192
    kw = dict(A=rnd(g, Z, M, K, dtype=torch.bfloat16), B=rnd(g, Z, N, K, dtype=torch.bfloat16), D=torch.full((2, M, 2 * 40), 3.0, dtype=torch.bfloat16),
              M=M, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=BF16, b_dtype=BF16, d_dtype=BF16, prec=0, lda=K, ldb=K, ldd=80, batch=Z, batch_inner=2,
              a_zo=2 * M * K, a_zi=M * K, b_zo=2 * N * K, b_zi=N * K, d_zo=M * 80, d_zi=40, alpha=1.0, n_store=40)
    cases.append(("gemm_fast_batched2", "gemm", kw, TOL_BF))
    # 1c. 256 x 256 / 8-wave direct-to-LDS kernel (chosen when >= 200 such tiles): ragged edges, batch, epilogue, row groups
    Z, M, N, K = 4, 2096, 2088, 192
    kw = dict(A=rnd(g, Z, M, K, dtype=torch.bfloat16), B=rnd(g, Z, N, K, dtype=torch.bfloat16), D=torch.zeros(Z, M, 2088, dtype=torch.bfloat16),
              M=M, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=BF16, b_dtype=BF16, d_dtype=BF16, prec=0, lda=K, ldb=K, ldd=2088, batch=Z, batch_inner=1,
              a_zo=M * K, b_zo=N * K, d_zo=M * 2088, alpha=1.0, colshift=rnd(g, Z, N), col_zo=N, act=1, n_store=N)
    cases.append(("gemm_fast256_batched_gelu", "gemm", kw, TOL_BF))
    Bn, Mb, K, N = 16, 517, 128, 1030
    XA = rnd(g, Bn, Mb + 3, K, dtype=torch.bfloat16)
    XT = rnd(g, Bn, Mb + 3, 1032)
    kw = dict(A=XA[:, 3:], B=rnd(g, N, K, dtype=torch.bfloat16), D=XT[:, 3:], M=Bn * Mb, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=BF16, b_dtype=BF16,
              d_dtype=F32, prec=0, lda=K, ldb=K, ldd=1032, a_mb=Mb, a_bs=(Mb + 3) * K, d_mb=Mb, d_bs=(Mb + 3) * 1032, batch=1, batch_inner=1,
              alpha=1.0, colshift=rnd(g, N), resid=XT[:, 3:], ldr=1032, r_mb=Mb, r_bs=(Mb + 3) * 1032,
              rowscale=torch.rand(Bn, 2, generator=g), n_prompt=5, n_store=1032)
    cases.append(("gemm_fast256_rowgroups_resid", "gemm", kw, TOL_BF))
    # 1d. the LDS-DMA kernel forced on small shapes (variant = 3): K tails (K % 64 != 0 -> general addressing, zero page), ragged M / N,
    #     one and many K tiles
    for (M, N, K) in ((300, 260, 200), (70, 700, 64), (513, 350, 1096), (256, 128, 8), (1000, 300, 72)):
        kw = base(M, N, K, BF16, BF16, BF16, 0, variant=3, colshift=rnd(g, N), act=1)
        cases.append((f"gemm_dma_forced_{M}x{N}x{K}", "gemm", kw, TOL_BF))
    # 1d'. the 128 x 128 LDS-DMA kernel (variant = 4; two workgroups per CU): K tails / fast addressing, ragged edges, one and many K tiles,
    #      task batches with an inner pair (the decoder's catpair output), A row groups + fp32 residual epilogue
    for (M, N, K) in ((300, 260, 200), (70, 700, 64), (513, 350, 1096), (256, 128, 8), (1000, 300, 72), (640, 300, 1024), (300, 350, 608)):
        kw = base(M, N, K, BF16, BF16, BF16, 0, variant=4, colshift=rnd(g, N), act=1)
        cases.append((f"gemm_dma128_{M}x{N}x{K}", "gemm", kw, TOL_BF))
    # narrow outputs: 32- and 64-column tiles (N = 1, 21, 40, 64), fp32 and bf16 stores, K tail
    for (M, N, K, ddt) in ((700, 21, 352, F32), (300, 1, 64, F32), (513, 40, 200, BF16), (260, 64, 128, BF16), (1000, 7, 72, BF16)):
        kw = base(M, N, K, BF16, BF16, ddt, 0, variant=4, colshift=rnd(g, N))
        cases.append((f"gemm_dma128_narrow_{M}x{N}x{K}", "gemm", kw, TOL_BF))
    Z, M, N, K = 6, 300, 44, 128
    kw = dict(A=rnd(g, Z, M, K, dtype=torch.bfloat16), B=rnd(g, Z, N, K, dtype=torch.bfloat16), D=torch.zeros(Z // 2, M, 96, dtype=torch.bfloat16),
              M=M, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=BF16, b_dtype=BF16, d_dtype=BF16, prec=0, lda=K, ldb=K, ldd=96, batch=Z, batch_inner=2,
              a_zo=2 * M * K, a_zi=M * K, b_zo=2 * N * K, b_zi=N * K, d_zo=M * 96, d_zi=48, alpha=1.0, n_store=48, variant=4,
              colshift=rnd(g, Z, N), col_zo=2 * N, col_zi=N)
    cases.append(("gemm_dma128_batched_pair", "gemm", kw, TOL_BF))
    Bn, Mb, K, N = 6, 117, 192, 300
    XA = rnd(g, Bn, Mb + 3, K, dtype=torch.bfloat16)
    XT = rnd(g, Bn, Mb + 3, 304)
    kw = dict(A=XA[:, 3:], B=rnd(g, N, K, dtype=torch.bfloat16), D=XT[:, 3:], M=Bn * Mb, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=BF16, b_dtype=BF16,
              d_dtype=F32, prec=0, lda=K, ldb=K, ldd=304, a_mb=Mb, a_bs=(Mb + 3) * K, d_mb=Mb, d_bs=(Mb + 3) * 304, batch=1, batch_inner=1,
              alpha=1.0, colshift=rnd(g, N), resid=XT[:, 3:], ldr=304, r_mb=Mb, r_bs=(Mb + 3) * 304,
              rowscale=torch.rand(Bn, 2, generator=g), n_prompt=5, n_store=304, variant=4)
    cases.append(("gemm_dma128_rowgroups_resid", "gemm", kw, TOL_BF))
    # many K tiles / both parities of the tile count (fast addressing: K % 64 == 0) and K tails (general addressing)
    for (M, N, K, v) in ((300, 520, 200, 3), (513, 600, 1096, 3), (520, 700, 1152, 3), (300, 512, 1088, 3), (256, 256, 64, 3), (256, 256, 128, 3)):
        cases.append((f"gemm_dma_sched{v}_{M}x{N}x{K}", "gemm", base(M, N, K, BF16, BF16, BF16, 0, variant=v, colshift=rnd(g, N)), TOL_BF))
    # 1e. specialised interior-tile epilogues (KIND 0..4 of gemm_epilogue_fast) on the DMA kernel (variant 3) and the general kernel
    #     (variant 1), next to edge tiles that take the general epilogue in the same launch; variant 11 = general epilogue everywhere
    for v in (3, 4, 1, 11):
        M, N, K = 600, 520, 136
        A16 = rnd(g, M, K, dtype=torch.bfloat16)
        B16 = rnd(g, N, K, dtype=torch.bfloat16)
        common = dict(A=A16, B=B16, M=M, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=BF16, b_dtype=BF16, prec=0, lda=K, ldb=K,
                      batch=1, batch_inner=1, alpha=1.0, variant=v)
        cases.append((f"gemm_epi_kind0_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528,
                                                               colshift=rnd(g, N), n_store=N), TOL_BF))
        cases.append((f"gemm_epi_kind1_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0), d_dtype=F32, ldd=528, n_store=N), TOL_BF))
        cases.append((f"gemm_epi_kind2_colscale_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528,
                                                                        colshift=rnd(g, N), colscale=rnd(g, N).abs() + 0.5, act=1, n_store=N), TOL_BF))
        cases.append((f"gemm_epi_kind2_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528,
                                                               colshift=rnd(g, N), act=1, aux_out=torch.full((M, 536), 3.0, dtype=torch.bfloat16),
                                                               aux_dtype=BF16, ldaux=536, n_store=N), TOL_BF))
        cases.append((f"gemm_epi_kind2_noaux_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528,
                                                                     colshift=rnd(g, N), act=1, n_store=N), TOL_BF))
        XT = rnd(g, M, 528)
        cases.append((f"gemm_epi_kind3_v{v}", "gemm", dict(common, D=XT, d_dtype=F32, ldd=528, d_mb=100, d_bs=100 * 528, colshift=rnd(g, N),
                                                               resid=XT, ldr=528, r_mb=100, r_bs=100 * 528, rowscale=torch.rand(6, 2, generator=g),
                                                               n_prompt=7, n_store=N), TOL_BF))
        cases.append((f"gemm_epi_kind3_norowscale_v{v}", "gemm", dict(common, D=torch.zeros(M, 528), d_dtype=F32, ldd=528, colshift=rnd(g, N),
                                                                          resid=rnd(g, M, 520), ldr=520, n_store=N), TOL_BF))
        cases.append((f"gemm_epi_kind4_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528, act=3,
                                                               aux_in=rnd(g, M, 536, dtype=torch.bfloat16), aux_dtype=BF16, ldaux=536, n_store=N), TOL_BF))
        # ABI 10: GELU with its derivative taken in the forward (act 5: D = GELU(z), aux_out = GELU'(z)) and the one-multiply backward epilogue
        # (act 6: D = acc * aux_in) — the specialised kinds 2 / 4 on interior tiles, the general epilogue on the ragged ones
        cases.append((f"gemm_epi_kind2_daux_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528,
                                                                    colshift=rnd(g, N), act=5, aux_out=torch.full((M, 536), 3.0, dtype=torch.bfloat16),
                                                                    aux_dtype=BF16, ldaux=536, n_store=N), TOL_BF))
        cases.append((f"gemm_epi_kind4_mulaux_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528, act=6,
                                                                      aux_in=rnd(g, M, 536, dtype=torch.bfloat16), aux_dtype=BF16, ldaux=536, n_store=N), TOL_BF))
        cases.append((f"gemm_colsum_kind4_mulaux_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528, act=6,
                                                                        aux_in=rnd(g, M, 536, dtype=torch.bfloat16), aux_dtype=BF16, ldaux=536, n_store=N,
                                                                        colsum_out=torch.full((N + 3,), 9.0),
                                                                        colsum_ws=scratch(((M + 127) // 128) * ((N + 7) // 8 * 8))), dict(TOL_BF, f32=2e-4)))
        # column sums of the stored tile from the epilogue (colsum_out: the bias gradient of the layer whose output gradient D is) —
        # specialised kinds 0 / 4 on interior tiles + the general epilogue on the ragged ones, fp32 D through the general epilogue
        # (a 1-ulp flip of one rounded bf16 element moves a column sum by ~1e-5 of its magnitude: the fp32 bound is looser than for plain stores)
        TOL_CS = dict(TOL_BF, f32=2e-4)
        csw = lambda: scratch(((M + 127) // 128) * ((N + 7) // 8 * 8))
        cases.append((f"gemm_colsum_kind4_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528, act=3,
                                                                 aux_in=rnd(g, M, 536, dtype=torch.bfloat16), aux_dtype=BF16, ldaux=536, n_store=N,
                                                                 colsum_out=torch.full((N + 3,), 9.0), colsum_ws=csw()), TOL_CS))
        cases.append((f"gemm_colsum_kind0_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0, dtype=torch.bfloat16), d_dtype=BF16, ldd=528,
                                                                 colshift=rnd(g, N), n_store=N, colsum_out=torch.full((N + 3,), 9.0), colsum_ws=csw()), TOL_CS))
        cases.append((f"gemm_colsum_f32_v{v}", "gemm", dict(common, D=torch.full((M, 528), 7.0), d_dtype=F32, ldd=528, n_store=N,
                                                               colsum_out=torch.full((N + 3,), 9.0), colsum_ws=csw()), TOL_CS))
    # 1e'. MTT_SPLIT operands (x = hi + lo bf16 planes) on the LDS-DMA kernel: the fp32-class product as one K-concatenated bf16 GEMM
    #      (gemm_dma_kernel<2>); D fp32 / split planes, every epilogue kind it serves (1, 3, 5, 6 + the general one on ragged tiles),
    #      row groups on A, task batches; K = 64 .. 1088 (both parities of 3 K / 64).  The emulator reads hi + lo in fp64.
    def planes(t):
        hi = t.to(torch.bfloat16)
        return hi, (t - hi.float()).to(torch.bfloat16)
    # round 5: N-edge tiles with <= 128 columns run the kernel's 4 x 2 wave layout (half the MFMAs, 128 B rows staged): 260 / 520 / 300 / 350 and
    # exactly 128 columns take it, 136 and 1024 do not; K = 64 (two steps: only the tail waits), 96 (three), 608 (19: odd), 1088
    for (M, N, K) in ((300, 260, 64), (600, 520, 192), (257, 300, 1088), (1000, 1024, 256), (512, 350, 608), (300, 128, 96), (300, 136, 96)):
        Ah, Al = planes(rnd(g, M, K + 8)); Bh, Bl = planes(rnd(g, N, K + 16))
        common = dict(A=Ah, A_lo=Al, B=Bh, B_lo=Bl, M=M, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=SPLIT, b_dtype=SPLIT, prec=1, lda=K + 8, ldb=K + 16,
                      batch=1, batch_inner=1, alpha=1.0)
        ldd = (N + 7) // 8 * 8 + 8
        cases.append((f"gemm_split_f32out_{M}x{N}x{K}", "gemm", dict(common, D=torch.full((M, ldd), 7.0), d_dtype=F32, ldd=ldd, colshift=rnd(g, N)), TOL_X3))
        cases.append((f"gemm_split_splitout_{M}x{N}x{K}", "gemm", dict(common, D=torch.full((M, ldd), 7.0, dtype=torch.bfloat16),
                                                                          D_lo=torch.full((M, ldd), 5.0, dtype=torch.bfloat16), d_dtype=SPLIT, ldd=ldd,
                                                                          colshift=rnd(g, N), n_store=(N + 7) // 8 * 8), TOL_SPLIT_D))
        cases.append((f"gemm_split_gelu_aux_{M}x{N}x{K}", "gemm", dict(common, D=torch.full((M, ldd), 7.0, dtype=torch.bfloat16),
                                                                        D_lo=torch.full((M, ldd), 5.0, dtype=torch.bfloat16), d_dtype=SPLIT, ldd=ldd,
                                                                        colshift=rnd(g, N), act=1, aux_out=torch.full((M, ldd + 8), 3.0, dtype=torch.bfloat16),
                                                                        aux_dtype=BF16, ldaux=ldd + 8), TOL_SPLIT_D))
        cases.append((f"gemm_split_gelu_daux_{M}x{N}x{K}", "gemm", dict(common, D=torch.full((M, ldd), 7.0, dtype=torch.bfloat16),
                                                                         D_lo=torch.full((M, ldd), 5.0, dtype=torch.bfloat16), d_dtype=SPLIT, ldd=ldd,
                                                                         colshift=rnd(g, N), act=5, aux_out=torch.full((M, ldd + 8), 3.0, dtype=torch.bfloat16),
                                                                         aux_dtype=BF16, ldaux=ldd + 8), TOL_SPLIT_D))
        XT = rnd(g, M, ldd)
        cases.append((f"gemm_split_resid_{M}x{N}x{K}", "gemm", dict(common, D=XT, d_dtype=F32, ldd=ldd, d_mb=100, d_bs=100 * ldd, colshift=rnd(g, N),
                                                                     resid=XT, ldr=ldd, r_mb=100, r_bs=100 * ldd, rowscale=torch.rand(M // 100 + 1, 2, generator=g),
                                                                     n_prompt=7), TOL_X3))
    Bn, Mb, K, N = 5, 117, 128, 520
    XAh, XAl = planes(rnd(g, Bn, Mb + 3, K))
    Bh, Bl = planes(rnd(g, N, K))
    kw = dict(A=XAh[:, 3:], A_lo=XAl[:, 3:], B=Bh, B_lo=Bl, D=torch.zeros(Bn * Mb, N), M=Bn * Mb, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=SPLIT,
              b_dtype=SPLIT, d_dtype=F32, prec=1, lda=K, ldb=K, ldd=N, a_mb=Mb, a_bs=(Mb + 3) * K, batch=1, batch_inner=1, alpha=1.0, colshift=rnd(g, N))
    cases.append(("gemm_split_rowgroups", "gemm", kw, TOL_X3))
    Z, M, N, K = 3, 300, 264, 192
    Ah, Al = planes(rnd(g, Z, M, K)); Bh, Bl = planes(rnd(g, Z, N, K))
    kw = dict(A=Ah, A_lo=Al, B=Bh, B_lo=Bl, D=torch.zeros(Z, M, N), M=M, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=SPLIT, b_dtype=SPLIT, d_dtype=F32, prec=1,
              lda=K, ldb=K, ldd=N, batch=Z, batch_inner=1, a_zo=M * K, b_zo=N * K, d_zo=M * N, alpha=1.0, colshift=rnd(g, Z, N), col_zo=N)
    cases.append(("gemm_split_batched", "gemm", kw, TOL_X3))
    # 1e+. the implicit-GEMM 3x3 conv on MTT_SPLIT planes (mtt_gemm variant 9, gemm_ring3_kernel<true>): channel pitch 32 / 96 / 352 (the
    #      fea_fuse width), dilation 1 / 2, mirrored taps, ragged M and N tiles, task batches, BN-folded GELU epilogue; K = 9 * Cp
    for (Bc, H, W, Ci, Cp, Co, dil, flip, Zc) in ((2, 9, 11, 30, 32, 40, 1, 0, 1), (1, 16, 20, 90, 96, 300, 2, 0, 2), (3, 8, 8, 350, 352, 350, 1, 1, 1)):
        rows = Bc * H * W
        xa = rnd(g, Zc, rows, Cp); xa[..., Ci:] = 0
        wa = rnd(g, Zc, Co, 9, Cp) * 0.2; wa[..., Ci:] = 0
        Ah, Al = planes(xa); Bh, Bl = planes(wa.reshape(Zc, Co, 9 * Cp))
        Cop = (Co + 7) // 8 * 8
        kw = dict(A=Ah, A_lo=Al, B=Bh, B_lo=Bl, D=torch.full((Zc, rows, Cop), 7.0), M=rows, N=Co, K=9 * Cp, a_op=OP_CONV_K, b_op=OP_K, a_dtype=SPLIT,
                  b_dtype=SPLIT, d_dtype=F32, prec=1, lda=Cp, ldb=9 * Cp, ldd=Cop, batch=Zc, batch_inner=1, a_zo=rows * Cp, b_zo=Co * 9 * Cp,
                  d_zo=rows * Cop, conv=dict(H=H, W=W, C=Ci, Cp=Cp, dil=dil, flip=flip), alpha=1.0, n_store=Cop, colshift=rnd(g, Zc, Co), col_zo=Co)
        cases.append((f"conv3_split_{Bc}x{H}x{W}_c{Ci}_dil{dil}_flip{flip}", "gemm", kw, TOL_X3))
        cases.append((f"conv3_split_bnfold_gelu_{Bc}x{H}x{W}_c{Ci}", "gemm", dict(kw, D=torch.full((Zc, rows, Cop), 7.0), colscale=rnd(g, Zc, Co), act=1), TOL_X3))
    # 1e+a. the prediction dgrad of TaskHeadsFn on the 128-row LDS-DMA kernel (forced variant 4): K = pad8(n) = 8 / 24 — less than one 64-deep K
    #       step, chunks past K read the zero page — a tall M, N = 352 / 176, bf16 output with n_store = pitch
    for (M, N, K) in ((5000, 352, 24), (3001, 176, 8)):
        kw = dict(A=rnd(g, M, K, dtype=torch.bfloat16), B=rnd(g, N, K, dtype=torch.bfloat16), D=torch.full((M, N), 7.0, dtype=torch.bfloat16), M=M, N=N, K=K,
                  a_op=OP_K, b_op=OP_K, a_dtype=BF16, b_dtype=BF16, d_dtype=BF16, prec=0, lda=K, ldb=K, ldd=N, batch=1, batch_inner=1, alpha=1.0, n_store=N, variant=4)
        cases.append((f"gemm_dma128_shortk_{M}x{N}x{K}", "gemm", kw, TOL_BF))
    # 1e+b. the same implicit-GEMM conv in bf16 on the LDS-DMA ring (mtt_gemm variant 12, gemm_ringc_kernel; round 5): forced by variant 3 on
    #       these small maps (AUTO takes it from 2048 pixel rows), channel pitch 32 / 96 / 352, dilation, mirrored taps (the dgrad form),
    #       ragged M and N tiles, task batches, bias / BN-folded GELU epilogues, bf16 and fp32 outputs
    for (Bc, H, W, Ci, Cp, Co, dil, flip, Zc) in ((2, 9, 11, 30, 32, 40, 1, 0, 1), (1, 16, 20, 90, 96, 300, 2, 0, 2), (3, 8, 8, 350, 352, 350, 1, 1, 1),
                                                  (5, 24, 20, 64, 64, 264, 1, 1, 1)):
        rows = Bc * H * W
        xa = rnd(g, Zc, rows, Cp, dtype=torch.bfloat16); xa[..., Ci:] = 0
        wa = (rnd(g, Zc, Co, 9, Cp) * 0.2).to(torch.bfloat16); wa[..., Ci:] = 0
        Cop = (Co + 7) // 8 * 8
        kw = dict(A=xa, B=wa.reshape(Zc, Co, 9 * Cp), D=torch.full((Zc, rows, Cop), 7.0, dtype=torch.bfloat16), M=rows, N=Co, K=9 * Cp, a_op=OP_CONV_K,
                  b_op=OP_K, a_dtype=BF16, b_dtype=BF16, d_dtype=BF16, prec=0, lda=Cp, ldb=9 * Cp, ldd=Cop, batch=Zc, batch_inner=1, a_zo=rows * Cp,
                  b_zo=Co * 9 * Cp, d_zo=rows * Cop, conv=dict(H=H, W=W, C=Ci, Cp=Cp, dil=dil, flip=flip), alpha=1.0, n_store=Cop,
                  colshift=rnd(g, Zc, Co), col_zo=Co, variant=3)
        cases.append((f"conv3_ring_bf16_{Bc}x{H}x{W}_c{Ci}_dil{dil}_flip{flip}", "gemm", kw, TOL_BF))
        cases.append((f"conv3_ring_bf16_bnfold_gelu_f32out_{Bc}x{H}x{W}_c{Ci}", "gemm",
                      dict(kw, D=torch.full((Zc, rows, Cop), 7.0), d_dtype=F32, colscale=rnd(g, Zc, Co), act=1), TOL_BF))
    # 1e++. tall GEMMs with a handful of outputs in the fp32-class modes (mtt_gemm variant 11, gemm_f32n_kernel: exact fp32 MFMA): the head
    #       predictions' shapes (K = 352, N = 1 / 21), N = 32, K = 8 and 1024, ragged M (not a multiple of 32 / of a workgroup's 512 rows),
    #       channel padding columns (n_store), task batches
    for (M, N, K, Zc) in ((5000, 21, 352, 1), (2049, 1, 352, 1), (4096, 32, 8, 1), (3000, 7, 1024, 2)):
        ldd = (N + 7) // 8 * 8 + 8
        kw = dict(A=rnd(g, Zc, M, K + 8), B=rnd(g, Zc, N, K), D=torch.full((Zc, M, ldd), 7.0), M=M, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=F32, b_dtype=F32,
                  d_dtype=F32, prec=1, lda=K + 8, ldb=K, ldd=ldd, batch=Zc, batch_inner=1, a_zo=M * (K + 8), b_zo=N * K, d_zo=M * ldd, alpha=1.0,
                  colshift=rnd(g, Zc, N), col_zo=N, n_store=(N + 7) // 8 * 8)
        cases.append((f"gemm_f32n_{M}x{N}x{K}_z{Zc}", "gemm", kw, TOL_X3))
    # ... and with the A PROLOGUE of ABI 10 (mtt_gemm_desc.a_scale): BatchNorm scale / shift + GELU / ReLU / nothing applied to the operand
    # while it is loaded, the transformed operand stored rounded to bf16 (a_aux16; untouched 3.0 fill in its pitch padding) — the head
    # predictions' shape, a ragged small M (the prologue form takes any row count), K = 1024 without the side copy
    for (M, N, K, a_act, aux) in ((5000, 21, 352, 1, True), (349, 7, 352, 1, True), (3000, 3, 1024, 2, False), (700, 32, 64, 0, True)):
        ldd = (N + 7) // 8 * 8 + 8
        sc = rnd(g, K).abs() + 0.3; sh = rnd(g, K); sc[K - 2:] = 0.0; sh[K - 2:] = 0.0          # (channel padding: scale = shift = 0)
        kw = dict(A=rnd(g, M, K + 8), B=rnd(g, N, K), D=torch.full((M, ldd), 7.0), M=M, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=F32, b_dtype=F32,
                  d_dtype=F32, prec=1, lda=K + 8, ldb=K, ldd=ldd, batch=1, batch_inner=1, alpha=1.0, colshift=rnd(g, N), n_store=(N + 7) // 8 * 8,
                  a_scale=sc, a_shift=sh, a_act=a_act, a_aux16=torch.full((M, K + 16), 3.0, dtype=torch.bfloat16) if aux else None,
                  ld_a16=K + 16 if aux else 0)
        cases.append((f"gemm_f32n_prologue_{M}x{N}x{K}_act{a_act}", "gemm", kw, TOL_X3))
    # 1e''. bf16 arithmetic on fp32-STORED operands (rounded while staged: general kernel MODE 3 = f32 x f32, MODE 4 = bf16 x f32), in the
    #       layouts the bf16 backward of the x3-forward training mode uses them: dgrad (OP_K x OP_R), wgrad (OP_R x OP_R), conv dgrad / wgrad
    for tag, adt, bdt in (("f32f32", F32, F32), ("bf16f32", BF16, F32)):
        cases.append((f"gemm_bf16prec_{tag}_plain", "gemm", base(200, 150, 136, adt, bdt, F32, 0), TOL_BF))
        kw = dict(A=rnd(g, 90, 72, dtype=DT[adt]), B=rnd(g, 72, 136, dtype=DT[bdt]), D=torch.zeros(90, 136), M=90, N=136, K=72,
                  a_op=OP_K, b_op=OP_R, a_dtype=adt, b_dtype=bdt, d_dtype=F32, prec=0, lda=72, ldb=136, ldd=136, batch=1, batch_inner=1, alpha=1.0)
        cases.append((f"gemm_bf16prec_{tag}_dgrad", "gemm", kw, TOL_BF))
        kw = dict(A=rnd(g, 203, 72, dtype=DT[adt]), B=rnd(g, 203, 40, dtype=DT[bdt]), D=torch.zeros(72, 40), M=72, N=40, K=203, a_op=OP_R, b_op=OP_R,
                  a_dtype=adt, b_dtype=bdt, d_dtype=F32, prec=0, lda=72, ldb=40, ldd=40, batch=1, batch_inner=1, alpha=1.0)
        cases.append((f"gemm_bf16prec_{tag}_wgrad", "gemm", kw, TOL_BF))
        Bn, H, W, Ci, Co = 2, 9, 7, 24, 40
        X = rnd(g, Bn * H * W, Ci, dtype=DT[bdt]); dY = rnd(g, Bn * H * W, Co, dtype=DT[adt])
        kw = dict(A=dY, B=X, D=torch.full((Co, 9 * Ci), 5.0), M=Co, N=9 * Ci, K=Bn * H * W, a_op=OP_R, b_op=OP_CONV_R, a_dtype=adt, b_dtype=bdt,
                  d_dtype=F32, prec=0, lda=Co, ldb=Ci, ldd=9 * Ci, batch=1, batch_inner=1, conv=dict(H=H, W=W, C=Ci, Cp=Ci, dil=1, flip=0), alpha=1.0)
        cases.append((f"gemm_bf16prec_{tag}_conv_wgrad", "gemm", kw, TOL_BF))
    # 1f. token-major weight-gradient kernel (gemm_tn_kernel, forced by variant 3 on small shapes): ragged M / N / K, several K tiles of
    #     both parities, batch slabs (split K), asymmetric operands (an M <-> N swap or a k permutation cannot pass), then the 3x3 conv
    #     weight gradient (implicit im2col^T on the B side; halo, dilation, channel padding)
    for (Mtok, Nf, Kf) in ((203, 72, 40), (64, 256, 256), (1000, 300, 520), (129, 264, 16), (448, 520, 264)):
        la = (Nf + 7) // 8 * 8 + 8
        kw = dict(A=rnd(g, Mtok, la, dtype=torch.bfloat16), B=rnd(g, Mtok, Kf + 16, dtype=torch.bfloat16), D=torch.full((Nf, Kf + 8), 5.0),
                  M=Nf, N=Kf, K=Mtok, a_op=OP_R, b_op=OP_R, a_dtype=BF16, b_dtype=BF16, d_dtype=F32, prec=0, lda=la, ldb=Kf + 16, ldd=Kf + 8,
                  batch=1, batch_inner=1, alpha=1.0, variant=3)
        cases.append((f"gemm_tn_{Mtok}x{Nf}x{Kf}", "gemm", kw, TOL_BF))
    Z, c, Nf, Kf = 3, 192, 300, 264
    kw = dict(A=rnd(g, Z * c, Nf + 4, dtype=torch.bfloat16), B=rnd(g, Z * c, Kf, dtype=torch.bfloat16), D=torch.full((Z, Nf, Kf), 5.0),
              M=Nf, N=Kf, K=c, a_op=OP_R, b_op=OP_R, a_dtype=BF16, b_dtype=BF16, d_dtype=F32, prec=0, lda=Nf + 4, ldb=Kf, ldd=Kf,
              batch=Z, batch_inner=1, a_zo=c * (Nf + 4), b_zo=c * Kf, d_zo=Nf * Kf, alpha=1.0, variant=3)
    cases.append(("gemm_tn_batched_slabs", "gemm", kw, TOL_BF))
    for dil, (Bn, H, W, Ci, Co) in ((1, (2, 9, 7, 20, 30)), (2, (2, 9, 7, 20, 30)), (1, (3, 20, 17, 44, 300)), (1, (2, 16, 16, 350, 350))):
        Cp, Cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
        X = rnd(g, Bn * H * W, Cp, dtype=torch.bfloat16); X[..., Ci:] = 0
        dY = rnd(g, Bn * H * W, Cop, dtype=torch.bfloat16); dY[..., Co:] = 0
        kw = dict(A=dY, B=X, D=torch.full((Co, 9 * Cp), 5.0), M=Co, N=9 * Cp, K=Bn * H * W, a_op=OP_R, b_op=OP_CONV_R,
                  a_dtype=BF16, b_dtype=BF16, d_dtype=F32, prec=0, lda=Cop, ldb=Cp, ldd=9 * Cp, batch=1, batch_inner=1,
                  conv=dict(H=H, W=W, C=Ci, Cp=Cp, dil=dil, flip=0), alpha=1.0, variant=3)
        cases.append((f"gemm_tn_conv3_wgrad_d{dil}_{Bn}x{H}x{W}x{Ci}to{Co}", "gemm", kw, TOL_BF))
    # two image slices as a (task, slice) batch, as Conv3x3Fn.backward launches it
    Bn, H, W, Ci, Co, S = 4, 8, 6, 24, 40, 2
    X = rnd(g, 2, Bn * H * W, Ci, dtype=torch.bfloat16)
    dY = rnd(g, 2, Bn * H * W, Co, dtype=torch.bfloat16)
    cc = Bn * H * W // S
    kw = dict(A=dY, B=X, D=torch.full((2, S, Co, 9 * Ci), 5.0), M=Co, N=9 * Ci, K=cc, a_op=OP_R, b_op=OP_CONV_R, a_dtype=BF16, b_dtype=BF16,
              d_dtype=F32, prec=0, lda=Co, ldb=Ci, ldd=9 * Ci, batch=2 * S, batch_inner=S, a_zo=Bn * H * W * Co, a_zi=cc * Co,
              b_zo=Bn * H * W * Ci, b_zi=cc * Ci, d_zo=S * Co * 9 * Ci, d_zi=Co * 9 * Ci, conv=dict(H=H, W=W, C=Ci, Cp=Ci, dil=1, flip=0),
              alpha=1.0, variant=3)
    cases.append(("gemm_tn_conv3_wgrad_slices", "gemm", kw, TOL_BF))
    # 2. asymmetric identity check (A = I) catches transposed C layout
    kw = base(128, 128, 128, F32, F32, F32, 1)
    kw["A"] = torch.eye(128, 136)
    cases.append(("gemm_identityA_x3", "gemm", kw, TOL_X3))
    # 3. epilogue: bias, colscale, gelu, aux_out, resid (aliasing D), rowscale with row groups, n_store, alpha
    for prec, adt in ((0, BF16), (1, F32)):
        M, N, K = 2 * 37, 70, 64
        Bn, Mb = 2, 37
        XT = rnd(g, Bn, Mb + 5, 80)
        kw = dict(A=rnd(g, M, K, dtype=DT[adt]), B=rnd(g, N, K, dtype=DT[adt]), D=XT[:, 5:], M=M, N=N, K=K, a_op=OP_K, b_op=OP_K,
                  a_dtype=adt, b_dtype=adt, d_dtype=F32, prec=prec, lda=K, ldb=K, ldd=80, d_mb=Mb, d_bs=(Mb + 5) * 80,
                  batch=1, batch_inner=1, alpha=0.5, colscale=rnd(g, N).abs() + 0.5, colshift=rnd(g, N), act=1,
                  aux_out=torch.zeros(M, 72, dtype=DT[adt]), aux_dtype=adt, ldaux=72,
                  rowscale=torch.tensor([[1.0, 0.5], [2.0, 0.0]]), n_prompt=3,
                  resid=XT[:, 5:], ldr=80, r_mb=Mb, r_bs=(Mb + 5) * 80, n_store=N)
        cases.append((f"gemm_epilogue_{'x3' if prec else 'bf16'}", "gemm", kw, TOL_X3 if prec else TOL_BF))
    # 4. batched two-level with broadcast resid rows and n_store padding
    for prec, adt in ((0, BF16), (1, F32)):
        Z, M, N, K = 4, 50, 20, 40
        kw = dict(A=rnd(g, Z, M, K, dtype=DT[adt]), B=rnd(g, Z, N, K, dtype=DT[adt]),
                  D=torch.full((2, M, 2 * 24), 3.0, dtype=DT[adt]), M=M, N=N, K=K, a_op=OP_K, b_op=OP_K,
                  a_dtype=adt, b_dtype=adt, d_dtype=adt, prec=prec, lda=K, ldb=K, ldd=48, batch=Z, batch_inner=2,
                  a_zo=2 * M * K, a_zi=M * K, b_zo=2 * N * K, b_zi=N * K, d_zo=M * 48, d_zi=24, alpha=1.0,
                  colshift=rnd(g, Z, N), col_zo=2 * N, col_zi=N, n_store=24)
        cases.append((f"gemm_batched2_{'x3' if prec else 'bf16'}", "gemm", kw, TOL_X3 if prec else TOL_BF))
    # 5. dgrad layout (A: K-contig, B: row-contig) with GELU_BWD on aux_in, and A row groups
    for prec, adt in ((0, BF16), (1, F32)):
        M, Nf, Kf = 90, 72, 136      # forward y[M,Nf] = x[M,Kf] W[Nf,Kf]^T ; dgrad: dx[M,Kf] = dy[M,Nf] W
        kw = dict(A=rnd(g, M, Nf, dtype=DT[adt]), B=rnd(g, Nf, Kf, dtype=DT[adt]), D=torch.zeros(M, Kf, dtype=DT[adt]),
                  M=M, N=Kf, K=Nf, a_op=OP_K, b_op=OP_R, a_dtype=adt, b_dtype=adt, d_dtype=adt, prec=prec,
                  lda=Nf, ldb=Kf, ldd=Kf, batch=1, batch_inner=1, alpha=1.0, act=3,
                  aux_in=rnd(g, M, Kf, dtype=DT[adt]), aux_dtype=adt, ldaux=Kf)
        cases.append((f"gemm_dgrad_{'x3' if prec else 'bf16'}", "gemm", kw, TOL_X3 if prec else TOL_BF))
        cases.append((f"gemm_dgrad_mulaux_{'x3' if prec else 'bf16'}", "gemm", dict(kw, act=6, D=torch.zeros(M, Kf, dtype=DT[adt])), TOL_X3 if prec else TOL_BF))
    # 6. wgrad layout (both row-contig, reduction over tokens not a multiple of 8)
    for prec, adt in ((0, BF16), (1, F32)):
        Mtok, Nf, Kf = 203, 72, 40
        kw = dict(A=rnd(g, Mtok, Nf, dtype=DT[adt]), B=rnd(g, Mtok, Kf, dtype=DT[adt]), D=torch.zeros(Nf, Kf),
                  M=Nf, N=Kf, K=Mtok, a_op=OP_R, b_op=OP_R, a_dtype=adt, b_dtype=adt, d_dtype=F32, prec=prec,
                  lda=Nf, ldb=Kf, ldd=Kf, batch=1, batch_inner=1, alpha=1.0)
        cases.append((f"gemm_wgrad_{'x3' if prec else 'bf16'}", "gemm", kw, TOL_X3 if prec else TOL_BF))
    # 7. implicit 3x3 conv forward / dgrad (flip) / dilation 2, C not a multiple of 8, batched over 2 "tasks"
    for prec, adt in ((0, BF16), (1, F32)):
        for dil, flip in ((1, 0), (2, 0), (1, 1)):
            Bn, H, W, Ci, Co = 2, 9, 7, 20, 30
            Cp, Cop = 24, 32
            X = rnd(g, 2, Bn * H * W, Cp, dtype=DT[adt]); X[..., Ci:] = 0
            Wt = rnd(g, 2, Co, 9, Cp, dtype=DT[adt]); Wt[..., Ci:] = 0
            kw = dict(A=X, B=Wt, D=torch.zeros(2, Bn * H * W, Cop, dtype=DT[adt]), M=Bn * H * W, N=Co, K=9 * Cp,
                      a_op=OP_CONV_K, b_op=OP_K, a_dtype=adt, b_dtype=adt, d_dtype=adt, prec=prec, lda=Cp, ldb=9 * Cp, ldd=Cop,
                      batch=2, batch_inner=1, a_zo=Bn * H * W * Cp, b_zo=Co * 9 * Cp, d_zo=Bn * H * W * Cop,
                      conv=dict(H=H, W=W, C=Ci, Cp=Cp, dil=dil, flip=flip), alpha=1.0,
                      colshift=rnd(g, 2, Co), col_zo=Co, act=1, n_store=Cop)
            cases.append((f"gemm_conv3_d{dil}f{flip}_{'x3' if prec else 'bf16'}", "gemm", kw, TOL_X3 if prec else TOL_BF))
    # 8. 3x3 conv wgrad (A = dY row-contig, B = gathered X)
    for prec, adt in ((0, BF16), (1, F32)):
        Bn, H, W, Ci, Co = 2, 9, 7, 20, 30
        Cp, Cop = 24, 32
        X = rnd(g, Bn * H * W, Cp, dtype=DT[adt]); X[..., Ci:] = 0
        dY = rnd(g, Bn * H * W, Cop, dtype=DT[adt])
        kw = dict(A=dY, B=X, D=torch.zeros(Co, 9 * Cp), M=Co, N=9 * Cp, K=Bn * H * W, a_op=OP_R, b_op=OP_CONV_R,
                  a_dtype=adt, b_dtype=adt, d_dtype=F32, prec=prec, lda=Cop, ldb=Cp, ldd=9 * Cp, batch=1, batch_inner=1,
                  conv=dict(H=H, W=W, C=Ci, Cp=Cp, dil=1, flip=0), alpha=1.0)
        cases.append((f"gemm_conv3_wgrad_{'x3' if prec else 'bf16'}", "gemm", kw, TOL_X3 if prec else TOL_BF))
    # 9. ConvTranspose 2x2 pixel-shuffle store
    for prec, adt in ((0, BF16), (1, F32)):
        Bn, H, W, Ci, Co = 2, 5, 6, 24, 10
        Cop = 16
        kw = dict(A=rnd(g, Bn * H * W, Ci, dtype=DT[adt]), B=rnd(g, 4 * Co, Ci, dtype=DT[adt]),
                  D=torch.zeros(Bn * 4 * H * W, Cop, dtype=DT[adt]), M=Bn * H * W, N=4 * Co, K=Ci, a_op=OP_K, b_op=OP_K,
                  a_dtype=adt, b_dtype=adt, d_dtype=adt, prec=prec, lda=Ci, ldb=Ci, ldd=Cop, batch=1, batch_inner=1, alpha=1.0,
                  colshift=rnd(g, 4 * Co), store_mode=1, ps_H=H, ps_W=W, ps_Co=Co)
        cases.append((f"gemm_pixshuf_{'x3' if prec else 'bf16'}", "gemm", kw, TOL_X3 if prec else TOL_BF))
    return cases


def attn_cases():
    cases = []
    g = torch.Generator().manual_seed(12)
    for (B, N, nH, T) in ((2, 150, 2, 6), (1, 30, 2, 4), (1, 257, 1, 0)):
        for prec, adt in ((0, BF16), (1, F32)):
            C = nH * 64
            kw = dict(qkv=rnd(g, B * N, 3 * C, dtype=DT[adt]), out=torch.zeros(B * N, C, dtype=DT[adt]),
                      rawlog=torch.zeros(B, nH, max(T, 1), N) if T else None, lse=torch.zeros(B, nH, N),
                      B=B, N=N, nH=nH, T=T, dtype=adt, prec=prec, scale=0.125)
            cases.append((f"attn_B{B}N{N}T{T}_{'x3' if prec else 'bf16'}", "attn_fwd", kw,
                          dict(f32=3e-5 if prec else 2e-3, bf16=6e-3)))
    # flash backward (bf16 mode): out / lse inputs come from the fp64 emulator of the forward
    for (B, N, nH, T) in ((2, 150, 2, 6), (1, 257, 1, 0), (1, 64, 1, 3)):
        C = nH * 64
        qkv = rnd(g, B * N, 3 * C, dtype=DT[BF16])
        fw = dict(qkv=qkv, out=torch.zeros(B * N, C, dtype=DT[BF16]), rawlog=None, lse=torch.zeros(B, nH, N), B=B, N=N, nH=nH, T=0,
                  dtype=BF16, prec=0, scale=0.125)
        abi_emul.call("attn_fwd", **fw)
        kw = dict(qkv=qkv, out=fw["out"], rawlog=None, lse=fw["lse"], B=B, N=N, nH=nH, T=T, dtype=BF16, prec=0, scale=0.125,
                  xargs=[rnd(g, B * N, C, dtype=DT[BF16]), rnd(g, B, nH, T, N) * 0.05 if T else None,
                         torch.zeros(B * N, 3 * C, dtype=DT[BF16]), torch.zeros(B, nH, 2, (N + 3) // 4 * 4)])
        cases.append((f"attn_bwd_B{B}N{N}T{T}", "attn_bwd", kw, dict(f32=5e-3, bf16=1.5e-2)))
    # the A/B variants of the flash kernels (mtt_attn_desc.variant 3 = register-staged tiles, 2 = the first form) stay covered
    for variant in (3, 2):
        B, N, nH, T = 2, 150, 2, 6
        C = nH * 64
        kw = dict(qkv=rnd(g, B * N, 3 * C, dtype=DT[BF16]), out=torch.zeros(B * N, C, dtype=DT[BF16]), rawlog=torch.zeros(B, nH, T, N),
                  lse=torch.zeros(B, nH, N), B=B, N=N, nH=nH, T=T, dtype=BF16, prec=0, scale=0.125, variant=variant)
        cases.append((f"attn_B{B}N{N}T{T}_bf16_variant{variant}", "attn_fwd", kw, dict(f32=2e-3, bf16=6e-3)))
        fw = dict(kw, rawlog=None, T=0, out=torch.zeros(B * N, C, dtype=DT[BF16]), lse=torch.zeros(B, nH, N))
        fw.pop("variant")
        abi_emul.call("attn_fwd", **fw)
        kb = dict(qkv=kw["qkv"], out=fw["out"], rawlog=None, lse=fw["lse"], B=B, N=N, nH=nH, T=T, dtype=BF16, prec=0, scale=0.125, variant=variant,
                  xargs=[rnd(g, B * N, C, dtype=DT[BF16]), rnd(g, B, nH, T, N) * 0.05, torch.zeros(B * N, 3 * C, dtype=DT[BF16]),
                         torch.zeros(B, nH, 2, (N + 3) // 4 * 4)])
        cases.append((f"attn_bwd_B{B}N{N}T{T}_variant{variant}", "attn_bwd", kb, dict(f32=5e-3, bf16=1.5e-2)))
    # x3 attention on split planes (qkv hi / lo in, out hi / lo + lse + raw prompt logits out): the x3f mode's forward.  Ragged ends:
    # N = 257 / 70 / 1030 end in a key tile of <= 16 keys and in a 16-row sub-block without query rows; 150 in neither
    for (B, N, nH, T) in ((2, 150, 2, 6), (1, 257, 1, 0), (1, 70, 2, 3), (1, 1030, 1, 6)):
        C = nH * 64
        q = rnd(g, B * N, 3 * C)
        qh = q.to(torch.bfloat16)
        kw = dict(qkv=qh, qkv_lo=(q - qh.float()).to(torch.bfloat16), out=torch.zeros(B * N, C, dtype=torch.bfloat16),
                  out_lo=torch.zeros(B * N, C, dtype=torch.bfloat16), rawlog=torch.zeros(B, nH, max(T, 1), N) if T else None,
                  lse=torch.zeros(B, nH, N), B=B, N=N, nH=nH, T=T, dtype=SPLIT, prec=1, scale=0.125)
        cases.append((f"attn_B{B}N{N}T{T}_split", "attn_fwd", kw, dict(f32=3e-5, bf16=6e-3, split=3e-5, split_pairs=[("out", "out_lo")])))
        kp = dict(kw, out=torch.zeros(B * N, C, dtype=torch.bfloat16), out_lo=torch.zeros(B * N, C, dtype=torch.bfloat16),
                  rawlog=torch.zeros(B, nH, max(T, 1), N) if T else None, lse=torch.zeros(B, nH, N), variant=1)      # the register-staged x3 kernel on planes
        cases.append((f"attn_B{B}N{N}T{T}_split_plain", "attn_fwd", kp, dict(f32=3e-5, bf16=6e-3, split=3e-5, split_pairs=[("out", "out_lo")])))
    # softmax spike (forces large running-max jumps across tiles)
    B, N, nH, T = 1, 200, 1, 2
    q = rnd(g, B * N, 3 * 64)
    q[70, 64:128] *= 12.0
    kw = dict(qkv=q, out=torch.zeros(B * N, 64), rawlog=torch.zeros(B, nH, T, N), lse=torch.zeros(B, nH, N),
              B=B, N=N, nH=nH, T=T, dtype=F32, prec=1, scale=0.125)
    cases.append(("attn_spike_x3", "attn_fwd", kw, dict(f32=3e-5, bf16=6e-3)))
    return cases


def row_cases():
    cases = []
    g = torch.Generator().manual_seed(13)
    # LayerNorm writing split planes (+ the optional fp32 copy), and the fp32 -> split cast
    for want32 in (False, True):
        rows, C = 37, 128
        kw = dict(x=rnd(g, rows, C + 4), y=torch.zeros(rows, C, dtype=torch.bfloat16), y_lo=torch.zeros(rows, C, dtype=torch.bfloat16),
                  gamma=rnd(g, C), beta=rnd(g, C), mean=torch.zeros(rows), rstd=torch.zeros(rows), rows=rows, C=C, ldx=C + 4, ldy=C, y_dtype=SPLIT, eps=1e-6)
        if want32:
            kw.update(y32=torch.zeros(rows, C + 8), ldy32=C + 8)
        cases.append((f"ln_fwd_split_{int(want32)}", "layernorm_fwd", kw, dict(f32=1e-5, bf16=5e-3, split=1e-5, split_pairs=[("y", "y_lo")])))
    for (rows, cols, lds) in ((37, 128, 132), (50, 45, 45), (9, 300, 304)):
        ldd = (cols + 7) // 8 * 8
        args = [rnd(g, rows, lds), torch.full((rows, ldd), 3.0, dtype=torch.bfloat16), torch.full((rows, ldd), 3.0, dtype=torch.bfloat16), rows, cols, lds, ldd]
        cases.append((f"split_cast_{rows}x{cols}", "split_cast", dict(args=args), dict(f32=1e-6, bf16=5e-3, split=1e-5, split_args=(1, 2))))
    for ydt in (F32, BF16):
        rows, C = 37, 128
        kw = dict(x=rnd(g, rows, C + 4), y=torch.zeros(rows, C, dtype=DT[ydt]), gamma=rnd(g, C), beta=rnd(g, C),
                  mean=torch.zeros(rows), rstd=torch.zeros(rows), rows=rows, C=C, ldx=C + 4, ldy=C, y_dtype=ydt, eps=1e-6)
        cases.append((f"ln_fwd_{ydt}", "layernorm_fwd", kw, TOL_ROW))
        # register-resident rows at the encoder width and a ragged one (C = 1024 / 300), more rows than one grid pass; C = 1500: the streaming kernel
        for (rows2, C2) in ((70, 1024), (9, 300), (5, 1500)):
            kw = dict(x=rnd(g, rows2, C2 + 4), y=torch.zeros(rows2, C2 + 8, dtype=DT[ydt]), gamma=rnd(g, C2), beta=rnd(g, C2),
                      mean=torch.zeros(rows2), rstd=torch.zeros(rows2), rows=rows2, C=C2, ldx=C2 + 4, ldy=C2 + 8, y_dtype=ydt, eps=1e-6)
            cases.append((f"ln_fwd_{ydt}_{rows2}x{C2}", "layernorm_fwd", kw, TOL_ROW))
        x = rnd(g, rows, C)
        mean = x.mean(-1); rstd = torch.rsqrt(x.var(-1, unbiased=False) + 1e-6)
        kw = dict(x=x, dy=rnd(g, rows, C, dtype=DT[ydt]), gamma=rnd(g, C), mean=mean, rstd=rstd, dx=rnd(g, rows, C),
                  dgamma=torch.full((C,), 3.0), dbeta=torch.full((C,), 3.0), rows=rows, C=C, ldx=C, ldy=C, y_dtype=ydt, eps=1e-6,
                  ws=scratch(2048 * 2 * C))
        cases.append((f"ln_bwd_{ydt}", "layernorm_bwd", kw, TOL_ROW))
        # parameter gradients only (no dx): the column-parallel kernel with 16 row lanes per block combined in LDS
        x = rnd(g, 300, 64)
        mean = x.mean(-1); rstd = torch.rsqrt(x.var(-1, unbiased=False) + 1e-6)
        kw = dict(x=x, dy=rnd(g, 300, 64, dtype=DT[ydt]), gamma=rnd(g, 64), mean=mean, rstd=rstd, dx=None,
                  dgamma=torch.full((64,), 3.0), dbeta=torch.full((64,), 3.0), rows=300, C=64, ldx=64, ldy=64, y_dtype=ydt, eps=1e-6,
                  ws=scratch(2048 * 2 * 64))
        cases.append((f"ln_bwd_params_only_{ydt}", "layernorm_bwd", kw, TOL_ROW))
        for rows2, C2 in ((301, 1024), (70, 2048)):          # fused single-pass kernel (C <= 1024, several row blocks) / two-kernel path
            x = rnd(g, rows2, C2)
            mean = x.mean(-1); rstd = torch.rsqrt(x.var(-1, unbiased=False) + 1e-6)
            kw = dict(x=x, dy=rnd(g, rows2, C2, dtype=DT[ydt]), gamma=rnd(g, C2), mean=mean, rstd=rstd, dx=rnd(g, rows2, C2),
                      dgamma=torch.full((C2,), 3.0), dbeta=torch.full((C2,), 3.0), rows=rows2, C=C2, ldx=C2, ldy=C2, y_dtype=ydt, eps=1e-6,
                      ws=scratch(2048 * 2 * C2))
            cases.append((f"ln_bwd_{ydt}_C{C2}", "layernorm_bwd", kw, dict(f32=2e-5, bf16=5e-3)))
            kw = dict(kw, dx=torch.full((rows2, C2), 3.0), dx_in=rnd(g, rows2, C2), dgamma=torch.zeros(C2), dbeta=torch.zeros(C2),
                      ws=scratch(2048 * 2 * C2))
            cases.append((f"ln_bwd_join_{ydt}_C{C2}", "layernorm_bwd", kw, dict(f32=2e-5, bf16=5e-3)))
    for sdt in (F32, BF16):
        rows, cols, ld = 50, 77, 80
        kw = dict(S=rnd(g, rows, ld, dtype=DT[sdt]), P=torch.full((rows, ld), 9.0, dtype=DT[sdt]), rows=rows, cols=cols, ld=ld,
                  s_dtype=sdt, p_dtype=sdt, scale=0.3)
        cases.append((f"softmax_fwd_{sdt}", "softmax_fwd", kw, TOL_ROW))
        P = torch.softmax(rnd(g, rows, ld), -1).to(DT[sdt])
        kw = dict(P=P, dP=rnd(g, rows, ld, dtype=DT[sdt]), dS=torch.zeros(rows, ld, dtype=DT[sdt]), extra=rnd(g, 2 * 3, cols + 3),
                  rows=rows, cols=cols, ld=ld, s_dtype=sdt, p_dtype=sdt, scale=0.3, rows_per_mat=25, extra_rows=3, extra_ld=cols + 3)
        cases.append((f"softmax_bwd_{sdt}", "softmax_bwd", kw, TOL_ROW))
    for odt in (F32, BF16):
        B, H, W = 2, 32, 48
        cases.append((f"patchify_{odt}", "patchify16",
                      dict(args=[rnd(g, B, 3, H, W), torch.zeros(B * 6, 768, dtype=DT[odt]), B, H, W, odt]), TOL_ROW))
    for dt in (F32, BF16):
        for (h, w, nh) in ((4, 6, 1), (4, 6, 2)):
            B, T, C = 2, 5, 128
            N = T + h * w
            ldq = (h * w + 7) // 8 * 8
            kw = dict(q=rnd(g, B * T, ldq, dtype=DT[dt]), xn=rnd(g, B * N, C, dtype=DT[dt]), rawchan=torch.full((B, T, nh * nh, C), 9.0),
                      B=B, T=T, N=N, C=C, h=h, w=w, nh=nh, nw=nh, dtype=dt, ldq=ldq, ws=scratch(64 * B * T * nh * nh * C))
            cases.append((f"chanlogit_{dt}_win{nh}", "chan_logits", kw, TOL_ROW))
            XT = rnd(g, B, N, C)
            kw = dict(x=XT[:, T:], x_ld=C, x_bs=N * C, rawlog=rnd(g, B, C // 64, T, N), rawchan=rnd(g, B, T, nh * nh, C),
                      out=torch.zeros(2 * T, B * h * w, C, dtype=DT[dt]), B=B, T=T, N=N, C=C, h=h, w=w, nh=nh, nw=nh, out_dtype=dt)
            cases.append((f"modulate_{dt}_win{nh}", "modulate", kw, TOL_ROW))
            if dt == BF16:          # ABI 8: MTT_SPLIT output (hi / lo planes for the split-plane fea_decode GEMM)
                ks = dict(kw, out=torch.zeros(2 * T, B * h * w, C, dtype=torch.bfloat16), out_lo=torch.zeros(2 * T, B * h * w, C, dtype=torch.bfloat16),
                          out_dtype=SPLIT)
                cases.append((f"modulate_split_win{nh}", "modulate", ks, dict(f32=1e-5, bf16=5e-3, split=1e-5, split_pairs=[("out", "out_lo")])))
        # several pixel splits per window (partials through the workspace, summed in split order): 16 x 16 patches
        B, T, C, h, w = 1, 3, 128, 16, 16
        N = T + h * w
        kw = dict(q=rnd(g, B * T, h * w, dtype=DT[dt]), xn=rnd(g, B * N, C, dtype=DT[dt]), rawchan=torch.full((B, T, 1, C), 9.0),
                  B=B, T=T, N=N, C=C, h=h, w=w, nh=1, nw=1, dtype=dt, ldq=h * w, ws=scratch(64 * B * T * C))
        cases.append((f"chanlogit_{dt}_splits", "chan_logits", kw, TOL_ROW))
        # windows whose width is a multiple of 8: the eight-pixels-per-lane kernel (windowed, non-square, padded query pitch, 7 tasks)
        for (B, T, h, w, nh, nw) in ((2, 6, 16, 16, 2, 2), (1, 7, 8, 24, 1, 3), (2, 6, 32, 32, 1, 1)):
            N = T + h * w
            ldq = h * w + 8
            kw = dict(q=rnd(g, B * T, ldq, dtype=DT[dt]), xn=rnd(g, B * N, C, dtype=DT[dt]), rawchan=torch.full((B, T, nh * nw, C), 9.0),
                      B=B, T=T, N=N, C=C, h=h, w=w, nh=nh, nw=nw, dtype=dt, ldq=ldq, ws=scratch(64 * B * T * nh * nw * C))
            cases.append((f"chanlogit_px8_{dt}_{h}x{w}_win{nh}x{nw}", "chan_logits", kw, TOL_ROW))
    # ABI 9: the normalised tokens as hi / lo planes (MTT_SPLIT; q fp32) — what LayerNorm wrote for the split-plane GEMMs, no fp32 copy:
    # the one-pixel kernel (6 x 4 windows) and the eight-pixel kernel (32 x 32, one window; 16 x 16 in 2 x 2 windows; padded query pitch)
    for (B, T, h, w, nh, nw, C) in ((2, 3, 6, 4, 1, 1, 128), (2, 6, 32, 32, 1, 1, 128), (1, 5, 16, 16, 2, 2, 64)):
        N = T + h * w
        ldq = h * w + 8
        x32 = rnd(g, B * N, C)
        xh = x32.to(torch.bfloat16)
        kw = dict(q=rnd(g, B * T, ldq), xn=xh, xn_lo=(x32 - xh.float()).to(torch.bfloat16), rawchan=torch.full((B, T, nh * nw, C), 9.0),
                  B=B, T=T, N=N, C=C, h=h, w=w, nh=nh, nw=nw, dtype=SPLIT, ldq=ldq, ws=scratch(64 * B * T * nh * nw * C))
        cases.append((f"chanlogit_split_{h}x{w}_win{nh}x{nw}", "chan_logits", kw, TOL_ROW))
    # backward-only kernels ((8, 8, *): window widths that are multiples of 4 take the token-grouped chan_logits_bwd kernel)
    for dt in (F32, BF16):
        for (h, w, nh) in ((4, 6, 1), (4, 6, 2), (8, 8, 1), (8, 8, 2)):
            B, T, C = 2, 5, 128
            N = T + h * w
            ldq = (h * w + 7) // 8 * 8
            XT = rnd(g, B, N, C)
            dXT = rnd(g, B, N, C)
            kw = dict(x=XT[:, T:], x_ld=C, x_bs=N * C, rawlog=rnd(g, B, C // 64, T, N), rawchan=rnd(g, B, T, nh * nh, C),
                      out=None, B=B, T=T, N=N, C=C, h=h, w=w, nh=nh, nw=nh, out_dtype=dt,
                      xargs=[rnd(g, 2 * T, B * h * w, C, dtype=DT[dt]), dXT[:, T:], torch.zeros(B, C // 64, T, N), torch.full((B, T, nh * nh, C), 9.0),
                             scratch(32 * B * T * nh * nh * C)])
            cases.append((f"modulate_bwd_{dt}_{h}x{w}_win{nh}", "modulate_bwd", kw, dict(f32=2e-5, bf16=5e-3)))
            kw = dict(q=rnd(g, B * T, ldq, dtype=DT[dt]), xn=rnd(g, B * N, C, dtype=DT[dt]), rawchan=None,
                      B=B, T=T, N=N, C=C, h=h, w=w, nh=nh, nw=nh, dtype=dt, ldq=ldq,
                      xargs=[rnd(g, B, T, nh * nh, C), torch.zeros(B * T, ldq, dtype=DT[dt]), dt, rnd(g, B * N, C)])
            cases.append((f"chanlogit_bwd_{dt}_{h}x{w}_win{nh}", "chan_logits_bwd", kw, dict(f32=2e-5, bf16=5e-3)))
        T, B, rpb, ld, C = 6, 2, 300, 56, 52
        fea = rnd(g, T, B * rpb, ld, dtype=DT[dt]); fea[..., C:] = 0
        kw = dict(fea=fea, out=None, wmix=None, T=T, B=B, rows_per_b=rpb, ld=ld, C=C, fea_dtype=dt, accumulate=0,
                  xargs=[rnd(g, T, B * rpb, ld), torch.full((B, T, T), 9.0), scratch(4096 * B * T * T)])      # dw is WRITTEN (stale 9.0 must vanish)
        cases.append((f"ctr_dw_{dt}", "ctr_dw", kw, dict(f32=2e-5, bf16=5e-3)))
        cases.append((f"rowscale_cast_vec_{dt}", "rowscale_cast",
                      dict(args=[rnd(g, 2 * 13, 24), torch.zeros(26, 32, dtype=DT[dt]), 26, 24, 24, 32, F32, dt,
                                 torch.tensor([[0.5, 2.0], [0.0, 1.5]]), 13, 3]), TOL_ROW))
        # cast + column sums of the stored values: several row blocks and 256-column panels (ragged last panel), with and without rowscale
        for (rows, cols, mb) in ((2 * 130, 24, 130), (3 * 1030, 1024, 1030), (700, 776, 0)):
            rs = torch.rand(rows // mb + 1, 2, generator=g) if mb else None
            cases.append((f"rowscale_cast_colsum_{dt}_{rows}x{cols}", "rowscale_cast_colsum",
                          dict(args=[rnd(g, rows, cols + 8), torch.zeros(rows, cols + 16, dtype=DT[dt]), rows, cols, cols + 8, cols + 16, F32, dt,
                                     rs, mb, 3, torch.full((cols + 2,), 9.0), scratch(1024 * (cols + 8))]), dict(TOL_ROW, f32=2e-4)))   # sums of rounded values: see TOL_CS
        cases.append((f"rowscale_cast_{dt}", "rowscale_cast",
                      dict(args=[rnd(g, 2 * 13, 24), torch.zeros(26, 32, dtype=DT[dt]), 26, 20, 24, 32, F32, dt,
                                 torch.tensor([[0.5, 2.0], [0.0, 1.5]]), 13, 3]), TOL_ROW))
    for dt in (F32, BF16):
        for accum in (0, 1):
            T, B, rpb, ld, C = 6, 2, 24, 56, 52
            fea = rnd(g, T, B * rpb, ld, dtype=DT[dt]); fea[..., C:] = 0
            kw = dict(fea=fea, out=rnd(g, T, B * rpb, ld), wmix=rnd(g, B, T, T), T=T, B=B, rows_per_b=rpb, ld=ld, C=C,
                      fea_dtype=dt, accumulate=accum)
            cases.append((f"ctr_mix_{dt}_acc{accum}", "ctr_mix", kw, TOL_ROW))
        kw = dict(fea=rnd(g, T, B * rpb, ld), out=torch.full((T, B * rpb, ld), 3.0, dtype=torch.bfloat16), wmix=rnd(g, B, T, T), T=T, B=B,
                  rows_per_b=rpb, ld=ld, C=C, fea_dtype=F32, accumulate=0, out_dtype=BF16)
        cases.append((f"ctr_mix_bf16out_{dt}", "ctr_mix", kw, TOL_ROW))
    for dt in (F32, BF16):
        for (Hi, Wi, Ho, Wo) in ((4, 6, 16, 24), (8, 8, 4, 4), (5, 7, 11, 9)):
            B, C, ld = 2, 20, 24
            x = rnd(g, B * Hi * Wi, ld, dtype=DT[dt])
            kw = {"in": x, "out": torch.zeros(B * Ho * Wo, ld, dtype=DT[dt]), "B": B, "C": C, "Hin": Hi, "Win": Wi, "Hout": Ho, "Wout": Wo,
                  "ld_in": ld, "ld_out": ld, "in_dtype": dt, "out_dtype": dt, "out_nchw": 0, "accumulate": 0}
            cases.append((f"bilinear_nhwc_{dt}_{Hi}x{Wi}to{Ho}x{Wo}", "bilinear_fwd", kw, TOL_ROW))
            kw = {"in": x, "out": torch.zeros(B, C, Ho, Wo), "B": B, "C": C, "Hin": Hi, "Win": Wi, "Hout": Ho, "Wout": Wo,
                  "ld_in": ld, "ld_out": 0, "in_dtype": dt, "out_dtype": F32, "out_nchw": 1, "accumulate": 0}
            cases.append((f"bilinear_nchw_{dt}_{Hi}x{Wi}to{Ho}x{Wo}", "bilinear_fwd", kw, TOL_ROW))
            kw = {"in": rnd(g, B * Ho * Wo, ld, dtype=DT[dt]), "out": rnd(g, B * Hi * Wi, ld), "B": B, "C": C, "Hin": Hi, "Win": Wi,
                  "Hout": Ho, "Wout": Wo, "ld_in": ld, "ld_out": ld, "in_dtype": dt, "out_dtype": F32, "out_nchw": 0, "accumulate": 1}
            cases.append((f"bilinear_bwd_{dt}_{Hi}x{Wi}to{Ho}x{Wo}", "bilinear_bwd", kw, TOL_ROW))
        kw = {"in": rnd(g, 2, 5, 16, 24), "out": torch.zeros(2 * 4 * 6, 8), "B": 2, "C": 5, "Hin": 4, "Win": 6, "Hout": 16, "Wout": 24,
              "ld_in": 8, "ld_out": 0, "in_dtype": F32, "out_dtype": F32, "out_nchw": 1, "accumulate": 1}
        cases.append((f"bilinear_bwd_nchw_{dt}", "bilinear_bwd", kw, TOL_ROW))
        # integer scales 4 / 2 of the NCHW forms (the head predictions -> image size): specialised kernels; C not a multiple of 4, 2-pixel maps
        for (Hi, Wi, sc, C) in ((8, 12, 4, 21), (6, 10, 2, 7), (2, 2, 4, 1), (16, 8, 4, 3)):
            ld = (C + 7) // 8 * 8
            kw = {"in": rnd(g, 2 * Hi * Wi, ld, dtype=DT[dt]), "out": torch.zeros(2, C, sc * Hi, sc * Wi), "B": 2, "C": C, "Hin": Hi, "Win": Wi,
                  "Hout": sc * Hi, "Wout": sc * Wi, "ld_in": ld, "ld_out": 0, "in_dtype": dt, "out_dtype": F32, "out_nchw": 1, "accumulate": 0}
            cases.append((f"bilinear_nchw_x{sc}_{dt}_{Hi}x{Wi}_c{C}", "bilinear_fwd", kw, TOL_ROW))
            kw = {"in": rnd(g, 2, C, sc * Hi, sc * Wi), "out": rnd(g, 2 * Hi * Wi, ld), "B": 2, "C": C, "Hin": Hi, "Win": Wi, "Hout": sc * Hi,
                  "Wout": sc * Wi, "ld_in": ld, "ld_out": 0, "in_dtype": F32, "out_dtype": F32, "out_nchw": 1, "accumulate": 1}
            cases.append((f"bilinear_bwd_nchw_x{sc}_{dt}_{Hi}x{Wi}_c{C}", "bilinear_bwd", kw, TOL_ROW))
        # integer scales of the NHWC form (InvPT stage resizes, the x4 / x2 feature upsamples), ragged 8-channel groups, 2 x 2 maps
        for (Hi, Wi, sc, C) in ((6, 10, 2, 20), (2, 2, 4, 9), (3, 5, 2, 8), (7, 3, 4, 33)):
            ld = (C + 7) // 8 * 8
            kw = {"in": rnd(g, 2 * sc * Hi * sc * Wi, ld, dtype=DT[dt]), "out": rnd(g, 2 * Hi * Wi, ld), "B": 2, "C": C, "Hin": Hi, "Win": Wi,
                  "Hout": sc * Hi, "Wout": sc * Wi, "ld_in": ld, "ld_out": ld, "in_dtype": dt, "out_dtype": F32, "out_nchw": 0, "accumulate": 1}
            cases.append((f"bilinear_bwd_nhwc_x{sc}_{dt}_{Hi}x{Wi}_c{C}", "bilinear_bwd", kw, TOL_ROW))
    for dt in (F32, BF16):
        rows, C, ld = 300, 52, 56
        x = rnd(g, rows, ld, dtype=DT[dt]); x[:, C:] = 0
        ws = scratch(1 << 16)
        kw = dict(x=x, mean_out=torch.zeros(C), m2_out=torch.zeros(C), rows=rows, C=C, ld=ld, dtype=dt, Z=1, x_zs=0, p_zs=C, xargs=[ws])
        cases.append((f"bn_stats_{dt}", "bn_stats", kw, TOL_ROW))
        mean, rstd, gam, bet = rnd(g, C) * 0.1, rnd(g, C).abs() + 0.5, rnd(g, C), rnd(g, C)
        for act in (0, 1, 2):
            kw = dict(x=x, y=torch.full((rows, ld), 5.0, dtype=DT[dt]), mean=mean, rstd=rstd, gamma=gam, beta=bet,
                      rows=rows, C=C, ld=ld, dtype=dt, act=act, Z=1, x_zs=0, p_zs=C)
            cases.append((f"bn_apply_{dt}_act{act}", "bn_apply", kw, TOL_ROW))
        dy = rnd(g, rows, ld, dtype=DT[dt])
        kw = dict(x=x, dy=dy, mean=mean, rstd=rstd, gamma=gam, beta=bet, dsum=torch.zeros(C), dsumxh=torch.zeros(C),
                  rows=rows, C=C, ld=ld, dtype=dt, act=1, Z=1, x_zs=0, p_zs=C, xargs=[ws])
        cases.append((f"bn_bwd_reduce_{dt}", "bn_bwd_reduce", kw, TOL_ROW))
        kw = dict(x=x, dy=dy, dx=torch.full((rows, ld), 5.0, dtype=DT[dt]), mean=mean, rstd=rstd, gamma=gam, beta=bet,
                  dsum=rnd(g, C), dsumxh=rnd(g, C), rows=rows, C=C, ld=ld, dtype=dt, act=1, Z=1, x_zs=0, p_zs=C)
        cases.append((f"bn_bwd_apply_{dt}", "bn_bwd_apply", kw, TOL_ROW))
        # Z-batched stack (3 maps, 5000 rows -> several row blocks), channel means far from zero: the centred statistics must
        # keep full accuracy where E[x^2] - E[x]^2 cancels (|mean| = 200 sigma)
        Zs, rows2 = 3, 5000
        xs = rnd(g, Zs, rows2, ld) * 0.05 + 10.0 * torch.arange(1, C + 1 + (ld - C))[None, None, :].float() / C
        xs[..., C:] = 0
        xs = xs.to(DT[dt])
        kw = dict(x=xs, mean_out=torch.zeros(Zs, C), m2_out=torch.zeros(Zs, C), rows=rows2, C=C, ld=ld, dtype=dt, Z=Zs, x_zs=rows2 * ld, p_zs=C,
                  xargs=[scratch(1 << 18)])
        cases.append((f"bn_stats_stack_offset_{dt}", "bn_stats", kw, TOL_ROW))
        means, rstds, gams, bets = rnd(g, Zs, C) * 0.1 + xs.float()[:, :, :C].mean(1), rnd(g, Zs, C).abs() + 0.5, rnd(g, Zs, C), rnd(g, Zs, C)
        dys = rnd(g, Zs, rows2, ld, dtype=DT[dt])
        kw = dict(x=xs, y=torch.full((Zs, rows2, ld), 5.0, dtype=DT[dt]), mean=means, rstd=rstds, gamma=gams, beta=bets,
                  rows=rows2, C=C, ld=ld, dtype=dt, act=1, Z=Zs, x_zs=rows2 * ld, p_zs=C)
        cases.append((f"bn_apply_stack_{dt}", "bn_apply", kw, TOL_ROW))
        kw = dict(x=xs, dy=dys, mean=means, rstd=rstds, gamma=gams, beta=bets, dsum=torch.zeros(Zs, C), dsumxh=torch.zeros(Zs, C),
                  rows=rows2, C=C, ld=ld, dtype=dt, act=2, Z=Zs, x_zs=rows2 * ld, p_zs=C, xargs=[scratch(1 << 18)])
        cases.append((f"bn_bwd_reduce_stack_{dt}", "bn_bwd_reduce", kw, TOL_ROW))
        kw = dict(x=xs, dy=dys, dx=torch.full((Zs, rows2, ld), 5.0, dtype=DT[dt]), mean=means, rstd=rstds, gamma=gams, beta=bets,
                  dsum=rnd(g, Zs, C), dsumxh=rnd(g, Zs, C), rows=rows2, C=C, ld=ld, dtype=dt, act=2, Z=Zs, x_zs=rows2 * ld, p_zs=C)
        cases.append((f"bn_bwd_apply_stack_{dt}", "bn_bwd_apply", kw, TOL_ROW))
        if dt == F32:
            # ABI 9 (mtt_bn_desc.g_dtype): x fp32, dy / dx bf16 with the same pitch and map stride in elements — the gradient maps of a
            # bf16-arithmetic backward next to the fp32-stored conv output of the x3f forward (ConvHeadFn)
            dyb = dys.to(torch.bfloat16)
            kw = dict(x=xs, dy=dyb, mean=means, rstd=rstds, gamma=gams, beta=bets, dsum=torch.zeros(Zs, C), dsumxh=torch.zeros(Zs, C),
                      rows=rows2, C=C, ld=ld, dtype=F32, g_dtype=BF16 + 1, act=1, Z=Zs, x_zs=rows2 * ld, p_zs=C, xargs=[scratch(1 << 18)])
            cases.append(("bn_bwd_reduce_stack_x32_g16", "bn_bwd_reduce", kw, TOL_ROW))
            kw = dict(x=xs, dy=dyb, dx=torch.full((Zs, rows2, ld), 5.0, dtype=torch.bfloat16), mean=means, rstd=rstds, gamma=gams, beta=bets,
                      dsum=rnd(g, Zs, C), dsumxh=rnd(g, Zs, C), rows=rows2, C=C, ld=ld, dtype=F32, g_dtype=BF16 + 1, act=1, Z=Zs, x_zs=rows2 * ld, p_zs=C)
            cases.append(("bn_bwd_apply_stack_x32_g16", "bn_bwd_apply", kw, TOL_ROW))
        cases.append((f"cast2d_{dt}", "cast2d", dict(args=[rnd(g, 30, 20), torch.full((30, 24), 4.0, dtype=DT[dt]), 30, 18, 20, 24, F32, dt, 1]), TOL_ROW))
        cases.append((f"colsum_{dt}", "colsum", dict(args=[rnd(g, 300, 56, dtype=DT[dt]), torch.full((52,), 3.0), 300, 52, 56, dt, scratch(300 * 56)]), TOL_ROW))
        for (r_, c_, ld_) in ((5000, 4104, 4112), (1031, 350, 352), (70, 1024, 1024)):     # two column chunks + ragged tail / row lanes / few rows
            cases.append((f"colsum_{dt}_{r_}x{c_}", "colsum",
                          dict(args=[rnd(g, r_, ld_, dtype=DT[dt]), torch.full((c_,), 3.0), r_, c_, ld_, dt, scratch(1536 * ld_)]), TOL_ROW))
        # Z maps per launch pair (task-batched bias gradients): plain stack, and the second half of a 'catpair' row (column offset)
        Zb, r_, c_, ld_ = 5, 700, 300, 2 * 304
        src = rnd(g, Zb, r_, ld_, dtype=DT[dt])
        cases.append((f"colsum_batched_{dt}", "colsum_batched",
                      dict(args=[src, torch.full((Zb, c_), 3.0), r_, c_, ld_, dt, Zb, r_ * ld_, c_, scratch(Zb * 1536 * ld_)]), TOL_ROW))
        cases.append((f"colsum_batched_half_{dt}", "colsum_batched",
                      dict(args=[src.reshape(-1)[304:], torch.full((Zb, c_), 3.0), r_, c_, ld_, dt, Zb, r_ * ld_, c_, scratch(Zb * 1536 * ld_)]), TOL_ROW))
        cases.append((f"add_rows_{dt}", "add_rows", dict(args=[rnd(g, 30, 24, dtype=DT[dt]), rnd(g, 30, 32), 30, 20, 24, 32, dt, 0.5]), TOL_ROW))
    # cross-task reweighting weights (ABI 10): the per-task MLP over the head dimension of the prompt<->prompt raw logits, forward and backward
    # (drawlog: only the first T columns are written — the 7.0 fill of the other columns must survive; weight gradients WRITTEN over stale 9.0)
    for (B, T, nH, N) in ((63, 6, 16, 1030), (3, 2, 2, 10), (5, 4, 12, 260)):
        raw = rnd(g, B, nH, T, N, scale=3.0)
        kw = dict(rawlog=raw, w0=rnd(g, T, nH * nH, scale=0.4), b0=rnd(g, T, nH), w2=rnd(g, T, nH, scale=0.5), b2=rnd(g, T, 1),
                  wmix=torch.full((B, T, T), 9.0), B=B, T=T, nH=nH, N=N)
        cases.append((f"ctr_weights_{B}x{T}x{nH}", "ctr_weights", kw, dict(f32=2e-6, bf16=5e-3)))
        kb = dict(kw, wmix=None, xargs=[rnd(g, B, T, T), torch.full((B, nH, T, N), 7.0), torch.full((T, nH * nH), 9.0), torch.full((T, nH), 9.0),
                                        torch.full((T, nH), 9.0), torch.full((T, 1), 9.0)])
        cases.append((f"ctr_weights_bwd_{B}x{T}x{nH}", "ctr_weights_bwd", kb, dict(f32=5e-6, bf16=5e-3)))
    # detection-branch losses (ABI 10): element-wise map + deterministic sum, and the gradient with a device scalar / an element-wise upstream
    for (kind, N, C, wmode) in ((0, 349, 10, 0), (0, 5000, 3, 1), (0, 777, 10, 2), (1, 349, 7, 0), (1, 4100, 2, 2), (1, 600, 9, 1)):
        pred = rnd(g, N, C, scale=2.5)
        target = torch.randint(0, C + 1, (N,), generator=g) if kind == 0 else pred + rnd(g, N, C, scale=0.3)
        weight = None if wmode == 0 else (torch.rand(N, generator=g) + 0.1 if wmode == 1 else torch.rand(N, C, generator=g) + 0.1)
        base = dict(pred=pred, target=target, weight=weight, N=N, C=C, kind=kind, wmode=wmode, gamma=2.0 if N % 2 else 1.5, alpha=0.25, beta=1.0 / 9.0)
        cases.append((f"detloss_fwd_k{kind}_w{wmode}_{N}", "detloss_fwd",
                      dict(base, out=torch.full((N, C), 9.0), sum=torch.full((1,), 9.0), ws=scratch(4096)), dict(f32=3e-6, bf16=5e-3)))
        cases.append((f"detloss_bwd_scalar_k{kind}_w{wmode}_{N}", "detloss_bwd",
                      dict(base, out=None, sum=None, ws=None, xargs=[torch.tensor([1.7]), None, 0.37, torch.full((N, C), 9.0)]), dict(f32=5e-6, bf16=5e-3)))
        cases.append((f"detloss_bwd_elem_k{kind}_w{wmode}_{N}", "detloss_bwd",
                      dict(base, out=None, sum=None, ws=None, xargs=[None, rnd(g, N, C), 2.0, torch.full((N, C), 9.0)]), dict(f32=5e-6, bf16=5e-3)))
    return cases


def invpt_cases():
    cases = []
    g = torch.Generator().manual_seed(14)
    for dt in (F32, BF16):
        Z, B, H, W, ld = 2, 2, 6, 4, 24
        kw = dict(x=rnd(g, Z, B * H * W, ld, dtype=DT[dt]), w=rnd(g, Z, 9, ld), y=torch.zeros(Z, B * 3 * 2, ld, dtype=DT[dt]),
                  scale=rnd(g, Z, ld), shift=rnd(g, Z, ld), Z=Z, B=B, H=H, W=W, ld=ld, dtype=dt)
        cases.append((f"dwconv_{dt}", "dwconv3x3s2", kw, TOL_ROW))
        kw = dict(kw, scale=None, shift=None, H=5, W=3, x=rnd(g, Z, B * 15, ld, dtype=DT[dt]), y=torch.zeros(Z, B * 3 * 2, ld, dtype=DT[dt]))
        cases.append((f"dwconv_odd_{dt}", "dwconv3x3s2", kw, TOL_ROW))
        for (H, W, k) in ((4, 6, 2), (5, 7, 4), (8, 8, 8)):
            Ho, Wo = -(-H // k), -(-W // k)
            kw = dict(x=rnd(g, 2 * H * W, 16, dtype=DT[dt]), y=torch.zeros(2 * Ho * Wo, 16, dtype=DT[dt]), B=2, H=H, W=W, k=k, ld=16, dtype=dt)
            cases.append((f"avgpool_{dt}_{H}x{W}k{k}", "avgpool_ceil", kw, TOL_ROW))
        T, rows, D = 3, 37, 20
        kw = dict(x=rnd(g, T, rows, 24), y=torch.full((T, rows, 24), 3.0, dtype=DT[dt]), gamma=rnd(g, T * D), beta=rnd(g, T * D),
                  rows=rows, T=T, D=D, ldx=24, ldy=24, y_dtype=dt, eps=1e-5)
        cases.append((f"ln_mt_{dt}", "layernorm_mt", kw, TOL_ROW))
        B, H, W, Cop = 2, 3, 4, 16
        kw = dict(yall=rnd(g, B * H * W, 9 * Cop, dtype=DT[dt]), out=torch.zeros(B * 4 * H * W, Cop, dtype=DT[dt]), bias=rnd(g, Cop),
                  B=B, H=H, W=W, Cop=Cop, dtype=dt, out_dtype=dt)
        cases.append((f"convt_gather_{dt}", "convt3x3s2_gather", kw, TOL_ROW))
    for dt in (F32, BF16):
        Z, B, H, W, ld = 2, 2, 6, 4, 24
        kw = dict(x=rnd(g, Z, B * H * W, ld, dtype=DT[dt]), w=rnd(g, Z, 9, ld), y=None, scale=None, shift=None, Z=Z, B=B, H=H, W=W, ld=ld, dtype=dt,
                  xargs=[rnd(g, Z, B * 3 * 2, ld, dtype=DT[dt]), torch.zeros(Z, B * H * W, ld, dtype=DT[dt]), torch.zeros(Z, 9, ld)])
        cases.append((f"dwconv_bwd_{dt}", "dwconv3x3s2_bwd", kw, dict(f32=2e-5, bf16=6e-3)))
        Z, B, H, W, ld = 3, 3, 9, 7, 136                     # odd map, more than one 64-channel group with a ragged last one, > 32 pixels per lane set
        kw = dict(x=rnd(g, Z, B * H * W, ld, dtype=DT[dt]), w=rnd(g, Z, 9, ld), y=None, scale=None, shift=None, Z=Z, B=B, H=H, W=W, ld=ld, dtype=dt,
                  xargs=[rnd(g, Z, B * 5 * 4, ld, dtype=DT[dt]), torch.zeros(Z, B * H * W, ld, dtype=DT[dt]), torch.zeros(Z, 9, ld)])
        cases.append((f"dwconv_bwd_ragged_{dt}", "dwconv3x3s2_bwd", kw, dict(f32=2e-5, bf16=6e-3)))
        kw = dict(x=None, y=None, B=2, H=5, W=7, k=4, ld=16, dtype=dt, xargs=[rnd(g, 2 * 2 * 2, 16, dtype=DT[dt]), torch.zeros(2 * 35, 16, dtype=DT[dt])])
        cases.append((f"avgpool_bwd_{dt}", "avgpool_ceil_bwd", kw, TOL_ROW))
        kw = dict(yall=None, out=None, bias=None, B=2, H=3, W=4, Cop=16, dtype=dt, out_dtype=dt,
                  xargs=[rnd(g, 2 * 4 * 12, 16, dtype=DT[dt]), torch.zeros(2 * 12, 9 * 16, dtype=DT[dt])])
        cases.append((f"convt_gather_bwd_{dt}", "convt3x3s2_gather_bwd", kw, TOL_ROW))
    B, heads, T, qh, qw, K = 2, 2, 3, 4, 2, 9
    Q = T * qh * qw
    kw = dict(cur=rnd(g, B, heads, Q, 16), prev=rnd(g, B, heads, Q // 4, 16), out=torch.zeros(B, heads, Q, 16), w=rnd(g, heads, 2 * heads),
              bias=rnd(g, heads), B=B, heads=heads, T=T, qh=qh, qw=qw, K=K, ldk=16, ldkp=16)
    cases.append(("attn_msg", "attn_msg", kw, TOL_ROW))
    for (B, heads, T, qh, qw, K) in ((2, 2, 3, 4, 2, 9), (1, 2, 6, 8, 8, 96)):
        Q = T * qh * qw
        Kp = (K + 7) // 8 * 8
        kw = dict(cur=rnd(g, B, heads, Q, Kp), prev=rnd(g, B, heads, Q // 4, Kp), out=None, w=rnd(g, heads, 2 * heads), bias=None,
                  B=B, heads=heads, T=T, qh=qh, qw=qw, K=K, ldk=Kp, ldkp=Kp,
                  xargs=[rnd(g, B, heads, Q, Kp), torch.zeros(B, heads, Q, Kp), torch.zeros(B, heads, Q, Kp), torch.full((heads, 2 * heads), 9.0),
                         torch.full((heads,), 9.0), scratch(2049 * 36)])
        cases.append((f"attn_msg_bwd_K{K}", "attn_msg_bwd", kw, dict(f32=2e-5, bf16=5e-3)))
    return cases


def upconv_cases():
    """mtt_upconv4_expand / _gather (upsample x4 + 3x3 conv, taps first): borders (1-pixel maps, single rows / columns), several column
    blocks per row, more than 8 images (every XCD slot of the block numbering + the ragged last group), bias / scale / activation."""
    cases = []
    g = torch.Generator().manual_seed(29)
    shapes = ((2, 2, 3, 5, 11), (1, 1, 1, 1, 8), (1, 3, 1, 4, 20), (1, 2, 5, 1, 12), (3, 4, 4, 6, 350), (1, 1, 32, 32, 350))
    for dt in (F32, BF16):
        for (Z, B, h, w, C) in shapes:
            Cp = (C + 7) // 8 * 8
            z = torch.zeros(Z, B * h * w, 9, Cp)
            z[..., :C] = rnd(g, Z, B * h * w, 9, C)
            for act, has_sc in ((0, False), (1, True)):
                kw = dict(z=z.to(DT[dt]), y=torch.full((Z, B * 16 * h * w, Cp), 7.0, dtype=DT[dt]), bias=rnd(g, Z, C),
                          colscale=rnd(g, Z, C) if has_sc else None, Z=Z, B=B, h=h, w=w, C=C, Cp=Cp, z_dtype=dt, y_dtype=dt, act=act)
                cases.append((f"upconv4_expand_{dt}_{Z}x{B}x{h}x{w}x{C}_act{act}", "upconv4_expand", kw, TOL_ROW))
            dy = torch.zeros(Z, B * 16 * h * w, Cp)
            dy[..., :C] = rnd(g, Z, B * 16 * h * w, C)
            kw = dict(z=torch.full((Z, B * h * w, 9 * Cp), 7.0, dtype=DT[dt]), y=dy.to(DT[dt]), bias=None, colscale=None,
                      Z=Z, B=B, h=h, w=w, C=C, Cp=Cp, z_dtype=dt, y_dtype=dt, act=0)
            cases.append((f"upconv4_gather_{dt}_{Z}x{B}x{h}x{w}x{C}", "upconv4_gather", kw, TOL_ROW))
    return cases


def swin_cases():
    """TaskPrompter-Swin forward entry points: patchify (P = 2 / 4 / 8, padded K), NCHW resize (down / up), row gather (shifted + padded
    window partition with prompts, its inverse, column-offset destination as in patch merging, dtype conversion), window attention
    (bf16 MFMA kernel at 2 / 4 / 6 / 10 key tiles and the exact fp32 kernel; with / without shift mask; padding pixels; 2 and 3 prompts;
    several windows per image and several images), channel attention (1 and 4 windows, kv bias, bf16 / fp32 kv), the stride-2 map conv
    and the modulation with 32-channel heads."""
    import importlib
    sw = importlib.import_module("multi-task-transformer_amd.taskprompter_swin")
    cases = []
    g = torch.Generator().manual_seed(31)
    for P, (B, H, W) in ((4, (2, 16, 24)), (2, (1, 8, 6)), (8, (1, 16, 16))):
        Kp = (3 * P * P + 7) // 8 * 8
        for dt in (F32, BF16):
            cases.append((f"patchify_P{P}_{dt}", "patchify", dict(args=[rnd(g, B, 3, H, W), torch.full((B * (H // P) * (W // P), Kp), 7.0, dtype=DT[dt]),
                                                                        B, H, W, P, Kp, dt]), TOL_ROW))
    for (pl, Hin, Win, Hout, Wout) in ((6, 16, 24, 12, 18), (3, 9, 7, 18, 14), (2, 32, 64, 24, 48)):
        cases.append((f"resize_nchw_{Hin}x{Win}_to_{Hout}x{Wout}", "resize_nchw",
                      dict(args=[rnd(g, pl, Hin, Win), torch.full((pl, Hout, Wout), 7.0), pl, Hin, Win, Hout, Wout]), TOL_ROW))
    # gather: the real window tables (shift 2 of window 5 on a 7 x 9 map -> padded to 10 x 10), 3 prompts
    for (res, window, shifted, T, C, B) in (((7, 9), 5, True, 3, 40, 2), ((8, 12), 4, True, 2, 64, 3), ((8, 12), 4, False, 2, 16, 1)):
        ws, shift, Hp, Wp = sw.block_geometry(res, window, shifted)
        part, pix, rev = sw.window_tables(res, ws, shift, Hp, Wp, T, "cpu")
        N, Nw, nW = T + res[0] * res[1], T + ws * ws, (Hp // ws) * (Wp // ws)
        for sdt, ddt in ((F32, BF16), (BF16, BF16), (F32, F32)):
            src = rnd(g, B, N, C, dtype=DT[sdt])
            kw = dict(src=src, dst=torch.full((B, nW * Nw, C), 7.0, dtype=DT[ddt]), idx=part, rows=nW * Nw, C=C, ld_src=C, ld_dst=C,
                      src_dtype=sdt, dst_dtype=ddt, B=B, src_bs=N * C, dst_bs=nW * Nw * C, idx_bs=0)
            cases.append((f"gather_partition_{res[0]}x{res[1]}_w{window}_s{shift}_{sdt}{ddt}", "gather_rows", kw, TOL_ROW))
        win = rnd(g, B, nW * Nw, C, dtype=torch.bfloat16)
        dst = torch.full((B, N, C + 8), 7.0)
        kw = dict(src=win, dst=dst[:, T:, 8:], idx=rev, rows=res[0] * res[1], C=C, ld_src=C, ld_dst=C + 8, src_dtype=BF16, dst_dtype=F32,
                  B=B, src_bs=nW * Nw * C, dst_bs=N * (C + 8), idx_bs=0)
        cases.append((f"gather_reverse_{res[0]}x{res[1]}_w{window}_s{shift}", "gather_rows", kw, TOL_ROW))
    # pixel shuffle of the ConvTranspose2d GEMM (ABI 12): odd Co (unaligned (dy, dx) blocks), wide output pitch, both dtypes
    for (B_, H_, W_, Co_, ldz_, ldo_, zdt, odt) in ((2, 3, 5, 18, 72, 24, F32, F32), (1, 4, 4, 225 // 9, 104, 32, F32, BF16), (2, 2, 3, 7, 32, 8, BF16, F32)):
        kw = dict(args=[rnd(g, B_ * H_ * W_, ldz_, dtype=DT[zdt]), torch.full((B_ * 4 * H_ * W_, ldo_), 7.0, dtype=DT[odt]), B_, H_, W_, Co_, ldz_, ldo_, zdt, odt])
        cases.append((f"pixshuf2_{B_}x{H_}x{W_}_c{Co_}_{zdt}{odt}", "pixshuf2", kw, TOL_ROW))
    # window attention
    for (res, window, shifted, T, nH, B, dts) in (((8, 12), 4, True, 2, 2, 2, (BF16, F32)), ((7, 9), 5, True, 3, 1, 2, (BF16, F32)),
                                                  ((7, 14), 7, False, 2, 3, 1, (BF16, F32)), ((9, 9), 9, False, 3, 2, 2, (BF16,)),
                                                  ((24, 24), 12, True, 2, 4, 1, (BF16, F32)), ((12, 24), 12, False, 3, 2, 2, (BF16,))):
        ws, shift, Hp, Wp = sw.block_geometry(res, window, shifted)
        part, pix, rev = sw.window_tables(res, ws, shift, Hp, Wp, T, "cpu")
        mask = sw.shift_attn_mask(Hp, Wp, ws, shift)
        ws2, nW = ws * ws, (Hp // ws) * (Wp // ws)
        N, Nw, Cc = T + res[0] * res[1], T + ws * ws, nH * 32
        for dt in dts:
            qkv = rnd(g, B * nW, Nw, 3 * Cc, scale=0.7).to(DT[dt])
            kw = dict(qkv=qkv, out=torch.full((B * nW, Nw, Cc), 7.0, dtype=DT[dt]), rawmap=torch.full((B, nH, T, N), 3.0),
                      bias=rnd(g, nH, ws2, ws2, scale=0.5), mask=mask, pix=pix, nwin=B * nW, nW=nW, nH=nH, T=T, ws2=ws2, dtype=dt,
                      scale=32 ** -0.5, map_ld=N, map_off=T)
            cases.append((f"winattn_{res[0]}x{res[1]}_w{window}_s{shift}_T{T}_h{nH}_{dt}", "winattn_fwd", kw, dict(f32=2e-5 if dt == F32 else 6e-3, bf16=1.5e-2)))
            if dt == F32:     # the matrix-core form on fp32 storage (ABI 11): 3 bf16 MFMAs per product on hi / lo split operands — fp32-class
                kw = dict(kw, out=torch.full((B * nW, Nw, Cc), 7.0), rawmap=torch.full((B, nH, T, N), 3.0), mfma=1)
                cases.append((f"winattn_x3mfma_{res[0]}x{res[1]}_w{window}_s{shift}_T{T}_h{nH}", "winattn_fwd", kw, dict(f32=5e-5, bf16=1.5e-2)))
    # window attention backward (exact VALU kernel for both dtypes): dqkv and the window-part dS; with / without mask and raw-logit gradient
    for (res, window, shifted, T, nH, B, dt, with_raw) in (((8, 12), 4, True, 2, 2, 2, F32, True), ((7, 9), 5, True, 3, 1, 2, F32, True),
                                                           ((7, 9), 5, False, 3, 2, 1, BF16, True), ((12, 24), 12, True, 2, 2, 1, F32, False),
                                                           ((12, 12), 12, False, 3, 1, 2, BF16, True), ((24, 24), 12, True, 2, 4, 1, BF16, True),
                                                           ((24, 12), 12, True, 2, 2, 2, F32, True), ((7, 14), 7, False, 2, 3, 1, BF16, False)):
        ws, shift, Hp, Wp = sw.block_geometry(res, window, shifted)
        part, pix, rev = sw.window_tables(res, ws, shift, Hp, Wp, T, "cpu")
        mask = sw.shift_attn_mask(Hp, Wp, ws, shift)
        ws2, nW = ws * ws, (Hp // ws) * (Wp // ws)
        N, Nw, Cc = T + res[0] * res[1], T + ws * ws, nH * 32
        qkv = rnd(g, B * nW, Nw, 3 * Cc, scale=0.7).to(DT[dt])
        fkw = dict(qkv=qkv, out=torch.zeros(B * nW, Nw, Cc, dtype=DT[dt]), rawmap=None, bias=rnd(g, nH, ws2, ws2, scale=0.5), mask=mask, pix=pix,
                   nwin=B * nW, nW=nW, nH=nH, T=T, ws2=ws2, dtype=dt, scale=32 ** -0.5, map_ld=N, map_off=T)
        abi_emul.call("winattn_fwd", **fkw)                      # the forward output the backward reads (computed by the emulator)
        kw = dict(fkw, xargs=[rnd(g, B * nW, Nw, Cc).to(DT[dt]), rnd(g, B, nH, T, N) if with_raw else None,
                              torch.full((B * nW, Nw, 3 * Cc), 7.0, dtype=DT[dt]), torch.full((B * nW, nH, ws2, ws2), 7.0)])
        cases.append((f"winattn_bwd_{res[0]}x{res[1]}_w{window}_s{shift}_T{T}_h{nH}_{dt}", "winattn_bwd", kw, dict(f32=3e-5 if dt == F32 else 8e-3, bf16=1.5e-2)))
        if dt == F32:         # the bf16 matrix-core backward on fp32 storage (the x3f mode's backward): operands rounded to bf16 while loaded
            kw = dict(kw, mfma=1, xargs=[kw["xargs"][0], kw["xargs"][1], torch.full((B * nW, Nw, 3 * Cc), 7.0), torch.full((B * nW, nH, ws2, ws2), 7.0)])
            cases.append((f"winattn_bwd_mfma_{res[0]}x{res[1]}_w{window}_s{shift}_T{T}_h{nH}", "winattn_bwd", kw, dict(f32=1.5e-2, bf16=1.5e-2)))
            kw = dict(kw, biasT=kw["bias"].transpose(1, 2).contiguous(),        # ABI 13: the transposed bias table (16-byte loads in the key-owner pass)
                      xargs=[kw["xargs"][0], kw["xargs"][1], torch.full((B * nW, Nw, 3 * Cc), 7.0), torch.full((B * nW, nH, ws2, ws2), 7.0)])
            cases.append((f"winattn_bwd_mfma_biasT_{res[0]}x{res[1]}_w{window}_s{shift}_T{T}_h{nH}", "winattn_bwd", kw, dict(f32=1.5e-2, bf16=1.5e-2)))
    # channel attention
    for (B, T, C, ce, nwin, kdt, has_b) in ((2, 2, 64, 16, 1, F32, True), (1, 3, 136, 64, 2, F32, True), (2, 2, 256, 256, 1, BF16, False), (1, 2, 1024, 256, 2, F32, True)):
        Cp = (C + 7) // 8 * 8
        kw = dict(q=rnd(g, B, T, ce, scale=0.3), kvT=rnd(g, B, 2 * ce, Cp, scale=0.5).to(DT[kdt]), rawchan=torch.full((B, T, nwin * nwin, C), 7.0),
                  cx=torch.full((B, T, ce), 7.0), B=B, T=T, C=C, ce=ce, nh=nwin, nw=nwin, kv_dtype=kdt, ldk=Cp, scale=ce ** -0.5,
                  kvbias=rnd(g, 2 * ce, scale=0.2) if has_b else None)
        cases.append((f"chanattn_B{B}T{T}C{C}ce{ce}w{nwin}_{kdt}", "chanattn_fwd", kw, TOL_ROW))
    # 3x3 stride-2 conv of the attention maps, in the [B, nH*T, T + hw] layout of the raw prompt logits
    for (B, Ci, H, W, T) in ((2, 4, 8, 12, 2), (1, 24, 6, 10, 3)):
        N, N2 = T + H * W, T + (H // 2) * (W // 2)
        kw = dict(x=rnd(g, B, Ci, N), w=rnd(g, Ci, Ci, 3, 3, scale=0.3), bias=rnd(g, Ci), y=torch.full((B, Ci, N2), 7.0), B=B, Ci=Ci, Co=Ci, H=H, W=W,
                  x_bs=Ci * N, x_cs=N, x_off=T, y_bs=Ci * N2, y_cs=N2, y_off=T)
        cases.append((f"conv3s2_{B}x{Ci}x{H}x{W}", "conv3s2_nchw", kw, TOL_ROW))
        # its backward: input / weight / bias gradients (dx covers the map columns only: the first T columns of every row stay as they were)
        kb = dict(x=kw["x"], w=kw["w"], bias=None, y=None, B=B, Ci=Ci, Co=Ci, H=H, W=W, x_bs=Ci * N, x_cs=N, x_off=T, y_bs=Ci * N2, y_cs=N2, y_off=T,
                  xargs=[rnd(g, B, Ci, N2), torch.full((B, Ci, N), 7.0), torch.full((Ci, Ci, 3, 3), 7.0), torch.full((Ci,), 7.0)])
        cases.append((f"conv3s2_bwd_{B}x{Ci}x{H}x{W}", "conv3s2_nchw_bwd", kb, dict(f32=2e-5, bf16=5e-3)))
    # channel attention backward (q / kT / vT gradients from the logit and the mixed-value gradients), forward logits from the emulator
    for (B, T, C, ce, nwin, with_raw) in ((2, 2, 64, 16, 1, True), (1, 3, 136, 64, 2, True), (2, 2, 256, 256, 1, False), (1, 2, 1024, 256, 2, True)):
        Cp = (C + 7) // 8 * 8
        fk = dict(q=rnd(g, B, T, ce, scale=0.3), kvT=rnd(g, B, 2 * ce, Cp, scale=0.5), rawchan=torch.zeros(B, T, nwin * nwin, C), cx=torch.zeros(B, T, ce),
                  B=B, T=T, C=C, ce=ce, nh=nwin, nw=nwin, kv_dtype=F32, ldk=Cp, scale=ce ** -0.5, kvbias=None)
        abi_emul.call("chanattn_fwd", **fk)
        kb = dict(fk, cx=None, xargs=[rnd(g, B, T, nwin * nwin, C, scale=0.3) if with_raw else None, rnd(g, B, T, ce), torch.full((B, T, ce), 7.0),
                                      torch.full((B, 2 * ce, Cp + 8), 7.0), Cp + 8, scratch(2 * B * T * nwin * nwin * C)])
        cases.append((f"chanattn_bwd_B{B}T{T}C{C}ce{ce}w{nwin}", "chanattn_bwd", kb, dict(f32=3e-5, bf16=5e-3)))
    # modulation with 32-channel heads (last Swin stage)
    B, T, h, w, C, hg = 2, 2, 4, 6, 64, 32
    N = T + h * w
    xt = rnd(g, B, N, C)
    kw = dict(x=xt[:, T:], x_ld=C, x_bs=N * C, rawlog=rnd(g, B, C // hg, T, N), rawchan=rnd(g, B, T, 4, C), out=torch.full((2 * T, B * h * w, C), 7.0, dtype=torch.bfloat16),
              B=B, T=T, N=N, C=C, h=h, w=w, nh=2, nw=2, out_dtype=BF16, hg=hg)
    cases.append(("modulate_hg32", "modulate", kw, TOL_ROW))
    return cases


def all_cases():
    return gemm_cases() + attn_cases() + row_cases() + invpt_cases() + upconv_cases() + swin_cases()
