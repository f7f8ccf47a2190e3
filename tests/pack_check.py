"""Shared checks of the persistent weight packs (ops.seg_pack / mtt_segcopy) against plain torch re-layouts: run on the CPU emulator by
tests/test_host_cpu.py and on the HIP kernel by tests/test_gpu_ops.py.  Casts are round-to-nearest-even on both sides: bit-exact."""
import torch

import mtt_amd
from mtt_amd import ops


def pad8(n):
    import mtt_amd
    return mtt_amd.ops.pad8(n)              # the product's channel-pitch rule (multiples of 8; of 32 from ops.PITCH32_FROM channels on)


def _params(device, shapes, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(device)) for s in shapes]


def _ref_linear(ws, dtype):
    N, K = ws[0].shape[0], ws[0].numel() // ws[0].shape[0]
    buf = torch.zeros(len(ws), N, pad8(K), dtype=torch.float32, device=ws[0].device)
    for z, w in enumerate(ws):
        buf[z, :, :K] = w.detach().reshape(N, K)
    return buf.to(dtype)


def _ref_conv3(ws, dtype, transpose):
    out = []
    for w in ws:
        w = w.detach()
        w = w.permute(1, 2, 3, 0) if transpose else w.permute(0, 2, 3, 1)
        R, _, _, Cin = w.shape
        buf = torch.zeros(R, 9, pad8(Cin), dtype=torch.float32, device=w.device)
        buf[:, :, :Cin] = w.reshape(R, 9, Cin)
        out.append(buf.reshape(R, 9 * pad8(Cin)))
    return torch.stack(out).to(dtype)


def _ref_up9(ws, dtype):
    out = []
    for w in ws:
        Co, Ci = w.shape[:2]
        buf = torch.zeros(9, pad8(Co), pad8(Ci), dtype=torch.float32, device=w.device)
        buf[:, :Co, :Ci] = w.detach().permute(2, 3, 0, 1).reshape(9, Co, Ci)
        out.append(buf.reshape(9 * pad8(Co), pad8(Ci)))
    return torch.stack(out).to(dtype)


def same_split(sp, ref32):
    """a Split pack against the fp32 reference layout: hi = bf16(x), lo = bf16(x - hi), both round-to-nearest-even"""
    hi = ref32.to(torch.bfloat16)
    same(sp.hi, hi)
    same(sp.lo, (ref32 - hi.float()).to(torch.bfloat16))


def same(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert torch.equal(a.cpu(), b.cpu()), float((a.float().cpu() - b.float().cpu()).abs().max())


def check_packs(device):
    ops.clear_pack_cache()
    bf, x3 = ops.Prec("bf16"), ops.Prec("x3")
    # linear stacks: vector path (K % 4 == 0), scalar path (K = 1030 / 301), 1x1 conv weights, several chunks per segment
    for shapes in ([(300, 1024)] * 3, [(44, 1030)] * 2, [(21, 52, 1, 1)], [(1024, 4096)], [(7, 301)] * 5):
        ws = _params(device, shapes)
        for prec in (bf, x3):
            got = ops.pack_linear(ws, prec, ('t', len(shapes), shapes[0]))
            same(got, _ref_linear(ws, prec.adt))
        sp = ops.pack_linear_split(ws, ('ts', shapes[0]))
        ref = _ref_linear(ws, torch.float32)
        hi = ref.to(torch.bfloat16)
        same(sp.hi, hi)
        same(sp.lo, (ref - hi.float()).to(torch.bfloat16))
    # conv layouts
    ws = _params(device, [(52, 44, 3, 3)] * 3, 1)
    for prec in (bf, x3):
        same(ops.pack_conv3(ws, prec, 'c'), _ref_conv3(ws, prec.adt, False))
        same(ops.pack_conv3(ws, prec, 'c', transpose=True), _ref_conv3(ws, prec.adt, True))
        same(ops.pack_upconv9(ws, prec, 'u'), _ref_up9(ws, prec.adt))
    # the same layouts as pre-split planes (x3f: the split-plane implicit-GEMM conv and the nine-tap head GEMM)
    same_split(ops.pack_conv3_split(ws, 'cs'), _ref_conv3(ws, torch.float32, False))
    same_split(ops.pack_upconv9_split(ws, 'us'), _ref_up9(ws, torch.float32))
    w32 = _params(device, [(24, 64, 3, 3)] * 2, 8)                # channel pitch % 32 == 0: pack_conv3 itself picks the split layout under x3f
    got = ops.pack_conv3(w32, ops.Prec("x3f"), 'c32')
    assert isinstance(got, ops.Split)
    same_split(got, _ref_conv3(w32, torch.float32, False))
    assert not isinstance(ops.pack_conv3(w32, ops.Prec("x3f"), 'c32', transpose=True), ops.Split)      # the dgrad operand stays one bf16 plane
    # padded concatenation (fea_fuse[0]) and bias stacks
    ws = _params(device, [(52, 88, 1, 1)] * 2, 2)
    got = ops.pack_kmap(ws, 52, 96, [(0, 0, 44), (48, 44, 44)], bf, 'k')
    ref = torch.zeros(2, 52, 96, device=device)
    for z, w in enumerate(ws):
        w2 = w.detach().reshape(52, 88)
        ref[z, :, :44] = w2[:, :44]
        ref[z, :, 48:92] = w2[:, 44:]
    same(got, ref.to(torch.bfloat16))
    same_split(ops.pack_kmap_split(ws, 52, 96, [(0, 0, 44), (48, 44, 44)], 'ks'), ref)
    bs = _params(device, [(301,)] * 4, 3)
    same(ops.stack_vec(bs, 'b'), torch.stack([b.detach() for b in bs]))
    # transposed packs: tiles with ragged edges, several tiles per chunk, and the small-matrix fallback
    for shape in ((1024, 4096), (300, 1030), (70, 130), (8, 24)):
        w, = _params(device, [shape], 4)
        same(ops.pack_linear_T(w, torch.bfloat16, ('T', shape)), w.detach().t().contiguous().to(torch.bfloat16))
        same(ops.pack_linear_T(w, torch.float32, ('T32', shape)), w.detach().t().contiguous())


def check_refresh(device):
    """After a parameter update every registered pack is refreshed by ONE launch, on the first stale lookup."""
    ops.clear_pack_cache()
    bf = ops.Prec("bf16")
    ws = _params(device, [(40, 52)] * 3, 5)
    cs = _params(device, [(12, 20, 3, 3)] * 2, 6)
    a0 = ops.pack_linear(ws, bf, 'r')
    c0 = ops.pack_conv3(cs, bf, 'rc')
    n0 = ops.pack_refreshes
    assert ops.pack_linear(ws, bf, 'r') is a0 and ops.pack_refreshes == n0          # unchanged parameters: served from the registry
    with torch.no_grad():
        for p in ws + cs:
            p.mul_(1.5)                                                          # torch bumps `_version`
    a1 = ops.pack_linear(ws, bf, 'r')
    assert a1 is a0 and ops.pack_refreshes == n0 + 1                             # same persistent buffer, refreshed in place
    same(a1, _ref_linear(ws, torch.bfloat16))
    c1 = ops.pack_conv3(cs, bf, 'rc')
    assert c1 is c0 and ops.pack_refreshes == n0 + 1                             # the one refresh covered this pack too
    same(c1, _ref_conv3(cs, torch.bfloat16, False))
    for p in ws:                                                                 # a raw-pointer update: only the epoch says so
        p.data.view(-1)[0:1].copy_(torch.full((1,), 7.0, device=device))
    ops.bump_param_epoch()
    same(ops.pack_linear(ws, bf, 'r'), _ref_linear(ws, torch.bfloat16))
    assert ops.pack_refreshes == n0 + 2
    # a re-allocated parameter (new storage) rebuilds its entry instead of copying from the stale address
    ws[0].data = ws[0].data.clone() * 2
    same(ops.pack_linear(ws, bf, 'r'), _ref_linear(ws, torch.bfloat16))
    ops.bump_param_epoch()
    same(ops.pack_conv3(cs, bf, 'rc'), _ref_conv3(cs, torch.bfloat16, False))
    same(ops.pack_linear(ws, bf, 'r'), _ref_linear(ws, torch.bfloat16))
    # two models in one process: a step on the parameters of ONE pack must not invalidate a pending backward that saved
    # the OTHER pack — the refresh rewrites every pack, but only the packs whose own parameters moved get their autograd version bumped
    a, c = ops.pack_linear(ws, bf, 'r'), ops.pack_conv3(cs, bf, 'rc')
    va, vc = a._version, c._version
    with torch.no_grad():
        for p in ws:
            p.mul_(0.5)
    ops.bump_param_epoch(ws)                                                      # (what FusedClipAdam.step does: it names its parameters)
    n1 = ops.pack_refreshes
    assert ops.pack_conv3(cs, bf, 'rc') is c and ops.pack_refreshes == n1 + 1     # the epoch moved: refreshed (identical bytes) ...
    assert c._version == vc and a._version > va                                  # ... but only the changed pack's version moved
    same(ops.pack_linear(ws, bf, 'r'), _ref_linear(ws, torch.bfloat16))
    # a raw-pointer writer torch cannot see (no `_version` change) — named parameters: their packs' versions move, the others' do not;
    # unknown origin: EVERY pack is treated as rewritten, so that a pending backward holding an old pack fails loudly
    va, vc = a._version, c._version
    for p in cs:
        p.data.view(-1)[0:1].copy_(torch.full((1,), 3.0, device=device))
    ops.bump_param_epoch(cs)
    assert ops.pack_conv3(cs, bf, 'rc') is c and c._version > vc and a._version == va
    same(c, _ref_conv3(cs, torch.bfloat16, False))
    va, vc = a._version, c._version
    ops.bump_param_epoch()
    assert ops.pack_linear(ws, bf, 'r') is a and a._version > va and c._version > vc


def check_unpack(device):
    """Gradient scatter (ops.unpack_grads): relative tables reused across fresh buffers; padding columns dropped."""
    ops.clear_pack_cache()
    g = torch.Generator().manual_seed(7)
    for _ in range(2):                                                           # second round: the cached program on new buffers
        Z, N, K, Kp = 3, 52, 44, 48
        dW = torch.randn(Z, N, Kp, generator=g).to(device)
        outs = ops.unpack_grads(dW, ('lin', N, Kp), [(N, K, 1, 1)] * Z,
                                lambda src, flat, offs: [ops.segment(src, z * N * Kp, flat, offs[z], (1, N, K), (0, Kp, 1), (0, K, 1)) for z in range(Z)])
        for z in range(Z):
            same(outs[z], dW[z, :, :K].reshape(N, K, 1, 1).contiguous())
            assert outs[z].is_contiguous()
        Co, Ci, Cip = 20, 12, 16
        dC = torch.randn(Z, Co, 9 * Cip, generator=g).to(device)
        outs = ops.unpack_grads(dC, 'conv', [(Co, Ci, 3, 3)] * Z,
                                lambda src, flat, offs: [ops.segment(src, z * Co * 9 * Cip, flat, offs[z], (Co, 9, Ci), (9 * Cip, Cip, 1), (Ci * 9, 1, 9))
                                                         for z in range(Z)])
        for z in range(Z):
            same(outs[z], dC[z].view(Co, 3, 3, Cip)[..., :Ci].permute(0, 3, 1, 2).contiguous())
        part = ops.unpack_grads(dW, ('part', N), [(N, 2 * K)], lambda src, flat, offs: [ops.segment(src, 0, flat, offs[0] + K, (1, N, K), (0, Kp, 1), (0, 2 * K, 1))],
                                partial=True)[0]
        ref = torch.zeros(N, 2 * K, device=device)
        ref[:, K:] = dW[0, :, :K]
        same(part, ref)
