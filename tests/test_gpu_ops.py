"""-m gpu: every C-ABI entry point of libmtt_hip.so vs the CPU emulator (bit-for-bit identical inputs)."""
import pytest
import torch

import gpu_cases

CASES = gpu_cases.all_cases()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_entry_point_matches_emulator(case):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cname, entry, kw, tol = case
    r = gpu_cases.run_case(entry, kw, None, tol)
    assert r["ok"], f"{cname}: {r['errs']}"


@pytest.mark.gpu
def test_attention_variants_are_bitwise_identical_and_repeatable():
    """tools/attn_variants_check.py: the LDS-DMA flash kernels (default) against both register-staged variants on ragged shapes, forward and
    backward, with repeats (a staging race shows as run-to-run differences)."""
    import os
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "attn_variants_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_segcopy_packs_refresh_and_gradient_scatter_on_device():
    """mtt_segcopy on the GPU: every persistent pack layout (casts, hi/lo planes, tap-major conv matrices, padded concatenations, LDS-tiled
    transposes), the one-launch refresh after a parameter update and the gradient scatter, bit-exact against plain torch re-layouts."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pack_check
    pack_check.check_packs("cuda")
    pack_check.check_refresh("cuda")
    pack_check.check_unpack("cuda")
