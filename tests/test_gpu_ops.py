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
