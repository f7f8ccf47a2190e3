"""Static ISA checks (hipcc only, no GPU): the inline-asm LDS transpose reads of the attention and weight-gradient kernels must not have
their destination registers touched, nor any control flow, before the wait that retires them (tools/check_tr_hazards.py)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_tr_hazards", os.path.join(ROOT, "tools", "check_tr_hazards.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_checker_flags_a_copy_of_an_in_flight_register():
    good = ["_Zk:", "\tds_read_b64_tr_b16 v[10:11], v5", "\tds_read_b64_tr_b16 v[12:13], v5 offset:2048", "\tv_fma_f32 v1, v2, v3, v4",
            "\ts_waitcnt lgkmcnt(0)", "\tv_mfma_f32_16x16x32_bf16 v[20:23], v[10:13], v[30:33], v[20:23]"]
    assert chk.check_asm(good) == (1, [])
    copy = good[:3] + ["\tv_mov_b64_e32 v[40:41], v[12:13]"] + good[3:]
    blocks, problems = chk.check_asm(copy)
    assert blocks == 1 and len(problems) == 1 and "in-flight" in problems[0]
    branch = good[:3] + [".LBB0_3:"] + good[3:]
    assert "control flow" in chk.check_asm(branch)[1][0]
    partial = good[:3] + ["\ts_waitcnt lgkmcnt(1)", "\tv_mov_b32_e32 v40, v10", "\tv_mov_b32_e32 v41, v12"] + good[3:]
    blocks, problems = chk.check_asm(partial)            # the first read is retired by lgkmcnt(1), the second is not
    assert len(problems) == 1 and "v41, v12" in problems[0]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_transpose_read_intervals_are_clean():
    total, problems = chk.main()
    assert total >= 14, total          # forward (2 stages), dQ (4 halves), dK/dV (4 halves), token-major weight gradient (4)
    assert problems == [], problems
