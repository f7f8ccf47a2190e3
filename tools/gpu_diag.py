"""Run every GPU parity case without stopping; write gpurun_out/diag.json + a readable log.
Usage (on the GPU box):  python tools/gpu_diag.py [substring filters...]"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import gpu_cases  # noqa: E402


def main():
    filt = sys.argv[1:]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    report = {}
    t0 = time.time()
    for cname, entry, kw, tol in gpu_cases.all_cases():
        if filt and not any(f in cname for f in filt):
            continue
        try:
            r = gpu_cases.run_case(entry, kw, None, tol)
            report[cname] = r
            print(("PASS " if r["ok"] else "FAIL ") + cname, {k: (f"{v[0]:.2e}", f"{v[1]:.2e}") if isinstance(v, tuple) else v for k, v in r["errs"].items()}, flush=True)
        except Exception as e:  # noqa: BLE001
            report[cname] = dict(ok=False, exc=repr(e))
            print("EXC  " + cname, repr(e), flush=True)
            traceback.print_exc()
            try:
                torch.cuda.synchronize()
            except Exception as e2:  # noqa: BLE001
                print("device unusable after exception:", e2)
                break
    n_ok = sum(1 for r in report.values() if r.get("ok"))
    print(f"{n_ok}/{len(report)} cases passed in {time.time() - t0:.1f}s")
    with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as f:
        json.dump(report, f, indent=1, default=str)


if __name__ == "__main__":
    main()
