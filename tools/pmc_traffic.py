"""HBM traffic of the dominant kernel from rocprofv3 PMC passes, as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE in
SEPARATE passes (TCC slots), FETCH_SIZE doubled on gfx950 (it tallies 128-byte requests at 64 bytes for wide coalesced reads),
both in KiB.  Run ON THE GPU BOX after the two passes:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch
    python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 'gemm_ring3_kernel' gpurun_out/pmc_traffic.json

Output: {kernel name: {launches, fetch_kib_raw, write_kib, hbm_bytes_per_launch, commit, csrc_sha, source}} merged into the JSON file named
last — committed as profiles/pmc_traffic.json, which bench.py reports as roofline.traffic AS LONG AS csrc_sha (hash of csrc/gemm.hip the
numbers were measured on) matches the tree; a stale record is reported as traffic = null with a note.
"""
import csv
import glob
import json
import os
import sqlite3
import sys


def per_kernel(outdir, counter, pat):
    vals = []
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and pat in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if not files:
        for db in glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True):
            con = sqlite3.connect(db)
            tabs = [t[0] for t in con.execute("select name from sqlite_master where type in ('table','view')")]
            view = [t for t in tabs if t.startswith("counters_collection")]
            if not view:
                continue
            cols = [c[0] for c in con.execute(f"select * from {view[0]} limit 1").description]
            kn = "kernel_name" if "kernel_name" in cols else "name"
            for name, cname, val in con.execute(f"select {kn}, counter_name, value from {view[0]}"):
                if cname == counter and pat in name:
                    vals.append(float(val))
    return vals


def main():
    import hashlib
    import subprocess
    fdir, wdir, pat = sys.argv[1], sys.argv[2], sys.argv[3]
    out_path = sys.argv[4] if len(sys.argv) > 4 else None            # JSON file to merge the record into (profiles/pmc_traffic.json)
    f, w = per_kernel(fdir, "FETCH_SIZE", pat), per_kernel(wdir, "WRITE_SIZE", pat)
    if not f or not w:
        print(json.dumps(dict(error="no records", fetch=len(f), write=len(w))))
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = hashlib.sha256(open(os.path.join(root, "multi-task-transformer_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]
    try:
        commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:  # noqa: BLE001
        commit = ""
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    rec = dict(kernel=pat, launches=len(f), fetch_kib_raw=round(fk, 1), write_kib=round(wk, 1),
               hbm_bytes_per_launch=int((2.0 * fk + wk) * 1024), commit=commit or os.environ.get("MTT_COMMIT", ""), csrc_sha=sha,
               source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over every launch of the kernel in one training step of "
                      "`bench.py`; FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md §HBM), KiB -> bytes; Infinity-Cache hits are counted")
    if out_path:
        try:
            allrec = json.load(open(out_path))
            if "kernel" in allrec and "hbm_bytes_per_launch" in allrec:       # round-3 single-record format
                allrec = {}
        except (OSError, ValueError):
            allrec = {}
        allrec[pat] = rec
        json.dump(allrec, open(out_path, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
