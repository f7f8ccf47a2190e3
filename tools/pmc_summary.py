"""Average PMC counter values per kernel from rocprofv3 --pmc output (counter_collection csv or results db).
Usage: python tools/pmc_summary.py <outdir> [kernel-substring ...]"""
import csv
import glob
import os
import sqlite3
import sys


def main():
    outdir, pats = sys.argv[1], sys.argv[2:]
    agg = {}
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:70], r["Counter_Name"])
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    if not files:
        for db in glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True):
            con = sqlite3.connect(db)
            tabs = [t[0] for t in con.execute("select name from sqlite_master where type in ('table','view')")]
            view = [t for t in tabs if t.startswith("counters_collection") or t == "counters_collection"]
            if not view:
                print("tables:", tabs)
                continue
            cur = con.execute(f"select * from {view[0]} limit 1")
            cols = [c[0] for c in cur.description]
            kn = "kernel_name" if "kernel_name" in cols else "name"
            for name, cname, val in con.execute(f"select {kn}, counter_name, value from {view[0]}"):
                a = agg.setdefault((name[:70], cname), [0, 0.0])
                a[0] += 1
                a[1] += float(val)
    names = sorted({k[0] for k in agg})
    for n in names:
        if pats and not any(p in n for p in pats):
            continue
        print(n)
        for (kn, c), a in sorted(agg.items()):
            if kn == n:
                print(f"    {c:32s} {a[1] / a[0]:16.1f}   (avg of {a[0]} dispatch records)")


if __name__ == "__main__":
    main()
