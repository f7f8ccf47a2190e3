"""Summarise a rocprofv3 --kernel-trace results DB (or kernel-trace CSV) into a text table for profiles/.
Usage: python tools/prof_summary.py <rocprof_out_dir> [n_iters] > profiles/<name>.txt"""
import csv
import glob
import os
import sqlite3
import sys


def rows_from(outdir):
    dbs = glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True)
    if dbs:
        con = sqlite3.connect(dbs[0])
        cur = con.cursor()
        return list(cur.execute("select name, end-start, grid_x/workgroup_x, grid_z, vgpr_count, accum_vgpr_count, lds_size from kernels"))
    out = []
    for f in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
                        int(r["Grid_Size_Z"]), int(r.get("VGPR_Count", 0) or 0), int(r.get("Accum_VGPR_Count", 0) or 0), int(r.get("LDS_Block_Size", 0) or 0)))
    return out


def main():
    outdir = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = rows_from(outdir)
    agg = {}
    for name, dur, gx, gz, vg, ag, lds in rows:
        a = agg.setdefault(name, [0, 0, vg, ag, lds])
        a[0] += 1
        a[1] += dur
    tot = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 --kernel-trace summary of {outdir}: {len(rows)} dispatches, total kernel time {tot / 1e6:.2f} ms over {iters} iteration(s)")
    print(f"# {'total_ms':>10} {'%':>6} {'calls':>7} {'avg_us':>10}  vgpr agpr   lds  kernel")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {a[1] / 1e6:10.3f} {100 * a[1] / tot:6.2f} {a[0]:7d} {a[1] / a[0] / 1e3:10.1f}  {a[2]:4d} {a[3]:4d} {a[4]:6d}  {name[:120]}")
    # per launch shape (workgroups x, grid z) of the GEMM-family kernels: which call sites the time belongs to
    shp = {}
    for name, dur, gx, gz, vg, ag, lds in rows:
        if "gemm" in name:
            key = (name.replace("(anonymous namespace)::", "").replace("void ", "")[:34], gx, gz)
            a = shp.setdefault(key, [0, 0])
            a[0] += 1
            a[1] += dur
    print("# GEMM-family kernels by launch shape:   total_ms  calls  avg_us  workgroups  grid_z  kernel")
    for (name, gx, gz), a in sorted(shp.items(), key=lambda kv: -kv[1][1])[:36]:
        print(f"  {a[1] / 1e6:10.3f} {a[0]:6d} {a[1] / a[0] / 1e3:9.1f} {gx:8d} {gz:5d}  {name}")


if __name__ == "__main__":
    main()
