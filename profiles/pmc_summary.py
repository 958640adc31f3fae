#!/usr/bin/env python3
"""Mean per-dispatch value of every counter collected for one kernel in gpurun_out/<tag>_pmc_*.

    python profiles/pmc_summary.py it1 [--kernel k_wave_episodes] [--out profiles/r02_x.csv]
"""
import argparse, collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def collect(tag, kernel, last=0):
    counters = collections.OrderedDict(); geom = None
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_pmc_*"))):
        if not os.path.isdir(d):
            continue
        # a pass collected twice leaves two files (named by pid): the newest one counts
        f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime, default=None)
        if f is None:
            continue
        per = collections.defaultdict(list)
        rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
        if last:        # only the last `last` dispatches of the kernel (e.g. the timed ones after a --pretrain phase)
            ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-last:]
            rows = [r for r in rows if int(r["Dispatch_Id"]) in ids]
        for r in rows:
            if True:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                geom = dict(grid=int(r["Grid_Size"]), workgroup=int(r["Workgroup_Size"]), lds=int(r["LDS_Block_Size"]),
                            vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), scratch=int(r["Scratch_Size"]))
        for k, v in per.items():
            counters[k] = (sum(v) / len(v), min(v), max(v), len(v))
    return counters, geom

def stats(tag, kernel):
    # (a pass collected twice leaves two files, named by pid: the newest one counts)
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Name"]:
                return dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), min_ns=float(r["MinNs"]), max_ns=float(r["MaxNs"]))
    return None

if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("tag"); ap.add_argument("--kernel", default="k_wave_episodes"); ap.add_argument("--out")
    ap.add_argument("--last", type=int, default=0)
    a = ap.parse_args()
    c, g = collect(a.tag, a.kernel, a.last)
    st = stats(a.tag, a.kernel)
    lines = ["counter,mean_per_dispatch,min,max,dispatches"] + ["%s,%.6g,%.6g,%.6g,%d" % ((k,) + v) for k, v in c.items()]
    print("\n".join(lines)); print("geometry", g); print("kernel stats", st)
    if a.out:
        open(a.out, "w").write("\n".join(lines) + "\n")
