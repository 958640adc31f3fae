#!/usr/bin/env python3
"""Turn raw rocprofv3 output (gpurun_out/<tag>_stats, <tag>_pmc_*) into the small
summaries committed under profiles/ and into profiles/traffic.json (read by bench.py).

    python profiles/summarize.py r01 --games 1048576 --episodes-per-launch 25

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are collected in
SEPARATE --pmc passes, are in KiB, and FETCH_SIZE under-reports wide coalesced reads by
exactly 2x on gfx950, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as is.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--games", type=int, default=1 << 20)
    ap.add_argument("--episodes-per-launch", type=int, default=25)
    ap.add_argument("--kernel", default="k_wave_episodes")
    args = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out")
    dst = os.path.join(ROOT, "profiles")
    tag = args.tag
    stats = sorted(glob.glob(os.path.join(src, tag + "_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime,
                   reverse=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    counters = collections.OrderedDict()
    geometry = None
    for f in sorted(glob.glob(os.path.join(src, tag + "_pmc_*", "*", "*_counter_collection.csv"))):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if args.kernel in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                geometry = dict(grid=int(r["Grid_Size"]), workgroup=int(r["Workgroup_Size"]),
                                lds_block=int(r["LDS_Block_Size"]), vgpr=int(r["VGPR_Count"]),
                                sgpr=int(r["SGPR_Count"]), scratch=int(r["Scratch_Size"]))
        for k, v in per.items():
            counters[k] = dict(mean=sum(v) / len(v), min=min(v), max=max(v), dispatches=len(v))
    with open(os.path.join(dst, tag + "_pmc_summary.csv"), "w") as f:
        f.write("counter,mean_per_dispatch,min,max,dispatches\n")
        for k, v in counters.items():
            f.write("%s,%.6g,%.6g,%.6g,%d\n" % (k, v["mean"], v["min"], v["max"], v["dispatches"]))
    out = dict(tag=tag, kernel="wave" if "wave" in args.kernel else "generic", kernel_symbol=args.kernel,
               games=args.games, episodes_per_launch=args.episodes_per_launch, geometry=geometry)
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        rd = 2.0 * counters["FETCH_SIZE"]["mean"] * 1024.0
        wr = counters["WRITE_SIZE"]["mean"] * 1024.0
        out.update(fetch_size_kib=counters["FETCH_SIZE"]["mean"], write_size_kib=counters["WRITE_SIZE"]["mean"],
                   hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr,
                   hbm_bytes_per_launch=rd + wr,
                   note="read = 2*FETCH_SIZE*1024 (gfx950 wide-read correction), write = WRITE_SIZE*1024; "
                        "separate --pmc passes; mean over the timed dispatches of the kernel")
    env_steps = args.games * 100 * args.episodes_per_launch
    if "SQ_INSTS_VALU" in counters:
        out["insts_per_env_step"] = {k[9:].lower(): counters[k]["mean"] / env_steps
                                     for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if k in counters}
    if "SQ_WAVE_CYCLES" in counters:
        w = counters["SQ_WAVE_CYCLES"]["mean"]
        out["wave_time_split"] = {k: counters[k]["mean"] / w for k in
                                  ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in counters}
    json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
    b = os.path.join(src, tag + "_bench.json")
    if os.path.exists(b):
        line = [l for l in open(b).read().splitlines() if l.startswith("{")][-1]
        open(os.path.join(dst, tag + "_bench.json"), "w").write(line + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
