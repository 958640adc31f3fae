#!/usr/bin/env python3
"""Round-2 summaries from the raw rocprofv3 output of profiles/collect_r02.sh (two launch sizes):

    python profiles/summarize_r02.py r02c25 25 r02c5 5

writes profiles/<tag>_kernel_stats.csv, <tag>_pmc_summary.csv, <tag>_bench.json for both tags and
profiles/traffic.json (read by bench.py):
  * model: HBM bytes per launch = G * (bytes_per_game_per_launch + bytes_per_game_per_episode * E), solved
    from the two launch sizes.  HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE come from
    SEPARATE --pmc passes, are in KiB, and FETCH_SIZE under-reports wide coalesced reads by exactly 2x on
    gfx950, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as is.
  * issue: the kernel's real bound from the SQ counters of the larger launch size.
"""
import json, os, shutil, sys, glob
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import collect, stats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, T = 1 << 20, 100
N_SIMD, N_CU, N_SE = 1024, 256, 32           # MI355X: 256 CUs x 4 SIMD-32; SQ_BUSY_CYCLES sums 32 shader engines


def one(tag, E, kernel="k_wave_episodes"):
    c, geom = collect(tag, kernel)
    st = stats(tag, kernel)
    with open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.csv"), "w") as f:
        f.write("counter,mean_per_dispatch,min,max,dispatches\n")
        for k, v in c.items():
            f.write("%s,%.6g,%.6g,%.6g,%d\n" % ((k,) + v))
    ks = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    b = os.path.join(ROOT, "gpurun_out", tag + "_bench.json")
    if os.path.exists(b):
        line = [l for l in open(b).read().splitlines() if l.startswith("{")][-1]
        open(os.path.join(ROOT, "profiles", tag + "_bench.json"), "w").write(line + "\n")
    m = {k: v[0] for k, v in c.items()}
    rd, wr = 2.0 * m["FETCH_SIZE"] * 1024.0, m["WRITE_SIZE"] * 1024.0
    return dict(tag=tag, episodes_per_launch=E, counters=m, geometry=geom, kernel_stats=st, hbm_read=rd, hbm_write=wr,
                hbm_bytes=rd + wr)


def main():
    a, ea, b, eb = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    A, B = one(a, ea), one(b, eb)
    per_ep = (A["hbm_bytes"] - B["hbm_bytes"]) / ((ea - eb) * G)
    per_launch = A["hbm_bytes"] / G - per_ep * ea
    m = A["counters"]
    t = A["kernel_stats"]["avg_ns"] * 1e-9
    cycles = m["SQ_BUSY_CYCLES"] / N_SE                       # shader cycles of one launch (= t * clock)
    env_steps = G * T * ea
    issue = dict(
        source="profiles/%s_pmc_summary.csv (rocprofv3 --pmc, %d episodes per launch)" % (a, ea),
        clock_ghz=cycles / t / 1e9,
        insts_per_env_step=dict(valu=m["SQ_INSTS_VALU"] / env_steps, salu=m["SQ_INSTS_SALU"] / env_steps,
                                lds=m["SQ_INSTS_LDS"] / env_steps, branch=m["SQ_INSTS_BRANCH"] / env_steps),
        # VALU issue: a SIMD issues at most one vector instruction per 4-cycle issue slot (what the counters
        # show here: SALU and VALU both come out at 4 cycles per instruction per SIMD); `valu_frac` = share of
        # those slots used; `valu_frac_2cyc` prices an instruction at the SIMD-32 execute time of 2 cycles
        # (MI355X_MICROARCH.md) -- the rate back-to-back independent FMAs from several waves would reach
        valu_frac=m["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * cycles),
        valu_frac_2cyc=m["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * cycles),
        # time waves spend executing VALU instructions (quad-cycles x 4), summed over a SIMD's 5 waves, per
        # SIMD cycle: multi-cycle instructions (float64, 32-bit multiplies, DPP / readlane hazards) included
        valu_active_frac=m["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * cycles),
        salu_frac=m["SQ_INSTS_SALU"] * 4.0 / (N_SIMD * cycles),     # scalar issue slots, same pricing
        lds_busy_frac=m["SQ_LDS_IDX_ACTIVE"] / (N_CU * cycles) / 4.0 if "SQ_LDS_IDX_ACTIVE" in m else None,
        lds_bank_conflict_share=m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"],
        # where a wave's time goes (disjoint): issuing / parked on s_waitcnt / stalled at issue
        issue_frac=m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        wait_frac=m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        stall_frac=m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        waves_per_simd=5)
    sc = os.path.join(ROOT, "profiles", "r02_wave_scaling.json")
    if os.path.exists(sc):
        issue["occupancy_scaling"] = json.load(open(sc))
    out = dict(kernel="wave", kernel_symbol="k_wave_episodes<float, 2, 1, false, false>", games=G,
               model=dict(bytes_per_game_per_launch=per_launch, bytes_per_game_per_episode=per_ep,
                          fitted_to=[dict(episodes_per_launch=x["episodes_per_launch"], hbm_bytes_per_launch=x["hbm_bytes"],
                                          read=x["hbm_read"], write=x["hbm_write"],
                                          avg_launch_ms=x["kernel_stats"]["avg_ns"] * 1e-6) for x in (A, B)],
                          note="read = 2*FETCH_SIZE*1024 (gfx950 wide-read correction), write = WRITE_SIZE*1024; separate "
                               "--pmc passes; mean over the dispatches of the kernel"),
               geometry=A["geometry"], issue=issue)
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
