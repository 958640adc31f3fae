#!/usr/bin/env python3
"""Round-3 summaries from the raw rocprofv3 output of profiles/collect_r03.sh / collect_nn_r03.sh.

    python profiles/summarize_r03.py headline r03c25 25 r03c5 5      -> profiles/<tag>_{kernel_stats,pmc_summary}.csv,
                                                                        <tag>_bench.json, traffic.json (+ r03_traffic.json)
    python profiles/summarize_r03.py nn r03nn                         -> profiles/r03_nn{rr,qr}_{kernel_stats,pmc_summary}.csv,
                                                                        r03_nn*_bench.json, nn_traffic.json

traffic.json (read by bench.py):
  * model: HBM bytes per launch = G * (bytes_per_game_per_launch + bytes_per_game_per_episode * E), solved from the two
    launch sizes.  HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE come from SEPARATE --pmc passes, are
    in KiB, and FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950: read = 2 * FETCH_SIZE * 1024.
  * issue: executed instructions per env-step (SQ_INSTS_*), the shader clock of the run (SQ_BUSY_CYCLES / 32 shader engines
    / kernel time; GRBM_GUI_ACTIVE / 8 beside it), the split of wave time.  bench.py prices the VALU count with the
    microbenchmark (profiles/r03_ubench_issue.md over profiles/isa_mix.json) -- NOT with SQ_ACTIVE_INST_VALU, which counts 4
    cycles per instruction of any class (profiles/r03_ubench_pmc_summary.csv).
  * wave / nn: hash of the kernel sources the binary the counters were collected on was built from (thrl_build_info); bench.py
    marks its roofline `stale` when the loaded library differs.
"""
import csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import collect, stats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "gpurun_out")
G, T = 1 << 20, 100
N_SIMD, N_CU, N_SE, N_XCD = 1024, 256, 32, 8


def lib_of(tag):
    p = os.path.join(GO, tag + "_library.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def write_summary(name, c):
    with open(os.path.join(ROOT, "profiles", name), "w") as f:
        f.write("counter,mean_per_dispatch,min,max,dispatches\n")
        for k, v in c.items():
            f.write("%s,%.6g,%.6g,%.6g,%d\n" % ((k,) + v))


def copy_stats(tag, name):
    ks = sorted(glob.glob(os.path.join(GO, tag + "_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(ROOT, "profiles", name))


def copy_bench(src, name):
    if os.path.exists(src):
        lines = [l for l in open(src).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(ROOT, "profiles", name), "w").write(lines[-1] + "\n")


def one(tag, E, kernel="k_wave_episodes"):
    c, geom = collect(tag, kernel)
    st = stats(tag, kernel)
    write_summary(tag + "_pmc_summary.csv", c)
    copy_stats(tag, tag + "_kernel_stats.csv")
    copy_bench(os.path.join(GO, tag + "_bench.json"), tag + "_bench.json")
    m = {k: v[0] for k, v in c.items()}
    rd, wr = 2.0 * m["FETCH_SIZE"] * 1024.0, m["WRITE_SIZE"] * 1024.0
    return dict(tag=tag, episodes_per_launch=E, counters=m, geometry=geom, kernel_stats=st, hbm_read=rd, hbm_write=wr,
                hbm_bytes=rd + wr, library=lib_of(tag))


def headline(a, ea, b, eb):
    A, B = one(a, ea), one(b, eb)
    per_ep = (A["hbm_bytes"] - B["hbm_bytes"]) / ((ea - eb) * G)
    per_launch = A["hbm_bytes"] / G - per_ep * ea
    m = A["counters"]
    t = A["kernel_stats"]["avg_ns"] * 1e-9
    cycles = m["SQ_BUSY_CYCLES"] / N_SE                       # shader cycles of one launch
    env_steps = G * T * ea
    issue = dict(
        source="profiles/%s_pmc_summary.csv + %s_kernel_stats.csv (rocprofv3 --pmc / --kernel-trace --stats, %d episodes per launch)" % (a, a, ea),
        clock_ghz=cycles / t / 1e9,
        clock_ghz_grbm=(m["GRBM_GUI_ACTIVE"] / N_XCD / t / 1e9) if "GRBM_GUI_ACTIVE" in m else None,
        avg_launch_ms=t * 1e3,
        insts_per_env_step=dict(valu=m["SQ_INSTS_VALU"] / env_steps, salu=m["SQ_INSTS_SALU"] / env_steps,
                                lds=m["SQ_INSTS_LDS"] / env_steps, branch=m["SQ_INSTS_BRANCH"] / env_steps),
        # where a wave's time goes (disjoint): issuing / parked on s_waitcnt / stalled at issue
        issue_frac=m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        wait_frac=m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        stall_frac=m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        waves_per_simd=5)
    if "SQ_LDS_IDX_ACTIVE" in m:
        issue["lds_busy_frac"] = m["SQ_LDS_IDX_ACTIVE"] / (N_CU * cycles) / 4.0
        issue["lds_bank_conflict_share"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if "SQ_ACTIVE_INST_VALU" in m:
        issue["sq_active_inst_valu_x4_per_inst"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]
    sc = os.path.join(ROOT, "profiles", "r02_wave_scaling.json")
    if os.path.exists(sc):
        issue["occupancy_scaling"] = json.load(open(sc))
        issue["occupancy_scaling"]["collected"] = "round 2 (THRL_WAVE_MAX_WAVES_PER_CU sweep); the kernel's code has not changed since"
    mix = json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json")))
    res = mix["kernels"]["k_wave_episodes<float,2,1> (headline)"]["resources"]
    geom = dict(A["geometry"] or {})
    geom.update(vgpr=res.get("NumVgprs"), sgpr=res.get("TotalNumSgprs"), scratch=res.get("ScratchSize"),
                note="grid / workgroup / LDS from the rocprofv3 dispatch record; registers and scratch from the kernel's ISA metadata "
                     "(the record's VGPR_Count field reads 48 for this 96-VGPR kernel)")
    out = dict(kernel="wave", kernel_symbol="k_wave_episodes<float, 2, 1, false, false, false, false>", games=G,
               src=A["library"].get("src"), wave=A["library"].get("wave"), library=A["library"],
               model=dict(bytes_per_game_per_launch=per_launch, bytes_per_game_per_episode=per_ep,
                          fitted_to=[dict(episodes_per_launch=x["episodes_per_launch"], hbm_bytes_per_launch=x["hbm_bytes"],
                                          read=x["hbm_read"], write=x["hbm_write"],
                                          avg_launch_ms=x["kernel_stats"]["avg_ns"] * 1e-6) for x in (A, B)],
                          note="read = 2*FETCH_SIZE*1024 (gfx950 wide-read correction), write = WRITE_SIZE*1024; separate "
                               "--pmc passes; mean over the dispatches of the kernel"),
               geometry=geom, issue=issue)
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_traffic.json"), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("wave", "model")}, indent=1)[:600])
    print(json.dumps({k: v for k, v in issue.items() if k != "occupancy_scaling"}, indent=1))


ISA_KEY = {("rr", "k_mixed_wave"): "k_mixed_wave<float,NR=2,24,2,memo> (2 x Reinforce)",
           ("qr", "k_mixed_wave"): "k_mixed_wave<float,NR=1,24,2,table> (QTable vs Reinforce)",
           ("rr", "k_ptuple_episodes"): "k_ptuple_episodes<float,NR=2,24,2,lds> (2 x Reinforce)",
           ("qr", "k_ptuple_episodes"): "k_ptuple_episodes<float,NR=1,24,2,hbm> (QTable vs Reinforce)",
           ("rr", "k_nn_reinforce_train"): "k_nn_reinforce_train<24,false>", ("qr", "k_nn_reinforce_train"): "k_nn_reinforce_train<24,false>",
           ("rr", "k_nn_returns"): "k_nn_reinforce_train<24,false>", ("qr", "k_nn_returns"): "k_nn_reinforce_train<24,false>",
           ("qa", "k_ptuple_episodes"): "k_ptuple_episodes<float,NR=1,24,2,hbm> (QTable vs Reinforce)",      # (the same kernel: the policy head)
           ("qa", "k_mixed_wave"): "k_mixed_wave<float,NR=1,24,2,table> (QTable vs Reinforce)",
           ("qa", "k_nn_reinforce_train"): "k_nn_reinforce_train<24,true> (ActorCritic)"}


def nn(tag):
    lib = lib_of(tag)
    Gn, steps = 65536, 40                                   # bench.py --workload nn --steps 40 --warmup 10
    out = dict(nn=lib.get("nn"), src=lib.get("src"), library=lib, games=Gn, pairings={})
    for p in ("rr", "qr", "qa", "qc"):
        copy_bench(os.path.join(GO, "%s_%s_bench.json" % (tag, p)), "r03_nn%s_bench.json" % p)
    for p in ("rr", "qr", "qa"):
        ptag = "%s_%s" % (tag, p)
        copy_stats(ptag, "r03_nn%s_kernel_stats.csv" % p)
        ks = {}
        total_ns = 0.0
        for kern in ("k_ptuple_episodes", "k_mixed_wave", "k_nn_reinforce_train", "k_nn_returns"):
            c, geom = collect(ptag, kern)
            st = stats(ptag, kern)
            if not c or not st:
                continue
            write_summary("r03_nn%s_%s_pmc_summary.csv" % (p, kern.replace("k_", "")), c)
            m = {k: v[0] for k, v in c.items()}
            ks[kern] = dict(m=m, st=st)
            total_ns += st["avg_ns"] * st["calls"]
        # env-steps covered by the profiled run: warm-up 10 + timed 40 episodes
        env_steps_run = Gn * T * (steps + 10)
        kernels, clock = {}, None
        for kern, d in ks.items():
            m, st = d["m"], d["st"]
            calls = st["calls"]
            t = st["avg_ns"] * 1e-9
            cyc = m["SQ_BUSY_CYCLES"] / N_SE
            clock = cyc / t / 1e9 if clock is None else clock
            kernels[kern] = dict(
                isa_key=ISA_KEY[(p, kern)], calls=calls, avg_ms=t * 1e3, time_share=st["avg_ns"] * calls / total_ns,
                valu_insts_per_env_step=m["SQ_INSTS_VALU"] * calls / env_steps_run,
                salu_insts_per_env_step=m["SQ_INSTS_SALU"] * calls / env_steps_run,
                hbm_bytes_per_env_step=(2.0 * m["FETCH_SIZE"] * 1024.0 + m["WRITE_SIZE"] * 1024.0) * calls / env_steps_run,
                hbm_bytes_per_launch=2.0 * m["FETCH_SIZE"] * 1024.0 + m["WRITE_SIZE"] * 1024.0,
                hbm_gbps=(2.0 * m["FETCH_SIZE"] * 1024.0 + m["WRITE_SIZE"] * 1024.0) / t / 1e9,
                wait_frac=m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], issue_frac=m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"],
                clock_ghz=cyc / t / 1e9)
        out["pairings"][p] = dict(kernels=kernels, clock_ghz=clock, nn=lib.get("nn"),
                                  source="profiles/r03_nn%s_*_pmc_summary.csv + r03_nn%s_kernel_stats.csv (40 + 10 episodes x 65,536 games)" % (p, p))
    json.dump(out, open(os.path.join(ROOT, "profiles", "nn_traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_nn_traffic.json"), "w"), indent=1)
    print(json.dumps(out["pairings"], indent=1)[:3000])


if __name__ == "__main__":
    if sys.argv[1] == "headline":
        headline(sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5]))
    else:
        nn(sys.argv[2])
