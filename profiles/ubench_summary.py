#!/usr/bin/env python3
"""Turns the raw output of profiles/ubench_issue.hip (gpurun_out/r03_ubench_issue.json, and the two rocprofv3 --pmc
passes over the same binary, gpurun_out/r03_ubench_pmc*/) into the committed tables:

    profiles/r03_ubench_issue.json   raw rows (copied)
    profiles/r03_ubench_issue.md     cycles per wave64 instruction per SIMD by class and resident waves per SIMD
    profiles/r03_ubench_pmc_summary.csv   what the SQ "active" counters report per instruction of each class
"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "gpurun_out")
WS = [1, 2, 3, 4, 5, 6, 8]


def main():
    src = os.path.join(GO, "r03_ubench_issue.json")
    dst = os.path.join(ROOT, "profiles", "r03_ubench_issue.json")
    if os.path.exists(src):
        shutil.copy(src, dst)
    d = json.load(open(dst))
    t = collections.OrderedDict()
    for r in d["rows"]:
        t.setdefault(r["class"], {})[r["waves_per_simd"]] = r
    with open(os.path.join(ROOT, "profiles", "r03_ubench_issue.md"), "w") as f:
        f.write("# gfx950 instruction-issue microbenchmark (profiles/ubench_issue.hip), %s, %d CUs\n\n" % (d["device"], d["cus"]))
        f.write("Shader cycles per wave64 instruction **per SIMD** (kernel wall time x shader clock / instructions issued on a SIMD),\n"
                "streams of independent instructions of one class, W resident waves per SIMD on every SIMD of the chip.\n"
                "`sat` = mean over W = 4, 5, 6, 8 (the issue cost a saturated SIMD sustains); shader clock %s MHz (s_memtime / s_memrealtime).\n\n"
                % "-".join(str(int(x)) for x in (min(r["shader_mhz"] for r in d["rows"]), max(r["shader_mhz"] for r in d["rows"]))))
        f.write("| class | " + " | ".join("W=%d" % w for w in WS) + " | sat |\n|---|" + "---|" * (len(WS) + 1) + "\n")
        for k, v in t.items():
            sat = sum(v[w]["wall_cycles_per_inst_per_simd"] for w in (4, 5, 6, 8)) / 4
            f.write("| %s | " % k + " | ".join("%.2f" % v[w]["wall_cycles_per_inst_per_simd"] for w in WS) + " | **%.2f** |\n" % sat)
        f.write("\nReading: (1) one wave alone issues a VALU instruction every ~4.8 cycles whatever its class; (2) with two or more waves a SIMD\n"
                "sustains ~2.3 cycles per instruction ONLY for plain one/two-source 32-bit operations whose sources are all VGPRs (v_and/add/mov/xor/\n"
                "add_f32/mul_f32, also in the VOP3 encoding); (3) everything else costs ~4.1 cycles per instruction per SIMD however many waves are\n"
                "resident: an SGPR or inline-constant source, three-source forms (v_fma_f32 3.7-3.8, v_mad_u32_u24, v_max3, v_add3), SDWA, DPP,\n"
                "v_readlane / v_writelane, compares, v_cndmask with an SGPR-pair mask, 32-bit multiplies, every float64 operation, v_pk_fma_f32;\n"
                "v_permlane32_swap 8.1; (4) SALU 4.2 per SIMD (one scalar instruction per cycle per CU); VALU and SALU streams overlap fully;\n"
                "(5) LDS: ds_read_b32/u16 8.1 (= 128 B/clk/CU), ds_read2_b32 16, ds_bpermute 24; (6) v_cndmask_b32 in the VOP2 encoding (mask in VCC)\n"
                "is fast-class (2.2) when interleaved with other instructions but a back-to-back stream of them runs at 23.5 cycles each;\n"
                "(7) the hand-scheduled play-chain group (12 instructions = 4 dependent steps) takes 151 cycles for one wave alone (38 per step: the\n"
                "latency floor of a game's serial chain) and ~4.0-4.3 cycles per instruction per SIMD once 4+ waves share the SIMD.\n")
    # ---- what the SQ counters say per instruction of each class
    classes = list(t.keys())
    rows = []
    for tag in ("r03_ubench_pmc", "r03_ubench_pmc2"):
        fs = glob.glob(os.path.join(GO, tag, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        disp = collections.OrderedDict()
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):      # (a pass collected twice leaves two files: the newest counts)
            disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        ks = sorted(disp)
        n_cls = len(ks) // (2 * len(WS))
        for ci in range(n_cls):
            for wi, W in enumerate(WS):
                c = disp[ks[(ci * len(WS) + wi) * 2 + 1]]          # the timed launch (the first of each pair is the warm-up)
                rows.append((tag, ci, W, c))
    if rows:
        # the PMC passes may have been taken with an older class list: name classes by position only when the counts match
        with open(os.path.join(ROOT, "profiles", "r03_ubench_pmc_summary.csv"), "w") as f:
            f.write("pass,class_index,class,waves_per_simd,counter,value,per_instruction_x4\n")
            for tag, ci, W, c in rows:
                n_cls = max(r[1] for r in rows if r[0] == tag) + 1
                name = classes[ci] if n_cls == len(classes) else "class#%d" % ci
                iv, isc, il = c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0)
                for k, v in sorted(c.items()):
                    den = {"SQ_ACTIVE_INST_VALU": iv, "SQ_ACTIVE_INST_SCA": isc, "SQ_ACTIVE_INST_LDS": il}.get(k, 0)
                    f.write("%s,%d,%s,%d,%s,%.6g,%s\n" % (tag, ci, name.replace(",", ";"), W, k, v, ("%.3f" % (4 * v / den)) if den else ""))
    print(open(os.path.join(ROOT, "profiles", "r03_ubench_issue.md")).read()[:3000])


if __name__ == "__main__":
    main()
