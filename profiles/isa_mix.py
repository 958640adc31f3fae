#!/usr/bin/env python3
"""Static instruction-class mix of the headline wave kernel, priced with the measured issue costs.

    python profiles/isa_mix.py [--ubench profiles/r03_ubench_issue.json] [--out profiles/r03_isa_mix.json]

Why: the SQ counters give the kernel's VALU instruction COUNT (SQ_INSTS_VALU) but not its split into the two
price classes the microbenchmark found on gfx950 (profiles/ubench_issue.hip): a plain two-source 32-bit VALU
operation whose sources are all VGPRs issues every ~2.3 cycles per SIMD, everything else this kernel is made of
(an SGPR / constant source, VOP3 three-source forms, SDWA, DPP, v_readlane / v_writelane, compares, v_cndmask,
float64) every ~4.1, v_permlane32_swap 8.1.  `SQ_ACTIVE_INST_VALU` cannot tell them apart either: it counts
exactly 4 cycles per instruction of ANY class (8 for the swap) -- measured under the same microbenchmark,
profiles/r03_ubench_pmc_summary.csv.  So the mix comes from the ISA: this script compiles the translation unit of
the headline variant (k_wave_episodes<float,2,1,false,false,false,false>, thrl_wave_f32.hip) to assembly with the
library's own flags and classifies every VALU instruction by its operand form.  The hot loops are straight-line
unrolled code, so the static mix of the kernel body is used as the estimate of the dynamic one; bench.py prints the
resulting price together with the two bounds (everything fast / everything slow).
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (translation unit, regular expression over the mangled kernel name, key in the output)
KERNELS = [("thrl_wave_f32.hip", "k_wave_episodesIfLi2ELi1ELb0ELb0ELb0ELb0EE", "k_wave_episodes<float,2,1> (headline)"),
           ("thrl_mixed.hip", "k_mixed_waveIfLi1ELi24ELi2ELb0ELi2EE", "k_mixed_wave<float,NR=1,24,2,table> (QTable vs Reinforce)"),
           ("thrl_mixed.hip", "k_mixed_waveIfLi2ELi24ELi2ELb0ELi1EE", "k_mixed_wave<float,NR=2,24,2,memo> (2 x Reinforce)"),
           ("thrl_nn.hip", "k_nn_reinforce_trainILi24ELb0EE", "k_nn_reinforce_train<24,false>"),
           ("thrl_nn.hip", "k_nn_reinforce_trainILi24ELb1EE", "k_nn_reinforce_train<24,true> (ActorCritic)"),
           ("thrl_ptuple.hip", r"k_ptuple_episodesIfLi2ELi24ELi2ELb1E(Lb0E)*E", "k_ptuple_episodes<float,NR=2,24,2,lds> (2 x Reinforce)"),
           ("thrl_ptuple.hip", r"k_ptuple_episodesIfLi1ELi24ELi2ELb0E(Lb0E)*E", "k_ptuple_episodes<float,NR=1,24,2,hbm> (QTable vs Reinforce)"),
           ("thrl_tuple_f32.hip", r"k_tuple_episodesIfLi3ELi1E(Lb0E)*E", "k_tuple_episodes<float,N=3,NSEG=1> (three players)")]
# two-source (or one-source) 32-bit operations measured at the fast price when every source is a VGPR
FAST_MEASURED = {"v_and_b32", "v_add_u32", "v_mov_b32", "v_add_f32", "v_mul_f32", "v_xor_b32"}
# same operand form, not measured one by one: assumed fast under the same condition (listed separately in the output)
FAST_ASSUMED = {"v_or_b32", "v_sub_u32", "v_subrev_u32", "v_max_f32", "v_min_f32", "v_sub_f32", "v_subrev_f32", "v_not_b32",
                "v_max_u32", "v_min_u32", "v_max_i32", "v_min_i32", "v_sub_co_u32", "v_add_co_u32"}


def asm_of_kernel(tu, symbol):
    from th_rl_amd import build
    out = os.path.join(ROOT, "build", tu.replace(".hip", ".s"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(build.CSRC, tu)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(build.CSRC, f)) for f in build.SOURCES + build.HEADERS):
        flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-S", "--cuda-device-only", "-o", out, src],
                              cwd=build.CSRC, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    # `symbol` is a regular expression over the mangled name (trailing all-false template flags are written (Lb0E)* so that a
    # new flag does not break the lookup)
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and l.rstrip().split(";")[0].strip().endswith(":") and re.search(symbol, l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    meta = {}
    for l in lines[end:end + 400]:
        m = re.match(r";\s*(NumVgprs|TotalNumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)\s*[:=]\s*(\d+)", l.strip())
        if m and m.group(1) not in meta:
            meta[m.group(1)] = int(m.group(2))
    return lines[start:end + 1], meta


def classify(line):
    """-> (unit, price_class) for one instruction line."""
    txt = line.split(";")[0].strip()
    if not txt or txt.startswith(".") or txt.endswith(":"):
        return None
    m = txt.split(None, 1)
    op = m[0]
    args = [a.strip() for a in (m[1].split(",") if len(m) > 1 else [])]
    if op.startswith("s_"):
        if op in ("s_nop", "s_waitcnt", "s_barrier", "s_sleep", "s_endpgm", "s_setprio", "s_sethalt") or op.startswith("s_waitcnt"):
            return ("sopp_nop_wait", None)
        if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64"):
            return ("branch", None)
        if op.startswith("s_load") or op.startswith("s_buffer_load") or op in ("s_memtime", "s_memrealtime", "s_dcache_inv"):
            return ("smem", None)
        return ("salu", "salu")
    if op.startswith("ds_"):
        return ("lds", "lds")
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return ("vmem", None)
    if not op.startswith("v_"):
        return ("other", None)
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if "permlane" in base:
        return ("valu", "swap8")
    if base in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
        return ("valu", "slow")
    if op.endswith(("_sdwa", "_dpp")) or " quad_perm:" in txt or " row_" in txt or "src0_sel" in txt:
        return ("valu", "slow")
    if base.startswith("v_cmp") or base.startswith("v_cndmask"):
        return ("valu", "slow" if not (base.startswith("v_cndmask") and op.endswith("_e32")) else "cndmask_vcc")
    if "64" in base.split("_", 1)[1]:
        return ("valu", "slow")                                     # float64 / 64-bit integer forms
    srcs = args[1:]
    all_vgpr = all(re.fullmatch(r"v\d+|v\[\d+:\d+\]", a) for a in srcs) and len(srcs) >= 1
    if len(srcs) >= 3:
        return ("valu", "slow")                                     # three-source VOP3 (fma, mad, max3, bfe, ...)
    if all_vgpr and base in FAST_MEASURED:
        return ("valu", "fast")
    if all_vgpr and base in FAST_ASSUMED:
        return ("valu", "fast_assumed")
    return ("valu", "slow")


def prices(ubench_path):
    """Saturated issue cost (cycles per wave64 instruction per SIMD, wall-clock based, >= 4 waves per SIMD)."""
    d = json.load(open(ubench_path))
    by = collections.defaultdict(dict)
    for r in d["rows"]:
        by[r["class"]][r["waves_per_simd"]] = r["wall_cycles_per_inst_per_simd"]
    sat = {k: sum(v[w] for w in (4, 5, 6, 8)) / 4.0 for k, v in by.items()}
    fast = [sat[k] for k in ("v_and_b32", "v_add_u32", "v_mov_b32", "v_add_f32", "v_mul_f32") if k in sat]
    slow = [sat[k] for k in ("v_and_b32 (SGPR src0)", "v_add_u32_sdwa", "v_mad_u32_u24", "v_max_f32_dpp(quad_perm)", "v_readlane_b32",
                             "v_writelane_b32", "v_max3_f32", "v_cmp_gt_f32(->vcc)", "v_fma_f64", "v_add_f64", "v_fma_f32",
                             "v_cndmask_b32_e64 (SGPR-pair mask)") if k in sat]
    out = {"fast": sum(fast) / len(fast), "slow": sum(slow) / len(slow), "swap8": sat.get("v_permlane32_swap_b32", 8.1),
           "salu": sat.get("s_add_u32", 4.16), "lds": sat.get("ds_read_b32", 8.1),
           "cndmask_vcc": sat.get("v_cndmask_b32", sat.get("v_cndmask_b32_e64 (SGPR-pair mask)", 4.2)), "table": sat}
    return out


def mix_of(tu, symbol, p):
    lines, meta = asm_of_kernel(tu, symbol)
    units, valu, ops = collections.Counter(), collections.Counter(), collections.defaultdict(collections.Counter)
    for l in lines:
        c = classify(l)
        if c is None:
            continue
        units[c[0]] += 1
        if c[0] == "valu":
            valu[c[1]] += 1
            ops[c[1]][l.split(";")[0].split()[0]] += 1
    n = sum(valu.values())
    share = {k: valu[k] / n for k in valu}
    fast_share = share.get("fast", 0.0) + share.get("fast_assumed", 0.0)
    price = (fast_share * p["fast"] + share.get("slow", 0.0) * p["slow"] + share.get("swap8", 0.0) * p["swap8"]
             + share.get("cndmask_vcc", 0.0) * p["fast"])     # fast when interleaved (2.2 cycles in a 1:1 mix with v_and_b32);
    # only a back-to-back stream of v_cndmask_b32_e32 stalls (23.5 cycles each): profiles/r03_ubench_issue.md
    return {"symbol": symbol, "translation_unit": tu, "resources": meta, "static_instructions": dict(units), "valu_classes": dict(valu),
            "valu_class_share": share, "top_ops": {k: dict(v.most_common(10)) for k, v in ops.items()},
            "valu_cycles_per_inst_static_mix": price}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ubench", default=os.path.join(ROOT, "profiles", "r03_ubench_issue.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "isa_mix.json"))
    a = ap.parse_args()
    from th_rl_amd import build
    p = prices(a.ubench)
    out = {"src": build.source_hash(), "wave": build.source_hash(build.WAVE_FILES), "nn": build.source_hash(build.NN_FILES),
           "price_cycles": {"fast": p["fast"], "slow": p["slow"], "swap8": p["swap8"], "salu": p["salu"], "lds_b32": p["lds"]},
           "valu_cycles_per_inst_bounds": [p["fast"], p["slow"]],
           "note": "static mix of each kernel body as the estimate of its dynamic mix (the hot loops are straight-line unrolled "
                   "code); prices = saturated wall-clock issue costs of profiles/r03_ubench_issue.json (mean over 4-8 waves per SIMD); "
                   "v_cndmask_b32_e32 (mask in vcc) priced at the fast class (2.2 cycles interleaved 1:1 with v_and_b32; a back-to-back "
                   "stream of them measures 23.5)",
           "kernels": {}}
    for tu, sym, key in KERNELS:
        out["kernels"][key] = mix_of(tu, sym, p)
        k = out["kernels"][key]
        print("%-55s VALU %5d  classes %s  price %.2f  %s" % (key, sum(k["valu_classes"].values()), k["valu_classes"],
                                                             k["valu_cycles_per_inst_static_mix"], k["resources"]))
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
