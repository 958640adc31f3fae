#!/usr/bin/env python3
"""Where the tuple-chain kernel's launch time goes: timing-only ablation builds (one phase skipped each, results wrong by
construction: python -m th_rl_amd.build --ablate-tuple MASK --out build/libthrl_tabl_MASK.so), each run in its own process
(THRL_LIB selects the library) on one shape of profiles/exp_tuple_one.py.

    python profiles/ablate_tuple.py [three|two] > gpurun_out/ablate_tuple.txt
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "full kernel", 1: "- replay", 2: "- play chain", 4: "- G table", 8: "- draws", 16: "- per-row argmax", 32: "- visit log + counters",
         64: "- log sums", 127: "everything above skipped (stream in / out, phase (e), loop skeleton)"}
shape = sys.argv[1] if len(sys.argv) > 1 else "three"
base = None
for m in (0, 1, 2, 4, 8, 16, 32, 64, 127):
    lib = os.path.join(ROOT, "build", "libthrl_tabl_%d.so" % m)
    if m and not os.path.exists(lib):
        continue
    env = dict(os.environ)
    if m:
        env["THRL_LIB"] = lib
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "profiles", "exp_tuple_one.py"), shape, "65536"], env=env, stderr=subprocess.DEVNULL).decode()
    ms = float(re.search(r"\(([\d.]+) ms per", out).group(1))
    base = ms if m == 0 else base
    print("%3d  %-75s %7.2f ms per 32 episodes  %6.2f ms saved (%4.1f %%)" % (m, NAMES[m], ms, base - ms, 100 * (base - ms) / base), flush=True)
