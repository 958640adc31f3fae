#!/usr/bin/env python3
"""Where the wave kernel's launch time goes: timing-only ablation builds (one phase removed each,
results wrong by construction: python -m th_rl_amd.build --ablate MASK --out build/libthrl_abl_MASK.so)
run the default bench workload in their own process (THRL_LIB selects the library).

    python profiles/ablate.py [--chunk 25] > gpurun_out/ablate.txt
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "full kernel", 1: "- replay passes", 2: "- play chain", 4: "- play tables (readlanes + LUT gather)", 6: "- play chain and tables",
         8: "- Philox", 16: "- per-row argmax", 32: "- visit counters (log + histogram)", 64: "- log sums", 128: "- replay schedule",
         255: "everything above removed (stream in / out, operands, loop skeleton)"}

def run(mask, chunk, extra):
    env = dict(os.environ)
    if mask:
        env["THRL_LIB"] = os.path.join(ROOT, "build", "libthrl_abl_%d.so" % mask)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(2 * chunk), "--warmup", str(chunk),
                                   "--chunk", str(chunk), "--no-cpu-baseline"] + extra, env=env, stderr=subprocess.DEVNULL)
    d = json.loads([l for l in out.decode().splitlines() if l.startswith("{")][-1])
    return d["roofline"]["avg_launch_ms"], d["value"]

if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--chunk", type=int, default=25); ap.add_argument("--extra", default="")
    a = ap.parse_args()
    extra = a.extra.split() if a.extra else []
    base = None
    for m in (0, 1, 2, 4, 6, 8, 16, 32, 64, 128, 255):
        if m and not os.path.exists(os.path.join(ROOT, "build", "libthrl_abl_%d.so" % m)):
            continue
        ms, v = run(m, a.chunk, extra)
        base = ms if m == 0 else base
        print("%3d  %-70s %8.2f ms/launch  %6.2f ms saved (%4.1f %%)  %.3e env-steps/s" % (m, NAMES[m], ms, base - ms, 100 * (base - ms) / base, v), flush=True)
