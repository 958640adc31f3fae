#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel of one translation unit, as the compiler reports them.
    python profiles/kernel_resources.py thrl_nn.hip [name-filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from th_rl_amd import build
tu = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
out = os.path.join(ROOT, "build", tu.replace(".hip", ".s"))
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-S", "--cuda-device-only", "-o", out, tu],
                      cwd=build.CSRC, stderr=subprocess.DEVNULL)
txt = open(out).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)", txt):
    if flt not in m.group(1):
        continue
    seg = txt[m.end():m.end() + 8000]
    g = lambda k: re.search(r";\s*%s\s*[:=]\s*(\d+)" % k, seg).group(1)
    print("%-70s vgpr %s total %s sgpr %s scratch %s occupancy %s" % (m.group(1)[:70], g("NumVgprs"), g("TotalNumVgprs"), g("TotalNumSgprs"),
                                                                       g("ScratchSize"), g("Occupancy")))
