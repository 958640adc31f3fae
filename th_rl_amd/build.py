"""Builds th_rl_amd/libthrl_hip.so (gfx950) in-tree with hipcc.

    python -m th_rl_amd.build [--verbose]

-ffp-contract=off is part of the numerics contract: the kernels and
the host-side formula restatements must round every multiply/add separately,
like numpy does, so results are bit-identical to the oracle.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libthrl_hip.so")
SOURCES = ["thrl_api.hip", "thrl_generic.hip", "thrl_ops.hip", "thrl_wave.hip", "thrl_nn.hip", "thrl_mixed.hip", "thrl_cac.hip"]
HEADERS = ["thrl_device.h", "thrl_kernels.h", "thrl_wave_lut.h", "thrl_policy.h", "thrl_cac.h", os.path.join("..", "..", "include", "thrl.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose="--verbose" in sys.argv)
    print("built", LIB)
