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
SOURCES = ["thrl_api.hip", "thrl_generic.hip", "thrl_ops.hip", "thrl_wave.hip", "thrl_wave_f32.hip", "thrl_wave_f32n.hip",
           "thrl_wave_f32s.hip", "thrl_wave_f64.hip", "thrl_wave_f64n.hip", "thrl_wave_f64s.hip", "thrl_nn.hip", "thrl_mixed.hip",
           "thrl_cac.hip"]
HEADERS = ["thrl_device.h", "thrl_kernels.h", "thrl_wave_lut.h", "thrl_wave_kernel.h", "thrl_policy.h", "thrl_cac.h", os.path.join("..", "..", "include", "thrl.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    """Each source is compiled to an object in parallel (one hipcc per source), then linked."""
    if not force and up_to_date():
        return LIB
    import concurrent.futures
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in FLAGS if f != "-shared"]
    if verbose:
        cflags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    with tempfile.TemporaryDirectory() as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, src.replace(".hip", ".o"))
            cmd = [hipcc] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
            return obj
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_one, SOURCES))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB], cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose="--verbose" in sys.argv)
    print("built", LIB)
