"""Builds th_rl_amd/libthrl_hip.so (gfx950) in-tree with hipcc.

    python -m th_rl_amd.build [--verbose] [--ablate MASK --out PATH]

-ffp-contract=off is part of the numerics contract: the kernels and
the host-side formula restatements must round every multiply/add separately,
like numpy does, so results are bit-identical to the oracle.

Objects are cached under build/obj (keyed by source, flags and the newest header), so an edit to
one kernel file recompiles one translation unit.  --ablate builds a TIMING-ONLY diagnostic variant
of the wave kernel (phases skipped by -DTHRL_ABLATE=MASK, results wrong by construction) into a
separate library that profiles/ablate.py loads through THRL_LIB; the product library never has it.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libthrl_hip.so")
OBJ_DIR = os.path.join(os.path.dirname(HERE), "build", "obj")
SOURCES = ["thrl_api.hip", "thrl_generic.hip", "thrl_ops.hip", "thrl_wave.hip", "thrl_wave_f32.hip", "thrl_wave_f32c.hip",
           "thrl_wave_f32g.hip", "thrl_wave_f32n.hip", "thrl_wave_f32nc.hip", "thrl_wave_f32s.hip", "thrl_wave_f64.hip", "thrl_wave_f64c.hip", "thrl_wave_f64g.hip",
           "thrl_wave_f64n.hip", "thrl_wave_f64nc.hip", "thrl_wave_f64s.hip", "thrl_nn.hip", "thrl_mixed.hip", "thrl_cac.hip",
           "thrl_tuple.hip", "thrl_tuple_f32.hip", "thrl_tuple_f64.hip", "thrl_tuple_f32_noise.hip", "thrl_tuple_f64_noise.hip", "thrl_tuple_f32_sweep.hip", "thrl_tuple_f64_sweep.hip",
           "thrl_ptuple.hip"]
HEADERS = ["thrl_device.h", "thrl_kernels.h", "thrl_wave_lut.h", "thrl_wave_kernel.h", "thrl_tuple_kernel.h", "thrl_policy.h", "thrl_cac.h",
           os.path.join("..", "..", "include", "thrl.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wno-pass-failed"]


def _deps():
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]


def up_to_date(lib=LIB):
    if not os.path.exists(lib):
        return False
    t = os.path.getmtime(lib)
    return all(os.path.getmtime(d) <= t for d in _deps())


WAVE_FILES = ["thrl_wave_kernel.h", "thrl_wave_f32.hip", "thrl_wave_lut.h", "thrl_kernels.h", "thrl_device.h"]
NN_FILES = ["thrl_mixed.hip", "thrl_ptuple.hip", "thrl_tuple_kernel.h", "thrl_nn.hip", "thrl_policy.h", "thrl_cac.h", "thrl_cac.hip", "thrl_kernels.h", "thrl_device.h"]


def source_hash(files=None):
    """sha1 (12 hex digits) over kernel sources, headers and compiler flags.  thrl_build_info() reports three:
    `src=` everything, `wave=` what the headline kernel k_wave_episodes<float,...> is compiled from, `nn=` the
    neural-agent kernels.  profiles/traffic.json / nn_traffic.json record the hash of the binary their PMC
    constants were collected on, so bench.py can tell when they are stale."""
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for f in sorted(files if files is not None else SOURCES + HEADERS):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def build(force=False, verbose=False, ablate=0, out=None, ablate_tuple=0):
    """Each source is compiled to an object in parallel (one hipcc per source), then linked."""
    lib = os.path.abspath(out) if out else LIB
    if not force and not ablate and not ablate_tuple and up_to_date(lib):
        return lib
    import concurrent.futures
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in FLAGS if f != "-shared"]
    if verbose:
        cflags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)

    src_hash, wave_hash, nn_hash = source_hash(), source_hash(WAVE_FILES), source_hash(NN_FILES)

    def compile_one(src):
        flags = list(cflags)
        if ablate and src == "thrl_wave_f32.hip":
            flags.append("-DTHRL_ABLATE=%d" % ablate)
        if ablate_tuple and src == "thrl_tuple_f32.hip":
            flags.append("-DTHRL_TUP_ABLATE=%d" % ablate_tuple)
        if src == "thrl_api.hip":           # thrl_build_info(): which binary is this
            flags += ["-DTHRL_BUILD_ABLATE=%d" % (ablate | (ablate_tuple << 16)), '-DTHRL_SRC_HASH="%s"' % src_hash,
                      '-DTHRL_WAVE_HASH="%s"' % wave_hash, '-DTHRL_NN_HASH="%s"' % nn_hash]
        key = hashlib.sha1((" ".join(flags) + hipcc).encode()).hexdigest()[:10]
        obj = os.path.join(OBJ_DIR, "%s-%s.o" % (src.replace(".hip", ""), key))
        path = os.path.join(CSRC, src)
        if (not force and not verbose and os.path.exists(obj)
                and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_time)):
            return obj
        cmd = [hipcc] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], cwd=CSRC)
    return lib


if __name__ == "__main__":
    ab, outp = 0, None
    if "--ablate" in sys.argv:
        ab = int(sys.argv[sys.argv.index("--ablate") + 1], 0)
    abt = int(sys.argv[sys.argv.index("--ablate-tuple") + 1], 0) if "--ablate-tuple" in sys.argv else 0
    if "--out" in sys.argv:
        outp = sys.argv[sys.argv.index("--out") + 1]
    print("built", build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, ablate=ab, out=outp, ablate_tuple=abt))
