#!/usr/bin/env python3
"""Builds the DIAGNOSTIC library build/libthrl_stamp.so: thrl_nn.hip with s_memtime stamps at the phase boundaries of the folded
network update (thread 0 of every block accumulates the time of each phase and writes 16 u64 into grad_out, whose gradient
writes are disabled).  The source file is restored afterwards; the product library is not touched.
    python profiles/make_stamp_build.py && gpurun -- 'bash profiles/gpu_stamps.sh'"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "th_rl_amd", "csrc", "thrl_nn.hip")
bak = "/tmp/thrl_nn.hip.orig"
shutil.copy(src, bak)
s = open(src).read()


def ins(marker, text):
    global s
    assert s.count(marker) == 1, marker
    s = s.replace(marker, text + marker)


try:
    s = s.replace('''    const int g = blockIdx.x, tid = threadIdx.x;
    const int Pp = 2 * kH + A * kH + A; ''', '''    const int g = blockIdx.x, tid = threadIdx.x;
    unsigned long long stamp_t[16];
    for (int i = 0; i < 16; i++) stamp_t[i] = 0;
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_t[i] += now_ - stamp_last; stamp_last = now_; } while (0)
    const int Pp = 2 * kH + A * kH + A; ''')
    ins('        // ---- State dedupe.  In a noise-free game', '        STAMP(0);\n')
    ins('            unsigned key[kParts];', '            STAMP(1);\n')
    ins('    const float invN = 1.0f / (float)N;', '    STAMP(2);\n')
    ins('        long long S0[kPad], S1[kPad];', '        STAMP(3);\n')
    ins('            {   // a unit whose threshold lies strictly inside', '            STAMP(4);\n')
    ins('            // running sums over the sorted states, A_k(s) and B_k(s) in place', '            STAMP(5);\n')
    ins('            // softmax, entropy, d loss / d logits folded over the transitions of each state: four threads per state', '            STAMP(6);\n')
    ins('            {   // inclusive prefix of d_s[k] (EA)', '            STAMP(7);\n')
    ins("            {   // this unit's range sums over the chunk", '            STAMP(8);\n')
    ins("        // the gradient of this unit's parameters, straight into the sweep", '        STAMP(9);\n')
    ins('    // Adam state of the first sweep iteration', '    STAMP(10);\n')
    ins('    const float norm = sqrtf(block_sum(sq, red));', '    STAMP(11);\n')
    assert '                if (grad_out) grad_out[(int64_t)g * P + idx] = grad;' in s
    s = s.replace('                if (grad_out) grad_out[(int64_t)g * P + idx] = grad;', '')
    i = s.index('// Discounted returns of Reinforce.train_net')
    k = s.rindex('}\n', 0, i)
    s = s[:k] + ('    STAMP(12);\n    if (grad_out && tid == 0) for (int i = 0; i < 16; i++) '
                 'reinterpret_cast<unsigned long long*>(grad_out)[(size_t)g * 16 + i] = stamp_t[i];\n') + s[k:]
    open(src, "w").write(s)
    subprocess.check_call([sys.executable, "-m", "th_rl_amd.build", "--out", os.path.join(ROOT, "build", "libthrl_stamp.so")], cwd=ROOT)
finally:
    shutil.copy(bak, src)
