// thrl_policy.h -- Reinforce.pi + Categorical.sample (agents.py:147-168) for ONE game per
// wavefront, the 1 -> 256 -> A network held entirely in registers:
//   lane l owns hidden units l, l+64, l+128, l+192 (w1, b1 and one column slice of every W2 row),
//   the A logits are reduced over the 64 lanes with a transposing butterfly (32 slots -> 1 per
//   lane pair in 70 ops instead of 6 per logit), softmax / inverse-CDF run one action per lane
//   pair.  No LDS, no barriers.  k_nn_act and the fused episode kernel share this function, so
//   their logits, probabilities and samples are bit-identical.
#pragma once
#include "thrl_device.h"

namespace thrl {

constexpr int kH = THRL_NN_HIDDEN;      // 256
constexpr int kMaxA = 32;

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false));
}
// lanes without a source (row_shr past the row start, rows masked off) read 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140;

// v_permlane16_swap: odd rows of a <-> even rows of b ; v_permlane32_swap: upper half of a <-> lower half of b
__device__ __forceinline__ void swap16(float& a, float& b) {
    const v2u_t r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned x = r.x, y = r.y;
    a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
__device__ __forceinline__ void swap32(float& a, float& b) {
    const v2u_t r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned x = r.x, y = r.y;
    a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float opaque(float v) { asm("" : "+v"(v)); return v; }

struct OpMax { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; } };
struct OpAdd { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };

// all-reduce over the 64 lanes; every lane ends with the same bits (each level combines a
// value with its partner's, and the ops are commutative)
template <typename Op>
__device__ __forceinline__ float wave_all(float v, Op op) {
    v = op(v, dpp_f<kQuadXor1>(v));
    v = op(v, dpp_f<kQuadXor2>(v));
    v = op(v, dpp_f<kRowHalfMirror>(v));
    v = op(v, dpp_f<kRowMirror>(v));
    { float a = v, b = opaque(v); swap16(a, b); v = op(a, b); }
    { float a = v, b = opaque(v); swap32(a, b); v = op(a, b); }
    return v;
}
__device__ __forceinline__ double wave_allmax(double v) {
    OpMax op;
    v = op(v, dpp_d<kQuadXor1>(v));
    v = op(v, dpp_d<kQuadXor2>(v));
    v = op(v, dpp_d<kRowHalfMirror>(v));
    v = op(v, dpp_d<kRowMirror>(v));
    {
        float al = __int_as_float(__double2loint(v)), ah = __int_as_float(__double2hiint(v));
        float bl = opaque(al), bh = opaque(ah);
        swap16(al, bl); swap16(ah, bh);
        v = op(__hiloint2double(__float_as_int(ah), __float_as_int(al)), __hiloint2double(__float_as_int(bh), __float_as_int(bl)));
    }
    {
        float al = __int_as_float(__double2loint(v)), ah = __int_as_float(__double2hiint(v));
        float bl = opaque(al), bh = opaque(ah);
        swap32(al, bl); swap32(ah, bh);
        v = op(__hiloint2double(__float_as_int(ah), __float_as_int(al)), __hiloint2double(__float_as_int(bh), __float_as_int(bl)));
    }
    return v;
}
__device__ __forceinline__ float wave_allmax(float v) { return wave_all(v, OpMax()); }

// Sum 32 per-lane slots over the wave; the total of slot s ends in lanes 2s and 2s+1.
// Level by level the lanes split the slots they keep by one lane bit and add the partner's copy.
__device__ __forceinline__ float wave_sum32_transposed(float (&v)[32], int lane) {
#pragma unroll
    for (int i = 0; i < 16; i++) { swap32(v[i], v[i + 16]); v[i] = v[i] + v[i + 16]; }      // slot bit4 = lane bit5
#pragma unroll
    for (int i = 0; i < 8; i++) { swap16(v[i], v[i + 8]); v[i] = v[i] + v[i + 8]; }          // slot bit3 = lane bit4
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {                                                            // slot bit2 = lane bit3
        const float keep = b3 ? v[i + 4] : v[i], give = b3 ? v[i] : v[i + 4];
        v[i] = keep + dpp_f<kRowMirror>(give);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {                                                            // slot bit1 = lane bit2
        const float keep = b2 ? v[i + 2] : v[i], give = b2 ? v[i] : v[i + 2];
        v[i] = keep + dpp_f<kRowHalfMirror>(give);
    }
    {                                                                                        // slot bit0 = lane bit1
        const float keep = b1 ? v[1] : v[0], give = b1 ? v[0] : v[1];
        v[0] = keep + dpp_f<kQuadXor2>(give);
    }
    return v[0] + dpp_f<kQuadXor1>(v[0]);
}

// the levels of wave_sum32_transposed below its first one (16 slots, already folded over lane bit 5)
__device__ __forceinline__ float wave_sum16_transposed(float (&v)[16], int lane) {
#pragma unroll
    for (int i = 0; i < 8; i++) { swap16(v[i], v[i + 8]); v[i] = v[i] + v[i + 8]; }          // slot bit3 = lane bit4
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {                                                            // slot bit2 = lane bit3
        const float keep = b3 ? v[i + 4] : v[i], give = b3 ? v[i] : v[i + 4];
        v[i] = keep + dpp_f<kRowMirror>(give);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {                                                            // slot bit1 = lane bit2
        const float keep = b2 ? v[i + 2] : v[i], give = b2 ? v[i] : v[i + 2];
        v[i] = keep + dpp_f<kRowHalfMirror>(give);
    }
    {                                                                                        // slot bit0 = lane bit1
        const float keep = b1 ? v[1] : v[0], give = b1 ? v[0] : v[1];
        v[0] = keep + dpp_f<kQuadXor2>(give);
    }
    return v[0] + dpp_f<kQuadXor1>(v[0]);
}

typedef float pf2 __attribute__((ext_vector_type(2)));

template <int APAD>
struct PolicyRegs {
    float w1[4], b1[4];
    pf2 w2[APAD / 2][4];       // fc_pi rows (2p, 2p+1) packed for v_pk_fma_f32
    float b2;                  // bias of action (lane >> 1)
};

// params layout of one game (thrl_nn_init): w1[256] b1[256] W2[A][256] b2[A]
template <int APAD>
__device__ __forceinline__ void policy_load(PolicyRegs<APAD>& r, const float* __restrict__ w, int A, int lane) {
#pragma unroll
    for (int j = 0; j < 4; j++) { r.w1[j] = w[lane + 64 * j]; r.b1[j] = w[kH + lane + 64 * j]; }
#pragma unroll
    for (int p = 0; p < APAD / 2; p++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            r.w2[p][j] = pf2{2 * p < A ? w[2 * kH + (2 * p) * kH + lane + 64 * j] : 0.0f,
                             2 * p + 1 < A ? w[2 * kH + (2 * p + 1) * kH + lane + 64 * j] : 0.0f};
    r.b2 = (lane >> 1) < A ? w[2 * kH + A * kH + (lane >> 1)] : 0.0f;
}

// Action probabilities of the policy for the float32 state x: lane 2k holds p_k (odd lanes and
// k >= A hold 0).
template <int APAD>
__device__ __forceinline__ float policy_probs(const PolicyRegs<APAD>& r, int A, float x, int lane) {
    float h[4];
#pragma unroll
    for (int j = 0; j < 4; j++) h[j] = fmaxf(__fmaf_rn(r.w1[j], x, r.b1[j]), 0.0f);
    // The 32 per-lane logit slots are produced pair by pair and folded at once with their partner 16 slots up
    // (the first level of wave_sum32_transposed: the same swaps and additions in the same order, so the same
    // bits) -- 16 live accumulators instead of 32: with two networks in registers those 16 decide between
    // fitting the register file and spilling in the step loop.
    float v[16];
    auto dot = [&](int p) {
        pf2 s = pf2{0.0f, 0.0f};
        if (2 * p < APAD) {
#pragma unroll
            for (int j = 0; j < 4; j++) s = __builtin_elementwise_fma(r.w2[p][j], pf2{h[j], h[j]}, s);
        }
        return s;
    };
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const pf2 lo = dot(p), hi = dot(p + 8);
        float a0 = lo.x, b0 = hi.x, a1 = lo.y, b1 = hi.y;
        swap32(a0, b0); swap32(a1, b1);                                                      // slot bit4 = lane bit5
        v[2 * p] = a0 + b0; v[2 * p + 1] = a1 + b1;
    }
    const float z = wave_sum16_transposed(v, lane) + r.b2;
    const int k = lane >> 1;
    const bool valid = k < A, mine = valid && !(lane & 1);
    const float m = wave_all(valid ? z : -INFINITY, OpMax());
    const float e = mine ? expf(z - m) : 0.0f;
    const float sum = wave_all(e, OpAdd());
    return e / sum;
}
// inclusive scan of the probabilities in lane order: lane 2k (and 2k+1) holds p_0 + ... + p_k
__device__ __forceinline__ float policy_cdf(float p) {
    float c = p;
    c = c + dpp_f0<0x111, 0xF>(c);                         // row_shr:1
    c = c + dpp_f0<0x112, 0xF>(c);
    c = c + dpp_f0<0x114, 0xF>(c);
    c = c + dpp_f0<0x118, 0xF>(c);
    c = c + dpp_f0<0x142, 0xA>(c);                         // row_bcast:15 into rows 1 and 3
    c = c + dpp_f0<0x143, 0xC>(c);                         // row_bcast:31 into rows 2 and 3
    return c;
}
// Categorical sample by inverse CDF: the first action whose cumulative probability exceeds uu
__device__ __forceinline__ int policy_pick(float c, float uu, int A, int lane) {
    const bool mine = (lane >> 1) < A && !(lane & 1);
    const unsigned long long hit = __ballot(mine && uu < c);
    return hit ? (int)(__builtin_ctzll(hit) >> 1) : A - 1;
}

// Returns the action to every lane (wave-uniform).  sample: inverse CDF on `uu`; else argmax
// (get_action, agents.py:165-168).  Lane 2k holds action k's probability in *prob (odd lanes 0).
template <int APAD>
__device__ __forceinline__ int policy_act(const PolicyRegs<APAD>& r, int A, float x, bool sample, float uu, int lane,
                                          float* prob) {
    const float p = policy_probs(r, A, x, lane);
    if (prob) *prob = p;
    if (sample) return policy_pick(policy_cdf(p), uu, A, lane);
    const bool mine = (lane >> 1) < A && !(lane & 1);
    const float pm = wave_all(p, OpMax());
    return (int)(__builtin_ctzll(__ballot(mine && p == pm)) >> 1);
}

}  // namespace thrl
