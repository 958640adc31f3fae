// thrl_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Arithmetic contract (DESIGN.md "Numerics"): every float operation below is a
// separately rounded IEEE op in the order the reference evaluates it (the
// library is compiled with -ffp-contract=off, and the explicit *_rn intrinsics
// pin the places that matter), so float64 mode reproduces numpy bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/thrl.h"

namespace thrl {

// ---------------------------------------------------------------- Philox4x32-10
// Counter layout: ctr = (step, episode, game_lo, (game_hi & 0xFFFFFF) | stream<<24),
// key = 64-bit seed.  Streams: agent pair p (agents 2p, 2p+1) = p; env noise 0x40;
// table init 0x80; state init 0x81; play_greedy reset 0x82.
constexpr uint32_t kStreamNoise = 0x40u;
constexpr uint32_t kStreamInitTable = 0x80u;
constexpr uint32_t kStreamInitState = 0x81u;
constexpr uint32_t kStreamPlayReset = 0x82u;

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 64-bit product per multiplier (v_mad_u64_u32) instead of mul_hi + mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ u32x4 draw(uint64_t seed, uint64_t game, uint32_t episode, uint32_t step,
                                      uint32_t stream) {
    return philox4x32_10(step, episode, (uint32_t)game,
                         (uint32_t)((game >> 32) & 0xFFFFFFu) | (stream << 24),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

__device__ __forceinline__ double u01_32(uint32_t x) { return __dmul_rn((double)x, 0x1p-32); }
__device__ __forceinline__ double u01_53(uint32_t hi, uint32_t lo) {
    return __dmul_rn(__dadd_rn(__dmul_rn((double)(hi >> 5), 67108864.0), (double)(lo >> 6)), 0x1p-53);
}

// ---------------------------------------------------------------- per-agent constants
struct AgentParams {          // one per agent, by value in the kernel argument block
    int32_t rows;             // states + 1
    int32_t n_states;         // `states`
    int32_t n_actions;
    int32_t min_memory;
    int32_t capacity;
    int32_t table_off;        // element offset inside a game's slab
    double  max_state;
    float   max_state_f;      // weak-scalar cast used by the float32 encode
    double  inv_max_state;    // RN(1 / max_state), RN_f32(1 / max_state_f): only for the fast encodes below,
    float   inv_max_state_f;  // which fall back to the exact division near a rounding boundary
    double  gamma, alpha, one_minus_alpha;
    float   gamma_f, alpha_f, one_minus_alpha_f;
    float   alpha_gamma_f;    // alpha_f * gamma_f, one rounded float product
    double  eps_end, eps_step;
    double  act_lo, act_span; // lo, (hi - lo)
    double  act_den;          // actions - 1.0
};

struct EnvParams {
    double a, b, ratio;       // ratio = a / b
    double noise_prob, noise_lo; // noise_lo = a * 0.7
};

// QTable.encode on the float64 state (agents.py:47-49 via train_net / get_action)
// (rows are clamped to the table for memory safety; the reference raises
// IndexError for a state past the last row, so valid runs never hit the clamp)
__device__ __forceinline__ int clamp_row(int r, const AgentParams& p) { return min(max(r, 0), p.n_states); }
__device__ __forceinline__ int encode64(double price, const AgentParams& p) {
    return clamp_row((int)rint(__dmul_rn(__ddiv_rn(price, p.max_state), (double)p.n_states)), p);
}
// QTable.encode on the float32-cast state (trainer.py:53 -> agents.py:88)
__device__ __forceinline__ int encode32(double price, const AgentParams& p) {
    return clamp_row((int)rintf(__fmul_rn(__fdiv_rn((float)price, p.max_state_f), (float)p.n_states)), p);
}
// The same two encodes without the division on the common path.  q' = price * RN(1/max_state) is within
// 1.5 * 2^-52 (relative) of the correctly rounded quotient q, so y' = RN(q' * states) is within 2.5 * 2^-52 |y|
// of the reference's y = RN(q * states): rint(y') == rint(y) unless a half-integer lies that close to y'.
// In that case (probability ~1e-13 per call in float64, ~1e-4 in float32) the exact form decides.
__device__ __forceinline__ int encode64_fast(double price, const AgentParams& p) {
    const double y = __dmul_rn(__dmul_rn(price, p.inv_max_state), (double)p.n_states);
    const double r = rint(y);
    if (fabs(fabs(y - r) - 0.5) <= 0x1p-49 * fabs(y)) return encode64(price, p);      // 8 * 2^-52 |y|: margin 3x
    return clamp_row((int)r, p);
}
__device__ __forceinline__ int encode32_fast(double price, const AgentParams& p) {
    const float y = __fmul_rn(__fmul_rn((float)price, p.inv_max_state_f), (float)p.n_states);
    const float r = rintf(y);
    if (fabsf(fabsf(y - r) - 0.5f) <= 0x1p-20f * fabsf(y)) return encode32(price, p);   // 8 * 2^-23 |y|
    return clamp_row((int)r, p);
}
// QTable.scale (agents.py:51-57)
__device__ __forceinline__ double scale_action(int action, const AgentParams& p) {
    return __dadd_rn(__dmul_rn(__ddiv_rn((double)action, p.act_den), p.act_span), p.act_lo);
}

// NoisyPriceState.step (environments.py:25-39): scaled[] -> price, rewards[]
template <int MAXN>
__device__ __forceinline__ double env_step(const EnvParams& e, int n, const double* scaled,
                                           double a_eff, double* rewards) {
    double A[MAXN];
    double Q = 0.0;
#pragma unroll
    for (int i = 0; i < MAXN; i++)
        if (i < n) { A[i] = __dmul_rn(e.ratio, scaled[i]); Q = __dadd_rn(Q, A[i]); }
    double p = __dsub_rn(a_eff, __dmul_rn(e.b, Q));
    if (!(p > 0.0)) p = 0.0;
#pragma unroll
    for (int i = 0; i < MAXN; i++)
        if (i < n) rewards[i] = __dmul_rn(p, A[i]);
    return p;
}

// TD target arithmetic (agents.py:72-74), float64 and float32 flavours
__device__ __forceinline__ double td_value(double ov, double re, double nm, const AgentParams& p) {
    const double t4 = __dmul_rn(p.one_minus_alpha, ov);
    const double t2 = __dadd_rn(re, __dmul_rn(p.gamma, nm));
    return __dadd_rn(t4, __dmul_rn(p.alpha, t2));
}
__device__ __forceinline__ float td_value(float ov, double re, float nm, const AgentParams& p) {
    // float32 mode (oracle_td_update_f32): one fused op depends on the live next_max
    const float t4 = __fmul_rn(p.one_minus_alpha_f, ov);
    const float b = __fmaf_rn(p.alpha_f, (float)re, t4);
    return __fmaf_rn(p.alpha_gamma_f, nm, b);
}

// the same with explicit coefficients (per-game sweeps: alpha / gamma come from arrays)
struct TdCoef {
    double alpha, gamma, one_minus_alpha;
    float alpha_f, one_minus_alpha_f, alpha_gamma_f;
};
__device__ __forceinline__ TdCoef td_coef(const AgentParams& p) {
    return TdCoef{p.alpha, p.gamma, p.one_minus_alpha, p.alpha_f, p.one_minus_alpha_f, p.alpha_gamma_f};
}
__device__ __forceinline__ TdCoef td_coef(double alpha, double gamma) {      // as fill_agents() derives them on the host
    const double oma = __dsub_rn(1.0, alpha);
    const float af = (float)alpha, gf = (float)gamma;
    return TdCoef{alpha, gamma, oma, af, (float)oma, __fmul_rn(af, gf)};
}
__device__ __forceinline__ double td_value(double ov, double re, double nm, const TdCoef& c) {
    const double t4 = __dmul_rn(c.one_minus_alpha, ov);
    const double t2 = __dadd_rn(re, __dmul_rn(c.gamma, nm));
    return __dadd_rn(t4, __dmul_rn(c.alpha, t2));
}
__device__ __forceinline__ float td_value(float ov, double re, float nm, const TdCoef& c) {
    const float t4 = __fmul_rn(c.one_minus_alpha_f, ov);
    const float b = __fmaf_rn(c.alpha_f, (float)re, t4);
    return __fmaf_rn(c.alpha_gamma_f, nm, b);
}

template <typename T>
__device__ __forceinline__ int argmax_row(const T* __restrict__ row, int n) {
    int b = 0;
    T bv = row[0];
    for (int k = 1; k < n; k++) {
        const T v = row[k];
        if (v > bv) { bv = v; b = k; }
    }
    return b;
}
template <typename T>
__device__ __forceinline__ T max_row(const T* __restrict__ row, int n) {
    T bv = row[0];
    for (int k = 1; k < n; k++) {
        const T v = row[k];
        if (v > bv) bv = v;
    }
    return bv;
}

}  // namespace thrl
