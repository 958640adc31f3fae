// thrl_cac.h -- CAC policy heads for one game per wavefront, shared by k_cac_act (weights in HBM)
// and the fused episode kernel (weights in LDS): identical arithmetic => identical actions.
#pragma once
#include <math.h>

#include "thrl_policy.h"

namespace thrl {

constexpr int kCacP = THRL_CAC_PARAMS;
constexpr int kCacW1 = 0, kCacB1 = kH, kCacWmu = 2 * kH, kCacBmu = 3 * kH, kCacWstd = 3 * kH + 1,
              kCacBstd = 4 * kH + 1, kCacWv = 4 * kH + 2, kCacBv = 5 * kH + 2;

__device__ __forceinline__ float softplus_f(float s) { return s > 20.0f ? s : log1pf(expf(s)); }   // torch threshold 20
__device__ __forceinline__ float sigmoid_f(float a) { return 1.0f / (1.0f + expf(-a)); }
// standard normal from two uniforms in [0,1): float64 Box-Muller, as the oracle
__device__ __forceinline__ float box_muller_f(double u1, double u2) {
    return (float)(sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2));
}

// mu = 4 tanh(fc_mu h), std = softplus(fc_std h) for the float32 state x (agents.py:360-364);
// `w` is any pointer (global or LDS) to the game's THRL_CAC_PARAMS floats.  Wave-uniform results.
template <typename P>
__device__ __forceinline__ void cac_policy(P w, float x, int lane, float& mu, float& sd) {
    float pm = 0.0f, ps = 0.0f;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const int j = lane + 64 * jj;
        const float h = fmaxf(__fmaf_rn(w[kCacW1 + j], x, w[kCacB1 + j]), 0.0f);
        pm = __fmaf_rn(w[kCacWmu + j], h, pm); ps = __fmaf_rn(w[kCacWstd + j], h, ps);
    }
    const float m = wave_all(pm, OpAdd()) + w[kCacBmu], s = wave_all(ps, OpAdd()) + w[kCacBstd];
    mu = 4.0f * tanhf(m); sd = softplus_f(s);
}

}  // namespace thrl
