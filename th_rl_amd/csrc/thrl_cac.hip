// thrl_cac.hip -- the reference's continuous actor-critic agent `CAC` (agents.py:333-442) for G games:
// shared hidden layer fc1 (1 -> 256) and three 256 -> 1 heads, mu = 4*tanh(fc_mu h),
// std = softplus(fc_std h), v = fc_v h.  Parameter vector per game (THRL_CAC_PARAMS = 1283 floats):
//   [fc1.weight 256 | fc1.bias 256 | fc_mu.weight 256 | fc_mu.bias | fc_std.weight 256 | fc_std.bias |
//    fc_v.weight 256 | fc_v.bias]
// float32 like torch; the O(N) closed forms of the [N,N] broadcasts are evaluated in float64.
#include "thrl_cac.h"
#include "thrl_kernels.h"

namespace thrl {
namespace {

constexpr int kP = kCacP;
constexpr int oW1 = kCacW1, oB1 = kCacB1, oWmu = kCacWmu, oBmu = kCacBmu, oWstd = kCacWstd, oBstd = kCacBstd,
              oWv = kCacWv, oBv = kCacBv;
constexpr uint32_t kStreamCacInit = 0x91u;

// torch.nn.Linear default init: U(-1/sqrt(fan_in), +) for weights and biases (fan_in 1 for fc1, 256 for the heads)
__global__ void __launch_bounds__(256) k_cac_init(int G, float* params, uint64_t seed, uint64_t game_offset, int agent) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)G * kP) return;
    const int g = (int)(idx / kP), j = (int)(idx - (int64_t)g * kP);
    const u32x4 x = draw(seed, game_offset + (uint64_t)g, (uint32_t)agent, (uint32_t)(j >> 2), kStreamCacInit);
    const uint32_t r = (j & 3) == 0 ? x.x : ((j & 3) == 1 ? x.y : ((j & 3) == 2 ? x.z : x.w));
    const float u = (float)((double)r * 0x1p-32);
    params[idx] = (2.0f * u - 1.0f) * (j < 2 * kH ? 1.0f : 1.0f / sqrtf((float)kH));
}

// pi() + sample_action (agents.py:360-381): one game per wavefront, 4 games per block.
// u1 == NULL: the mean action sigmoid(mu) (the reference's get_action raises, see include/thrl.h).
__global__ void __launch_bounds__(256) k_cac_act(int G, const float* __restrict__ params, const double* __restrict__ price,
        const double* __restrict__ u1, const double* __restrict__ u2, float* __restrict__ action_out,
        float* __restrict__ mu_out, float* __restrict__ std_out, float* __restrict__ v_out) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const float* w = params + (int64_t)g * kP;
    const float x = (float)price[g];
    float mu, sd;
    cac_policy(w, x, lane, mu, sd);
    float a = mu;
    if (u1) a = mu + sd * box_muller_f(u1[g], u2[g]);
    if (lane == 0) {
        action_out[g] = sigmoid_f(a);
        if (mu_out) mu_out[g] = mu;
        if (std_out) std_out[g] = sd;
    }
    if (v_out) {
        float pv = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int j = lane + 64 * jj;
            pv = __fmaf_rn(w[oWv + j], fmaxf(__fmaf_rn(w[oW1 + j], x, w[oB1 + j]), 0.0f), pv);
        }
        const float v = wave_all(pv, OpAdd()) + w[oBv];
        if (lane == 0) v_out[g] = v;
    }
}

__device__ __forceinline__ float block_sum_f(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

// train_net (agents.py:389-416) as the reference executes it: rewards / actions [N] against
// mu / std / v [N,1] broadcast to [N,N]:
//   adv[i,j] = r_j + gamma*v'_i - v_i ,  log_prob[i,j] = log N(y_j; mu_i, std_i),  y = logit(5e-5 + (1-1e-4)*a)
//   loss = mean_ij(adv^2 - log_prob*adv.detach()) + ent*(-mean_i H_i)
// The per-row sums S0 = sum_j adv, S1 = sum_j adv*(y_j-mu_i), S2 = sum_j adv*(y_j-mu_i)^2 have closed
// forms in the five batch sums R, Y, YY, RY, RYY (float64 here), so the update is O(N).
__global__ void __launch_bounds__(256) k_cac_train(int G, float* __restrict__ params, float* __restrict__ adam_m,
        float* __restrict__ adam_v, int step, int N, int ld, const double* __restrict__ price, const float* __restrict__ action,
        const double* __restrict__ reward, const double* __restrict__ nprice, float gamma, float ent_coef, float lr,
        const double* __restrict__ gamma_g, const double* __restrict__ ent_g, float* __restrict__ grad_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cac[];
    if (gamma_g) gamma = (float)gamma_g[blockIdx.x];        // per-game sweeps
    if (ent_g) ent_coef = (float)ent_g[blockIdx.x];
    double* redd = reinterpret_cast<double*>(smem_cac);     // [4]
    float* red = reinterpret_cast<float*>(redd + 4);        // [8]
    float* ws = red + 8;                                    // [5][kH]: w1 b1 wmu wstd wv
    float* xs = ws + 5 * kH;                                // [N] each below
    float* xps = xs + N; float* rs = xps + N; float* ys = rs + N;
    float* ms = ys + N; float* ss = ms + N; float* cs = ss + N;
    const int g = blockIdx.x, tid = threadIdx.x;
    float* w = params + (int64_t)g * kP;
    ws[tid] = w[oW1 + tid]; ws[kH + tid] = w[oB1 + tid]; ws[2 * kH + tid] = w[oWmu + tid];
    ws[3 * kH + tid] = w[oWstd + tid]; ws[4 * kH + tid] = w[oWv + tid];
    const float bmu = w[oBmu], bstd = w[oBstd], bv = w[oBv];
    for (int n = tid; n < N; n += 256) {
        // the replayed transitions of game g are one contiguous row [ld] per array (game-major rings)
        const size_t m = (size_t)g * ld + n;
        xs[n] = (float)price[m]; xps[n] = (float)nprice[m];
        rs[n] = (float)reward[m];
        const float a2 = 5e-5f + (1.0f - 1e-4f) * action[m];
        ys[n] = logf(a2 / (1.0f - a2));
    }
    __syncthreads();
    // ---- phase 1 (thread = transition): heads on s and s', batch sums
    double R = 0, Y = 0, YY = 0, RY = 0, RYY = 0;
    for (int n = tid; n < N; n += 256) {
        const float x = xs[n], xp = xps[n];
        float m = bmu, s = bstd, v = bv, vp = bv;
        for (int j = 0; j < kH; j++) {
            const float w1 = ws[j], b1 = ws[kH + j], wv = ws[4 * kH + j];
            const float h = fmaxf(__fmaf_rn(w1, x, b1), 0.0f), hp = fmaxf(__fmaf_rn(w1, xp, b1), 0.0f);
            m = __fmaf_rn(ws[2 * kH + j], h, m); s = __fmaf_rn(ws[3 * kH + j], h, s);
            v = __fmaf_rn(wv, h, v); vp = __fmaf_rn(wv, hp, vp);
        }
        ms[n] = m; ss[n] = s; cs[n] = gamma * vp - v;
        const double r = rs[n], y = ys[n];
        R += r; Y += y; YY += y * y; RY += r * y; RYY += r * y * y;
    }
    R = block_sum_d(R, redd); Y = block_sum_d(Y, redd); YY = block_sum_d(YY, redd);
    RY = block_sum_d(RY, redd); RYY = block_sum_d(RYY, redd);
    // ---- phase 2 (thread = transition): d loss / d (pre-tanh m_i, pre-softplus s_i, v_i)
    const double dN = (double)N, inv_n2 = 1.0 / (dN * dN);
    float pbm = 0.0f, pbs = 0.0f, pbv = 0.0f;
    for (int n = tid; n < N; n += 256) {
        const double c = cs[n], m = ms[n], s = ss[n];
        const double mu = 4.0 * (double)tanhf((float)m), sd = (double)softplus_f((float)s);
        const double S0 = R + dN * c;
        const double S1 = RY + c * Y - mu * S0;
        const double S2 = RYY - 2.0 * mu * RY + mu * mu * R + c * (YY - 2.0 * mu * Y + dN * mu * mu);
        const double dmu = -(S1 / (sd * sd)) * inv_n2;
        const double dsd = -((S2 / (sd * sd * sd)) - S0 / sd) * inv_n2 - (double)ent_coef / (dN * sd);
        const double th = tanh(m);
        const float dm = (float)(dmu * 4.0 * (1.0 - th * th));
        const float ds = (float)(dsd / (1.0 + exp(-s)));
        const float gv = (float)(-2.0 * inv_n2 * S0);
        ms[n] = dm; ss[n] = ds; cs[n] = gv;
        pbm += dm; pbs += ds; pbv += gv;
    }
    const float gbmu = block_sum_f(pbm, red), gbstd = block_sum_f(pbs, red);
    const float gbv = (1.0f - gamma) * block_sum_f(pbv, red);            // gv' = -gamma * gv
    // ---- phase 3 (thread = hidden unit)
    const float w1 = ws[tid], b1 = ws[kH + tid], wmu = ws[2 * kH + tid], wsd = ws[3 * kH + tid], wv = ws[4 * kH + tid];
    float gw1 = 0.0f, gb1 = 0.0f, gwmu = 0.0f, gwsd = 0.0f, gwv = 0.0f;
    for (int n = 0; n < N; n++) {
        const float x = xs[n], xp = xps[n], dm = ms[n], ds = ss[n], gv = cs[n], gvp = -gamma * gv;
        const float pre = __fmaf_rn(w1, x, b1), prp = __fmaf_rn(w1, xp, b1);
        const float h = fmaxf(pre, 0.0f), hp = fmaxf(prp, 0.0f);
        gwmu = __fmaf_rn(dm, h, gwmu); gwsd = __fmaf_rn(ds, h, gwsd);
        gwv = __fmaf_rn(gv, h, gwv); gwv = __fmaf_rn(gvp, hp, gwv);
        if (pre > 0.0f) { const float dh = dm * wmu + ds * wsd + gv * wv; gw1 = __fmaf_rn(dh, x, gw1); gb1 += dh; }
        if (prp > 0.0f) { const float dh = gvp * wv; gw1 = __fmaf_rn(dh, xp, gw1); gb1 += dh; }
    }
    // clip_grad_norm_(1.0), Adam
    float sq = gw1 * gw1 + gb1 * gb1 + gwmu * gwmu + gwsd * gwsd + gwv * gwv;
    if (tid == 0) sq += gbmu * gbmu + gbstd * gbstd + gbv * gbv;
    const float norm = sqrtf(block_sum_f(sq, red));
    const float coef = fminf(1.0f, 1.0f / (norm + 1e-6f));
    const float t = (float)(step + 1);
    const float bc1 = 1.0f - powf(0.9f, t), bc2s = sqrtf(1.0f - powf(0.999f, t));
    const float step_size = lr / bc1;
    float* mg = adam_m + (int64_t)g * kP;
    float* vg = adam_v + (int64_t)g * kP;
    auto upd = [&](int idx, float grad) {
        grad *= coef;
        if (grad_out) grad_out[(int64_t)g * kP + idx] = grad;
        const float m = 0.9f * mg[idx] + 0.1f * grad;
        const float v = 0.999f * vg[idx] + 0.001f * grad * grad;
        mg[idx] = m; vg[idx] = v;
        w[idx] = w[idx] - step_size * (m / (sqrtf(v) / bc2s + 1e-8f));
    };
    upd(oW1 + tid, gw1); upd(oB1 + tid, gb1); upd(oWmu + tid, gwmu); upd(oWstd + tid, gwsd); upd(oWv + tid, gwv);
    if (tid == 0) { upd(oBmu, gbmu); upd(oBstd, gbstd); upd(oBv, gbv); }
}

}  // namespace

int launch_cac_init(int G, float* params, uint64_t seed, uint64_t off, int agent, hipStream_t s) {
    const int64_t n = (int64_t)G * kP;
    hipLaunchKernelGGL(k_cac_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, G, params, seed, off, agent);
    return (int)hipGetLastError();
}
int launch_cac_act(int G, const float* params, const double* price, const double* u1, const double* u2, float* action,
                   float* mu, float* sd, float* v, hipStream_t s) {
    hipLaunchKernelGGL(k_cac_act, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, s, G, params, price, u1, u2, action, mu, sd, v);
    return (int)hipGetLastError();
}
size_t cac_train_lds_bytes(int N) { return 4 * sizeof(double) + sizeof(float) * (8 + 5 * (size_t)kH + 7 * (size_t)N); }
int launch_cac_train(int G, float* params, float* m, float* v, int step, int N, int ld, const double* price, const float* action,
                     const double* reward, const double* nprice, float gamma, float ent, float lr,
                     const double* gamma_g, const double* ent_g, float* grad, hipStream_t s) {
    const size_t lds = cac_train_lds_bytes(N);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cac_train), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_cac_train, dim3(G), dim3(256), lds, s, G, params, m, v, step, N, ld, price, action, reward, nprice,
                       gamma, ent, lr, gamma_g, ent_g, grad);
    return (int)hipGetLastError();
}

}  // namespace thrl
