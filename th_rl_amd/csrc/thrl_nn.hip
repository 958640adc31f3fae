// thrl_nn.hip -- the reference's `Reinforce` agent (agents.py:119-220) for G games:
// one 256-thread block per game; a 1 -> 256 -> A MLP, float32, no MFMA (BASELINE
// config #4: per-game independent weights make every product a tiny GEMV).
#include <math.h>

#include "thrl_policy.h"
#include "thrl_kernels.h"

namespace thrl {

constexpr uint32_t kStreamNnInit = 0x90u;

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

// torch.nn.Linear.reset_parameters: weight ~ kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +),
// bias ~ U(-1/sqrt(fan_in), +)
__global__ void __launch_bounds__(256) k_nn_init(int G, int A, float* params, uint64_t seed,
                                                  uint64_t game_offset, int agent) {
    const int P = 2 * kH + A * kH + A;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)G * P) return;
    const int g = (int)(idx / P), j = (int)(idx - (int64_t)g * P);
    const u32x4 x = draw(seed, game_offset + (uint64_t)g, (uint32_t)agent, (uint32_t)(j >> 2), kStreamNnInit);
    const uint32_t r = (j & 3) == 0 ? x.x : ((j & 3) == 1 ? x.y : ((j & 3) == 2 ? x.z : x.w));
    const float u = (float)((double)r * 0x1p-32);                      // [0,1)
    const float bound = j < 2 * kH ? 1.0f : 1.0f / sqrtf((float)kH);   // fan_in = 1 for fc1, 256 for fc_pi
    params[idx] = (2.0f * u - 1.0f) * bound;
}

// pi() + categorical sample / argmax: one game per wavefront (thrl_policy.h), 4 games per block
template <int APAD>
__global__ void __launch_bounds__(256) k_nn_act(int G, int A, const float* __restrict__ params,
                                                 const double* __restrict__ price, const double* __restrict__ u,
                                                 int32_t* __restrict__ action_out, float* __restrict__ prob_out) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int P = 2 * kH + A * kH + A;
    PolicyRegs<APAD> r;
    policy_load(r, params + (int64_t)g * P, A, lane);
    float prob;
    const int a = policy_act(r, A, (float)price[g], u != nullptr, u ? (float)u[g] : 0.0f, lane, &prob);
    if (lane == 0) action_out[g] = a;
    if (prob_out && !(lane & 1) && (lane >> 1) < A) prob_out[(int64_t)g * A + (lane >> 1)] = prob;
}

// train_net for one game per block (agents.py:171-193).
// The n transitions are processed in chunks of kChunk: pass A (thread = transition of the chunk)
// does the forward pass and dL/dlogits into LDS, pass B (thread = hidden unit) accumulates that
// unit's column of every gradient over the chunk.  Chunking keeps LDS at ~50 KB per block, so
// three blocks (12 waves) share a CU instead of one.
constexpr int kChunk = 256;

__global__ void __launch_bounds__(256) k_nn_reinforce_train(int G, int A, float* __restrict__ params,
        float* __restrict__ adam_m, float* __restrict__ adam_v, int step, int N,
        const double* __restrict__ price, const int32_t* __restrict__ action, const double* __restrict__ reward,
        float gamma, float ent_coef, float lr, float* __restrict__ grad_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_nn[];
    float* W2 = reinterpret_cast<float*>(smem_nn);          // [A][kH]
    float* w1s = W2 + A * kH;                               // [kH]
    float* b1s = w1s + kH;                                  // [kH]
    float* b2s = b1s + kH;                                  // [kMaxA]
    float* xs = b2s + kMaxA;                                // [N]
    float* Gs = xs + N;                                     // [N]
    float* dz = Gs + N;                                     // [kChunk][A]
    float* red = dz + (size_t)kChunk * A;                   // [8]
    const int g = blockIdx.x, tid = threadIdx.x;
    const int P = 2 * kH + A * kH + A;
    float* w = params + (int64_t)g * P;

    for (int k = tid; k < A * kH; k += 256) W2[k] = w[2 * kH + k];
    w1s[tid] = w[tid]; b1s[tid] = w[kH + tid];
    if (tid < A) b2s[tid] = w[2 * kH + A * kH + tid];
    for (int n = tid; n < N; n += 256) {
        xs[n] = (float)price[(size_t)n * G + g];
        Gs[n] = (float)reward[(size_t)n * G + g];
    }
    __syncthreads();
    // discounted return: the reference's serial recurrence, last to first (agents.py:178-181)
    if (tid == 0)
        for (int n = N - 2; n >= 0; n--) Gs[n] = __fadd_rn(Gs[n], __fmul_rn(gamma, Gs[n + 1]));
    __syncthreads();
    float part = 0.0f;
    for (int n = tid; n < N; n += 256) part += Gs[n];
    const float mean = block_sum(part, red) / (float)N;
    part = 0.0f;
    for (int n = tid; n < N; n += 256) { const float d = Gs[n] - mean; part += d * d; }
    const float sd = sqrtf(block_sum(part, red) / (float)(N - 1));      // torch.std: unbiased
    for (int n = tid; n < N; n += 256) Gs[n] = (Gs[n] - mean) / sd;
    __syncthreads();

    const float invN = 1.0f / (float)N;
    float gW2[kMaxA], col[kMaxA];
#pragma unroll
    for (int k = 0; k < kMaxA; k++) { gW2[k] = 0.0f; col[k] = k < A ? W2[k * kH + tid] : 0.0f; }
    float gw1 = 0.0f, gb1 = 0.0f, gb2 = 0.0f;
    const float w1j = w1s[tid], b1j = b1s[tid];

    for (int c0 = 0; c0 < N; c0 += kChunk) {
        const int cn = min(kChunk, N - c0);
        // pass A (thread = transition): forward, d loss / d logits
        if (tid < cn) {
            const int n = c0 + tid;
            const float x = xs[n];
            float zz[kMaxA];
#pragma unroll
            for (int k = 0; k < kMaxA; k++) zz[k] = k < A ? b2s[k] : -INFINITY;
            for (int j = 0; j < kH; j++) {
                const float hj = fmaxf(__fmaf_rn(w1s[j], x, b1s[j]), 0.0f);
#pragma unroll
                for (int k = 0; k < kMaxA; k++)
                    if (k < A) zz[k] = __fmaf_rn(W2[k * kH + j], hj, zz[k]);
            }
            float m = zz[0];
#pragma unroll
            for (int k = 1; k < kMaxA; k++) if (k < A) m = fmaxf(m, zz[k]);
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < kMaxA; k++) if (k < A) { zz[k] = expf(zz[k] - m); sum += zz[k]; }
            float Hn = 0.0f;
            float lp[kMaxA];
#pragma unroll
            for (int k = 0; k < kMaxA; k++)
                if (k < A) {
                    zz[k] = zz[k] / sum;
                    lp[k] = logf(fminf(fmaxf(zz[k], 1.1920929e-07f), 1.0f - 1.1920929e-07f));
                    Hn -= zz[k] * lp[k];
                }
            const int a_n = action[(size_t)n * G + g];
            const float Gn = Gs[n];
#pragma unroll
            for (int k = 0; k < kMaxA; k++)
                if (k < A)
                    dz[tid * A + k] = (Gn * (zz[k] - (k == a_n ? 1.0f : 0.0f)) + ent_coef * zz[k] * (lp[k] + Hn)) * invN;
        }
        __syncthreads();
        // pass B (thread = hidden unit j): fc_pi.weight[:, j], fc1.weight[j], fc1.bias[j]
        for (int i = 0; i < cn; i++) {
            const float x = xs[c0 + i];
            const float pre = __fmaf_rn(w1j, x, b1j);
            const float hj = fmaxf(pre, 0.0f);
            float dh = 0.0f;
#pragma unroll
            for (int k = 0; k < kMaxA; k++)
                if (k < A) {
                    const float d = dz[i * A + k];
                    gW2[k] = __fmaf_rn(d, hj, gW2[k]);
                    dh = __fmaf_rn(col[k], d, dh);
                }
            if (pre > 0.0f) { gw1 = __fmaf_rn(dh, x, gw1); gb1 += dh; }
        }
        if (tid < A) for (int i = 0; i < cn; i++) gb2 += dz[i * A + tid];
        __syncthreads();
    }

    // clip_grad_norm_(1.0) (agents.py:192)
    float sq = gw1 * gw1 + gb1 * gb1 + (tid < A ? gb2 * gb2 : 0.0f);
#pragma unroll
    for (int k = 0; k < kMaxA; k++) if (k < A) sq += gW2[k] * gW2[k];
    const float norm = sqrtf(block_sum(sq, red));
    const float coef = fminf(1.0f, 1.0f / (norm + 1e-6f));

    // Adam (torch.optim.Adam defaults, lr from the caller)
    const float t = (float)(step + 1);
    const float bc1 = 1.0f - powf(0.9f, t), bc2s = sqrtf(1.0f - powf(0.999f, t));
    const float step_size = lr / bc1;
    float* mg = adam_m + (int64_t)g * P;
    float* vg = adam_v + (int64_t)g * P;
    auto upd = [&](int idx, float grad) {
        grad *= coef;
        if (grad_out) grad_out[(int64_t)g * P + idx] = grad;
        const float m = 0.9f * mg[idx] + 0.1f * grad;
        const float v = 0.999f * vg[idx] + 0.001f * grad * grad;
        mg[idx] = m; vg[idx] = v;
        w[idx] = w[idx] - step_size * (m / (sqrtf(v) / bc2s + 1e-8f));
    };
    upd(tid, gw1);
    upd(kH + tid, gb1);
#pragma unroll
    for (int k = 0; k < kMaxA; k++) if (k < A) upd(2 * kH + k * kH + tid, gW2[k]);
    if (tid < A) upd(2 * kH + A * kH + tid, gb2);
}

// Philox draws of one lockstep step (same counters as the episode kernels)
__global__ void __launch_bounds__(256) k_op_draws(int G, int N, uint64_t seed, uint64_t game_offset,
        uint32_t episode, uint32_t step, double env_a, double noise_lo, int32_t nA0, int32_t nA1, int32_t nA2,
        int32_t nA3, int32_t nA4, int32_t nA5, int32_t nA6, int32_t nA7, double* u_out, int8_t* choice_out,
        double* noise_u_out, double* noise_a_out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int32_t nA[8] = {nA0, nA1, nA2, nA3, nA4, nA5, nA6, nA7};
    const uint64_t gid = game_offset + (uint64_t)g;
    u32x4 x = {0, 0, 0, 0};
    for (int i = 0; i < N; i++) {
        if ((i & 1) == 0) x = draw(seed, gid, episode, step, (uint32_t)(i >> 1));
        const uint32_t xu = (i & 1) ? x.z : x.x, xc = (i & 1) ? x.w : x.y;
        u_out[(size_t)i * G + g] = u01_32(xu);
        choice_out[(size_t)i * G + g] = (int8_t)__umulhi(xc, (uint32_t)nA[i]);
    }
    if (noise_u_out) {
        const u32x4 xn = draw(seed, gid, episode, step, kStreamNoise);
        noise_u_out[g] = u01_32(xn.x);
        noise_a_out[g] = __dadd_rn(noise_lo, __dmul_rn(__dsub_rn(env_a, noise_lo), u01_32(xn.y)));
    }
}

int launch_nn_init(int G, int A, float* params, uint64_t seed, uint64_t off, int agent, hipStream_t s) {
    const int64_t n = (int64_t)G * (2 * kH + A * kH + A);
    hipLaunchKernelGGL(k_nn_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, G, A, params, seed, off, agent);
    return (int)hipGetLastError();
}
int launch_nn_act(int G, int A, const float* params, const double* price, const double* u, int32_t* act,
                  float* prob, hipStream_t s) {
    const dim3 grid((unsigned)((G + 3) / 4)), block(256);
    if (A <= 8) hipLaunchKernelGGL(k_nn_act<8>, grid, block, 0, s, G, A, params, price, u, act, prob);
    else if (A <= 24) hipLaunchKernelGGL(k_nn_act<24>, grid, block, 0, s, G, A, params, price, u, act, prob);
    else hipLaunchKernelGGL(k_nn_act<32>, grid, block, 0, s, G, A, params, price, u, act, prob);
    return (int)hipGetLastError();
}
size_t nn_train_lds_bytes(int A, int N) {
    return sizeof(float) * ((size_t)A * kH + 2 * kH + kMaxA + 2 * (size_t)N + (size_t)kChunk * A + 8);
}
int launch_nn_train(int G, int A, float* params, float* m, float* v, int step, int N, const double* price,
                    const int32_t* action, const double* reward, float gamma, float ent, float lr, float* grad,
                    hipStream_t s) {
    const size_t lds = nn_train_lds_bytes(A, N);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_reinforce_train),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_nn_reinforce_train, dim3(G), dim3(256), lds, s, G, A, params, m, v, step, N, price,
                       action, reward, gamma, ent, lr, grad);
    return (int)hipGetLastError();
}
int launch_op_draws(int G, int N, uint64_t seed, uint64_t off, uint32_t episode, uint32_t step, double env_a,
                    double noise_lo, const int32_t* nA, double* u, int8_t* ch, double* nu, double* na, hipStream_t s) {
    hipLaunchKernelGGL(k_op_draws, dim3((G + 255) / 256), dim3(256), 0, s, G, N, seed, off, episode, step, env_a,
                       noise_lo, nA[0], nA[1], nA[2], nA[3], nA[4], nA[5], nA[6], nA[7], u, ch, nu, na);
    return (int)hipGetLastError();
}

}  // namespace thrl
