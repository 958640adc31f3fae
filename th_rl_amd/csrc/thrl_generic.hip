// thrl_generic.hip -- generic episode kernel: ONE THREAD PER GAME, tables in HBM.
//
// Handles every QTable/NoisyPriceState configuration the reference accepts
// (N <= 8 agents with individual grids, noise, any min_memory/capacity, float32
// or float64 tables, injected or Philox draws).  It is the parity anchor (the
// float64 + injected-draws instantiation reproduces the reference bit for bit)
// and the fallback for configs the fused wave kernel (thrl_wave.hip) rejects.
//
// Restates trainer.train_one's loop body (th_rl/trainer.py:46-70); per step
// QTable.sample_action (agents.py:80-89), scale (:51-57), NoisyPriceState.step
// (environments.py:25-39), ReplayBuffer.append (buffers.py:18-19); per episode
// QTable.train_net (agents.py:59-78).
#include "thrl_kernels.h"

namespace thrl {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T>
__global__ void __launch_bounds__(256) k_generic_episodes(const GenericArgs a) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = g < a.G;
    const int gc = active ? g : a.G - 1;
    const int N = a.N, G = a.G, T_steps = a.T;
    const uint64_t gid = a.game_offset + (uint64_t)gc;
    T* __restrict__ qg = reinterpret_cast<T*>(a.q) + (int64_t)gc * a.stride;
    int32_t* __restrict__ cg = a.counter ? a.counter + (int64_t)gc * a.stride : nullptr;

    double eps[THRL_MAXA];
    int cnt[THRL_MAXA];
#pragma unroll
    for (int i = 0; i < THRL_MAXA; i++) {
        eps[i] = (a.sw_eps && i < N) ? a.sw_eps[(size_t)i * G + gc] : a.eps0[i];      // per-game sweeps (thrl_buffers.sweep_*)
        cnt[i] = a.cnt0[i];
    }
    const double noise_prob_g = a.sw_noise_prob ? a.sw_noise_prob[gc] : a.env.noise_prob;
    double price = a.state[gc];

    for (int e = 0; e < a.n_episodes; e++) {
        const uint32_t eg = (uint32_t)(a.first_episode + (uint64_t)e);
        double rlog[THRL_MAXA], alog[THRL_MAXA];
#pragma unroll
        for (int i = 0; i < THRL_MAXA; i++) { rlog[i] = 0.0; alog[i] = 0.0; }

        for (int t = 0; t < T_steps; t++) {
            int act[THRL_MAXA];
            double scaled[THRL_MAXA], rew[THRL_MAXA];
            u32x4 x = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < THRL_MAXA; i++) {
                if (i >= N) break;
                const AgentParams& p = a.ag[i];
                double u; int ch;
                if (a.inj_u) {
                    const size_t k = (((size_t)e * T_steps + t) * N + i) * G + gc;
                    u = a.inj_u[k]; ch = a.inj_choice[k];
                } else {
                    if ((i & 1) == 0) x = draw(a.seed, gid, eg, (uint32_t)t, (uint32_t)(i >> 1));
                    const uint32_t xu = (i & 1) ? x.z : x.x, xc = (i & 1) ? x.w : x.y;
                    u = u01_32(xu);
                    ch = (int)__umulhi(xc, (uint32_t)p.n_actions);
                }
                int aa;
                if (u < eps[i]) {
                    aa = ch;
                } else {
                    const int row = encode32(price, p);
                    aa = argmax_row(qg + p.table_off + (int64_t)row * p.n_actions, p.n_actions);
                }
                act[i] = aa;
                scaled[i] = scale_action(aa, p);
            }
            double a_eff = a.env.a;
            if (a.env.noise_prob > 0.0) {
                double nu, na;
                if (a.inj_u) {
                    const size_t k = ((size_t)e * T_steps + t) * G + gc;
                    nu = a.inj_noise_u[k]; na = a.inj_noise_a[k];
                } else {
                    const u32x4 xn = draw(a.seed, gid, eg, (uint32_t)t, kStreamNoise);
                    nu = u01_32(xn.x);
                    na = __dadd_rn(a.env.noise_lo, __dmul_rn(__dsub_rn(a.env.a, a.env.noise_lo), u01_32(xn.y)));
                }
                if (nu < noise_prob_g) a_eff = na;
            }
            const double next_price = env_step<THRL_MAXA>(a.env, N, scaled, a_eff, rew);
#pragma unroll
            for (int i = 0; i < THRL_MAXA; i++) {
                if (i >= N) break;
                const AgentParams& p = a.ag[i];
                if (p.capacity > 0 && active) {
                    const int pos = cnt[i] % p.capacity;
                    const size_t m = ((size_t)pos * N + i) * G + g;
                    a.mem.s[m] = (int16_t)encode64(price, p);
                    a.mem.ns[m] = (int16_t)encode64(next_price, p);
                    a.mem.a[m] = (int16_t)act[i];
                    a.mem.r[m] = rew[i];
                }
                if (p.capacity > 0) {
                    cnt[i] += 1;
                    if (cnt[i] >= 2 * p.capacity) cnt[i] -= p.capacity;
                }
                rlog[i] = __dadd_rn(rlog[i], __ddiv_rn(rew[i], (double)T_steps));
                alog[i] = __dadd_rn(alog[i], __ddiv_rn(scaled[i], (double)T_steps));
            }
            price = next_price;
        }

        // ---- train_net for every agent (agents.py:59-78)
#pragma unroll
        for (int i = 0; i < THRL_MAXA; i++) {
            if (i >= N) break;
            const AgentParams& p = a.ag[i];
            const int len = cnt[i] < p.capacity ? cnt[i] : p.capacity;
            if (len >= p.min_memory) {
                if (active) {
                    const int start = cnt[i] <= p.capacity ? 0 : cnt[i] % p.capacity;
                    T* __restrict__ tab = qg + p.table_off;
                    const TdCoef tc = (a.sw_alpha || a.sw_gamma)
                        ? td_coef(a.sw_alpha ? a.sw_alpha[(size_t)i * G + gc] : p.alpha, a.sw_gamma ? a.sw_gamma[(size_t)i * G + gc] : p.gamma)
                        : td_coef(p);
                    int pos = start;
                    for (int k = 0; k < len; k++) {            // old_value snapshot (:67)
                        const size_t m = ((size_t)pos * N + i) * G + g;
                        a.mem.ov[m] = (double)tab[(int)a.mem.s[m] * p.n_actions + (int)a.mem.a[m]];
                        pos = pos + 1 == p.capacity ? 0 : pos + 1;
                    }
                    pos = start;
                    for (int k = 0; k < len; k++) {            // sequential live update (:68-76)
                        const size_t m = ((size_t)pos * N + i) * G + g;
                        const int st = a.mem.s[m], ac = a.mem.a[m], ns = a.mem.ns[m];
                        const T nm = max_row(tab + ns * p.n_actions, p.n_actions);
                        tab[st * p.n_actions + ac] = td_value((T)a.mem.ov[m], a.mem.r[m], nm, tc);
                        if (cg) cg[p.table_off + st * p.n_actions + ac] += 1;
                        pos = pos + 1 == p.capacity ? 0 : pos + 1;
                    }
                }
                cnt[i] = 0;                                     // memory.empty() (:77)
            }
            const double eend = a.sw_eps_end ? a.sw_eps_end[(size_t)i * G + gc] : p.eps_end;
            const double estep = a.sw_eps_step ? a.sw_eps_step[(size_t)i * G + gc] : p.eps_step;
            eps[i] = __dadd_rn(eend, __dmul_rn(__dsub_rn(eps[i], eend), estep));   // (:78)
        }

        // ---- logs (trainer.py:65-66 rows; mean over games is an API extension)
#pragma unroll
        for (int i = 0; i < THRL_MAXA; i++) {
            if (i >= N) break;
            if (active) {
                const size_t k = ((size_t)e * N + i) * G + g;
                if (a.game_reward_log) a.game_reward_log[k] = rlog[i];
                if (a.game_action_log) a.game_action_log[k] = alog[i];
            }
            if (a.sum_reward) {
                const double sr = wave_sum(active ? rlog[i] : 0.0);
                const double sa = wave_sum(active ? alog[i] : 0.0);
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(&a.sum_reward[(size_t)e * N + i], sr);
                    atomicAdd(&a.sum_action[(size_t)e * N + i], sa);
                }
            }
        }
    }
    if (active) {
        a.state[g] = price;
        if (a.sw_eps)
            for (int i = 0; i < N; i++) a.sw_eps[(size_t)i * G + g] = eps[i];
    }
}

__global__ void k_finalize_logs(double* sr, double* sa, int n, double G) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        if (sr) sr[k] = __ddiv_rn(sr[k], G);
        if (sa) sa[k] = __ddiv_rn(sa[k], G);
    }
}

int launch_generic(const GenericArgs& a, int q_dtype, hipStream_t s) {
    const int block = 256;
    const int grid = (a.G + block - 1) / block;
    if (q_dtype == 1)
        hipLaunchKernelGGL(k_generic_episodes<double>, dim3(grid), dim3(block), 0, s, a);
    else
        hipLaunchKernelGGL(k_generic_episodes<float>, dim3(grid), dim3(block), 0, s, a);
    return (int)hipGetLastError();
}

int launch_finalize_logs(double* sum_reward, double* sum_action, int n, int G, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize_logs, dim3((n + 255) / 256), dim3(256), 0, s, sum_reward, sum_action, n,
                       (double)G);
    return (int)hipGetLastError();
}

}  // namespace thrl
