// thrl_ops.hip -- init, greedy evaluation rollout and the unfused batched
// operator forms of the reference methods (one thread per game).
#include "thrl_kernels.h"

namespace thrl {

// QTable.__init__ (agents.py:29,45) + NoisyPriceState.reset (environments.py:50-53)
// for G games from Philox: one thread per table element, Box-Muller in float64.
template <typename T>
__global__ void __launch_bounds__(256) k_init(const InitArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)a.G * a.stride;
    if (idx < total) {
        const int g = (int)(idx / a.stride);
        const int64_t jj = idx - (int64_t)g * a.stride;
        const uint64_t gid = a.game_offset + (uint64_t)g;
        int i = 0;
#pragma unroll
        for (int k = 1; k < THRL_MAXA; k++)
            if (k < a.N && jj >= a.ag[k].table_off) i = k;
        const double gamma = a.sw_gamma ? a.sw_gamma[(size_t)i * a.G + g] : a.ag[i].gamma;
        const double base = __ddiv_rn(12.5, __dsub_rn(1.0, gamma));
        const u32x4 x = draw(a.seed, gid, 0xFFFFFFFFu, (uint32_t)(jj >> 2), kStreamInitTable);
        const uint32_t xa = (jj & 2) ? x.z : x.x, xb = (jj & 2) ? x.w : x.y;
        const double u1 = ((double)xa + 0.5) * 0x1p-32;
        const double u2 = ((double)xb + 0.5) * 0x1p-32;
        const double r = sqrt(-2.0 * log(u1));
        const double ang = 6.283185307179586476925286766559 * u2;
        const double z = (jj & 1) ? r * sin(ang) : r * cos(ang);
        reinterpret_cast<T*>(a.q)[idx] = (T)(base + z);
        if (a.counter) a.counter[idx] = 0;
    }
    if (idx < a.G) {
        const uint64_t gid = a.game_offset + (uint64_t)idx;
        const u32x4 x = draw(a.seed, gid, 0xFFFFFFFFu, 0u, kStreamInitState);
        a.state[idx] = __dmul_rn(a.env_a, u01_53(x.x, x.y));
    }
}

int launch_init(const InitArgs& a, int q_dtype, hipStream_t s) {
    const int64_t total = (int64_t)a.G * a.stride;
    const int64_t n = total > a.G ? total : a.G;
    const int grid = (int)((n + 255) / 256);
    if (q_dtype == 1) hipLaunchKernelGGL(k_init<double>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_init<float>, dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

// utils.play_game (utils.py:27-47): reset, then greedy get_action (agents.py:91-92,
// float64 encode), scale, env.step; no learning.
template <typename T>
__global__ void __launch_bounds__(256) k_play_greedy(const PlayArgs a) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G) return;
    const int N = a.N;
    const uint64_t gid = a.game_offset + (uint64_t)g;
    const T* __restrict__ qg = reinterpret_cast<const T*>(a.q) + (int64_t)g * a.stride;
    for (int it = 0; it < a.iters; it++) {
        double price;
        if (a.state0) price = a.state0[(size_t)it * a.G + g];
        else {
            const u32x4 x = draw(a.seed, gid, (uint32_t)it, 0u, kStreamPlayReset);
            price = __dmul_rn(a.env.a, u01_53(x.x, x.y));
        }
        double rs[THRL_MAXA], as[THRL_MAXA];
#pragma unroll
        for (int i = 0; i < THRL_MAXA; i++) { rs[i] = 0.0; as[i] = 0.0; }
        for (int t = 0; t < a.T; t++) {
            double scaled[THRL_MAXA], rew[THRL_MAXA];
#pragma unroll
            for (int i = 0; i < THRL_MAXA; i++) {
                if (i >= N) break;
                const AgentParams& p = a.ag[i];
                const int row = encode64(price, p);
                const int aa = argmax_row(qg + p.table_off + (int64_t)row * p.n_actions, p.n_actions);
                scaled[i] = scale_action(aa, p);
            }
            double a_eff = a.env.a;
            if (a.env.noise_prob > 0.0) {
                const u32x4 xn = draw(a.seed, gid, (uint32_t)it, (uint32_t)t, kStreamNoise + 1u);
                if (u01_32(xn.x) < a.env.noise_prob)
                    a_eff = __dadd_rn(a.env.noise_lo,
                                      __dmul_rn(__dsub_rn(a.env.a, a.env.noise_lo), u01_32(xn.y)));
            }
            price = env_step<THRL_MAXA>(a.env, N, scaled, a_eff, rew);
#pragma unroll
            for (int i = 0; i < THRL_MAXA; i++) {
                if (i >= N) break;
                rs[i] = __dadd_rn(rs[i], rew[i]);
                as[i] = __dadd_rn(as[i], scaled[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < THRL_MAXA; i++) {
            if (i >= N) break;
            a.mean_reward[((size_t)it * N + i) * a.G + g] = __ddiv_rn(rs[i], (double)a.T);
            a.mean_action[((size_t)it * N + i) * a.G + g] = __ddiv_rn(as[i], (double)a.T);
        }
    }
}

int launch_play(const PlayArgs& a, int q_dtype, hipStream_t s) {
    const int grid = (a.G + 255) / 256;
    if (q_dtype == 1) hipLaunchKernelGGL(k_play_greedy<double>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_play_greedy<float>, dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

// QTable.sample_action / get_action (agents.py:80-92), batched over games
template <typename T>
__global__ void __launch_bounds__(256) k_op_sample(const OpArgs a) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G) return;
    const AgentParams& p = a.ag[a.agent];
    int aa;
    if (a.u && a.u[g] < a.eps) {
        aa = a.choice[g];
    } else {
        const int row = a.encode32 ? encode32(a.price[g], p) : encode64(a.price[g], p);
        const T* qg = reinterpret_cast<const T*>(a.q) + (int64_t)g * a.stride + p.table_off;
        aa = argmax_row(qg + (int64_t)row * p.n_actions, p.n_actions);
    }
    a.action_out[g] = aa;
}

// QTable.scale (agents.py:51-57), batched over games
__global__ void __launch_bounds__(256) k_op_scale(const OpArgs a) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G) return;
    a.scaled_out[g] = scale_action(a.action[g], a.ag[a.agent]);
}

// QTable.encode (agents.py:47-49), batched over games
__global__ void __launch_bounds__(256) k_op_encode(const OpArgs a) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G) return;
    const AgentParams& p = a.ag[a.agent];
    a.action_out[g] = a.encode32 ? encode32_fast(a.price[g], p) : encode64_fast(a.price[g], p);
}

// NoisyPriceState.step (environments.py:25-39) on scaled actions [N][G]
__global__ void __launch_bounds__(256) k_op_env_step(const OpArgs a) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G) return;
    double scaled[THRL_MAXA], rew[THRL_MAXA];
#pragma unroll
    for (int i = 0; i < THRL_MAXA; i++) {
        if (i >= a.N) break;
        scaled[i] = a.scaled[(size_t)i * a.G + g];
    }
    double a_eff = a.env.a;
    if (a.noise_u && a.noise_u[g] < a.env.noise_prob) a_eff = a.noise_a[g];
    const double p = env_step<THRL_MAXA>(a.env, a.N, scaled, a_eff, rew);
    a.price_out[g] = p;
#pragma unroll
    for (int i = 0; i < THRL_MAXA; i++) {
        if (i >= a.N) break;
        a.reward_out[(size_t)i * a.G + g] = rew[i];
    }
}

// QTable.train_net's update loop (agents.py:61-76) on n transitions per game.
// The snapshot is taken into a per-thread pass first; since old_value only
// matters when a later transition revisits (s,a), the snapshot is kept exact by
// gathering all old values before any write (two passes over the inputs; the
// first pass stores into the `reward_out` scratch the caller provides).
template <typename T>
__global__ void __launch_bounds__(256) k_op_td(const OpArgs a) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G) return;
    const AgentParams& p = a.ag[a.agent];
    T* tab = reinterpret_cast<T*>(a.q) + (int64_t)g * a.stride + p.table_off;
    int32_t* cn = a.counter ? a.counter + (int64_t)g * a.stride + p.table_off : nullptr;
    double* ov = a.reward_out;     // scratch [n][G]
    for (int k = 0; k < a.n; k++) {
        const size_t m = (size_t)k * a.G + g;
        ov[m] = (double)tab[encode64(a.price[m], p) * p.n_actions + a.action[m]];
    }
    for (int k = 0; k < a.n; k++) {
        const size_t m = (size_t)k * a.G + g;
        const int st = encode64(a.price[m], p), ns = encode64(a.next_price[m], p), ac = a.action[m];
        const T nm = max_row(tab + ns * p.n_actions, p.n_actions);
        tab[st * p.n_actions + ac] = td_value((T)ov[m], a.reward[m], nm, p);
        if (cn) cn[st * p.n_actions + ac] += 1;
    }
}

int launch_op_sample(const OpArgs& a, int q_dtype, hipStream_t s) {
    const int grid = (a.G + 255) / 256;
    if (q_dtype == 1) hipLaunchKernelGGL(k_op_sample<double>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_op_sample<float>, dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
int launch_op_env_step(const OpArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_op_env_step, dim3((a.G + 255) / 256), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
int launch_op_encode(const OpArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_op_encode, dim3((a.G + 255) / 256), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
int launch_op_scale(const OpArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_op_scale, dim3((a.G + 255) / 256), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
int launch_op_td(const OpArgs& a, int q_dtype, hipStream_t s) {
    const int grid = (a.G + 255) / 256;
    if (q_dtype == 1) hipLaunchKernelGGL(k_op_td<double>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_op_td<float>, dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace thrl
