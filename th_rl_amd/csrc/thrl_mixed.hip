// thrl_mixed.hip -- fused episodes for games whose agents are any mix of QTable and Reinforce
// (the pairing of the reference's example configs): trainer.train_one's step loop
// (trainer.py:46-70) with QTable.train_net (agents.py:59-78) at every episode end, for
// `n_episodes` per launch.  ONE WAVEFRONT PER GAME:
//   * each Reinforce network (1 -> 256 -> A) lives in registers for the whole launch
//     (thrl_policy.h), so the 23.6 KB of weights per agent-game are read from HBM once per
//     launch instead of once per step;
//   * QTable tables live in LDS for the launch (row argmax / row max are one lane per action);
//   * Philox draws are produced 64 steps at a time, one step per lane; the scaled-action tables
//     and the two log accumulators are lane-parallel, so one float64 division per step remains;
//   * Reinforce transitions are appended to that agent's HBM replay ring ([G][buf_len]: a game's slots are
//     contiguous, for the 16-step flushes here and for the update kernels); the host launches
//     k_nn_reinforce_train when an update is due and sizes n_episodes so none falls inside.
// Same Philox streams and the same arithmetic as the unfused operator loop: bit-identical.
#include "thrl_cac.h"
#include "thrl_kernels.h"

namespace thrl {
namespace {

constexpr int kMemo = 64;          // at most this many memoised policy CDFs per network (one lane of the tag register each)

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ double lane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float lane_val(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_val(double v, int l) { return lane_f64(v, l); }

template <typename T> __device__ __forceinline__ T neg_inf();
template <> __device__ __forceinline__ float neg_inf<float>() { return -INFINITY; }
template <> __device__ __forceinline__ double neg_inf<double>() { return -(double)INFINITY; }

// Per-agent state lives in LANES (lane i = agent i: epsilon, append count, this step's action /
// scaled action / reward) and is picked with v_readlane, so the large bodies -- the policy, the
// TD update -- exist once in the code, not once per agent slot.
// Draw layout: one Philox batch covers 16 steps x 4 agent pairs, lane = pair * 16 + (step & 15).
// MEMO: 0 = every step evaluates the policy; 1 = CDFs memoised in LDS by a tag search (small price grids);
//       2 = CDFs in a per-game HBM table indexed by the ACTION TUPLE of the previous step (see below).
template <typename T, int NR, int APAD, int NA, bool CAC, int MEMO>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NR == 1 ? 3 : (NR == 0 ? 4 : 2))))
k_mixed_wave(const MixedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_mx[];
    double* const sc_tab = reinterpret_cast<double*>(smem_mx);           // [N][64] scaled action of (agent, action)
    T* const lds = reinterpret_cast<T*>(smem_mx + (size_t)a.N * 64 * sizeof(double));
    float* const lds_cac = reinterpret_cast<float*>(smem_mx + a.cac_lds_byte0);     // [n_cac][THRL_CAC_PARAMS]
    // Policy memo: the networks do not change inside a launch and the price lives on a small grid
    // (noise-free games: one value per pair of actions), so the CDF the policy returns for a state
    // is kept -- kMemo entries per network, tag = the float32 state's bits, lane e of `tag` = entry e.
    float* const lds_memo = reinterpret_cast<float*>(smem_mx + a.memo_lds_byte0);   // [NR][kMemo][APAD]
    double* const st_price = reinterpret_cast<double*>(smem_mx + a.stage_lds_byte0);       // [16] state before the step
    double* const st_rew = st_price + 16;                                                   // [16][N]
    int32_t* const st_act = reinterpret_cast<int32_t*>(st_rew + 16 * a.N);                  // [16][N]
    unsigned tag0 = 0u, tag1 = 0u;
    int nmemo0 = 0, nmemo1 = 0;
    // MEMO == 2, the policy table: the networks are frozen inside a launch and in a noise-free game the state after a step
    // is a function of that step's action tuple alone, so the CDF the policy returns for it is computed ONCE per launch
    // and tuple -- by the same code on the same input, hence the same bits -- and kept in HBM scratch
    // [G][NR][tuples][APAD] (written and read by this wave only; L2 / MALL resident).  A step then costs one 96-byte load
    // instead of the ~400-instruction evaluation.  Direct-mapped: the index is known from the previous step's actions, no
    // search; lane l of `pvalid` holds the valid bits of tuples 32l .. 32l+31.  prev_tuple < 0 (first step of a launch, a
    // noisy step): the state is off the grid and the policy is evaluated as before.
    unsigned pvalid0 = 0u, pvalid1 = 0u;
    int prev_tuple = -1;
    const int g = blockIdx.x, lane = threadIdx.x;
    const int N = a.N, G = a.G, Tn = a.T;
    const uint64_t gid = a.game_offset + (uint64_t)g;
    T* __restrict__ qg = reinterpret_cast<T*>(a.q) + (int64_t)g * a.stride;
    int32_t* __restrict__ cg = a.counter ? a.counter + (int64_t)g * a.stride : nullptr;

    // ---- launch prologue: tables -> LDS, networks -> registers, scale tables, per-lane agent state
    PolicyRegs<APAD> net0, net1;
    double eps_l = 0.0;
    int cnt_l = 0, cap_l = 0;
    for (int i = 0; i < N; i++) {
        const AgentParams& p = a.ag[i];
        if (a.kind[i] == 0) {
            const int n = p.rows * p.n_actions;
            for (int e = lane; e < n; e += 64) lds[a.lds_off[i] + e] = qg[p.table_off + e];
            sc_tab[i * 64 + lane] = scale_action(lane, p);
        } else if (CAC && a.kind[i] == 3) {
            const float* w = a.nn_params[i] + (int64_t)g * kCacP;
            for (int e = lane; e < kCacP; e += 64) lds_cac[a.lds_off[i] + e] = w[e];
        } else {
            // Reinforce.scale (agents.py:153-157): action / actions * (hi - lo) + lo
            sc_tab[i * 64 + lane] = __dadd_rn(__dmul_rn(__ddiv_rn((double)lane, (double)p.n_actions), p.act_span), p.act_lo);
        }
        if (lane == i) { eps_l = a.sw_eps ? a.sw_eps[(size_t)i * a.G + g] : a.eps0[i]; cnt_l = a.count0[i]; cap_l = a.buf_len[i]; }
    }
    // per-game sweeps of the QTable agents' schedule (lane i = agent i) and of the env noise
    double eend_l = 0.0, estep_l = 0.0;
    if (lane < a.N) {
        eend_l = a.sw_eps_end ? a.sw_eps_end[(size_t)lane * a.G + g] : a.ag[lane].eps_end;
        estep_l = a.sw_eps_step ? a.sw_eps_step[(size_t)lane * a.G + g] : a.ag[lane].eps_step;
    }
    const double noise_prob_g = a.sw_noise_prob ? a.sw_noise_prob[g] : a.env.noise_prob;
    // Replay-ring state of agent i lives in lane i: its four buffer pointers and its write index (kept
    // incrementally: count % capacity).  The append is then ONE predicated block of four stores -- no per-agent
    // loop re-loading pointers from the kernel arguments (that was 8 dependent scalar-load round trips per step).
    // (Only where registers are free: with a policy network in registers the eight pointer registers spill,
    // and a spill reload waits for the appends in flight -- measured 10-18 % slower; those variants keep the loop.)
    constexpr bool kLanePointers = NR == 0;
    double* my_bp = nullptr; int32_t* my_ba = nullptr; double* my_br = nullptr; double* my_bn = nullptr;
    if (kLanePointers)
        for (int i = 0; i < N; i++)
            if (lane == i) { my_bp = a.buf_price[i]; my_ba = a.buf_action[i]; my_br = a.buf_reward[i]; my_bn = a.buf_nprice[i]; }
    int widx_l = cap_l > 0 ? cnt_l % cap_l : 0;
    if (NR >= 1) {
        const int A = a.ag[a.ragent[0]].n_actions;
        policy_load(net0, a.nn_params[a.ragent[0]] + (int64_t)g * a.nn_stride[a.ragent[0]], A, lane);
    }
    if (NR >= 2) {
        const int A = a.ag[a.ragent[1]].n_actions;
        policy_load(net1, a.nn_params[a.ragent[1]] + (int64_t)g * a.nn_stride[a.ragent[1]], A, lane);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    double price = a.state[g];
    const bool noisy = a.env.noise_prob > 0.0;
    const double Td = (double)Tn;
    // the environment's constants as register values the optimiser cannot re-load from the kernel arguments inside
    // the step loop (an s_load + s_waitcnt per use and per step otherwise)
    double env_a = a.env.a, env_b = a.env.b, env_ratio = a.env.ratio;
    asm volatile("" : "+s"(env_a), "+s"(env_b), "+s"(env_ratio));
    const int my_agent = lane & 31;

    for (int e = 0; e < a.n_episodes; e++) {
        const uint32_t eg = (uint32_t)(a.first_episode + (uint64_t)e);
        double acc = 0.0;                          // lane i: reward log of agent i ; lane 32+i: action log
        u32x4 xs = {0, 0, 0, 0}, xn = {0, 0, 0, 0};
        float z_even = 0.0f, z_odd = 0.0f;         // CAC: standard normals of this lane's (step, pair)
        for (int t = 0; t < Tn; t++) {
            const int tl = t & 15;
            if (tl == 0) {                         // draws for steps t .. t+15 of every agent pair
                xs = draw(a.seed, gid, eg, (uint32_t)(t + (lane & 15)), (uint32_t)(lane >> 4));
                if (noisy) xn = draw(a.seed, gid, eg, (uint32_t)(t + (lane & 15)), kStreamNoise);
                if (CAC) {
                    z_even = box_muller_f(u01_32(xs.x), u01_32(xs.y));
                    z_odd = box_muller_f(u01_32(xs.z), u01_32(xs.w));
                }
            }
            int act_l = 0;                         // lane i: action of agent i
            double scaled_l = 0.0;                 // lanes i and 32+i: scaled action of agent i
            // ---- QTable.sample_action (agents.py:80-89)
#pragma unroll
            for (int i = 0; i < NA; i++) {
                if (i >= N || a.kind[i] != 0) continue;
                const AgentParams& p = a.ag[i];
                const int dl = (i >> 1) * 16 + tl;
                const uint32_t xu = lane_u32((i & 1) ? xs.z : xs.x, dl), xc = lane_u32((i & 1) ? xs.w : xs.y, dl);
                int aa;
                if (u01_32(xu) < lane_f64(eps_l, i)) {
                    aa = (int)__umulhi(xc, (uint32_t)p.n_actions);
                } else {
                    const int row = encode32(price, p);
                    const T v = lane < p.n_actions ? lds[a.lds_off[i] + row * p.n_actions + lane] : neg_inf<T>();
                    const T m = wave_allmax(v);
                    aa = (int)__builtin_ctzll(__ballot(v == m && lane < p.n_actions));       // first max wins
                }
                aa = rfl(aa);
                const double sc = sc_tab[i * 64 + aa];
                if (my_agent == i) { act_l = aa; scaled_l = sc; }
            }
            // ---- Reinforce.sample_action (agents.py:159-163)
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int i = a.ragent[r];
                const int dl = (i >> 1) * 16 + tl;
                const uint32_t xu = lane_u32((i & 1) ? xs.z : xs.x, dl);
                const int A = a.ag[i].n_actions;
                const float x = (float)price;
                const unsigned key = __float_as_uint(x);
                unsigned& tag = r == 0 ? tag0 : tag1;
                int& nmemo = r == 0 ? nmemo0 : nmemo1;
                const int K = a.memo_k;                                      // entries in use (<= kMemo)
                float* memo = lds_memo + r * K * APAD;
                const unsigned long long found = MEMO == 1 ? __ballot(tag == key && lane < min(nmemo, K)) : 0ull;
                float c;
                if (MEMO == 2) {
                    unsigned& pvalid = r == 0 ? pvalid0 : pvalid1;
                    float* const row = a.policy_tab + (((size_t)g * NR + r) * (size_t)a.ptab_tuples + (size_t)max(prev_tuple, 0)) * APAD;
                    const bool known = prev_tuple >= 0 && ((lane_u32(pvalid, prev_tuple >> 5) >> (prev_tuple & 31)) & 1u);
                    if (known) {
                        // (agent scope: served by the L2, where this wave's own earlier store of the row is)
                        c = __hip_atomic_load(row + min(lane >> 1, APAD - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        c = policy_cdf(policy_probs(r == 0 ? net0 : net1, A, x, lane));
                        if (prev_tuple >= 0) {
                            if (!(lane & 1) && (lane >> 1) < APAD)
                                __hip_atomic_store(row + (lane >> 1), c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (lane == (prev_tuple >> 5)) pvalid |= 1u << (prev_tuple & 31);
                            __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the row is in the L2 before any later read
                        }
                    }
                } else if (MEMO == 1 && found) {
                    c = memo[(int)__builtin_ctzll(found) * APAD + min(lane >> 1, APAD - 1)];
                } else {
                    c = policy_cdf(policy_probs(r == 0 ? net0 : net1, A, x, lane));
                    if (MEMO == 1) {
                        const int slot = nmemo >= K ? nmemo - K : nmemo;     // round-robin replacement
                        if (!(lane & 1) && (lane >> 1) < APAD) memo[slot * APAD + (lane >> 1)] = c;
                        if (lane == slot) tag = key;
                        nmemo += 1;
                        if (nmemo >= 2 * K) nmemo -= K;
                    }
                }
                int aa = policy_pick(c, (float)u01_32(xu), A, lane);
                aa = rfl(aa);
                const double sc = sc_tab[i * 64 + aa];
                if (my_agent == i) { act_l = aa; scaled_l = sc; }
            }
            // ---- CAC.sample_action (agents.py:377-381): the action is a float in (0,1), kept as its bits
            if (CAC) {
                for (int i = 0; i < N; i++) {
                    if (a.kind[i] != 3) continue;
                    float mu, sd;
                    cac_policy(lds_cac + a.lds_off[i], (float)price, lane, mu, sd);
                    const float z = lane_val((i & 1) ? z_odd : z_even, (i >> 1) * 16 + tl);
                    const float action = sigmoid_f(mu + sd * z);
                    const double sc = __dadd_rn(__dmul_rn((double)action, a.ag[i].act_span), a.ag[i].act_lo);   // CAC.scale
                    if (my_agent == i) { act_l = __float_as_int(action); scaled_l = sc; }
                }
            }
            // ---- NoisyPriceState.step (environments.py:25-39)
            double a_eff = env_a;
            bool on_grid = true;
            if (noisy) {
                const uint32_t nx = lane_u32(xn.x, tl), ny = lane_u32(xn.y, tl);
                if (u01_32(nx) < noise_prob_g) {
                    a_eff = __dadd_rn(a.env.noise_lo, __dmul_rn(__dsub_rn(env_a, a.env.noise_lo), u01_32(ny)));
                    on_grid = false;
                }
            }
            if (MEMO == 2) {          // the tuple that produces the next state (mixed radix over the agents' action counts)
                int tup = 0;
#pragma unroll
                for (int i = 0; i < NA; i++)
                    if (i < N) tup = tup * a.ag[i].n_actions + __builtin_amdgcn_readlane(act_l, i);
                prev_tuple = on_grid ? tup : -1;
            }
            const double A_l = __dmul_rn(env_ratio, scaled_l);
            double Q = 0.0;
#pragma unroll
            for (int i = 0; i < NA; i++)
                if (i < N) Q = __dadd_rn(Q, lane_f64(A_l, i));
            double next_price = __dsub_rn(a_eff, __dmul_rn(env_b, Q));
            if (!(next_price > 0.0)) next_price = 0.0;
            const double rew_l = __dmul_rn(next_price, A_l);
            // ---- memory.append (trainer.py:62), lane i for agent i
            if (kLanePointers) {
                if (lane < N && cap_l > 0) {
                    const size_t m = (size_t)g * cap_l + widx_l;
                    my_bp[m] = price; my_ba[m] = act_l;
                    my_br[m] = rew_l; my_bn[m] = next_price;
                }
            } else {
                // Network(s) in registers: the appends are STAGED in LDS and go to the rings 16 steps at a time
                // (lane = step).  A store in flight makes every spill reload of this register-heavy variant wait
                // for HBM (loads and stores share vmcnt); staged, the step loop has none in flight.
                if (lane == 0) st_price[tl] = price;
                if (lane < N && cap_l > 0) { st_rew[tl * N + lane] = rew_l; st_act[tl * N + lane] = act_l; }
                if (tl == 15 || t == Tn - 1) {
                    const int nb = tl + 1;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int i = 0; i < N; i++) {
                        const int cap = a.buf_len[i];
                        if (cap <= 0) continue;
                        int w0 = __builtin_amdgcn_readlane(widx_l, i) - tl;        // ring slot of the batch's first step
                        while (w0 < 0) w0 += cap;
                        // (a ring shorter than the batch: only the last `cap` steps survive, as with one store per step)
                        if (lane < nb && lane >= nb - cap) {
                            int slot = w0 + lane;
                            while (slot >= cap) slot -= cap;
                            const size_t m = (size_t)g * cap + slot;      // 16 consecutive slots: coalesced
                            a.buf_price[i][m] = st_price[lane];
                            a.buf_action[i][m] = st_act[lane * N + i];
                            a.buf_reward[i][m] = st_rew[lane * N + i];
                            a.buf_nprice[i][m] = lane + 1 < nb ? st_price[lane + 1] : next_price;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (cap_l > 0) {
                cnt_l += 1;
                if (cnt_l >= 2 * cap_l) cnt_l -= cap_l;
                widx_l += 1;
                if (widx_l == cap_l) widx_l = 0;
            }
            const double val = lane < N ? rew_l : ((lane >= 32 && lane < 32 + N) ? scaled_l : 0.0);
            acc = __dadd_rn(acc, __ddiv_rn(val, Td));                   // trainer.py:65-66
            price = next_price;
        }
        // ---- [A.train_net() for A in agents]: the QTable agents (agents.py:59-78)
        for (int i = 0; i < N; i++) {
            if (a.kind[i] != 0) continue;
            const AgentParams& p = a.ag[i];
            const int cap = a.buf_len[i];
            const int cnt = __builtin_amdgcn_readlane(cnt_l, i);
            const int len = cnt < cap ? cnt : cap;
            if (cap > 0 && len >= a.min_memory[i]) {
                T* const tab = lds + a.lds_off[i];
                const TdCoef tc = (a.sw_alpha || a.sw_gamma)
                    ? td_coef(a.sw_alpha ? a.sw_alpha[(size_t)i * G + g] : p.alpha, a.sw_gamma ? a.sw_gamma[(size_t)i * G + g] : p.gamma)
                    : td_coef(p);
                const int start = cnt <= cap ? 0 : cnt % cap;
                const int A = p.n_actions;
                const double* __restrict__ bp = a.buf_price[i]; const double* __restrict__ bn = a.buf_nprice[i];
                const double* __restrict__ br = a.buf_reward[i]; const int32_t* __restrict__ ba = a.buf_action[i];
                double* __restrict__ bo = a.buf_ov[i];
                __threadfence_block();             // the appends -> visible to the other lanes
                for (int base = 0; base < len; base += 64) {            // old_value snapshot (agents.py:67)
                    const int j = base + lane;
                    if (j < len) {
                        const size_t m = (size_t)g * cap + (start + j) % cap;
                        bo[m] = (double)tab[encode64(bp[m], p) * A + ba[m]];
                    }
                }
                for (int base = 0; base < len; base += 64) {
                    const int j = base + lane;
                    int st = 0, ns = 0, ac = 0; double re = 0.0; T ov = (T)0;
                    if (j < len) {
                        const size_t m = (size_t)g * cap + (start + j) % cap;
                        st = encode64(bp[m], p); ns = encode64(bn[m], p);
                        ac = ba[m]; re = br[m]; ov = (T)bo[m];
                    }
                    const int n = min(64, len - base);
                    for (int k = 0; k < n; k++) {                        // serial: later entries see earlier writes
                        const int st_k = __builtin_amdgcn_readlane(st, k), ns_k = __builtin_amdgcn_readlane(ns, k);
                        const int ac_k = __builtin_amdgcn_readlane(ac, k);
                        const double re_k = lane_f64(re, k);
                        const T ov_k = lane_val(ov, k);
                        const T nm = wave_allmax(lane < A ? tab[ns_k * A + lane] : neg_inf<T>());
                        const T nv = td_value(ov_k, re_k, nm, tc);
                        if (lane == 0) {
                            tab[st_k * A + ac_k] = nv;
                            if (cg) atomicAdd(&cg[p.table_off + st_k * A + ac_k], 1);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (lane == i) { cnt_l = 0; widx_l = 0; }
            }
            if (lane == i) eps_l = __dadd_rn(eend_l, __dmul_rn(__dsub_rn(eps_l, eend_l), estep_l));
        }
        if (lane < N) a.game_reward_log[((size_t)e * N + lane) * G + g] = acc;
        if (lane >= 32 && lane < 32 + N) a.game_action_log[((size_t)e * N + (lane - 32)) * G + g] = acc;
    }
    // ---- epilogue: tables back to HBM
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < N; i++) {
        if (a.kind[i] != 0) continue;
        const AgentParams& p = a.ag[i];
        const int n = p.rows * p.n_actions;
        for (int e = lane; e < n; e += 64) qg[p.table_off + e] = lds[a.lds_off[i] + e];
    }
    if (lane == 0) a.state[g] = price;
    if (a.sw_eps && lane < N && a.kind[lane] == 0) a.sw_eps[(size_t)lane * G + g] = eps_l;
}

template <typename T, int NR, int APAD, int NA, bool CAC, int MEMO>
int launch_cac(const MixedArgs& a, hipStream_t s) {
    auto kern = k_mixed_wave<T, NR, APAD, NA, CAC, MEMO>;
    if (a.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           a.lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(a.G), dim3(64), (size_t)a.lds_bytes, s, a);
    return (int)hipGetLastError();
}

// the CAC code (Box-Muller, LDS-resident heads) is compiled only into the variants that need it
template <typename T, int NR, int APAD, int NA>
int launch_one(const MixedArgs& a, hipStream_t s) {
    if (a.n_cac > 0) return launch_cac<T, NR, APAD, NA, true, 0>(a, s);
    if (NR > 0 && a.memo_on) return launch_cac<T, NR, APAD, NA, false, 1>(a, s);
    if (NR > 0 && a.ptab_on) return launch_cac<T, NR, APAD, NA, false, 2>(a, s);
    return launch_cac<T, NR, APAD, NA, false, 0>(a, s);
}

template <typename T, int NA>
int launch_na(const MixedArgs& a, hipStream_t s) {
    int amax = 0;
    for (int r = 0; r < a.n_r; r++) amax = a.ag[a.ragent[r]].n_actions > amax ? a.ag[a.ragent[r]].n_actions : amax;
    if (a.n_r == 0) return launch_one<T, 0, 24, NA>(a, s);
    if (a.n_r == 1) return amax <= 24 ? launch_one<T, 1, 24, NA>(a, s) : launch_one<T, 1, 32, NA>(a, s);
    return amax <= 24 ? launch_one<T, 2, 24, NA>(a, s) : launch_one<T, 2, 32, NA>(a, s);
}
template <typename T>
int launch_t(const MixedArgs& a, hipStream_t s) {
    return a.N <= 2 ? launch_na<T, 2>(a, s) : launch_na<T, THRL_MAXA>(a, s);
}

}  // namespace

int plan_mixed(MixedArgs& a, int q_dtype, const char** why) {
    const int esz = q_dtype == 1 ? 8 : 4;
    a.n_r = 0; a.ragent[0] = a.ragent[1] = -1;
    int off = 0;
    for (int i = 0; i < a.N; i++) {
        a.lds_off[i] = 0;
        if (a.kind[i] == 3) continue;                       // CAC: weights in LDS, placed below
        if (a.kind[i] != 0) {
            if (a.n_r == 2) { *why = "more than two discrete neural agents"; return -1; }
            if (a.ag[i].n_actions > kMaxA) { *why = "neural agent with more than 32 actions"; return -1; }
            a.ragent[a.n_r++] = i;
        } else {
            if (a.ag[i].n_actions > 64) { *why = "QTable agent with more than 64 actions"; return -1; }
            a.lds_off[i] = off;
            off += (a.ag[i].rows * a.ag[i].n_actions + 3) & ~3;
        }
    }
    a.lds_bytes = a.N * 64 * 8 + off * esz;            // scale tables, then the Q-tables, then CAC networks
    a.lds_bytes = (a.lds_bytes + 15) & ~15;
    a.cac_lds_byte0 = a.lds_bytes;
    a.n_cac = 0;
    for (int i = 0; i < a.N; i++)
        if (a.kind[i] == 3) { a.lds_off[i] = a.n_cac * ((kCacP + 3) & ~3); a.n_cac++; }
    a.lds_bytes += a.n_cac * ((kCacP + 3) & ~3) * 4;
    // Policy memo only where it pays: every agent discrete, little env noise, and the whole price
    // grid (distinct float32 prices over the action product; two agents on the same grid: 41) fits
    // the entries the LDS leaves at the occupancy the registers allow anyway (12 waves/CU with one
    // network, 8 with two).  A grid that does not fit was measured: QTable vs Reinforce has 441
    // prices, a 40-entry memo hits ~35 % of the steps early in training and the lookups cost more
    // than that saves (9.8 vs 9.2 ms per episode of 65,536 games), so it stays off there.
    // staging of 16 steps of appends: price [16] f64, then per step and agent reward f64 [16][N], action i32 [16][N]
    a.stage_lds_byte0 = a.lds_bytes;
    a.lds_bytes += 16 * 8 + 16 * a.N * 12;
    a.lds_bytes = (a.lds_bytes + 15) & ~15;
    a.memo_lds_byte0 = a.lds_bytes;
    a.memo_on = 0; a.memo_k = 0;
    if (a.n_r > 0 && a.n_cac == 0 && a.env.noise_prob <= 0.1) {
        long combos = 1;
        for (int i = 0; i < a.N && combos <= 4096; i++) combos *= a.ag[i].n_actions;
        int amax = 0;
        for (int r = 0; r < a.n_r; r++) amax = a.ag[a.ragent[r]].n_actions > amax ? a.ag[a.ragent[r]].n_actions : amax;
        const int entry = (amax <= 24 ? 24 : 32) * 4 * a.n_r;                 // bytes per memo entry over all networks
        const int budget = 163840 / (a.n_r == 1 ? 12 : 8) - a.lds_bytes;
        int k = budget / entry;
        k = k > kMemo ? kMemo : (k & ~7);
        if (combos <= 4096 && k >= 16) {
            float seen[kMemo + 1];
            int n_seen = 0;
            int kk[THRL_MAXA] = {0};
            for (long c = 0; c < combos && n_seen <= k; c++) {
                double Q = 0.0;
                for (int i = 0; i < a.N; i++) {
                    const AgentParams& p = a.ag[i];
                    const double den = a.kind[i] == 0 ? p.act_den : (double)p.n_actions;   // QTable / Reinforce scale
                    Q = Q + a.env.ratio * ((double)kk[i] / den * p.act_span + p.act_lo);
                }
                double pr = a.env.a - a.env.b * Q;
                if (!(pr > 0.0)) pr = 0.0;
                const float x = (float)pr;
                int j = 0;
                while (j < n_seen && seen[j] != x) j++;
                if (j == n_seen) seen[n_seen++] = x;
                for (int i = 0; i < a.N; i++) { if (++kk[i] < a.ag[i].n_actions) break; kk[i] = 0; }
            }
            if (n_seen <= k) { a.memo_on = 1; a.memo_k = k; a.lds_bytes += k * entry; }
        }
    }
    if (a.lds_bytes > 64 * 1024) { *why = "tables and CAC networks of one game exceed 64 KiB of LDS"; return -1; }
    // Policy table in HBM (MEMO == 2) where the LDS memo does not apply: every agent discrete, at most 2,048 action tuples
    // (64 lanes x 32 valid bits); the caller provides the scratch (thrl_mixed.policy_tab), without it the policy is
    // evaluated every step as before.
    a.ptab_on = 0; a.ptab_tuples = 0;
    if (a.n_r > 0 && a.n_cac == 0 && !a.memo_on) {
        long combos = 1;
        for (int i = 0; i < a.N && combos <= 2048; i++) combos *= a.ag[i].n_actions;
        if (combos <= 2048) {
            a.ptab_tuples = (int)combos;
            int amax = 0;
            for (int r = 0; r < a.n_r; r++) amax = a.ag[a.ragent[r]].n_actions > amax ? a.ag[a.ragent[r]].n_actions : amax;
            const size_t need = (size_t)a.G * a.n_r * (size_t)combos * (amax <= 24 ? 24 : 32) * sizeof(float);
            a.ptab_need_bytes = need;
            if (a.policy_tab && a.policy_tab_bytes >= need) a.ptab_on = 1;
        }
    }
    return 0;
}

int launch_mixed(const MixedArgs& a, int q_dtype, hipStream_t s) {
    return q_dtype == 1 ? launch_t<double>(a, s) : launch_t<float>(a, s);
}

}  // namespace thrl
