// thrl_ptuple.hip -- fused episodes for TWO-agent games with discrete neural-policy agents (Reinforce / ActorCritic,
// agents.py:119-330), alone or against a QTable: the tuple-chain design of thrl_tuple_kernel.h applied to the
// reference's example pairing (QTable vs Reinforce, example_config.json) and to BASELINE configs[3] (two policies).
// Launched by thrl_mixed_episodes where it applies (plan_ptuple), otherwise k_mixed_wave (thrl_mixed.hip) runs; results are
// bit-identical to it and to the unfused operator loop: same Philox streams, same arithmetic, per-game logs accumulated
// in step order (rewards_log += reward / max_steps, trainer.py:65-66).
//
// ONE WAVEFRONT PER GAME.  Unless the intercept was redrawn (env noise: NOISE variants, the state then carries its price and
// the policies are evaluated on it directly) the state after a step is a function of that step's action pair
// tau = a0 * A1 + a1, so
//   * per block, in LDS: pid[tau] = index of tau's distinct float32 price (the policy's input, trainer.py:53), the
//     QTable agent's window-local rows per tau, per-action quantities; the float64 price per tau stays in HBM;
//   * the networks are frozen inside a launch: the CDF of Reinforce.pi (agents.py:147-152) for price id p is computed ONCE
//     per launch by the code of thrl_policy.h (same lane layout => same bits as k_nn_act / k_mixed_wave) and looked up
//     afterwards -- two policies (small price grid): both tables filled eagerly at launch start, kept in LDS; one policy
//     against a QTable (21 x 21 prices): filled on first use with the network in registers, kept in HBM scratch;
//   * play = a serial chain per step: QTable action from the per-episode composed greedy table G[tau] (as in the tuple
//     kernel), policy action = inverse CDF of the step's uniform on the looked-up row; everything else is lane-parallel
//     over the steps: Philox draws, prices, rewards, replay-ring appends (coalesced: a game's ring slots are contiguous),
//     the QTable agent's old-value snapshot; train_net of the QTable agent as in the tuple kernel;
//   * logs: reward / T and scaled / T per step lane-parallel, then summed in step order (four lanes, one quantity each).
#include "thrl_policy.h"
#include "thrl_tuple_kernel.h"

namespace thrl {
namespace {

using tup::lds_addr;
using tup::lds_load;
using tup::lds_store;
using tup::bperm;
using tup::dpp32;
using tup::Ops;

__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t wrlane(uint32_t old, uint32_t val, int lane) {
    // (both operands are wave-uniform; readfirstlane says so to the compiler where its analysis gives up)
    asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0"
        : "+v"(old) : "s"(__builtin_amdgcn_readfirstlane((int)val)), "s"(__builtin_amdgcn_readfirstlane(lane)));
    return old;
}

// T: table dtype; NR: number of policy agents (1: the other agent is a QTable; 2: none is); APAD: padded action count of the
// CDF rows; NSEG: 64-step segments per episode; TLDS: CDF tables in LDS (else HBM scratch)
template <typename QT, int NR, int APAD, int NSEG, bool TLDS, bool NOISE, bool SWEEP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NR == 1 ? 3 : 4)))
k_ptuple_episodes(const PTupleArgs a) {
    static_assert(!SWEEP || (NR == 1 && NOISE), "per-game sweeps: the QTable agent's variant, compiled with the noise path");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int T = a.T, tuples = a.tuples, npid = a.npid;
    constexpr bool HASQ = NR == 1;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.lut);
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem);
        for (int k = threadIdx.x; k < (a.lut_lds_bytes >> 2); k += blockDim.x) dst[k] = src[k];
    }
    __syncthreads();
    const unsigned short* pid = reinterpret_cast<const unsigned short*>(smem);                    // [tuples]
    const unsigned short* qrows = reinterpret_cast<const unsigned short*>(smem + a.qrows_off);   // [tuples] play | train << 8 (QTable agent)
    const float* xf = reinterpret_cast<const float*>(smem + a.xf_off);                             // [npid] float32 price of a price id
    const double* lut_aq = reinterpret_cast<const double*>(smem + a.aq_off);                       // [2][64] (a/b) * scaled
    const double* lut_sc = lut_aq + 128;                                                           // [2][64] scaled action
    const double* price_lut = reinterpret_cast<const double*>(a.lut + a.price_off);               // [tuples] HBM / L2
    const double* qsum_lut = reinterpret_cast<const double*>(a.lut + a.qsum_off);                 // [tuples] total quantity (NOISE)
    unsigned char* game = smem + a.lut_lds_bytes + (size_t)wib * a.game_lds_bytes;
    QT* const tab = reinterpret_cast<QT*>(game);
    unsigned char* const am = game + a.am_off;
    unsigned char* const gt = game + a.g_off;                                                      // [tuples + 1] greedy action of the QTable agent
    typedef unsigned __attribute__((may_alias)) hist_u32;
    hist_u32* const hist = reinterpret_cast<hist_u32*>(game + a.hist_off);
    float* const cdf_lds = reinterpret_cast<float*>(game + a.cdf_off);                             // [NR][npid + 1][APAD] (TLDS)
    double* const logsT = reinterpret_cast<double*>(game + a.logs_off);                            // [64][4]
    const int qi = a.qi;                                     // index of the QTable agent (HASQ), policy agents a.ri[0..NR-1]
    const double Td = (double)T;

    // QTable agent's replay constants (lanes 0-15 own it, as agent 0 of the tuple kernel)
    const AgentParams& pq = a.ag[HASQ ? qi : 0];
    const int Aq = pq.n_actions;
    const int l16 = lane & 15;
    const unsigned a_bytes = (unsigned)Aq * (unsigned)sizeof(QT);
    const unsigned tab_me = lds_addr(tab);
    const unsigned col_b0 = (unsigned)min(l16, Aq - 1) * (unsigned)sizeof(QT), col_b1 = (unsigned)min(l16 + 16, Aq - 1) * (unsigned)sizeof(QT);
    const unsigned col_b2 = (unsigned)min(l16 + 32, Aq - 1) * (unsigned)sizeof(QT), col_b3 = (unsigned)min(l16 + 48, Aq - 1) * (unsigned)sizeof(QT);
    const int ncol = (Aq + 15) >> 4;
    const bool storer = HASQ && lane == 0;
    const unsigned long long smask = __ballot(storer);
    const unsigned tc0 = tab_me + col_b0, tc1 = tab_me + col_b1, tc2 = tab_me + col_b2, tc3 = tab_me + col_b3;
    const TdCoef tcq = td_coef(pq);
    const QT alpha_q = std::is_same<QT, float>::value ? (QT)tcq.alpha_f : (QT)tcq.alpha;
    const QT gamma_q = (QT)pq.gamma;
    const QT ag_q = std::is_same<QT, float>::value ? (QT)tcq.alpha_gamma_f : (QT)0;

    int g_claim = 0;
    if (lane == 0) g_claim = atomicAdd(a.next_game, 1);
    for (;;) {
        const int g = __builtin_amdgcn_readfirstlane(g_claim);
        if (g >= a.G) break;
        if (lane == 0) g_claim = atomicAdd(a.next_game, 1);
        const uint64_t gid = a.game_offset + (uint64_t)g;
        const double price0 = a.state[g];
        QT* __restrict__ qg = reinterpret_cast<QT*>(a.q) + (int64_t)g * a.stride;

        // ---- QTable agent: window -> LDS, initial state -> local rows
        int init_play = 0, init_train = 0, spill_p = -1, spill_t = -1;
        if (HASQ) {
            const int W = a.win_rows, lo = a.row_lo;
            const QT* src = qg + pq.table_off + lo * Aq;
            for (int k = lane; k < W * Aq; k += 64) tab[k] = src[k];
            const int sp = __builtin_amdgcn_readfirstlane(encode32(price0, pq)), st = __builtin_amdgcn_readfirstlane(encode64(price0, pq));
            if (sp >= lo && sp < lo + W) init_play = sp - lo; else { spill_p = sp; init_play = W; }
            if (st == sp) init_train = init_play;
            else if (st >= lo && st < lo + W) init_train = st - lo;
            else { spill_t = st; init_train = W + 1; }
            if (lane < Aq) {
                tab[W * Aq + lane] = qg[pq.table_off + (spill_p >= 0 ? spill_p : 0) * Aq + lane];
                tab[(W + 1) * Aq + lane] = qg[pq.table_off + (spill_t >= 0 ? spill_t : 0) * Aq + lane];
            }
            for (int k = lane; k < a.hist_dwords; k += 64) hist[k] = 0u;
        }
        // ---- policies: CDF tables.  Entry npid = the launch's initial state (off the grid).
        PolicyRegs<APAD> net;                    // NR == 1: the network stays in registers (rows are filled on first use)
        float* const ptab = TLDS ? cdf_lds : a.policy_tab + (size_t)g * NR * (size_t)(npid + 1) * APAD;
        unsigned pvalid = 0u;                    // NR == 1: lane l = valid bits of price ids 32l .. 32l + 31
        bool store_pending = false;
        if (NR == 2) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int i = a.ri[r], A = a.ag[i].n_actions;
                policy_load(net, a.nn_params[i] + (int64_t)g * a.nn_stride[i], A, lane);
                for (int p = 0; p <= npid; p++) {
                    const float x = p < npid ? xf[p] : (float)price0;
                    const float c = policy_cdf(policy_probs(net, A, x, lane));
                    if (!(lane & 1) && (lane >> 1) < APAD) ptab[((size_t)r * (npid + 1) + p) * APAD + (lane >> 1)] = c;
                }
            }
        } else {
            const int i = a.ri[0];
            policy_load(net, a.nn_params[i] + (int64_t)g * a.nn_stride[i], a.ag[i].n_actions, lane);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();

        int tau = tuples;                        // `tuples` = the launch's initial (off-grid) state
        // NOISE: a step whose intercept was redrawn (environments.py:28-31) leaves the action grid: the state after it is "off" --
        // its price is carried, the policies are evaluated on it directly, the QTable agent reads its row by encoding it.
        bool off = false;
        double p_off = price0;
        int off_qp = 0;
        double eps_q = HASQ ? a.eps0[qi] : 0.0;
        // SWEEP: this game's own alpha / gamma / epsilon schedule of the QTable agent and noise_prob (thrl_mixed.sweep_*), derived
        // exactly as fill_agents() derives the scalars -- as k_mixed_wave does
        TdCoef tcq_g = tcq;
        QT alpha_g = alpha_q, gamma_g = gamma_q, ag_g = ag_q;
        double eend_g = pq.eps_end, estep_g = pq.eps_step, np_g = a.env.noise_prob;
        if (SWEEP) {
            const size_t k = (size_t)qi * (size_t)a.G + (size_t)g;
            if (a.sw_alpha || a.sw_gamma) {
                const double ga = a.sw_gamma ? a.sw_gamma[k] : pq.gamma;
                tcq_g = td_coef(a.sw_alpha ? a.sw_alpha[k] : pq.alpha, ga);
                alpha_g = std::is_same<QT, float>::value ? (QT)tcq_g.alpha_f : (QT)tcq_g.alpha;
                gamma_g = (QT)ga;
                ag_g = std::is_same<QT, float>::value ? (QT)tcq_g.alpha_gamma_f : (QT)0;
            }
            if (a.sw_eps) eps_q = a.sw_eps[k];
            if (a.sw_eps_end) eend_g = a.sw_eps_end[k];
            if (a.sw_eps_step) estep_g = a.sw_eps_step[k];
            if (a.sw_noise_prob) np_g = a.sw_noise_prob[g];
        }
        int cnt[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) cnt[r] = a.count0[a.ri[r]];

        for (int e = 0; e < a.n_episodes; e++) {
            const uint32_t eg = (uint32_t)(a.first_episode + (uint64_t)e);
            const int tau_in = tau;                                  // state the episode starts in
            if (HASQ) {
                // (a) greedy action of every local row, lane = row; (b) G[tau] = greedy action of the QTable agent in state tau
                const int R = a.win_rows + 2;
                for (int base = 0; base < R; base += 64) {
                    const int row = min(base + lane, R - 1);
                    const QT* r = tab + row * Aq;
                    QT b = r[0];
                    int bi = 0, j = 1;
                    for (; j + 8 <= Aq; j += 8) {
                        QT v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = r[j + u];
#pragma unroll
                        for (int u = 0; u < 8; u++) if (v[u] > b) { b = v[u]; bi = j + u; }
                    }
                    for (; j < Aq; j++) { const QT v = r[j]; if (v > b) { b = v; bi = j; } }
                    if (base + lane < R) am[row] = (unsigned char)bi;
                }
                __builtin_amdgcn_wave_barrier();
                for (int base = 0; base <= tuples; base += 256) {
                    int row[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int tq = min(base + u * 64 + lane, tuples);
                        row[u] = tq == tuples ? init_play : (int)(qrows[tq] & 0xFFu);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (base + u * 64 + lane <= tuples) gt[base + u * 64 + lane] = am[row[u]];
                }
                __builtin_amdgcn_wave_barrier();
            }

            // ---- (c) draws, lane = step (agents 0 and 1 share Philox stream 0: words x,y / z,w -- agents.py:81-82, :161)
            uint32_t Ex[NSEG], Ch[NSEG];                 // QTable agent: explores?, random choice
            float UU[NSEG][NR];                          // policy agents: the uniform of the categorical draw
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                const int tt = min(seg * 64 + lane, T - 1);
                const u32x4 x = draw(a.seed, gid, eg, (uint32_t)tt, 0u);
                Ex[seg] = 0u; Ch[seg] = 0u;
                if (HASQ) {
                    const uint32_t xu = (qi & 1) ? x.z : x.x, xc = (qi & 1) ? x.w : x.y;
                    Ex[seg] = u01_32(xu) < eps_q ? 1u : 0u;
                    Ch[seg] = __umulhi(xc, (uint32_t)Aq);
                }
#pragma unroll
                for (int r = 0; r < NR; r++) UU[seg][r] = (float)u01_32((a.ri[r] & 1) ? x.z : x.x);
            }

            // NOISE: the env's draw of every step, lane = step: nzm bit t = intercept redrawn, NA = its value
            uint64_t nzm[NSEG];
            double NA[NSEG];
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                nzm[seg] = 0ull; NA[seg] = a.env.a;
                if (NOISE) {
                    const int tt = min(seg * 64 + lane, T - 1);
                    const u32x4 xn = draw(a.seed, gid, eg, (uint32_t)tt, kStreamNoise);
                    // (a noise_prob sweep moves the probability of a noisy env: with noise_prob 0 in the config there are no draws, as
                    //  in the other kernels)
                    nzm[seg] = __ballot(seg * 64 + lane < T && a.env.noise_prob > 0.0 && u01_32(xn.x) < (SWEEP ? np_g : a.env.noise_prob));
                    NA[seg] = __dadd_rn(a.env.noise_lo, __dmul_rn(__dsub_rn(a.env.a, a.env.noise_lo), u01_32(xn.y)));
                }
            }
            const bool off_in = off;                                 // is the state the episode starts in off the grid, and its price
            const double p_in = p_off;

            // ---- (d) play: the serial chain.  seq lane t = state step t was played in, acts lane t = a0 | a1 << 8
            uint32_t seq[NSEG], acts[NSEG];
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                seq[seg] = 0u; acts[seg] = 0u;
                const int n = min(64, T - seg * 64);
                for (int tl = 0; tl < n; tl++) {
                    uint32_t act2[2] = {0u, 0u};
                    if (NOISE && off) {
                        if (HASQ) {
                            const uint32_t gq = (uint32_t)__builtin_amdgcn_readfirstlane((int)am[off_qp]);
                            act2[qi] = rdlane(Ex[seg], tl) ? rdlane(Ch[seg], tl) : gq;
                        }
                        const float x = (float)p_off;                // trainer.py:53: the policy sees the float32 state
#pragma unroll
                        for (int r = 0; r < NR; r++) {
                            const int i = a.ri[r], A = a.ag[i].n_actions;
                            if (NR == 2) policy_load(net, a.nn_params[i] + (int64_t)g * a.nn_stride[i], A, lane);   // (not resident: 5 % of the steps)
                            const float c = policy_cdf(policy_probs(net, A, x, lane));
                            const float uu = __builtin_bit_cast(float, rdlane(__builtin_bit_cast(uint32_t, UU[seg][r]), tl));
                            act2[i] = (uint32_t)__builtin_amdgcn_readfirstlane(policy_pick(c, uu, A, lane));
                        }
                    } else {
                    if (HASQ) {
                        const uint32_t gq = (uint32_t)__builtin_amdgcn_readfirstlane((int)gt[tau]);
                        act2[qi] = rdlane(Ex[seg], tl) ? rdlane(Ch[seg], tl) : gq;
                    }
                    const int p = tau == tuples ? npid : (int)__builtin_amdgcn_readfirstlane((int)pid[min(tau, tuples - 1)]);
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        const int i = a.ri[r], A = a.ag[i].n_actions;
                        float* const row = ptab + ((size_t)r * (npid + 1) + p) * APAD;
                        float c;
                        if (NR == 2) {
                            c = row[min(lane >> 1, APAD - 1)];
                        } else {
                            const bool known = p < npid && ((rdlane(pvalid, p >> 5) >> (p & 31)) & 1u);
                            if (known) {
                                // (a row stored earlier by this wave must have reached the L2 before it is read back: the wait
                                //  is taken here, where it is usually over already, instead of right behind the store)
                                if (store_pending) { __builtin_amdgcn_s_waitcnt(0x0F70); store_pending = false; }      // vmcnt(0)
                                c = __hip_atomic_load(row + min(lane >> 1, APAD - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
                                const float x = p < npid ? xf[p] : (float)price0;
                                c = policy_cdf(policy_probs(net, A, x, lane));
                                if (p < npid) {
                                    if (!(lane & 1) && (lane >> 1) < APAD)
                                        __hip_atomic_store(row + (lane >> 1), c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    if (lane == (p >> 5)) pvalid |= 1u << (p & 31);
                                    store_pending = true;
                                }
                            }
                        }
                        const float uu = __builtin_bit_cast(float, rdlane(__builtin_bit_cast(uint32_t, UU[seg][r]), tl));
                        act2[i] = (uint32_t)__builtin_amdgcn_readfirstlane(policy_pick(c, uu, A, lane));
                    }
                    }
                    seq[seg] = wrlane(seq[seg], (uint32_t)tau, tl);
                    acts[seg] = wrlane(acts[seg], act2[0] | (act2[1] << 8), tl);
                    tau = (int)(act2[0] * (uint32_t)a.ag[1].n_actions + act2[1]);
                    if (NOISE) {
                        off = __builtin_amdgcn_readfirstlane((int)((nzm[seg] >> tl) & 1ull)) != 0;
                        if (off) {                                   // environments.py:29-33 with the redrawn intercept
                            const double na = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(NA[seg]), tl),
                                                               __builtin_amdgcn_readlane(__double2loint(NA[seg]), tl));
                            double pr = __dsub_rn(na, __dmul_rn(a.env.b, qsum_lut[tau]));
                            if (!(pr > 0.0)) pr = 0.0;
                            p_off = pr;
                            if (HASQ) off_qp = __builtin_amdgcn_readfirstlane(min(max(encode32(pr, pq) - a.row_lo, 0), a.win_rows - 1));
                        }
                    }
                }
            }
            const int tau_end = tau;

            // ---- (e) lane-parallel over the steps: prices, rewards, ring appends, the QTable agent's snapshot; logs
            uint32_t wr[NSEG], wc[NSEG];           // byte offsets inside the table: next-state row, rewritten cell
            Ops<QT> ops[NSEG];
            double acc = 0.0;                            // lane k < 4: reward of agent k (k < 2), scaled action of agent k - 2
            double p_carry = 0.0;                        // NOISE: the price after the previous segment's last step
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                const int tt = seg * 64 + lane;
                const bool valid = tt < T;
                const int tq = valid ? (int)seq[seg] : tau_end;
                int nxt = __shfl_down((int)seq[seg], 1, 64);
                if (seg + 1 < NSEG) { if (lane == 63) nxt = __builtin_amdgcn_readlane((int)seq[seg + 1 < NSEG ? seg + 1 : seg], 0); }
                if (tt + 1 >= T) nxt = tau_end;
                double p_next = price_lut[nxt];
                double p_prev = tq == tuples ? price0 : price_lut[min(tq, tuples - 1)];
                bool s_off = false, n_off = false;                   // NOISE: my state / the state after my step is off the grid
                if (NOISE) {
                    n_off = valid && ((nzm[seg] >> lane) & 1ull);
                    if (n_off) {
                        p_next = __dsub_rn(NA[seg], __dmul_rn(a.env.b, qsum_lut[nxt]));
                        if (!(p_next > 0.0)) p_next = 0.0;
                    }
                    s_off = lane > 0 ? (bool)((nzm[seg] >> (lane - 1)) & 1ull) : (seg > 0 ? (bool)((nzm[seg > 0 ? seg - 1 : 0] >> 63) & 1ull) : off_in);
                    double pp = __shfl_up(p_next, 1, 64);
                    if (lane == 0) pp = seg > 0 ? p_carry : p_in;
                    p_carry = __shfl(p_next, 63, 64);
                    if (s_off && valid) p_prev = pp;
                    if (!valid) s_off = false;
                }
                double rew[2], sca[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const uint32_t act = valid ? ((acts[seg] >> (8 * i)) & 0xFFu) : 0u;
                    rew[i] = __dmul_rn(p_next, lut_aq[i * 64 + act]);
                    sca[i] = lut_sc[i * 64 + act];
                }
#pragma unroll
                for (int r = 0; r < NR; r++) {           // memory.append of the policy agents (trainer.py:62)
                    const int i = a.ri[r], cap = a.buf_len[i];
                    if (valid && cap > 0) {
                        int slot = (cnt[r] + tt) % cap;
                        const size_t m = (size_t)g * cap + slot;
                        a.buf_price[i][m] = p_prev; a.buf_action[i][m] = (int32_t)((acts[seg] >> (8 * i)) & 0xFFu);
                        a.buf_reward[i][m] = rew[i]; a.buf_nprice[i][m] = p_next;
                    }
                }
                if (HASQ) {
                    const uint32_t act = valid ? ((acts[seg] >> (8 * qi)) & 0xFFu) : 0u;
                    uint32_t srow = tq == tuples ? (uint32_t)init_train : (uint32_t)(qrows[min(tq, tuples - 1)] >> 8);
                    uint32_t ns = (uint32_t)(qrows[nxt] >> 8);
                    if (NOISE) {                                     // the train rows of off-grid states: float64 encode of their price
                        if (s_off) srow = (uint32_t)min(max(encode64(p_prev, pq) - a.row_lo, 0), a.win_rows - 1);
                        if (n_off) ns = (uint32_t)min(max(encode64(p_next, pq) - a.row_lo, 0), a.win_rows - 1);
                    }
                    const uint32_t cell = valid ? srow * (uint32_t)Aq + act : 0u;
                    ops[seg].set(tab[cell], rew[qi], SWEEP ? tcq_g : tcq);
                    wr[seg] = ns * a_bytes;
                    wc[seg] = cell * (uint32_t)sizeof(QT);
                    if (valid && a.counter)
                        __hip_atomic_fetch_add(&hist[cell >> 1], 1u << ((cell & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                // logs: reward / T and scaled / T of this segment's steps (trainer.py:65-66), summed in step order below
                if (valid) {
                    logsT[lane * 4 + 0] = __ddiv_rn(rew[0], Td); logsT[lane * 4 + 1] = __ddiv_rn(rew[1], Td);
                    logsT[lane * 4 + 2] = __ddiv_rn(sca[0], Td); logsT[lane * 4 + 3] = __ddiv_rn(sca[1], Td);
                }
                __builtin_amdgcn_wave_barrier();
                {
                    const int n = min(64, T - seg * 64);
                    for (int t = 0; t < n; t++) acc = __dadd_rn(acc, logsT[t * 4 + (lane & 3)]);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (lane < 2) a.game_reward_log[((size_t)e * 2 + lane) * a.G + g] = acc;
            else if (lane < 4) a.game_action_log[((size_t)e * 2 + (lane - 2)) * a.G + g] = acc;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int cap = a.buf_len[a.ri[r]];
                if (cap > 0) { cnt[r] += T; while (cnt[r] >= 2 * cap) cnt[r] -= cap; }
            }

            // ---- (f) train_net of the QTable agent (agents.py:59-78): lanes 0-15 are its row
            if (HASQ) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int seg = 0; seg < NSEG; seg++) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int base_t = seg * 64 + b * 16;
                        if (base_t >= T) break;
                        const unsigned sel = (unsigned)(b * 16 + l16) << 2;
                        const uint32_t xr = bperm(sel, wr[seg]), xc = bperm(sel, wc[seg]);
                        Ops<QT> xo = ops[seg];
                        xo.gather(sel, ops[seg], true);
                        const int nb = min(16, T - base_t);
                        const QT agx = SWEEP ? ag_g : ag_q, alx = SWEEP ? alpha_g : alpha_q, gax = SWEEP ? gamma_g : gamma_q;
#define THRL_PT_BLOCK(NC)                                                                                                          \
                        if (nb == 16) tup::replay_block<QT, NC, true>(nb, xr, xc, xo, tab_me, tc0, tc1, tc2, tc3, storer, smask, agx, alx, gax); \
                        else tup::replay_block<QT, NC, false>(nb, xr, xc, xo, tab_me, tc0, tc1, tc2, tc3, storer, smask, agx, alx, gax);
                        if (ncol == 1) { THRL_PT_BLOCK(1) } else if (ncol == 2) { THRL_PT_BLOCK(2) } else { THRL_PT_BLOCK(4) }
#undef THRL_PT_BLOCK
                    }
                }
                eps_q = SWEEP ? __dadd_rn(eend_g, __dmul_rn(__dsub_rn(eps_q, eend_g), estep_g))
                              : __dadd_rn(pq.eps_end, __dmul_rn(__dsub_rn(eps_q, pq.eps_end), pq.eps_step));       // agents.py:78
            }
            (void)tau_in;
        }

        // ---- epilogue: table and counters back, env state
        if (HASQ) {
            const int W = a.win_rows, lo = a.row_lo;
            QT* dst = qg + pq.table_off + lo * Aq;
            for (int k = lane; k < W * Aq; k += 64) dst[k] = tab[k];
            if (lane < Aq) {
                if (spill_p >= 0) qg[pq.table_off + spill_p * Aq + lane] = tab[W * Aq + lane];
                if (spill_t >= 0) qg[pq.table_off + spill_t * Aq + lane] = tab[(W + 1) * Aq + lane];
            }
            if (a.counter) {
                int32_t* cg = a.counter + (int64_t)g * a.stride + pq.table_off;
                for (int k = lane; k < W * Aq; k += 64) {
                    const unsigned n = (hist[k >> 1] >> ((k & 1) << 4)) & 0xFFFFu;
                    if (n) cg[lo * Aq + k] += (int32_t)n;
                }
                if (lane < 2 * Aq) {
                    const int which = lane >= Aq, col = lane - which * Aq, k = (W + which) * Aq + col;
                    const int grow = which ? spill_t : spill_p;
                    const unsigned n = (hist[k >> 1] >> ((k & 1) << 4)) & 0xFFFFu;
                    if (grow >= 0 && n) cg[grow * Aq + col] += (int32_t)n;
                }
            }
        }
        if (lane == 0 && a.n_episodes > 0) a.state[g] = (NOISE && off) ? p_off : price_lut[tau];
        if (SWEEP && HASQ && a.sw_eps && lane == 0) a.sw_eps[(size_t)qi * (size_t)a.G + (size_t)g] = eps_q;
        __builtin_amdgcn_wave_barrier();
    }
}

// LUT image: pid u16 [tuples] | qrows u16 [tuples] | xf f32 [npid] | aq f64 [2][64] | sc f64 [2][64] || price f64 [tuples] | qsum f64 [tuples]
// ONE block of 1,024 threads.  Price ids: pid[tau] = rank, in order of first occurrence, of tau's float32 price among the
// distinct float32 prices of the grid (the host runs the same enumeration for the count: plan_ptuple).
__global__ void __launch_bounds__(1024) k_ptuple_lut(const PTupleArgs a, unsigned char* out) {
    __shared__ unsigned xb[kTupMaxTuples];          // float32 price bits per tuple
    __shared__ unsigned short canon[kTupMaxTuples]; // first tuple with the same price
    __shared__ unsigned char first[kTupMaxTuples];
    auto scaled = [&](int i, int k) {
        const AgentParams& p = a.ag[i];
        return a.kind[i] == 0 ? scale_action(k, p)
                              : __dadd_rn(__dmul_rn(__ddiv_rn((double)k, (double)p.n_actions), p.act_span), p.act_lo);   // Reinforce.scale (agents.py:153-157)
    };
    for (int idx = threadIdx.x; idx < a.tuples; idx += blockDim.x) {
        const int a0 = idx / a.ag[1].n_actions, a1 = idx - a0 * a.ag[1].n_actions;
        double sc[2] = {scaled(0, a0), scaled(1, a1)}, rew[2];
        const double price = env_step<2>(a.env, 2, sc, a.env.a, rew);
        reinterpret_cast<double*>(out + a.price_off)[idx] = price;
        reinterpret_cast<double*>(out + a.qsum_off)[idx] = __dadd_rn(__dadd_rn(0.0, __dmul_rn(a.env.ratio, sc[0])), __dmul_rn(a.env.ratio, sc[1]));   // as env_step sums it
        xb[idx] = __float_as_uint((float)price);
        unsigned short qr = 0;
        if (a.qi >= 0) {
            const AgentParams& p = a.ag[a.qi];
            const int rp = min(max(encode32(price, p) - a.row_lo, 0), a.win_rows - 1);
            const int rt = min(max(encode64(price, p) - a.row_lo, 0), a.win_rows - 1);
            qr = (unsigned short)(rp | (rt << 8));
        }
        reinterpret_cast<unsigned short*>(out + a.qrows_off)[idx] = qr;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < a.tuples; idx += blockDim.x) {
        int c = idx;
        for (int s = 0; s < idx; s++) if (xb[s] == xb[idx]) { c = s; break; }
        canon[idx] = (unsigned short)c; first[idx] = c == idx;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < a.tuples; idx += blockDim.x) {
        const int c = canon[idx];
        int rank = 0;
        for (int s = 0; s < c; s++) rank += first[s];
        reinterpret_cast<unsigned short*>(out)[idx] = (unsigned short)min(rank, a.npid - 1);
        if (c == idx && rank < a.npid) reinterpret_cast<float*>(out + a.xf_off)[rank] = __uint_as_float(xb[idx]);
    }
    const int idx = threadIdx.x;
    if (idx < 128) {
        const int i = idx >> 6, k = min(idx & 63, a.ag[i].n_actions - 1);
        const double sc = scaled(i, k);
        reinterpret_cast<double*>(out + a.aq_off)[idx] = __dmul_rn(a.env.ratio, sc);
        reinterpret_cast<double*>(out + a.aq_off)[128 + idx] = sc;
    }
}

template <typename QT, int NR, int APAD, bool TLDS, bool NOISE, bool SWEEP = false>
int launch_seg(const PTupleArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    const int nseg = (a.T + 63) / 64;
#define THRL_PT_LAUNCH(NS)                                                                                           \
    {                                                                                                                \
        auto kern = k_ptuple_episodes<QT, NR, APAD, NS, TLDS, NOISE, SWEEP>;                                         \
        if (lds > 64 * 1024) {                                                                                       \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                            \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
            if (e != hipSuccess) return (int)e;                                                                      \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, a);                                                \
        return (int)hipGetLastError();                                                                               \
    }
    if (nseg <= 1) THRL_PT_LAUNCH(1)
    if (nseg == 2) THRL_PT_LAUNCH(2)
    THRL_PT_LAUNCH(4)
#undef THRL_PT_LAUNCH
}

template <typename QT, bool NOISE>
int launch_n(const PTupleArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    int amax = 0;
    for (int r = 0; r < a.n_r; r++) amax = a.ag[a.ri[r]].n_actions > amax ? a.ag[a.ri[r]].n_actions : amax;
    if (a.n_r == 2) return amax <= 24 ? launch_seg<QT, 2, 24, true, NOISE>(a, grid, block, lds, s) : launch_seg<QT, 2, 32, true, NOISE>(a, grid, block, lds, s);
    return amax <= 24 ? launch_seg<QT, 1, 24, false, NOISE>(a, grid, block, lds, s) : launch_seg<QT, 1, 32, false, NOISE>(a, grid, block, lds, s);
}
template <typename QT>
int launch_t(const PTupleArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    if (a.sweep) {                          // (plan_ptuple: only with a QTable agent in the game)
        int amax = a.ag[a.ri[0]].n_actions;
        return amax <= 24 ? launch_seg<QT, 1, 24, false, true, true>(a, grid, block, lds, s) : launch_seg<QT, 1, 32, false, true, true>(a, grid, block, lds, s);
    }
    return a.env.noise_prob > 0.0 ? launch_n<QT, true>(a, grid, block, lds, s) : launch_n<QT, false>(a, grid, block, lds, s);
}

}  // namespace

int launch_ptuple_lut(const PTupleArgs& a, unsigned char* out, hipStream_t s) {
    hipLaunchKernelGGL(k_ptuple_lut, dim3(1), dim3(1024), 0, s, a, out);
    return (int)hipGetLastError();
}

int launch_ptuple(const PTupleArgs& a, int q_dtype, int grid, int block, size_t lds, hipStream_t s) {
    return q_dtype == 1 ? launch_t<double>(a, grid, block, lds, s) : launch_t<float>(a, grid, block, lds, s);
}

}  // namespace thrl
