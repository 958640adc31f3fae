/*
 * thrl_oracle.c -- CPU restatement of the reference's iterated-pricing-game hot
 * path.  TEST INFRASTRUCTURE ONLY: it may be called from tests/, from
 * __graft_entry__.smoke() and from bench.py's cpu_baseline leg, never from the
 * product (th_rl_amd/), which must fail loudly without its HIP library.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file
 * bit-for-bit (float64 mode, injected draws) against fixtures produced by
 * running the reference's own train_one (tests/golden/make_golden.py).
 *
 * Each function cites the reference code (paths relative to /root/reference)
 * whose arithmetic and operation ORDER it restates.  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC
 * (no FMA contraction: numpy/CPython round every multiply and add separately).
 *
 * Two table precisions: q_dtype 1 = float64 (the reference's), q_dtype 0 =
 * float32 tables with the TD arithmetic in float32 (the GPU performance path;
 * the GPU kernels must match THIS mode bit-for-bit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/thrl.h"

/* ------------------------------------------------------------------ Philox */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 constants). */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

/* Counter layout shared with the HIP kernels (DESIGN.md "RNG"):
 *   ctr = (step, episode, game_lo, (game_hi & 0xFFFFFF) | stream << 24), key = seed */
#define STREAM_AGENT_PAIR(p) ((uint32_t)(p))      /* agents 2p, 2p+1 */
#define STREAM_NOISE 0x40u
#define STREAM_INIT_TABLE 0x80u
#define STREAM_INIT_STATE 0x81u
#define STREAM_PLAY_RESET 0x82u

static inline void draw(uint64_t seed, uint64_t game, uint32_t episode, uint32_t step,
                        uint32_t stream, uint32_t out[4]) {
    uint32_t ctr[4] = {step, episode, (uint32_t)game,
                       (uint32_t)((game >> 32) & 0xFFFFFFu) | (stream << 24)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    oracle_philox4x32_10(ctr, key, out);
}

static inline double u01_32(uint32_t x) { return (double)x * 0x1p-32; }
static inline double u01_53(uint32_t hi, uint32_t lo) {
    /* same construction as numpy's legacy random_sample: 27 + 26 bits */
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * 0x1p-53;
}

/* ------------------------------------------------------------------ layout */
static int rows_of(const thrl_cfg* c, int i) { return c->n_states[i] + 1; }

size_t oracle_table_stride(const thrl_cfg* c) {
    size_t s = 0;
    for (int i = 0; i < c->n_agents; i++) s += (size_t)rows_of(c, i) * c->n_actions[i];
    return s;
}
size_t oracle_table_offset(const thrl_cfg* c, int agent) {
    size_t s = 0;
    for (int i = 0; i < agent; i++) s += (size_t)rows_of(c, i) * c->n_actions[i];
    return s;
}

/* ------------------------------------------------------------------ encode */
/* QTable.encode on a float64 array (th_rl/agents.py:47-49 as called from
 * train_net, agents.py:62,66): numpy.round == round-half-to-even == rint. */
int64_t oracle_encode64(double price, double max_state, int states) {
    double x = price / max_state;
    x = x * (double)states;
    return (int64_t)rint(x);
}
/* QTable.encode as called from sample_action (agents.py:88): the trainer has
 * already cast the state to float32 (th_rl/trainer.py:53); with numpy >= 2 the
 * Python scalars max_state / states are weak, so the arithmetic stays float32. */
int64_t oracle_encode32(double price, double max_state, int states) {
    float x = (float)price;
    x = x / (float)max_state;
    x = x * (float)states;
    return (int64_t)rintf(x);
}

/* QTable.scale (th_rl/agents.py:51-57): actions/(A-1.0)*(hi-lo)+lo in float64 */
double oracle_scale(int action, int n_actions, double lo, double hi) {
    double x = (double)action / ((double)n_actions - 1.0);
    x = x * (hi - lo);
    return x + lo;
}

/* first-max argmax, numpy.argmax semantics (agents.py:88,92) */
static int argmax_f64(const double* row, int n) {
    int b = 0;
    for (int k = 1; k < n; k++) if (row[k] > row[b]) b = k;
    return b;
}
static int argmax_f32(const float* row, int n) {
    int b = 0;
    for (int k = 1; k < n; k++) if (row[k] > row[b]) b = k;
    return b;
}

/* ------------------------------------------------------------------ env step */
/* NoisyPriceState.step (th_rl/environments.py:25-39) + scale_actions (:22-23).
 * scaled[i] are the already-scaled actions; noisy != 0 means the uniform(0,1)
 * draw was < noise_prob and new_a is the uniform(0.7a, a) draw. */
void oracle_env_step(const thrl_cfg* c, const double* scaled, int noisy, double new_a,
                     double* price_out, double* rewards_out) {
    double A[THRL_MAXA];
    double ratio = c->env_a / c->env_b;            /* self.a/self.b */
    double Q = 0.0;
    for (int i = 0; i < c->n_agents; i++) {
        A[i] = ratio * scaled[i];
        Q = Q + A[i];                              /* sum(): left to right from 0 */
    }
    double a_eff = noisy ? new_a : c->env_a;
    double p = a_eff - c->env_b * Q;
    if (!(p > 0.0)) p = 0.0;                       /* numpy.max([0, p]) */
    for (int i = 0; i < c->n_agents; i++) rewards_out[i] = p * A[i];
    *price_out = p;
}

/* NoisyPriceState.get_optimal (environments.py:41-48) */
void oracle_get_optimal(const thrl_cfg* c, double* nash, double* cartel) {
    int n = c->n_agents;
    double ratio = c->env_a / c->env_b;
    double an = ratio * 1.0 / (double)(n + 1), sum = 0.0;
    for (int i = 0; i < n; i++) sum += an;
    double p = c->env_a - c->env_b * sum; if (!(p > 0)) p = 0;
    double r = 0; for (int i = 0; i < n; i++) r += p * an;
    *nash = r;
    double ac = ratio * 0.5 * 1.0 / (double)n; sum = 0.0;
    for (int i = 0; i < n; i++) sum += ac;
    p = c->env_a - c->env_b * sum; if (!(p > 0)) p = 0;
    r = 0; for (int i = 0; i < n; i++) r += p * ac;
    *cartel = r;
}

/* ------------------------------------------------------------------ TD update */
/* The body of QTable.train_net once len(memory) >= min_memory
 * (th_rl/agents.py:61-76): old_value is a SNAPSHOT gathered before the loop
 * (:67), next_max reads the LIVE table (:71), writes are sequential (:75-76). */
void oracle_td_update_f64(double* table, int32_t* counter, int n_actions, int n,
                          const int32_t* st, const int32_t* ac, const double* rw,
                          const int32_t* ns, double alpha, double gamma) {
    double* ov = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int k = 0; k < n; k++) ov[k] = table[(size_t)st[k] * n_actions + ac[k]];
    for (int k = 0; k < n; k++) {
        const double* row = table + (size_t)ns[k] * n_actions;
        double nm = row[0];
        for (int j = 1; j < n_actions; j++) if (row[j] > nm) nm = row[j];
        double t4 = (1.0 - alpha) * ov[k];
        double t1 = gamma * nm;
        double t2 = rw[k] + t1;
        double t3 = alpha * t2;
        table[(size_t)st[k] * n_actions + ac[k]] = t4 + t3;
        if (counter) counter[(size_t)st[k] * n_actions + ac[k]] += 1;
    }
    free(ov);
}
/* float32-table variant (the GPU performance path; the reference itself is float64).  The target
 * (1-alpha)*ov + alpha*(r + gamma*next_max) is evaluated as
 *     t4 = c1*ov ;  b = fma(af, r, t4) ;  new = fma(af*gf, next_max, b)
 * with c1 = (float)(1-alpha), af = (float)alpha, gf = (float)gamma, af*gf one rounded float product:
 * only ONE operation depends on the live next_max, which is what the serial replay chain of the
 * wave kernel executes per step.  Constants are converted once. */
void oracle_td_update_f32(float* table, int32_t* counter, int n_actions, int n,
                          const int32_t* st, const int32_t* ac, const double* rw,
                          const int32_t* ns, double alpha, double gamma) {
    float* ov = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    const float c1 = (float)(1.0 - alpha), af = (float)alpha, gf = (float)gamma;
    const float agf = af * gf;
    for (int k = 0; k < n; k++) ov[k] = table[(size_t)st[k] * n_actions + ac[k]];
    for (int k = 0; k < n; k++) {
        const float* row = table + (size_t)ns[k] * n_actions;
        float nm = row[0];
        for (int j = 1; j < n_actions; j++) if (row[j] > nm) nm = row[j];
        float t4 = c1 * ov[k];
        float b = fmaf(af, (float)rw[k], t4);
        table[(size_t)st[k] * n_actions + ac[k]] = fmaf(agf, nm, b);
        if (counter) counter[(size_t)st[k] * n_actions + ac[k]] += 1;
    }
    free(ov);
}

/* epsilon decay, every train_net call (agents.py:78) */
double oracle_eps_decay(double eps, double eps_end, double eps_step) {
    return eps_end + (eps - eps_end) * eps_step;
}

/* ------------------------------------------------------------------ init */
/* QTable.__init__ table (agents.py:29: 12.5/(1-gamma) + randn) and
 * NoisyPriceState.sample_state (environments.py:15-16: uniform(0, a)), drawn
 * from Philox instead of numpy's global MT19937 (the reference is unseeded, so
 * only the distribution is pinned).  Box-Muller in float64. */
void oracle_init(const thrl_cfg* c, void* q, int32_t* counter, double* state,
                 uint64_t seed, uint64_t game_offset) {
    size_t stride = oracle_table_stride(c);
    const double two_pi = 6.283185307179586476925286766559;
    for (int g = 0; g < c->n_games; g++) {
        uint64_t gid = game_offset + (uint64_t)g;
        for (int i = 0; i < c->n_agents; i++) {
            size_t off = oracle_table_offset(c, i);
            size_t cnt = (size_t)rows_of(c, i) * c->n_actions[i];
            double base = 12.5 / (1.0 - c->gamma[i]);
            for (size_t j = 0; j < cnt; j++) {
                size_t jj = off + j;
                uint32_t x[4];
                draw(seed, gid, 0xFFFFFFFFu, (uint32_t)(jj >> 2), STREAM_INIT_TABLE, x);
                uint32_t xa = x[(jj & 2)], xb = x[(jj & 2) + 1];
                double u1 = ((double)xa + 0.5) * 0x1p-32;
                double u2 = ((double)xb + 0.5) * 0x1p-32;
                double r = sqrt(-2.0 * log(u1));
                double z = (jj & 1) ? r * sin(two_pi * u2) : r * cos(two_pi * u2);
                double v = base + z;
                if (c->q_dtype == 1) ((double*)q)[(size_t)g * stride + jj] = v;
                else ((float*)q)[(size_t)g * stride + jj] = (float)v;
                if (counter) counter[(size_t)g * stride + jj] = 0;
            }
        }
        uint32_t x[4];
        draw(seed, gid, 0xFFFFFFFFu, 0, STREAM_INIT_STATE, x);
        state[g] = c->env_a * u01_53(x[0], x[1]);
    }
}

/* ------------------------------------------------------------------ episodes */
typedef struct {
    int32_t* s;    /* [G][N][capmax] encode64(state)      */
    int32_t* a;    /* action index                        */
    int32_t* ns;   /* encode64(next_state)                */
    double*  r;    /* reward                              */
    int32_t  capmax;
} oracle_mem;

/*
 * trainer.train_one's loop (th_rl/trainer.py:46-70) for G independent games.
 * Draw sources: injected (inj_u != NULL; arrays [E][T][N][G] / [E][T][G]) or
 * Philox.  mem_count[i] = appends since the last memory.empty() (uniform over
 * games).  Logs: game_*_log [E][N][G] rows exactly as rewards_log/actions_log
 * (trainer.py:65-66); mean_*_log [E][N] = mean over games (sequential sum / G).
 */
int oracle_episodes(const thrl_cfg* c, void* q, int32_t* counter, double* state,
                    double* eps, int32_t* mem_count,
                    int32_t* mem_s, int32_t* mem_a, int32_t* mem_ns, double* mem_r, int32_t capmax,
                    uint64_t seed, uint64_t game_offset, uint64_t first_episode, int32_t n_episodes,
                    const double* inj_u, const int8_t* inj_choice,
                    const double* inj_noise_u, const double* inj_noise_a,
                    double* game_reward_log, double* game_action_log,
                    double* mean_reward_log, double* mean_action_log,
                    int32_t* trace_actions /* [E][T][N][G] or NULL */,
                    double* trace_price /* [E][T][G] or NULL */,
                    /* per-game sweeps, [N][G] (noise_prob [G]); NULL = the config's scalar */
                    const double* sw_gamma, const double* sw_alpha, const double* sw_eps_end,
                    const double* sw_eps_step, double* sw_eps, const double* sw_noise_prob) {
    const int G = c->n_games, N = c->n_agents, T = c->max_steps;
    const size_t stride = oracle_table_stride(c);
    size_t off[THRL_MAXA];
    for (int i = 0; i < N; i++) {
        off[i] = oracle_table_offset(c, i);
        if (c->capacity[i] > capmax) return -1;
        if (c->n_actions[i] < 2) return -1;
    }
    if (mean_reward_log) memset(mean_reward_log, 0, sizeof(double) * (size_t)n_episodes * N);
    if (mean_action_log) memset(mean_action_log, 0, sizeof(double) * (size_t)n_episodes * N);
    int32_t* tmp_s = (int32_t*)malloc(sizeof(int32_t) * (size_t)(capmax + 1) * 3);
    double* tmp_r = (double*)malloc(sizeof(double) * (size_t)(capmax + 1));
    int32_t *tmp_a = tmp_s + (capmax + 1), *tmp_ns = tmp_s + 2 * (capmax + 1);

    int32_t count_after[THRL_MAXA];
    for (int g = 0; g < G; g++) {
        const uint64_t gid = game_offset + (uint64_t)g;
        double eps_g[THRL_MAXA];
        int32_t cnt[THRL_MAXA];
        for (int i = 0; i < N; i++) { eps_g[i] = sw_eps ? sw_eps[(size_t)i * G + g] : eps[i]; cnt[i] = mem_count[i]; }
        const double noise_prob_g = sw_noise_prob ? sw_noise_prob[g] : c->noise_prob;
        double price = state[g];
        for (int e = 0; e < n_episodes; e++) {
            const uint32_t eg = (uint32_t)(first_episode + (uint64_t)e);
            double rlog[THRL_MAXA], alog[THRL_MAXA];
            for (int i = 0; i < N; i++) { rlog[i] = 0.0; alog[i] = 0.0; }
            for (int t = 0; t < T; t++) {
                int act[THRL_MAXA];
                double scaled[THRL_MAXA], rew[THRL_MAXA];
                /* --- QTable.sample_action per agent, in agent order (trainer.py:52-55) */
                for (int i = 0; i < N; i++) {
                    double u; int ch;
                    if (inj_u) {
                        size_t k = (((size_t)e * T + t) * N + i) * G + g;
                        u = inj_u[k]; ch = inj_choice[k];
                    } else {
                        uint32_t x[4];
                        draw(seed, gid, eg, (uint32_t)t, STREAM_AGENT_PAIR(i >> 1), x);
                        u = u01_32(x[(i & 1) * 2]);
                        ch = (int)(((uint64_t)x[(i & 1) * 2 + 1] * (uint64_t)c->n_actions[i]) >> 32);
                    }
                    if (u < eps_g[i]) {
                        act[i] = ch;                               /* agents.py:81-82 */
                    } else {
                        int64_t row = oracle_encode32(price, c->max_state[i], c->n_states[i]);
                        if (row < 0 || row > c->n_states[i]) { free(tmp_s); free(tmp_r); return -2; }
                        size_t base = (size_t)g * stride + off[i] + (size_t)row * c->n_actions[i];
                        act[i] = (c->q_dtype == 1) ? argmax_f64((double*)q + base, c->n_actions[i])
                                                   : argmax_f32((float*)q + base, c->n_actions[i]);
                    }
                    scaled[i] = oracle_scale(act[i], c->n_actions[i], c->act_lo[i], c->act_hi[i]);
                    if (trace_actions) trace_actions[(((size_t)e * T + t) * N + i) * G + g] = act[i];
                }
                /* --- NoisyPriceState.step (environments.py:25-39) */
                int noisy = 0; double new_a = c->env_a;
                if (c->noise_prob > 0.0) {
                    double nu, na;
                    if (inj_u) {
                        size_t k = ((size_t)e * T + t) * G + g;
                        nu = inj_noise_u[k]; na = inj_noise_a[k];
                    } else {
                        uint32_t x[4];
                        draw(seed, gid, eg, (uint32_t)t, STREAM_NOISE, x);
                        nu = u01_32(x[0]);
                        double lo = c->env_a * 0.7;
                        na = lo + (c->env_a - lo) * u01_32(x[1]);
                    }
                    if (nu < noise_prob_g) { noisy = 1; new_a = na; }
                }
                double next_price;
                oracle_env_step(c, scaled, noisy, new_a, &next_price, rew);
                if (trace_price) trace_price[((size_t)e * T + t) * G + g] = next_price;
                /* --- memory.append (trainer.py:61-62; buffers.py:18-19, deque maxlen) */
                for (int i = 0; i < N; i++) {
                    int cap = c->capacity[i];
                    size_t mb = ((size_t)g * N + i) * (size_t)capmax;
                    if (cap > 0) {
                        int pos = cnt[i] % cap;
                        mem_s[mb + pos] = (int32_t)oracle_encode64(price, c->max_state[i], c->n_states[i]);
                        mem_a[mb + pos] = act[i];
                        mem_r[mb + pos] = rew[i];
                        mem_ns[mb + pos] = (int32_t)oracle_encode64(next_price, c->max_state[i], c->n_states[i]);
                        cnt[i] += 1;
                        if (cnt[i] >= 2 * cap) cnt[i] -= cap;
                    }
                    rlog[i] = rlog[i] + rew[i] / (double)T;            /* trainer.py:65 */
                    alog[i] = alog[i] + scaled[i] / (double)T;         /* trainer.py:66 */
                }
                price = next_price;                                    /* trainer.py:67 */
            }
            /* --- [A.train_net() for A in agents] (trainer.py:70; agents.py:59-78) */
            for (int i = 0; i < N; i++) {
                int cap = c->capacity[i];
                int len = cnt[i] < cap ? cnt[i] : cap;
                if (len >= c->min_memory[i]) {
                    int start = cnt[i] <= cap ? 0 : cnt[i] % cap;
                    size_t mb = ((size_t)g * N + i) * (size_t)capmax;
                    for (int k = 0; k < len; k++) {
                        int p = (start + k) % cap;
                        tmp_s[k] = mem_s[mb + p]; tmp_a[k] = mem_a[mb + p];
                        tmp_ns[k] = mem_ns[mb + p]; tmp_r[k] = mem_r[mb + p];
                        if (tmp_s[k] < 0 || tmp_s[k] > c->n_states[i] || tmp_ns[k] < 0 ||
                            tmp_ns[k] > c->n_states[i]) { free(tmp_s); free(tmp_r); return -2; }
                    }
                    size_t base = (size_t)g * stride + off[i];
                    int32_t* cn = counter ? counter + base : NULL;
                    const double alpha_g = sw_alpha ? sw_alpha[(size_t)i * G + g] : c->alpha[i];
                    const double gamma_g = sw_gamma ? sw_gamma[(size_t)i * G + g] : c->gamma[i];
                    if (c->q_dtype == 1)
                        oracle_td_update_f64((double*)q + base, cn, c->n_actions[i], len, tmp_s, tmp_a,
                                             tmp_r, tmp_ns, alpha_g, gamma_g);
                    else
                        oracle_td_update_f32((float*)q + base, cn, c->n_actions[i], len, tmp_s, tmp_a,
                                             tmp_r, tmp_ns, alpha_g, gamma_g);
                    cnt[i] = 0;                                         /* memory.empty() */
                }
                eps_g[i] = oracle_eps_decay(eps_g[i], sw_eps_end ? sw_eps_end[(size_t)i * G + g] : c->eps_end[i],
                                            sw_eps_step ? sw_eps_step[(size_t)i * G + g] : c->eps_step[i]);
            }
            for (int i = 0; i < N; i++) {
                size_t k = ((size_t)e * N + i) * G + g;
                if (game_reward_log) game_reward_log[k] = rlog[i];
                if (game_action_log) game_action_log[k] = alog[i];
                if (mean_reward_log) mean_reward_log[(size_t)e * N + i] += rlog[i];
                if (mean_action_log) mean_action_log[(size_t)e * N + i] += alog[i];
            }
        }
        state[g] = price;
        if (sw_eps) for (int i = 0; i < N; i++) sw_eps[(size_t)i * G + g] = eps_g[i];
        if (g == G - 1) {
            for (int i = 0; i < N; i++) { count_after[i] = cnt[i]; }
            for (int i = 0; i < N; i++) eps[i] = eps_g[i];
        }
    }
    if (G > 0) for (int i = 0; i < N; i++) mem_count[i] = count_after[i];
    for (size_t k = 0; k < (size_t)n_episodes * N; k++) {
        if (mean_reward_log) mean_reward_log[k] /= (double)G;
        if (mean_action_log) mean_action_log[k] /= (double)G;
    }
    free(tmp_s); free(tmp_r);
    return 0;
}

/* ------------------------------------------------------------------ greedy play */
/* utils.play_game (th_rl/utils.py:27-47): env.reset(), then greedy get_action
 * (agents.py:91-92: encode on the float64 state, no float32 cast), scale, step.
 * state0 [iters][G] supplies the reset() draws (NULL: Philox).  Outputs are the
 * per-iteration means over the T steps, [iters][N][G]. */
int oracle_play_greedy(const thrl_cfg* c, const void* q, const double* state0, int32_t iters,
                       uint64_t seed, uint64_t game_offset, double* mean_reward, double* mean_action) {
    const int G = c->n_games, N = c->n_agents, T = c->max_steps;
    const size_t stride = oracle_table_stride(c);
    for (int g = 0; g < G; g++) {
        uint64_t gid = game_offset + (uint64_t)g;
        for (int it = 0; it < iters; it++) {
            double price;
            if (state0) price = state0[(size_t)it * G + g];
            else {
                uint32_t x[4];
                draw(seed, gid, (uint32_t)it, 0, STREAM_PLAY_RESET, x);
                price = c->env_a * u01_53(x[0], x[1]);
            }
            double rs[THRL_MAXA] = {0}, as[THRL_MAXA] = {0};
            for (int t = 0; t < T; t++) {
                double scaled[THRL_MAXA], rew[THRL_MAXA];
                for (int i = 0; i < N; i++) {
                    int64_t row = oracle_encode64(price, c->max_state[i], c->n_states[i]);
                    if (row < 0 || row > c->n_states[i]) return -2;
                    size_t base = (size_t)g * stride + oracle_table_offset(c, i) + (size_t)row * c->n_actions[i];
                    int a = (c->q_dtype == 1) ? argmax_f64((const double*)q + base, c->n_actions[i])
                                              : argmax_f32((const float*)q + base, c->n_actions[i]);
                    scaled[i] = oracle_scale(a, c->n_actions[i], c->act_lo[i], c->act_hi[i]);
                }
                int noisy = 0; double new_a = c->env_a;
                if (c->noise_prob > 0.0) {
                    uint32_t x[4];
                    draw(seed, gid, (uint32_t)it, (uint32_t)t, STREAM_NOISE + 1, x);
                    double lo = c->env_a * 0.7;
                    if (u01_32(x[0]) < c->noise_prob) { noisy = 1; new_a = lo + (c->env_a - lo) * u01_32(x[1]); }
                }
                double np_;
                oracle_env_step(c, scaled, noisy, new_a, &np_, rew);
                for (int i = 0; i < N; i++) { rs[i] += rew[i]; as[i] += scaled[i]; }
                price = np_;
            }
            for (int i = 0; i < N; i++) {
                mean_reward[((size_t)it * N + i) * G + g] = rs[i] / (double)T;
                mean_action[((size_t)it * N + i) * G + g] = as[i] / (double)T;
            }
        }
    }
    return 0;
}
