// thrl_kernels.h -- kernel argument blocks + host-side launchers (internal).
#pragma once
#include "thrl_device.h"

namespace thrl {

// Replay memory (ReplayBuffer contents, buffers.py) as structure-of-arrays,
// entry-major / game-minor so a wave's accesses coalesce: [capmax][N][G].
struct ReplayMem {
    int16_t* s;    // encode64(state)       (agents.py:62)
    int16_t* ns;   // encode64(next_state)  (agents.py:66)
    int16_t* a;    // action index
    double*  r;    // reward (float64 as stored by trainer.py:62)
    double*  ov;   // scratch: old_value snapshot (agents.py:67)
};

struct GenericArgs {
    int32_t G, N, T, capmax;
    int64_t stride;
    EnvParams env;
    AgentParams ag[THRL_MAXA];
    void* q;
    int32_t* counter;
    double* state;
    ReplayMem mem;
    double* sum_reward;        // [E][N] sums over games (finalised to means afterwards) or null
    double* sum_action;
    double* game_reward_log;   // [E][N][G] or null
    double* game_action_log;
    const double* inj_u;
    const int8_t* inj_choice;
    const double* inj_noise_u;
    const double* inj_noise_a;
    uint64_t seed, game_offset, first_episode;
    int32_t n_episodes;
    double eps0[THRL_MAXA];
    int32_t cnt0[THRL_MAXA];
    // per-game sweeps (null = the scalar parameters above), [N][G] except noise_prob [G]
    const double* sw_gamma; const double* sw_alpha; const double* sw_eps_end; const double* sw_eps_step;
    double* sw_eps; const double* sw_noise_prob;
};

// ---- fused wave-per-game kernel (thrl_wave.hip) -----------------------------
constexpr int kWaveMaxEpisodes = 32;   // 32 episodes x 4 log values = the 64 lanes of two accumulators

struct WaveArgs {
    int32_t G, T, A, rows;          // homogeneous 2-agent game; T = max_steps
    int32_t epk;                    // episodes per training cycle: the buffer reaches min_memory every epk-th
                                    // episode (buffers.py / agents.py:60); epk * T <= 256 transitions
    int32_t replay_from;            // first transition of a cycle that is still in the deque when it trains
                                    // (epk*T - min(epk*T, capacity); == epk*T: the agents never train)
    int32_t row_lo, win_rows;       // LDS window = rows [row_lo, row_lo+win_rows) + 1 spill row
    int32_t n_episodes;
    int32_t waves_per_block;
    int32_t total_waves;
    int32_t force_variant;          // 0: plain / GREEDY by epsilon; 1: plain; 2: GREEDY (thrl_kernel WAVE_PLAIN / WAVE_GREEDY)
    int32_t lut_bytes;              // LDS bytes of the payoff LUT region
    int32_t game_lds_bytes;         // LDS bytes per wave (tables)
    int64_t stride;
    EnvParams env;
    AgentParams ag[2];
    void* q;                        // float or double tables (the kernel variant's QT)
    int32_t* counter;
    double* state;
    const unsigned char* lut_ns;    // device: payoff LUT image (thrl_wave_lut.h), lut_bytes long
    long long* partial;             // device [total_waves][kWaveMaxEpisodes][4] per-wave log sums, fixed point
    double log_scale[2];            // fixed-point scales of the (reward, action) log sums: powers of two, sized by the host
                                    // so that G games cannot overflow 2^62; integer sums are order independent
    uint32_t* tlog;                 // device [total_waves][32 episodes][NSEG][64] packed transitions
    int32_t* next_game;             // device: work counter of the launch (games are handed out dynamically), zeroed by the host
    const double* inj_u;            // parity mode: device [n_episodes][T][2][G] uniforms, or null (Philox)
    const int8_t* inj_choice;       // parity mode: device [n_episodes][T][2][G] random.choice indices
    const double* inj_noise_u;      // parity mode with noise: device [n_episodes][T][G]
    const double* inj_noise_a;
    // per-game sweeps (null = the scalar parameters above), [2][G] except noise_prob [G]
    const double* sw_gamma; const double* sw_alpha; const double* sw_eps_end; const double* sw_eps_step;
    double* sw_eps; const double* sw_noise_prob;
    uint64_t seed, game_offset, first_episode;
    double eps[kWaveMaxEpisodes][2];
};

// ---- fused kernel for 1-4 QTable agents with individual grids (thrl_tuple_kernel.h): state = action tuple
constexpr int kTupMaxN = 4;
constexpr int kTupMaxEpisodes = 32;
constexpr int kTupMaxTuples = 4096;
struct TupleArgs {
    int32_t G, N, T, n_episodes, tuples;
    int32_t waves_per_block, total_waves;
    int32_t lut_lds_bytes;          // LDS-staged part of the LUT image: prow / trow [tuples] u32 each, then aq / sct [N][64] doubles
    int32_t aq_off, price_off;      // byte offsets in the image: aq (inside the staged part), price [tuples] (HBM only)
    int32_t qsum_off;               // total quantity per tuple [tuples] doubles (HBM only; games with env noise)
    int32_t game_lds_bytes;         // per wave: tables (the visit histogram overlays them after write-back) | greedy-action bytes | G table
    int32_t tab_off[kTupMaxN];      // element offset of agent i's window (+ 2 spill rows) in the per-game table region
    int32_t am_off, am_off_i[kTupMaxN];      // byte offsets
    int32_t g_off;                  // byte offset of G [tuples + 1] u16: the agents' greedy actions as bit fields
    int32_t act_sh[kTupMaxN], act_bits[kTupMaxN];          // agent i's field in an action word: shift, width = ceil(log2 A_i)
                                                           // (the widths sum to < log2(tuples) + N <= 16)
    int32_t hist_off_i[kTupMaxN], hist_dwords;   // histogram (u16 per cell, at the table region's base): per-agent dword offsets, total
    void* vlog;                     // device: visit log, per wave [n_episodes][T] words of N cells (u16 each; 4 bytes for N <= 2, else 8)
    int64_t vlog_wave_bytes;        // bytes of one wave's log
    int32_t row_lo[kTupMaxN], win_rows[kTupMaxN];
    int64_t stride;
    EnvParams env;
    AgentParams ag[kTupMaxN];
    void* q; int32_t* counter; double* state;
    const unsigned char* lut;       // device: the LUT image
    double* sum_reward; double* sum_action;      // device [n_episodes][N] sums over games (zeroed by the host) or null
    int32_t* next_game;             // device: work counter of the launch, zeroed by the host
    const double* inj_u; const int8_t* inj_choice;   // parity mode: [n_episodes][T][N][G], or null (Philox)
    const double* inj_noise_u; const double* inj_noise_a;      // parity mode with noise: [n_episodes][T][G]
    // per-game sweeps (null = the scalar parameters), [N][G] except noise_prob [G]; `sweep` = any of them given
    const double* sw_gamma; const double* sw_alpha; const double* sw_eps_end; const double* sw_eps_step;
    double* sw_eps; const double* sw_noise_prob;
    int32_t sweep;
    uint64_t seed, game_offset, first_episode;
    double eps[kTupMaxEpisodes][kTupMaxN];
};
int launch_tuple_lut(const TupleArgs& a, unsigned char* out, hipStream_t s);
int launch_tuple(const TupleArgs& a, int q_dtype, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_tuple_f32(const TupleArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_tuple_f64(const TupleArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_tuple_f32_noise(const TupleArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_tuple_f64_noise(const TupleArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_tuple_f32_sweep(const TupleArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_tuple_f64_sweep(const TupleArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);

// ---- two-agent games with discrete neural-policy agents on the tuple-chain design (thrl_ptuple.hip)
struct PTupleArgs {
    int32_t G, T, n_episodes, tuples, npid;
    int32_t kind[2];                // 0 = QTable, 1 = Reinforce, 2 = ActorCritic
    int32_t qi, n_r, ri[2];         // the QTable agent (or -1), the policy agents
    int32_t waves_per_block, lut_lds_bytes, game_lds_bytes;
    int32_t qrows_off, xf_off, aq_off, price_off;        // byte offsets in the LUT image (price: HBM only)
    int32_t qsum_off;               // total quantity per action pair [tuples] doubles (HBM only; games with env noise)
    int32_t am_off, g_off, hist_off, hist_dwords, cdf_off, logs_off;      // byte offsets in the per-game LDS region
    int32_t row_lo, win_rows;       // the QTable agent's row window
    int64_t stride;
    EnvParams env;
    AgentParams ag[2];
    void* q; int32_t* counter; double* state;
    const unsigned char* lut;
    const float* nn_params[2]; int32_t nn_stride[2];
    double* buf_price[2]; int32_t* buf_action[2]; double* buf_reward[2]; double* buf_nprice[2];
    int32_t buf_len[2], count0[2];
    double eps0[2];
    float* policy_tab;              // HBM CDF rows [G][npid + 1][APAD] (one policy against a QTable)
    // per-game sweeps of the QTable agent / the env (null = the scalars), [2][G] except noise_prob [G]; `sweep` = the kernel's
    // sweep variant is needed (a QTable agent in the game and any of them given)
    const double* sw_gamma; const double* sw_alpha; const double* sw_eps_end; const double* sw_eps_step;
    double* sw_eps; const double* sw_noise_prob;
    int32_t sweep;
    double* game_reward_log; double* game_action_log;    // [n_episodes][2][G]
    int32_t* next_game;
    uint64_t seed, game_offset, first_episode;
};
int launch_ptuple_lut(const PTupleArgs& a, unsigned char* out, hipStream_t s);
int launch_ptuple(const PTupleArgs& a, int q_dtype, int grid, int block, size_t lds_bytes, hipStream_t s);

int launch_generic(const GenericArgs& a, int q_dtype, hipStream_t s);
int launch_finalize_logs(double* sum_reward, double* sum_action, int n, int G, hipStream_t s);
int launch_wave_lut(const WaveArgs& a, unsigned char* out, hipStream_t s);
int launch_wave(const WaveArgs& a, int q_dtype, int grid, int block, size_t lds_bytes, hipStream_t s);
// one translation unit per (table type, NOISE, SWEEP) family of k_wave_episodes instantiations
int launch_wave_f32_plain(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f32_noise(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f32_sweep(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f64_plain(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f64_noise(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f64_sweep(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
// CYCLE = true variants: training cycles of several episodes and / or transitions dropped from the deque
// GREEDY: the play loop skips the per-step table of groups of four steps in which nobody explores
int launch_wave_f32_plain_greedy(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f64_plain_greedy(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f32_plain_cycle(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f32_noise_cycle(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f64_plain_cycle(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_f64_noise_cycle(const WaveArgs& a, int grid, int block, size_t lds_bytes, hipStream_t s);
int launch_wave_reduce(const long long* partial, const double* log_scale, int total_waves, int n_episodes, int G,
                       double* reward_log, double* action_log, hipStream_t s);

struct InitArgs {
    int32_t G, N;
    int64_t stride;
    double env_a;
    AgentParams ag[THRL_MAXA];
    void* q; int32_t* counter; double* state;
    uint64_t seed, game_offset;
    const double* sw_gamma;         // per-game gamma [N][G] (the table offset 12.5/(1-gamma), agents.py:29) or null
};
int launch_init(const InitArgs& a, int q_dtype, hipStream_t s);

struct PlayArgs {
    int32_t G, N, T, iters;
    int64_t stride;
    EnvParams env;
    AgentParams ag[THRL_MAXA];
    const void* q;
    const double* state0;
    double* mean_reward; double* mean_action;
    uint64_t seed, game_offset;
};
int launch_play(const PlayArgs& a, int q_dtype, hipStream_t s);

struct OpArgs {
    int32_t G, N, agent, n, encode32;
    int64_t stride;
    EnvParams env;
    AgentParams ag[THRL_MAXA];
    void* q; int32_t* counter;
    const double* price; const double* next_price; const double* u; const int8_t* choice;
    const int32_t* action; const double* reward; const double* noise_u; const double* noise_a;
    const double* scaled;
    double eps;
    int32_t* action_out; double* price_out; double* reward_out; double* scaled_out;
};
int launch_op_sample(const OpArgs& a, int q_dtype, hipStream_t s);
int launch_op_env_step(const OpArgs& a, hipStream_t s);
int launch_op_scale(const OpArgs& a, hipStream_t s);
int launch_op_encode(const OpArgs& a, hipStream_t s);
int launch_op_td(const OpArgs& a, int q_dtype, hipStream_t s);

// ---- neural policy agent (thrl_nn.hip)
struct MixedArgs {
    int32_t G, N, T, n_episodes;
    int64_t stride;
    EnvParams env;
    AgentParams ag[THRL_MAXA];            // QTable parameters; for a Reinforce slot: n_actions, act_lo, act_span
    int32_t kind[THRL_MAXA];              // 0 = QTable, 1 = Reinforce, 2 = ActorCritic (same policy head), 3 = CAC
    void* q; int32_t* counter; double* state;
    const float* nn_params[THRL_MAXA];    // [G][nn_stride] per Reinforce / ActorCritic agent
    int32_t nn_stride[THRL_MAXA];
    double* buf_price[THRL_MAXA]; int32_t* buf_action[THRL_MAXA];      // replay rings [G][buf_len] (game-major)
    double* buf_reward[THRL_MAXA]; double* buf_nprice[THRL_MAXA]; double* buf_ov[THRL_MAXA];
    int32_t buf_len[THRL_MAXA]; int32_t min_memory[THRL_MAXA]; int32_t count0[THRL_MAXA];
    double eps0[THRL_MAXA];
    double* game_reward_log; double* game_action_log;                  // [n_episodes][N][G]
    uint64_t seed, game_offset, first_episode;
    int32_t n_r; int32_t ragent[2];                                    // the (at most 2) Reinforce agents
    int32_t lds_off[THRL_MAXA];                                        // QTable: element offset of the table in LDS; CAC: float offset
    int32_t lds_bytes;
    int32_t memo_lds_byte0, memo_on, memo_k;                                   // memoised policy CDFs [n_r][64][APAD] floats
    float* policy_tab; size_t policy_tab_bytes, ptab_need_bytes;               // HBM policy table [G][n_r][tuples][APAD] (caller's scratch)
    int32_t ptab_on, ptab_tuples;
    int32_t n_cac, cac_lds_byte0;                                      // CAC networks (kind 3) live in LDS after the tables
    int32_t stage_lds_byte0;                                           // 16 steps of transitions staged before they go to the replay rings
    // per-game sweeps of the QTable agents / the env (null = the scalars above), [N][G] except noise_prob [G]
    const double* sw_gamma; const double* sw_alpha; const double* sw_eps_end; const double* sw_eps_step;
    double* sw_eps; const double* sw_noise_prob;
};
// fills n_r / ragent / lds_off / lds_bytes; returns 0 or -1 with a reason when the config does not fit
int plan_mixed(MixedArgs& a, int q_dtype, const char** why);
int launch_mixed(const MixedArgs& a, int q_dtype, hipStream_t s);
int launch_nn_init(int G, int A, float* params, uint64_t seed, uint64_t off, int agent, int value_head, hipStream_t s);
int launch_nn_act(int G, int A, const float* params, int P, const double* price, const double* u, int32_t* act,
                  float* prob, hipStream_t s);
size_t nn_train_lds_bytes(int A, int N, int value_head);
// nprice != NULL: ActorCritic update (value head, params stride P + 257); NULL: Reinforce
int launch_nn_train(int G, int A, float* params, float* m, float* v, int step, int N, int ld, const double* price,
                    const int32_t* action, const double* reward, const double* nprice, float gamma, float ent, float lr,
                    const double* gamma_g, const double* ent_g,      // per-game gamma / entropy coefficient [G] or null
                    float* grad,
                    float* returns_scratch,                            // [G][ld] floats or null (Reinforce: returns by a pre-pass)
                    hipStream_t s);
// ---- continuous actor-critic agent CAC (thrl_cac.hip)
int launch_cac_init(int G, float* params, uint64_t seed, uint64_t off, int agent, hipStream_t s);
int launch_cac_act(int G, const float* params, const double* price, const double* u1, const double* u2, float* action,
                   float* mu, float* sd, float* v, hipStream_t s);
size_t cac_train_lds_bytes(int N);
int launch_cac_train(int G, float* params, float* m, float* v, int step, int N, int ld, const double* price, const float* action,
                     const double* reward, const double* nprice, float gamma, float ent, float lr,
                     const double* gamma_g, const double* ent_g, float* grad, hipStream_t s);
int launch_op_draws(int G, int N, uint64_t seed, uint64_t off, uint32_t episode, uint32_t step, double env_a,
                    double noise_lo, const int32_t* nA, double* u, int8_t* ch, double* u2, double* nu, double* na,
                    hipStream_t s);

}  // namespace thrl
