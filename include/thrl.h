/*
 * thrl.h -- C ABI of libthrl_hip.so: the MI355X (gfx950) implementation of the
 * iterated-pricing-game hot path of HakimNessah/th_rl.
 *
 * The reference has NO FFI / plugin layer: its boundary is the
 * Python duck-typed protocol that th_rl/trainer.py:46-70 drives.  Each entry
 * point below therefore cites the reference *Python* code it replaces.  All
 * pointers marked "device" are device (HBM) pointers owned by the caller
 * (e.g. torch tensors); the library allocates nothing persistent, never
 * throws, never exits; every function returns 0 on success or a negative
 * thrl_err, and thrl_last_error() gives a thread-local message.  Work is
 * enqueued on the caller's HIP stream (passed as void*, 0 = default stream)
 * and is asynchronous; the caller synchronises.
 *
 * Re-entrant and thread-safe: calls on different threads / streams / devices do
 * not share mutable state.  The only process-wide state is (i) the thread-local
 * error string, (ii) a read-only cache of per-device figures (CU count, LDS per
 * CU, resident waves per CU from hipDeviceProp_t), keyed by the HIP device id
 * and filled once per device under a lock, and (iii) the tuning knobs
 * THRL_WAVE_MAX_WAVES_PER_CU and THRL_GREEDY_EPS (measurement only: the epsilon
 * below which the greedy-regime variants of the fused kernel are launched;
 * results do not depend on it), read from the environment once per process
 * (per call: thrl_run.kernel = THRL_KERNEL_WAVE_PLAIN / THRL_KERNEL_WAVE_GREEDY).
 * Launch geometry and thrl_workspace_bytes() refer to the CURRENT HIP device of
 * the calling thread (hipSetDevice / torch.cuda.device).
 *
 * Plain C: no torch / HIP types in any signature.
 */
#ifndef THRL_H
#define THRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THRL_ABI_VERSION 3          /* 2: per-game sweeps in thrl_qtable_init, thrl_mixed and the *_train entry points;
                                       3: thrl_build_info, THRL_KERNEL_WAVE_PLAIN / _GREEDY, replay rings [G][buf_len] */
#define THRL_MAXA 8          /* max agents per game (reference configs use 2) */
#define THRL_MAX_EPISODES_PER_LAUNCH 32

typedef enum {
    THRL_OK = 0,
    THRL_ERR_BAD_CONFIG = -1,   /* shape / range error in thrl_cfg              */
    THRL_ERR_NULL = -2,         /* a required pointer is NULL                   */
    THRL_ERR_UNSUPPORTED = -3,  /* requested kernel cannot run this config      */
    THRL_ERR_HIP = -4,          /* HIP runtime error (message has the string)   */
    THRL_ERR_WORKSPACE = -5     /* workspace / replay memory too small          */
} thrl_err;

typedef enum {
    THRL_KERNEL_AUTO = 0,       /* fused wave kernel when eligible, else generic */
    THRL_KERNEL_GENERIC = 1,    /* one thread per game, tables in HBM, f32/f64  */
    THRL_KERNEL_WAVE = 2,       /* one wavefront per game, tables in LDS, f32 / f64 */
    /* The wave kernel has two code variants with IDENTICAL results: the plain one and the greedy-regime one
     * (per-episode composed greedy tables, cyclic segments as register recurrences); THRL_KERNEL_WAVE picks by
     * epsilon (the greedy one once every agent's epsilon <= 0.05).  These two ids pin the variant for one call
     * -- measurements and the parity tests that cover each variant in each regime.  _GREEDY fails with
     * THRL_ERR_UNSUPPORTED where no greedy variant exists (noise, sweeps, multi-episode training cycles). */
    THRL_KERNEL_WAVE_PLAIN = 3,
    THRL_KERNEL_WAVE_GREEDY = 4,
    /* one wavefront per game, tables in LDS, for 1-4 QTable agents with INDIVIDUAL state / action grids (any
     * nplayers and any QTable per agent: trainer.py:21-23): games (with or without env noise) whose replay buffers
     * train once per episode; the state is carried as the action tuple of the last step, or as explicit table rows
     * after a step with a redrawn intercept; per-game sweeps (thrl_buffers.sweep_*) are taken, an epsilon-schedule
     * sweep needs sweep_eps.  AUTO picks it where the two-agent wave kernel does not apply. */
    THRL_KERNEL_TUPLE = 5
} thrl_kernel;

/*
 * Game description.  One thrl_cfg == the JSON config blocks that
 * trainer.create_game() splats into the constructors (trainer.py:13-26):
 *   per agent  : QTable.__init__ kwargs           (agents.py:13-28)
 *   environment: NoisyPriceState.__init__ kwargs  (environments.py:5)
 */
typedef struct {
    int32_t n_games;                 /* G: games stepped in lockstep on this device      */
    int32_t n_agents;                /* N = environment.nplayers = len(agents)           */
    int32_t max_steps;               /* T = environment.max_steps                        */
    int32_t q_dtype;                 /* 0 = float32 tables, 1 = float64 tables           */
    double  env_a, env_b;            /* demand intercept / slope (environments.py:5)     */
    double  noise_prob;              /* environments.py:28                               */
    int32_t n_states[THRL_MAXA];     /* QTable `states`  (table has states+1 rows)       */
    int32_t n_actions[THRL_MAXA];    /* QTable `actions`                                 */
    int32_t min_memory[THRL_MAXA];   /* agents.py:26,60                                  */
    int32_t capacity[THRL_MAXA];     /* ReplayBuffer deque maxlen (buffers.py:12)        */
    double  max_state[THRL_MAXA];    /* agents.py:21,48                                  */
    double  gamma[THRL_MAXA], alpha[THRL_MAXA];
    double  eps_end[THRL_MAXA], eps_step[THRL_MAXA];
    double  act_lo[THRL_MAXA], act_hi[THRL_MAXA];   /* action_range                      */
} thrl_cfg;

/*
 * HBM layout (all game-major so one game's data is one contiguous slab):
 *   q        [G][stride]  stride = sum_i (n_states[i]+1)*n_actions[i]; agent i's
 *                         block starts at off_i = sum_{j<i} rows_j*A_j and is the
 *                         reference's `QTable.table` (agents.py:29) row-major.
 *   counter  [G][stride]  int32, `QTable.counter` (agents.py:45,76); may be NULL.
 *   state    [G]          float64 env state = last price (environments.py:36).
 *   replay_mem            opaque, thrl_replay_mem_bytes(); the ReplayBuffer
 *                         contents that survive between episodes (buffers.py).
 */
typedef struct {
    void*    q;                      /* device, f32 or f64 per cfg.q_dtype               */
    int32_t* counter;                /* device or NULL                                   */
    double*  state;                  /* device                                           */
    void*    replay_mem;             /* device, generic kernel only (may be NULL for WAVE)*/
    size_t   replay_mem_bytes;
    double*  reward_log;             /* device [n_episodes][N]: mean over the G games of
                                        rewards_log[e,:] (trainer.py:65); or NULL        */
    double*  action_log;             /* device [n_episodes][N] (trainer.py:66); or NULL  */
    double*  game_reward_log;        /* device [n_episodes][N][G] per-game rows; or NULL */
    double*  game_action_log;        /* device [n_episodes][N][G]; or NULL               */
    /* parity mode: the reference's recorded random draws, or NULL for Philox       */
    const double* inj_u;             /* device [n_episodes][T][N][G] random.uniform(0,1) (agents.py:81) */
    const int8_t* inj_choice;        /* device [n_episodes][T][N][G] random.choice idx   (agents.py:82) */
    const double* inj_noise_u;       /* device [n_episodes][T][G]    (environments.py:28); NULL if noise_prob<=0 */
    const double* inj_noise_a;       /* device [n_episodes][T][G]    (environments.py:29) */
    void*    workspace;              /* device scratch, thrl_workspace_bytes()           */
    size_t   workspace_bytes;
    /* Per-game hyper-parameter sweeps (optional; NULL = thrl_cfg's scalar for every game): the
     * reference sweeps configs x runs one process at a time (main.py:13-21); here a sweep is a
     * per-game array.  Device, layout [N][G] (agent-major); honoured by both episode kernels. */
    const double* sweep_gamma;       /* QTable gamma  (agents.py:30)                     */
    const double* sweep_alpha;       /* QTable alpha  (agents.py:31)                     */
    const double* sweep_eps_end;     /* agents.py:37                                     */
    const double* sweep_eps_step;    /* agents.py:36                                     */
    double*       sweep_eps;         /* in/out: current epsilon per (agent, game); when given it
                                        replaces thrl_run.eps (agents.py:35,78)          */
    const double* sweep_noise_prob;  /* device [G]; thrl_cfg.noise_prob must be > 0 if any entry is */
} thrl_buffers;

/* Host-side run state that is identical for every game (so it never lives in HBM). */
typedef struct {
    uint64_t seed;                   /* Philox key                                       */
    uint64_t game_offset;            /* global id of local game 0 (sharding-invariant RNG)*/
    uint64_t first_episode;          /* global episode index of the first episode        */
    int32_t  n_episodes;             /* episodes to run in this call                     */
    int32_t  kernel;                 /* thrl_kernel                                      */
    double   eps[THRL_MAXA];         /* in/out: QTable.epsilon (agents.py:35,78)         */
    int32_t  mem_count[THRL_MAXA];   /* in/out: appends since the last memory.empty();
                                        len(agent.memory) == min(mem_count, capacity)
                                        (buffers.py:12-19,40); kept in [0, 2*capacity)   */
    int32_t  kernel_used;            /* out: thrl_kernel actually launched               */
} thrl_run;

int         thrl_version(void);
const char* thrl_last_error(void);
/* What this binary is: "abi=3;ablate=<mask>;src=<hash of all kernel sources>;wave=<hash of the sources of the
 * headline kernel>;nn=<hash of the neural-agent kernels' sources>".  ablate != 0 marks a
 * TIMING-ONLY diagnostic build (phases of the fused kernel compiled out, results wrong by construction;
 * profiles/ablate.py) -- callers that report results or throughput must refuse it (bench.py does). */
const char* thrl_build_info(void);
/* the ablation mask alone (0 = the product library) */
int         thrl_ablate_mask(void);

/* elements per game in q / counter (sum_i rows_i*A_i); 0 on bad config */
size_t thrl_table_stride(const thrl_cfg* cfg);
/* element offset of agent i's table inside one game's slab */
size_t thrl_table_offset(const thrl_cfg* cfg, int agent);
size_t thrl_replay_mem_bytes(const thrl_cfg* cfg);
/* scratch the episode kernels need for THIS config on the current device (payoff LUT image, per-wave
 * log partials and transition log of the wave kernel's persistent grid; LUT image and per-wave visit log
 * of the tuple-chain kernel's); 0 on a bad config */
size_t thrl_workspace_bytes(const thrl_cfg* cfg);
/* which kernel THRL_KERNEL_AUTO would pick for this config (thrl_kernel) */
int    thrl_select_kernel(const thrl_cfg* cfg, int injected);
/* Training cycle of the wave kernel: the replay buffers (buffers.py:12-19) reach min_memory every k-th
 * episode (agents.py:60), k = ceil(min_memory / max_steps); a call runs on the wave kernel when its
 * n_episodes is a multiple of k and the buffers are empty on entry (thrl_run.mem_count == 0), otherwise on
 * the generic kernel, which keeps the buffers in replay_mem.  Returns k >= 1, or 0 when the config is the
 * generic kernel's anyway. */
int    thrl_training_cycle(const thrl_cfg* cfg);

/*
 * Replaces QTable.__init__ table/counter init (agents.py:29,45) and
 * NoisyPriceState.reset() (environments.py:50-53, called once at trainer.py:45)
 * for all G games: q = 12.5/(1-gamma_i) + N(0,1), counter = 0, state ~ U(0,a),
 * from Philox4x32-10 keyed by (seed, global game id).  sweep_gamma: device [N][G] per-game gamma
 * (the table offset of a config sweep) or NULL for thrl_cfg.gamma.
 */
int thrl_qtable_init(const thrl_cfg* cfg, void* q, int32_t* counter, double* state,
                     uint64_t seed, uint64_t game_offset, const double* sweep_gamma, void* stream);

/*
 * THE hot path: replaces the body of trainer.train_one's loop (trainer.py:46-70)
 * for `run->n_episodes` episodes of all G games: per step QTable.sample_action
 * (agents.py:80-89), QTable.scale (:51-57), NoisyPriceState.step
 * (environments.py:25-39), ReplayBuffer.append (buffers.py:18-19), the log
 * accumulation (trainer.py:65-66); per episode QTable.train_net
 * (agents.py:59-78) = ReplayBuffer.replay/empty + snapshot-TD + epsilon decay.
 */
int thrl_qtable_episodes(const thrl_cfg* cfg, const thrl_buffers* bufs, thrl_run* run,
                         void* stream);

/*
 * Greedy evaluation rollout, replaces utils.play_game (utils.py:27-47):
 * env.reset() then `iters` episodes of get_action (agents.py:91-92) / scale /
 * env.step with no learning.  state0 [iters][G] are the reset() draws
 * (device, or NULL to draw from Philox); outputs are per-game episode means
 * mean_reward/mean_action [iters][N][G] (device).
 */
int thrl_play_greedy(const thrl_cfg* cfg, const void* q, const double* state0,
                     int32_t iters, uint64_t seed, uint64_t game_offset,
                     double* mean_reward, double* mean_action, void* stream);

/*
 * Unfused, batched-over-games operator forms of the reference methods (same
 * argument meaning, one call = one reference call applied to G games).  They
 * exist so a caller that drives the step loop itself (the duck-typed protocol)
 * still runs on the device.
 */
/* QTable.sample_action / get_action for agent `agent` (agents.py:80-92):
 * u/choice device [G] (u==NULL => greedy get_action); price device [G];
 * encode32 != 0 applies the trainer's float32 cast first (trainer.py:53). */
int thrl_op_sample_action(const thrl_cfg* cfg, int agent, const void* q, const double* price,
                          double eps, const double* u, const int8_t* choice, int encode32,
                          int32_t* action_out, void* stream);
/* QTable.encode for agent `agent` (agents.py:47-49): price [G] f64 -> row index [G] int32;
 * as_float32 != 0 evaluates it on the float32-cast state as sample_action does (trainer.py:53). */
int thrl_op_encode(const thrl_cfg* cfg, int agent, const double* price, int as_float32,
                   int32_t* row_out, void* stream);
/* QTable.scale for agent `agent` (agents.py:51-57): action index [G] int32 -> scaled [G] f64 */
int thrl_op_scale(const thrl_cfg* cfg, int agent, const int32_t* action, double* scaled_out, void* stream);
/* NoisyPriceState.step for G games (environments.py:25-39): scaled device [N][G] f64 = the
 * `actions` argument of the reference (already scaled); noise_u/noise_a device [G] or NULL
 * (the two numpy.random.uniform draws of :28-29); outputs price [G], reward [N][G]. */
int thrl_op_env_step(const thrl_cfg* cfg, const double* scaled, const double* noise_u,
                     const double* noise_a, double* price_out, double* reward_out, void* stream);
/* QTable.train_net's table update for agent `agent` on n transitions per game
 * (agents.py:61-76): price/next_price [n][G] f64, action [n][G] int32, reward [n][G] f64;
 * scratch: device [n][G] f64 (holds the old_value snapshot of agents.py:67). */
int thrl_op_td_update(const thrl_cfg* cfg, int agent, void* q, int32_t* counter, int32_t n,
                      const double* price, const int32_t* action, const double* reward,
                      const double* next_price, double* scratch, void* stream);

/*
 * Neural policy agent `Reinforce` (agents.py:119-220): a 1 -> 256 -> A MLP per game.
 * Parameter vector per game, P = thrl_nn_param_count(A) floats:
 *   [fc1.weight (256) | fc1.bias (256) | fc_pi.weight (A x 256 row-major) | fc_pi.bias (A)]
 * All arrays are device pointers and game-major: [G][P] for parameters and Adam moments, [G][ld] for the replayed
 * buffer (row g = game g's transitions in insertion order, the first n of its ld entries; ABI v3 -- v2 was
 * transition-major [n][G], which put one game's batch 8*G bytes apart: one 64-byte sector per transition).
 * float32 arithmetic as in torch.
 */
#define THRL_NN_HIDDEN 256
#define THRL_NN_MAX_TRANSITIONS 1400
size_t thrl_nn_param_count(int n_actions);
/* torch.nn.Linear default init, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases
 * (agents.py:136-137), from Philox keyed by (seed, global game id, agent). */
int thrl_nn_init(int n_games, int n_actions, float* params, uint64_t seed, uint64_t game_offset,
                 int agent, void* stream);
/* Reinforce.pi + sample_action / get_action (agents.py:147-168) for G games: price [G] f64
 * (cast to float32 as the trainer does, trainer.py:53); u [G] uniforms in [0,1) for the
 * categorical draw (inverse CDF), or NULL for the greedy get_action; outputs action [G] and
 * optionally the probabilities [G][A]. */
int thrl_nn_act(int n_games, int n_actions, const float* params, const double* price, const double* u,
                int32_t* action_out, float* prob_out, void* stream);
/* Reinforce.train_net's update (agents.py:171-193) for G games on n replayed transitions each:
 * discounted returns, z-score (unbiased std), policy-gradient + entropy loss, gradient-norm clip
 * at 1.0, one Adam step (lr, betas 0.9/0.999, eps 1e-8).  step = Adam step count BEFORE the call.
 * price / action / reward: device [G][ld], n <= ld valid entries per row (a replay ring of thrl_mixed can be
 * passed as it is: ld = buf_len).  sweep_gamma / sweep_entropy: device [G] per-game values (a config sweep as
 * one batch) or NULL for the scalars.  grad_out [G][P] (optional) receives the clipped gradient.  returns_scratch
 * (optional): device [G][ld] floats; when given, the discounted returns (:178-181: a serial recurrence per game) are computed
 * by a pre-pass with one lane per game instead of by one thread of each game's block -- the same operations in the same
 * order, so the same bits; NULL keeps the in-kernel form. */
int thrl_nn_reinforce_train(int n_games, int n_actions, float* params, float* adam_m, float* adam_v,
                            int32_t step, int32_t n, int32_t ld, const double* price, const int32_t* action,
                            const double* reward, double gamma, double entropy_coef, double lr,
                            const double* sweep_gamma, const double* sweep_entropy,
                            float* grad_out, float* returns_scratch, void* stream);
/*
 * `ActorCritic` (agents.py:222-330): Reinforce's network plus a value head fc_v (256 -> 1, bias
 * initialised to 1000, :243-244) on the shared hidden layer.  Parameter vector per game,
 * thrl_ac_param_count(A) = thrl_nn_param_count(A) + 257 floats:
 *   [Reinforce layout | fc_v.weight (256) | fc_v.bias (1)]
 * thrl_ac_act = pi + sample_action / get_action (:262-272, identical to Reinforce's).
 * thrl_ac_train = train_net (:274-305) AS THE REFERENCE EXECUTES IT: `rewards` is [N] while v and
 * v_prime are [N,1], so advantage = (rewards + gamma*v_prime) - v broadcasts to [N,N]
 * (advantage[i,j] = r_j + gamma*v'_i - v_i); loss = mean_ij(advantage^2 - log p_j(a_j)*advantage)
 * + entropy_coef * (-mean H); v_prime is not detached.  next_price [G][ld] = the replayed new_state.
 */
size_t thrl_ac_param_count(int n_actions);
int thrl_ac_init(int n_games, int n_actions, float* params, uint64_t seed, uint64_t game_offset,
                 int agent, void* stream);
int thrl_ac_act(int n_games, int n_actions, const float* params, const double* price, const double* u,
                int32_t* action_out, float* prob_out, void* stream);
int thrl_ac_train(int n_games, int n_actions, float* params, float* adam_m, float* adam_v,
                  int32_t step, int32_t n, int32_t ld, const double* price, const int32_t* action,
                  const double* reward, const double* next_price, double gamma, double entropy_coef,
                  double lr, const double* sweep_gamma, const double* sweep_entropy, float* grad_out, void* stream);
/*
 * `CAC`, the continuous actor-critic (agents.py:333-442): fc1 (1 -> 256) and three 256 -> 1 heads,
 * mu = 4*tanh(fc_mu h), std = softplus(fc_std h), v = fc_v h.  THRL_CAC_PARAMS floats per game:
 *   [fc1.weight 256 | fc1.bias 256 | fc_mu.weight 256 | fc_mu.bias | fc_std.weight 256 | fc_std.bias |
 *    fc_v.weight 256 | fc_v.bias]
 * thrl_cac_act = pi + sample_action (:360-381): a = mu + std*z, z from u1,u2 [G] by Box-Muller,
 *   action = sigmoid(a) in (0,1) (float32 [G]); scale is action*(hi-lo)+lo (:371-375).
 *   u1 == NULL gives the MEAN action sigmoid(mu): the reference's get_action (:383-387) builds
 *   Normal(mu, 0), which torch rejects (ValueError, recorded in tests/golden/g9_cac.npz), so its play
 *   path cannot run; the mean action is what it intends.  mu/std/v outputs are optional.
 * thrl_cac_train = train_net (:389-416) AS THE REFERENCE EXECUTES IT: rewards / actions [N] against
 *   mu / std / v [N,1] broadcast to [N,N] -- advantage[i,j] = r_j + gamma*v'_i - v_i and
 *   log_prob[i,j] = log N(logit(a_j); mu_i, std_i); loss = mean_ij(advantage^2 - log_prob*advantage)
 *   + entropy_coef*(-mean entropy); clip 1.0; Adam.  action [G][ld] float32 as stored by the trainer.
 */
#define THRL_CAC_PARAMS 1283
int thrl_cac_init(int n_games, float* params, uint64_t seed, uint64_t game_offset, int agent, void* stream);
int thrl_cac_act(int n_games, const float* params, const double* price, const double* u1, const double* u2,
                 float* action_out, float* mu_out, float* std_out, float* v_out, void* stream);
int thrl_cac_train(int n_games, float* params, float* adam_m, float* adam_v, int32_t step, int32_t n, int32_t ld,
                   const double* price, const float* action, const double* reward, const double* next_price,
                   double gamma, double entropy_coef, double lr, const double* sweep_gamma, const double* sweep_entropy,
                   float* grad_out, void* stream);
/*
 * Fused episodes for games whose agents are any mix of QTable and Reinforce (the pairing of the
 * reference's example configs): trainer.train_one's loop (trainer.py:46-70) with QTable.train_net
 * inside the kernel.  Reinforce / ActorCritic transitions go to that agent's replay buffer; the
 * CALLER runs thrl_nn_reinforce_train / thrl_ac_train when len(memory) >= min_memory and must size n_episodes so that no
 * network update falls inside one call.  Replay buffers are rings [G][buf_len] per agent (game-major since ABI v3:
 * the 16-step flushes of the kernel and a game's whole batch in the update kernels are contiguous).
 */
typedef struct {
    int32_t kind[THRL_MAXA];             /* 0 = QTable, 1 = Reinforce, 2 = ActorCritic, 3 = CAC */
    const float* nn_params[THRL_MAXA];   /* device [G][P] for the neural agents              */
    double*  buf_price[THRL_MAXA];       /* device [G][buf_len] state  (trainer.py:62)       */
    int32_t* buf_action[THRL_MAXA];      /* CAC agents: the float32 action's bits             */
    double*  buf_reward[THRL_MAXA];
    double*  buf_nprice[THRL_MAXA];      /* next state                                       */
    double*  buf_scratch[THRL_MAXA];     /* QTable agents: old_value snapshot (agents.py:67) */
    int32_t  buf_len[THRL_MAXA];
    int32_t  min_memory[THRL_MAXA];
    int32_t  count[THRL_MAXA];           /* in/out: appends since the last memory.empty()    */
    /* per-game sweeps of the QTable agents and the env, as in thrl_buffers: device [N][G] (noise_prob
     * [G]) or NULL.  Rows of neural agents are ignored here (their gamma / entropy sweep goes to the
     * *_train calls). */
    const double* sweep_gamma;
    const double* sweep_alpha;
    const double* sweep_eps_end;
    const double* sweep_eps_step;
    double*       sweep_eps;             /* in/out: current epsilon per (agent, game)        */
    const double* sweep_noise_prob;
    /* Optional scratch for the POLICY TABLE (ABI v3): inside one call the networks are frozen and, in a game
     * whose agents are all discrete, the state after a step is a function of that step's action tuple, so the
     * kernel evaluates Reinforce.pi / ActorCritic.pi (agents.py:147-152) once per call and tuple and looks the
     * result up afterwards (identical bits).  Device, thrl_mixed_policy_table_bytes() bytes, contents need not
     * survive between calls; NULL = evaluate the policy at every step (same results, slower). */
    float*   policy_tab;
    size_t   policy_tab_bytes;
    int32_t  flags;                      /* THRL_MIXED_* */
} thrl_mixed;
/* Two-agent games whose neural agents are discrete run on the tuple-chain kernel (the state is carried as the action pair of
 * the last step, or as its price after a step with a redrawn intercept; needs policy_tab; per-game sweeps are taken): same
 * results as the general kernel.  This flag keeps the general one. */
#define THRL_MIXED_NO_TUPLE_KERNEL 1
/* bytes of thrl_mixed.policy_tab this configuration can use on this many games (0: the table does not apply --
 * no discrete neural agent, a continuous agent in the game, more than 2,048 action tuples, or a price grid small
 * enough for the in-LDS memo) */
size_t thrl_mixed_policy_table_bytes(const thrl_cfg* cfg, const thrl_mixed* mx);
int thrl_mixed_episodes(const thrl_cfg* cfg, thrl_mixed* mx, void* q, int32_t* counter, double* state,
                        thrl_run* run, double* game_reward_log, double* game_action_log, void* stream);

/* Philox draws for one lockstep step of a caller-driven loop: u [N][G] f64 uniforms and
 * choice [N][G] int8 indices (agents.py:81-82), optionally u2 [N][G] (the word behind `choice` as a
 * uniform: CAC's second Box-Muller input) and the env's two noise draws [G]
 * (environments.py:28-29); same streams/counters as thrl_qtable_episodes uses internally. */
int thrl_op_draws(const thrl_cfg* cfg, uint64_t seed, uint64_t game_offset, uint64_t episode, int32_t step,
                  double* u_out, int8_t* choice_out, double* u2_out, double* noise_u_out, double* noise_a_out,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* THRL_H */
