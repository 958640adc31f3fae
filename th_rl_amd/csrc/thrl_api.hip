// thrl_api.hip -- the extern "C" surface of libthrl_hip.so (see include/thrl.h).
// Host logic only: validation, kernel selection, launch geometry.  No torch types.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "thrl_kernels.h"
#include "thrl_wave_lut.h"

using namespace thrl;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(int e, const char* what) {
    return fail(THRL_ERR_HIP, "%s: %s", what, hipGetErrorString((hipError_t)e));
}

// ---- host restatements of the tiny scalar formulas (IEEE double/float; this
// file is built with -ffp-contract=off so they round exactly like the device).
double h_scale(int k, const thrl_cfg* c, int i) {
    double x = (double)k / ((double)c->n_actions[i] - 1.0);
    x = x * (c->act_hi[i] - c->act_lo[i]);
    return x + c->act_lo[i];
}
int h_encode64(double price, const thrl_cfg* c, int i) {
    return (int)rint(price / c->max_state[i] * (double)c->n_states[i]);
}
int h_encode32(double price, const thrl_cfg* c, int i) {
    float x = (float)price;
    x = x / (float)c->max_state[i];
    x = x * (float)c->n_states[i];
    return (int)rintf(x);
}
int validate(const thrl_cfg* c) {
    if (!c) return fail(THRL_ERR_NULL, "cfg is NULL");
    if (c->n_games < 1) return fail(THRL_ERR_BAD_CONFIG, "n_games=%d must be >= 1", c->n_games);
    if (c->n_agents < 1 || c->n_agents > THRL_MAXA)
        return fail(THRL_ERR_BAD_CONFIG, "n_agents=%d out of [1,%d]", c->n_agents, THRL_MAXA);
    if (c->max_steps < 1) return fail(THRL_ERR_BAD_CONFIG, "max_steps=%d must be >= 1", c->max_steps);
    if (c->q_dtype != 0 && c->q_dtype != 1) return fail(THRL_ERR_BAD_CONFIG, "q_dtype=%d", c->q_dtype);
    if (!(c->env_b != 0.0)) return fail(THRL_ERR_BAD_CONFIG, "env_b must be non-zero");
    for (int i = 0; i < c->n_agents; i++) {
        if (c->n_states[i] < 1 || c->n_states[i] > 32000)
            return fail(THRL_ERR_BAD_CONFIG, "agent %d: states=%d out of [1,32000]", i, c->n_states[i]);
        if (c->n_actions[i] < 2 || c->n_actions[i] > 32000)
            return fail(THRL_ERR_BAD_CONFIG, "agent %d: actions=%d out of [2,32000]", i, c->n_actions[i]);
        if (c->capacity[i] < 0 || c->min_memory[i] < 0)
            return fail(THRL_ERR_BAD_CONFIG, "agent %d: negative capacity/min_memory", i);
        if (!(c->max_state[i] > 0.0)) return fail(THRL_ERR_BAD_CONFIG, "agent %d: max_state must be > 0", i);
    }
    return THRL_OK;
}

void fill_agents(const thrl_cfg* c, AgentParams* ag, EnvParams* env) {
    int off = 0;
    for (int i = 0; i < THRL_MAXA; i++) {
        AgentParams p;
        memset(&p, 0, sizeof(p));
        if (i < c->n_agents) {
            p.rows = c->n_states[i] + 1;
            p.n_states = c->n_states[i];
            p.n_actions = c->n_actions[i];
            p.min_memory = c->min_memory[i];
            p.capacity = c->capacity[i];
            p.table_off = off;
            off += p.rows * p.n_actions;
            p.max_state = c->max_state[i];
            p.max_state_f = (float)c->max_state[i];
            p.inv_max_state = 1.0 / c->max_state[i];
            p.inv_max_state_f = 1.0f / p.max_state_f;
            p.gamma = c->gamma[i]; p.alpha = c->alpha[i];
            p.one_minus_alpha = 1.0 - c->alpha[i];
            p.gamma_f = (float)c->gamma[i]; p.alpha_f = (float)c->alpha[i];
            p.one_minus_alpha_f = (float)(1.0 - c->alpha[i]);
            p.alpha_gamma_f = p.alpha_f * p.gamma_f;
            p.eps_end = c->eps_end[i]; p.eps_step = c->eps_step[i];
            p.act_lo = c->act_lo[i]; p.act_span = c->act_hi[i] - c->act_lo[i];
            p.act_den = (double)c->n_actions[i] - 1.0;
        }
        ag[i] = p;
    }
    if (env) {
        env->a = c->env_a; env->b = c->env_b; env->ratio = c->env_a / c->env_b;
        env->noise_prob = c->noise_prob; env->noise_lo = c->env_a * 0.7;
    }
}

// largest number of buffered transitions any agent can hold (bounds the ring)
int eff_capacity(const thrl_cfg* c, int i) {
    const int cap = c->capacity[i], mm = c->min_memory[i], T = c->max_steps;
    if (cap <= 0) return 0;
    if (cap < mm) return cap;                        // never trains, ring wraps at cap
    const long need = (long)T * ((mm + T - 1) / T > 0 ? (mm + T - 1) / T : 1);
    return (int)(need < cap ? need : cap);
}
int capmax_of(const thrl_cfg* c) {
    int m = 1;
    for (int i = 0; i < c->n_agents; i++) { int e = eff_capacity(c, i); if (e > m) m = e; }
    return m;
}

struct WavePlan {
    bool ok;
    char why[200];
    int row_lo, win_rows;
    int epk, replay_from;           // training cycle: episodes per train_net that trains, first kept transition
    int lut_bytes, game_lds_bytes, waves_per_block, blocks_per_cu;
};

// Per-device figures the launch geometry needs.  A read-only cache keyed by the HIP device id (filled
// once per device under a mutex; the only process-wide state besides the thread-local error string).
// Without a visible device (the CPU-side tests of the host logic) the MI355X figures are assumed.
struct DevInfo { int cus, lds_per_cu, waves_per_cu; };
constexpr DevInfo kMi355x = {256, 160 * 1024, 32};
constexpr int kMaxDevices = 64;

DevInfo dev_info() {
    static std::mutex mu;
    static DevInfo cache[kMaxDevices];
    static bool have[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        (void)hipGetLastError();
        return kMi355x;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!have[dev]) {
        DevInfo d = kMi355x;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return kMi355x; }
        if (prop.multiProcessorCount > 0) d.cus = prop.multiProcessorCount;       // 256; 32 per XCD partition
        if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) d.lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
        if (prop.maxThreadsPerMultiProcessor >= 64) d.waves_per_cu = prop.maxThreadsPerMultiProcessor / 64;
        cache[dev] = d;
        have[dev] = true;
    }
    return cache[dev];
}

// THRL_WAVE_MAX_WAVES_PER_CU: tuning / diagnostic knob, read ONCE per process (not per call)
int env_wave_cap() {
    static std::once_flag once;
    static int cap = 0;
    std::call_once(once, [] {
        if (const char* e = getenv("THRL_WAVE_MAX_WAVES_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 64) cap = v; }
    });
    return cap;
}

// Can the fused wave kernel run this config?  (DESIGN.md "wave kernel: eligibility")
WavePlan plan_wave(const thrl_cfg* c, const thrl_run* run, bool injected) {
    WavePlan p;
    memset(&p, 0, sizeof(p));
#define NO(msg) do { snprintf(p.why, sizeof(p.why), "%s", msg); return p; } while (0)
    if (c->n_agents != 2) NO("needs exactly 2 agents");
    if (c->n_states[0] != c->n_states[1] || c->n_actions[0] != c->n_actions[1] ||
        c->max_state[0] != c->max_state[1]) NO("agents must share the state/action grid sizes");
    const int A = c->n_actions[0], T = c->max_steps;
    if (A > 32) NO("actions > 32");
    // Training cycle (buffers.py:12-19, agents.py:60,77): after every episode len(memory) = min(appends,
    // capacity); train_net trains -- and empties the buffer -- once that reaches min_memory, i.e. every
    // epk = ceil(min_memory / T)-th episode, on the last min(epk*T, capacity) transitions.  Both agents
    // must share the cycle and the kept suffix (or both never train: capacity < min_memory).
    int epk = 0, keep = 0;
    for (int i = 0; i < 2; i++) {
        const int cap = c->capacity[i], mm = c->min_memory[i] > 0 ? c->min_memory[i] : 1;
        const int k_i = cap >= mm && cap > 0 ? (mm + T - 1) / T : 0;           // 0: never trains
        const int keep_i = k_i ? (k_i * T < cap ? k_i * T : cap) : 0;
        if (i == 0) { epk = k_i; keep = keep_i; }
        else if (k_i != epk || keep_i != keep) NO("the agents' replay buffers fill / train on different cycles");
        if (run && run->mem_count[i] != 0) NO("non-empty replay memory on entry");
    }
    const bool never = epk == 0;
    if (never) epk = 1;
    if ((long)epk * T > 256) NO("more than 256 transitions per training cycle");
    if (epk > kWaveMaxEpisodes) NO("more than 32 episodes per training cycle");
    if (run && run->n_episodes % epk != 0) NO("episodes of this call are not a multiple of the training cycle");
    p.epk = epk;
    p.replay_from = never ? epk * T : epk * T - keep;
    // Row window = every row a step can land in: both encodes (play: float32, train: float64)
    // of the price on the whole action grid; with noise the intercept ranges over [0.7a, a).
    int lo = 1 << 30, hi = -1;
    const double ratio = c->env_a / c->env_b;
    for (int a0 = 0; a0 < A; a0++)
        for (int a1 = 0; a1 < A; a1++) {
            double Q = 0.0;
            Q = Q + ratio * h_scale(a0, c, 0);
            Q = Q + ratio * h_scale(a1, c, 1);
            const double intercepts[2] = {c->env_a, c->env_a * 0.7};
            for (int z = 0; z < (c->noise_prob > 0.0 ? 2 : 1); z++) {
                double price = intercepts[z] - c->env_b * Q;
                if (!(price > 0.0)) price = 0.0;
                const int r64 = h_encode64(price, c, 0), r32 = h_encode32(price, c, 0);
                if (r64 < 0 || r64 > c->n_states[0] || r32 < 0 || r32 > c->n_states[0])
                    NO("price outside the table on the action grid");
                if (r64 < lo) lo = r64;
                if (r32 < lo) lo = r32;
                if (r64 > hi) hi = r64;
                if (r32 > hi) hi = r32;
            }
        }
    p.row_lo = lo;
    p.win_rows = hi - lo + 1;
    if (p.win_rows + 2 > 128) NO("reachable row window > 126 rows");
    const WaveLut L = wave_lut_layout(A);
    p.lut_bytes = L.lds_bytes;          // LDS-staged part of the LUT image
    p.game_lds_bytes = 2 * (p.win_rows + 2) * A * (c->q_dtype == 1 ? 8 : 4);
    if (c->q_dtype == 0 && !(c->noise_prob > 0.0)) p.game_lds_bytes += 256;       // per-step words of half a segment (thrl_wave_kernel.h kLdsMK)
    // choose waves/block to maximise resident waves per CU (LDS-bound); a block may take the whole CU's LDS
    int best_w = 0, best_total = 0, best_b = 0;
    const DevInfo dv = dev_info();
    int cap_waves = dv.waves_per_cu;
    if (env_wave_cap() > 0 && env_wave_cap() < cap_waves) cap_waves = env_wave_cap();
    // Most resident waves wins (throughput is latency-bound and scales with them).  A block's waves
    // are dealt to the CU's 4 SIMDs in turn, so among equal totals blocks of 4k waves are preferred
    // (they load the SIMDs evenly); with games handed out dynamically an uneven split only costs the
    // crowded SIMD's waves some speed (noise window: 3 x 5 waves measured 12 % faster than 3 x 4).
    // register limit: the float32 variants are compiled for 5 waves/SIMD (4 with noise or T > 128),
    // the float64 ones (twice the LDS per game) for 3
    const int reg_waves = 4 * (c->q_dtype == 1 ? 3 : ((c->noise_prob > 0.0 || c->max_steps * p.epk > 128) ? 4 : 5));
    if (cap_waves > reg_waves) cap_waves = reg_waves;
    for (int w = 1; w <= 16; w++) {
        const int lds = p.lut_bytes + w * p.game_lds_bytes;
        if (lds > dv.lds_per_cu) break;
        int b = dv.lds_per_cu / (((lds + 511) / 512) * 512);
        if (b * w > cap_waves) b = cap_waves / w;
        if (b < 1) continue;
        const int total = b * w;
        const bool even = (w & 3) == 0, best_even = best_w > 0 && (best_w & 3) == 0;
        if (total > best_total || (total == best_total && ((even && !best_even) || (even == best_even && w < best_w)))) {
            best_total = total; best_w = w; best_b = b;
        }
    }
    if (best_w == 0) NO("table window does not fit LDS");
    p.waves_per_block = best_w;
    p.blocks_per_cu = best_b;
    p.ok = true;
    return p;
#undef NO
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- the tuple-chain kernel (thrl_tuple_kernel.h): 1-4 QTable agents with individual grids, state = action tuple
struct TuplePlan {
    bool ok;
    char why[200];
    TupleArgs a;                    // geometry fields filled: tuples, windows, LDS layout, LUT offsets
    int waves_per_block, blocks_per_cu;
    size_t lut_image_bytes;
};
constexpr size_t kTupleWsBytes = 160 * 1024;       // LUT image (< 128 KiB) + the launch's work counter in the last 64 bytes;
                                                   // the visit log of the resident waves follows (tuple_workspace)

TuplePlan plan_tuple(const thrl_cfg* c, const thrl_run* run) {
    TuplePlan p;
    memset(&p, 0, sizeof(p));
#define NO(msg) do { snprintf(p.why, sizeof(p.why), "%s", msg); return p; } while (0)
    const int N = c->n_agents, T = c->max_steps;
    if (N < 1 || N > kTupMaxN) NO("more than 4 agents");
    if (T > 256) NO("more than 256 steps per episode");
    long tuples = 1;
    for (int i = 0; i < N; i++) {
        if (c->n_actions[i] > 64) NO("more than 64 actions");
        tuples *= c->n_actions[i];
        if (tuples > kTupMaxTuples) NO("more than 4,096 action tuples");
        // train_net must train -- and empty the buffer -- after every episode, on exactly that episode's transitions
        // (buffers.py:12-19, agents.py:60,77): min_memory <= T <= capacity
        if (!(c->min_memory[i] <= T && T <= c->capacity[i])) NO("replay buffer does not fill / train once per episode");
        if (run && run->mem_count[i] != 0) NO("non-empty replay memory on entry");
    }
    TupleArgs& a = p.a;
    a.N = N; a.T = T; a.tuples = (int)tuples;
    // row windows: both encodes of the price of every tuple; with noise the intercept ranges over [0.7a, a]
    // (environments.py:29-31) and both encodes are monotonic in the price, so the two ends cover the range
    int lo[kTupMaxN], hi[kTupMaxN];
    for (int i = 0; i < N; i++) { lo[i] = 1 << 30; hi[i] = -1; }
    const double ratio = c->env_a / c->env_b;
    int digit[kTupMaxN] = {0, 0, 0, 0};
    for (long t = 0; t < tuples; t++) {
        double Q = 0.0;
        for (int i = 0; i < N; i++) Q = Q + ratio * h_scale(digit[i], c, i);
        const double intercepts[2] = {c->env_a, c->env_a * 0.7};
        for (int v = 0; v < (c->noise_prob > 0.0 ? 2 : 1); v++) {
            double price = intercepts[v] - c->env_b * Q;
            if (!(price > 0.0)) price = 0.0;
            for (int i = 0; i < N; i++) {
                const int r64 = h_encode64(price, c, i), r32 = h_encode32(price, c, i);
                if (r64 < 0 || r64 > c->n_states[i] || r32 < 0 || r32 > c->n_states[i]) NO("price outside a table on the action grid");
                lo[i] = r64 < lo[i] ? r64 : lo[i]; lo[i] = r32 < lo[i] ? r32 : lo[i];
                hi[i] = r64 > hi[i] ? r64 : hi[i]; hi[i] = r32 > hi[i] ? r32 : hi[i];
            }
        }
        for (int i = N - 1; i >= 0; i--) { if (++digit[i] < c->n_actions[i]) break; digit[i] = 0; }     // last agent = fastest digit
    }
    const int esz = c->q_dtype == 1 ? 8 : 4;
    int elems = 0, am = 0, hd = 0;
    for (int i = 0; i < N; i++) {
        a.row_lo[i] = lo[i]; a.win_rows[i] = hi[i] - lo[i] + 1;
        if (a.win_rows[i] + 2 > 256) NO("reachable row window > 254 rows");
        const int cells = (a.win_rows[i] + 2) * c->n_actions[i];
        if (cells > 65535) NO("more than 65,535 resident table cells per agent");
        a.tab_off[i] = elems; elems += (cells + 3) & ~3;
        a.am_off_i[i] = am; am += (a.win_rows[i] + 2 + 3) & ~3;
        a.hist_off_i[i] = hd; hd += (cells + 1) / 2;
    }
    a.am_off = (int)align_up((size_t)elems * esz, 16);
    a.g_off = (int)align_up((size_t)a.am_off + am, 16);
    a.hist_dwords = hd;
    if ((size_t)hd * 4 > (size_t)elems * esz) NO("visit histogram does not fit the table region");      // (never: u16 per cell)
    a.game_lds_bytes = (int)align_up((size_t)a.g_off + 2 * ((size_t)tuples + 1), 16);
    // action words: agent i's action in a bit field of ceil(log2 A_i) bits; the widths sum to < log2(tuples) + N <= 16
    // (the fields start at bit 1 and the LAST agent's comes first: its field, masked, is its action x 2 = the byte offset of
    // its term in a u16 table indexed by the tuple -- the play chain of thrl_tuple_kernel.h adds it without a multiply)
    int sh = 1;
    for (int i = N - 1; i >= 0; i--) {
        int bits = 0;
        while ((1 << bits) < c->n_actions[i]) bits++;
        a.act_sh[i] = sh; a.act_bits[i] = bits; sh += bits;
    }
    if (sh > 16) NO("action word wider than 16 bits");
    a.vlog_wave_bytes = (int64_t)kTupMaxEpisodes * T * (N <= 2 ? 4 : 8);
    a.aq_off = (int)align_up((size_t)tuples * 8, 16);           // prow [tuples] u32, trow [tuples] u32
    a.lut_lds_bytes = a.aq_off + N * 64 * 8 * 2;
    a.price_off = a.lut_lds_bytes;
    a.qsum_off = a.price_off + 8 * (int)tuples;
    p.lut_image_bytes = (size_t)a.qsum_off + 8 * (size_t)tuples;
    if (p.lut_image_bytes > kTupleWsBytes - 64) NO("LUT image too large");
    const DevInfo dv = dev_info();
    int cap_waves = dv.waves_per_cu < 16 ? dv.waves_per_cu : 16;         // the kernel is compiled for 4 waves per SIMD
    int best_w = 0, best_total = 0, best_b = 0;
    for (int w = 1; w <= 16; w++) {
        const int lds = a.lut_lds_bytes + w * a.game_lds_bytes;
        if (lds > dv.lds_per_cu) break;
        int b = dv.lds_per_cu / (((lds + 511) / 512) * 512);
        if (b * w > cap_waves) b = cap_waves / w;
        if (b < 1) continue;
        const int total = b * w;
        const bool even = (w & 3) == 0, best_even = best_w > 0 && (best_w & 3) == 0;
        if (total > best_total || (total == best_total && ((even && !best_even) || (even == best_even && w < best_w)))) {
            best_total = total; best_w = w; best_b = b;
        }
    }
    if (best_w == 0) NO("tables of one game do not fit LDS");
    p.waves_per_block = best_w; p.blocks_per_cu = best_b;
    p.ok = true;
    return p;
#undef NO
}

// Workspace of the tuple-chain kernel: [0, kTupleWsBytes) LUT image + work counter, then the visit log of every resident wave
struct TupleWs { int grid, total_waves; size_t bytes; };
TupleWs tuple_workspace(const thrl_cfg* c, const TuplePlan& p) {
    TupleWs w;
    w.grid = (c->n_games + p.waves_per_block - 1) / p.waves_per_block;
    const int max_grid = dev_info().cus * p.blocks_per_cu;
    if (w.grid > max_grid) w.grid = max_grid;
    w.total_waves = w.grid * p.waves_per_block;
    w.bytes = kTupleWsBytes + align_up((size_t)w.total_waves * (size_t)p.a.vlog_wave_bytes, 256);
    return w;
}

constexpr size_t kLutRegion = 16384;            // workspace bytes reserved for the LUT image

// Workspace of the wave kernel, sized from the config and the current device:
//   [0, kLutRegion)   payoff-LUT image (+ the launch's work counter in its last 64 bytes)
//   partial           [total_waves][kWaveMaxEpisodes][4] fixed-point log sums
//   tlog              [total_waves][kWaveMaxEpisodes][NSEG][64] packed transitions (visit counters)
// total_waves = the persistent grid: resident blocks per CU x CUs, or fewer when G is small.
struct WaveWs { int grid, total_waves; size_t partial_off, tlog_off, bytes; };
WaveWs wave_workspace(const thrl_cfg* c, const WavePlan& p) {
    WaveWs w;
    w.grid = (c->n_games + p.waves_per_block - 1) / p.waves_per_block;
    const int max_grid = dev_info().cus * p.blocks_per_cu;
    if (w.grid > max_grid) w.grid = max_grid;
    w.total_waves = w.grid * p.waves_per_block;
    const size_t nseg = (size_t)(c->max_steps * p.epk + 63) / 64;
    w.partial_off = kLutRegion;
    w.tlog_off = w.partial_off + align_up((size_t)w.total_waves * 4 * kWaveMaxEpisodes * sizeof(long long), 256);
    w.bytes = w.tlog_off + align_up((size_t)w.total_waves * kWaveMaxEpisodes * nseg * 64 * sizeof(uint32_t), 256);
    return w;
}

}  // namespace

extern "C" {

int thrl_version(void) { return THRL_ABI_VERSION; }
const char* thrl_last_error(void) { return g_err; }

// th_rl_amd/build.py passes the ablation mask of the build and a hash of the sources
#ifndef THRL_BUILD_ABLATE
#define THRL_BUILD_ABLATE 0
#endif
#ifndef THRL_SRC_HASH
#define THRL_SRC_HASH "unknown"
#endif
#ifndef THRL_WAVE_HASH
#define THRL_WAVE_HASH "unknown"
#endif
#ifndef THRL_NN_HASH
#define THRL_NN_HASH "unknown"
#endif
#define THRL_STR2(x) #x
#define THRL_STR(x) THRL_STR2(x)
const char* thrl_build_info(void) {
    return "abi=" THRL_STR(THRL_ABI_VERSION) ";ablate=" THRL_STR(THRL_BUILD_ABLATE) ";src=" THRL_SRC_HASH
           ";wave=" THRL_WAVE_HASH ";nn=" THRL_NN_HASH;
}
int thrl_ablate_mask(void) { return THRL_BUILD_ABLATE; }

size_t thrl_table_stride(const thrl_cfg* c) {
    if (!c || c->n_agents < 1 || c->n_agents > THRL_MAXA) return 0;
    size_t s = 0;
    for (int i = 0; i < c->n_agents; i++) s += (size_t)(c->n_states[i] + 1) * (size_t)c->n_actions[i];
    return s;
}
size_t thrl_table_offset(const thrl_cfg* c, int agent) {
    if (!c || agent < 0 || agent > c->n_agents) return 0;
    size_t s = 0;
    for (int i = 0; i < agent; i++) s += (size_t)(c->n_states[i] + 1) * (size_t)c->n_actions[i];
    return s;
}
size_t thrl_replay_mem_bytes(const thrl_cfg* c) {
    if (validate(c) != THRL_OK) return 0;
    const size_t n = (size_t)capmax_of(c) * (size_t)c->n_agents * (size_t)c->n_games;
    return align_up(n * 2, 256) * 3 + align_up(n * 8, 256) * 2;
}
size_t thrl_workspace_bytes(const thrl_cfg* c) {
    if (validate(c) != THRL_OK) return 0;
    const WavePlan p = plan_wave(c, nullptr, false);
    const TuplePlan tp = plan_tuple(c, nullptr);
    const size_t tuple_ws = tp.ok ? tuple_workspace(c, tp).bytes : kLutRegion;       // (the generic kernel keeps nothing there)
    if (!p.ok) return tuple_ws;
    const size_t wave_ws = wave_workspace(c, p).bytes;                               // (either kernel can be asked for by id)
    return wave_ws > tuple_ws ? wave_ws : tuple_ws;
}
int thrl_select_kernel(const thrl_cfg* c, int injected) {
    if (validate(c) != THRL_OK) return THRL_ERR_BAD_CONFIG;
    const WavePlan p = plan_wave(c, nullptr, injected != 0);
    if (!p.ok) {
        const TuplePlan tp = plan_tuple(c, nullptr);
        if (tp.ok) return THRL_KERNEL_TUPLE;
        snprintf(g_err, sizeof(g_err), "generic kernel: %s; %s", p.why, tp.why);
        return THRL_KERNEL_GENERIC;
    }
    return THRL_KERNEL_WAVE;
}

int thrl_training_cycle(const thrl_cfg* c) {
    if (validate(c) != THRL_OK) return 0;
    const WavePlan p = plan_wave(c, nullptr, false);
    return p.ok ? p.epk : 0;
}

int thrl_qtable_init(const thrl_cfg* c, void* q, int32_t* counter, double* state, uint64_t seed,
                     uint64_t game_offset, const double* sweep_gamma, void* stream) {
    int rc = validate(c);
    if (rc) return rc;
    if (!q || !state) return fail(THRL_ERR_NULL, "q/state is NULL");
    InitArgs a;
    memset(&a, 0, sizeof(a));
    a.G = c->n_games; a.N = c->n_agents; a.stride = (int64_t)thrl_table_stride(c); a.env_a = c->env_a;
    fill_agents(c, a.ag, nullptr);
    a.q = q; a.counter = counter; a.state = state; a.seed = seed; a.game_offset = game_offset;
    a.sw_gamma = sweep_gamma;
    const int e = launch_init(a, c->q_dtype, (hipStream_t)stream);
    return e ? hip_fail(e, "k_init launch") : THRL_OK;
}

static int run_generic(const thrl_cfg* c, const thrl_buffers* b, thrl_run* run, hipStream_t s) {
    if ((b->sweep_eps_end || b->sweep_eps_step) && !b->sweep_eps)
        return fail(THRL_ERR_NULL, "sweep_eps_end / sweep_eps_step need the per-game epsilon state sweep_eps");
    if (b->sweep_noise_prob && !(c->noise_prob > 0.0))
        return fail(THRL_ERR_BAD_CONFIG, "sweep_noise_prob needs cfg.noise_prob > 0 (it switches the noise draws on)");
    const int capmax = capmax_of(c);
    const size_t need = thrl_replay_mem_bytes(c);
    if (!b->replay_mem || b->replay_mem_bytes < need)
        return fail(THRL_ERR_WORKSPACE, "replay_mem too small: have %zu need %zu", b->replay_mem_bytes, need);
    const bool injected = b->inj_u != nullptr;
    if (injected && !b->inj_choice) return fail(THRL_ERR_NULL, "inj_u given without inj_choice");
    if (injected && c->noise_prob > 0.0 && (!b->inj_noise_u || !b->inj_noise_a))
        return fail(THRL_ERR_NULL, "noise_prob > 0 with injected draws needs inj_noise_u/inj_noise_a");
    GenericArgs a;
    memset(&a, 0, sizeof(a));
    a.G = c->n_games; a.N = c->n_agents; a.T = c->max_steps; a.capmax = capmax;
    a.stride = (int64_t)thrl_table_stride(c);
    fill_agents(c, a.ag, &a.env);
    a.q = b->q; a.counter = b->counter; a.state = b->state;
    {
        const size_t n = (size_t)capmax * (size_t)c->n_agents * (size_t)c->n_games;
        char* base = (char*)b->replay_mem;
        const size_t s2 = align_up(n * 2, 256), s8 = align_up(n * 8, 256);
        a.mem.s = (int16_t*)base; a.mem.ns = (int16_t*)(base + s2); a.mem.a = (int16_t*)(base + 2 * s2);
        a.mem.r = (double*)(base + 3 * s2); a.mem.ov = (double*)(base + 3 * s2 + s8);
    }
    a.sum_reward = b->reward_log; a.sum_action = b->action_log;
    if ((a.sum_reward == nullptr) != (a.sum_action == nullptr))
        return fail(THRL_ERR_NULL, "reward_log and action_log must both be given or both NULL");
    a.game_reward_log = b->game_reward_log; a.game_action_log = b->game_action_log;
    a.inj_u = b->inj_u; a.inj_choice = b->inj_choice; a.inj_noise_u = b->inj_noise_u; a.inj_noise_a = b->inj_noise_a;
    a.seed = run->seed; a.game_offset = run->game_offset; a.first_episode = run->first_episode;
    a.n_episodes = run->n_episodes;
    for (int i = 0; i < THRL_MAXA; i++) { a.eps0[i] = run->eps[i]; a.cnt0[i] = run->mem_count[i]; }
    a.sw_gamma = b->sweep_gamma; a.sw_alpha = b->sweep_alpha; a.sw_eps_end = b->sweep_eps_end;
    a.sw_eps_step = b->sweep_eps_step; a.sw_eps = b->sweep_eps; a.sw_noise_prob = b->sweep_noise_prob;
    const int nlog = run->n_episodes * c->n_agents;
    if (a.sum_reward) {
        hipError_t e1 = hipMemsetAsync(a.sum_reward, 0, sizeof(double) * nlog, s);
        hipError_t e2 = hipMemsetAsync(a.sum_action, 0, sizeof(double) * nlog, s);
        if (e1 != hipSuccess || e2 != hipSuccess) return hip_fail(e1 != hipSuccess ? e1 : e2, "log memset");
    }
    int e = launch_generic(a, c->q_dtype, s);
    if (e) return hip_fail(e, "k_generic_episodes launch");
    if (a.sum_reward) {
        e = launch_finalize_logs(a.sum_reward, a.sum_action, nlog, c->n_games, s);
        if (e) return hip_fail(e, "k_finalize_logs launch");
    }
    // host mirror of the per-episode bookkeeping that is identical for all games
    for (int ep = 0; ep < run->n_episodes; ep++)
        for (int i = 0; i < c->n_agents; i++) {
            const int cap = c->capacity[i];
            if (cap > 0)
                for (int t = 0; t < c->max_steps; t++) {
                    run->mem_count[i] += 1;
                    if (run->mem_count[i] >= 2 * cap) run->mem_count[i] -= cap;
                }
            const int len = run->mem_count[i] < cap ? run->mem_count[i] : cap;
            if (len >= c->min_memory[i]) run->mem_count[i] = 0;
            run->eps[i] = c->eps_end[i] + (run->eps[i] - c->eps_end[i]) * c->eps_step[i];
        }
    run->kernel_used = THRL_KERNEL_GENERIC;
    return THRL_OK;
}

static int run_wave(const thrl_cfg* c, const thrl_buffers* b, thrl_run* run, const WavePlan& p, int force_variant, hipStream_t s) {
    if (b->inj_u && !b->inj_choice) return fail(THRL_ERR_NULL, "inj_u given without inj_choice");
    if (b->inj_u && c->noise_prob > 0.0 && (!b->inj_noise_u || !b->inj_noise_a))
        return fail(THRL_ERR_NULL, "noise_prob > 0 with injected draws needs inj_noise_u/inj_noise_a");
    if (!b->workspace || b->workspace_bytes < thrl_workspace_bytes(c))
        return fail(THRL_ERR_WORKSPACE, "workspace too small: have %zu need %zu", b->workspace_bytes,
                    thrl_workspace_bytes(c));
    if (wave_lut_layout(c->n_actions[0]).bytes > (int)kLutRegion - 64) return fail(THRL_ERR_UNSUPPORTED, "LUT image too large");
    WaveArgs a;
    memset(&a, 0, sizeof(a));
    a.G = c->n_games; a.T = c->max_steps; a.A = c->n_actions[0]; a.rows = c->n_states[0] + 1;
    a.row_lo = p.row_lo; a.win_rows = p.win_rows;
    a.epk = p.epk; a.replay_from = p.replay_from;
    a.waves_per_block = p.waves_per_block;
    a.force_variant = force_variant;
    a.lut_bytes = p.lut_bytes; a.game_lds_bytes = p.game_lds_bytes;
    a.stride = (int64_t)thrl_table_stride(c);
    AgentParams ag[THRL_MAXA];
    fill_agents(c, ag, &a.env);
    a.ag[0] = ag[0]; a.ag[1] = ag[1];
    a.q = b->q; a.counter = b->counter; a.state = b->state;
    unsigned char* lut = (unsigned char*)b->workspace;
    a.lut_ns = lut;
    const WaveWs ws = wave_workspace(c, p);
    a.partial = (long long*)((char*)b->workspace + ws.partial_off);
    {   // fixed-point scales of the log sums: 2^s with G * (bound of one game's episode mean) * 2^s <= 2^62
        double hi = 0.0;
        for (int i = 0; i < 2; i++) hi = fmax(hi, fmax(fabs(c->act_lo[i]), fabs(c->act_hi[i])));
        const double bound[2] = {fmax(c->env_a * (c->env_a / c->env_b) * hi, 1e-300), fmax(hi, 1e-300)};
        for (int k = 0; k < 2; k++) {
            int ex = 0;
            frexp((double)c->n_games * bound[k], &ex);          // G*bound < 2^ex
            int sh = 62 - ex;
            sh = sh > 60 ? 60 : (sh < 0 ? 0 : sh);
            a.log_scale[k] = ldexp(1.0, sh);
        }
    }
    a.tlog = (uint32_t*)((char*)b->workspace + ws.tlog_off);
    a.next_game = (int32_t*)((char*)b->workspace + kLutRegion - 64);       // the LUT image is < 16 KiB - 64
    a.seed = run->seed; a.game_offset = run->game_offset;
    a.sw_gamma = b->sweep_gamma; a.sw_alpha = b->sweep_alpha; a.sw_eps_end = b->sweep_eps_end;
    a.sw_eps_step = b->sweep_eps_step; a.sw_eps = b->sweep_eps; a.sw_noise_prob = b->sweep_noise_prob;
    if ((b->sweep_eps_end || b->sweep_eps_step) && !b->sweep_eps)
        return fail(THRL_ERR_NULL, "sweep_eps_end / sweep_eps_step need the per-game epsilon state sweep_eps");
    if (b->sweep_noise_prob && !(c->noise_prob > 0.0))
        return fail(THRL_ERR_BAD_CONFIG, "sweep_noise_prob needs cfg.noise_prob > 0 (it sizes the row window)");
    const int chunk_max = kWaveMaxEpisodes / p.epk * p.epk;        // whole training cycles per launch
    if (force_variant == 2 && (p.epk > 1 || p.replay_from > 0 || c->noise_prob > 0.0 || b->sweep_gamma || b->sweep_alpha ||
                               b->sweep_eps_end || b->sweep_eps_step || b->sweep_eps || b->sweep_noise_prob))
        return fail(THRL_ERR_UNSUPPORTED, "THRL_KERNEL_WAVE_GREEDY: no greedy-regime variant for noise, sweeps or "
                                          "multi-episode training cycles");

    const int block = p.waves_per_block * 64;
    const int grid = ws.grid;                             // persistent grid; games are handed out by a work counter
    a.total_waves = ws.total_waves;
    const size_t lds = (size_t)p.lut_bytes + (size_t)p.waves_per_block * p.game_lds_bytes;

    int e = launch_wave_lut(a, lut, s);
    if (e) return hip_fail(e, "k_wave_lut launch");
    int done = 0;
    while (done < run->n_episodes) {
        const int n = run->n_episodes - done < chunk_max ? run->n_episodes - done : chunk_max;
        a.n_episodes = n;
        a.first_episode = run->first_episode + (uint64_t)done;
        if (b->inj_u) {                                  // parity mode: this chunk's slice of the draws
            const size_t per_ep = (size_t)c->max_steps * 2 * (size_t)c->n_games;
            a.inj_u = b->inj_u + (size_t)done * per_ep;
            a.inj_choice = b->inj_choice + (size_t)done * per_ep;
            if (c->noise_prob > 0.0) {
                a.inj_noise_u = b->inj_noise_u + (size_t)done * (per_ep / 2);
                a.inj_noise_a = b->inj_noise_a + (size_t)done * (per_ep / 2);
            }
        }
        for (int ep = 0; ep < n; ep++)
            for (int i = 0; i < 2; i++) {
                a.eps[ep][i] = run->eps[i];
                run->eps[i] = c->eps_end[i] + (run->eps[i] - c->eps_end[i]) * c->eps_step[i];
            }
        if (hipMemsetAsync(a.next_game, 0, sizeof(int32_t), s) != hipSuccess) return hip_fail((int)hipGetLastError(), "hipMemsetAsync");
        e = launch_wave(a, c->q_dtype, grid, block, lds, s);
        if (e) return hip_fail(e, "k_wave_episodes launch");
        if (b->reward_log || b->action_log) {
            e = launch_wave_reduce(a.partial, a.log_scale, a.total_waves, n, c->n_games,
                                   b->reward_log ? b->reward_log + (size_t)done * 2 : nullptr,
                                   b->action_log ? b->action_log + (size_t)done * 2 : nullptr, s);
            if (e) return hip_fail(e, "k_wave_reduce launch");
        }
        done += n;
    }
    run->kernel_used = THRL_KERNEL_WAVE;
    return THRL_OK;
}

static int run_tuple(const thrl_cfg* c, const thrl_buffers* b, thrl_run* run, TuplePlan& p, hipStream_t s) {
    if (b->inj_u && !b->inj_choice) return fail(THRL_ERR_NULL, "inj_u given without inj_choice");
    if (b->inj_u && c->noise_prob > 0.0 && (!b->inj_noise_u || !b->inj_noise_a))
        return fail(THRL_ERR_NULL, "injected draws with noise_prob > 0 need inj_noise_u / inj_noise_a");
    const TupleWs ws = tuple_workspace(c, p);
    if (!b->workspace || b->workspace_bytes < ws.bytes)
        return fail(THRL_ERR_WORKSPACE, "workspace too small: have %zu need %zu", b->workspace_bytes, ws.bytes);
    if ((b->reward_log == nullptr) != (b->action_log == nullptr))
        return fail(THRL_ERR_NULL, "reward_log and action_log must both be given or both NULL");
    TupleArgs& a = p.a;
    a.G = c->n_games;
    a.stride = (int64_t)thrl_table_stride(c);
    AgentParams ag[THRL_MAXA];
    fill_agents(c, ag, &a.env);
    for (int i = 0; i < a.N; i++) a.ag[i] = ag[i];
    a.q = b->q; a.counter = b->counter; a.state = b->state;
    a.sw_gamma = b->sweep_gamma; a.sw_alpha = b->sweep_alpha; a.sw_eps_end = b->sweep_eps_end;
    a.sw_eps_step = b->sweep_eps_step; a.sw_eps = b->sweep_eps; a.sw_noise_prob = b->sweep_noise_prob;
    unsigned char* lut = (unsigned char*)b->workspace;
    a.lut = lut;
    a.next_game = (int32_t*)((char*)b->workspace + kTupleWsBytes - 64);
    a.vlog = (char*)b->workspace + kTupleWsBytes;
    a.seed = run->seed; a.game_offset = run->game_offset;
    a.waves_per_block = p.waves_per_block;
    const int grid = ws.grid;
    a.total_waves = ws.total_waves;
    const int block = p.waves_per_block * 64;
    const size_t lds = (size_t)a.lut_lds_bytes + (size_t)p.waves_per_block * a.game_lds_bytes;
    int e = launch_tuple_lut(a, lut, s);
    if (e) return hip_fail(e, "k_tuple_lut launch");
    const int nlog = run->n_episodes * c->n_agents;
    if (b->reward_log) {
        hipError_t e1 = hipMemsetAsync(b->reward_log, 0, sizeof(double) * nlog, s);
        hipError_t e2 = hipMemsetAsync(b->action_log, 0, sizeof(double) * nlog, s);
        if (e1 != hipSuccess || e2 != hipSuccess) return hip_fail(e1 != hipSuccess ? e1 : e2, "log memset");
    }
    int done = 0;
    while (done < run->n_episodes) {
        const int n = run->n_episodes - done < kTupMaxEpisodes ? run->n_episodes - done : kTupMaxEpisodes;
        a.n_episodes = n;
        a.first_episode = run->first_episode + (uint64_t)done;
        a.sum_reward = b->reward_log ? b->reward_log + (size_t)done * a.N : nullptr;
        a.sum_action = b->action_log ? b->action_log + (size_t)done * a.N : nullptr;
        if (b->inj_u) {
            const size_t per_ep = (size_t)c->max_steps * a.N * (size_t)c->n_games;
            a.inj_u = b->inj_u + (size_t)done * per_ep;
            a.inj_choice = b->inj_choice + (size_t)done * per_ep;
            if (c->noise_prob > 0.0) {
                a.inj_noise_u = b->inj_noise_u + (size_t)done * c->max_steps * (size_t)c->n_games;
                a.inj_noise_a = b->inj_noise_a + (size_t)done * c->max_steps * (size_t)c->n_games;
            }
        }
        for (int ep = 0; ep < n; ep++)
            for (int i = 0; i < a.N; i++) {
                a.eps[ep][i] = run->eps[i];
                run->eps[i] = c->eps_end[i] + (run->eps[i] - c->eps_end[i]) * c->eps_step[i];       // agents.py:78
            }
        if (hipMemsetAsync(a.next_game, 0, sizeof(int32_t), s) != hipSuccess) return hip_fail((int)hipGetLastError(), "hipMemsetAsync");
        e = launch_tuple(a, c->q_dtype, grid, block, lds, s);
        if (e) return hip_fail(e, "k_tuple_episodes launch");
        done += n;
    }
    if (b->reward_log) {
        e = launch_finalize_logs(b->reward_log, b->action_log, nlog, c->n_games, s);
        if (e) return hip_fail(e, "k_finalize_logs launch");
    }
    run->kernel_used = THRL_KERNEL_TUPLE;
    return THRL_OK;
}

int thrl_qtable_episodes(const thrl_cfg* c, const thrl_buffers* b, thrl_run* run, void* stream) {
    int rc = validate(c);
    if (rc) return rc;
    if (!b || !run) return fail(THRL_ERR_NULL, "bufs/run is NULL");
    if (!b->q || !b->state) return fail(THRL_ERR_NULL, "q/state is NULL");
    if (run->n_episodes < 0) return fail(THRL_ERR_BAD_CONFIG, "n_episodes < 0");
    run->kernel_used = 0;
    if (run->n_episodes == 0) return THRL_OK;
    const bool injected = b->inj_u != nullptr;
    const bool per_game_logs = b->game_reward_log || b->game_action_log;
    int k = run->kernel;
    if (k != THRL_KERNEL_AUTO && k != THRL_KERNEL_GENERIC && k != THRL_KERNEL_WAVE && k != THRL_KERNEL_WAVE_PLAIN &&
        k != THRL_KERNEL_WAVE_GREEDY && k != THRL_KERNEL_TUPLE)
        return fail(THRL_ERR_BAD_CONFIG, "unknown kernel id %d", k);
    const bool any_sweep = b->sweep_gamma || b->sweep_alpha || b->sweep_eps_end || b->sweep_eps_step || b->sweep_eps || b->sweep_noise_prob;
    if (k == THRL_KERNEL_TUPLE || k == THRL_KERNEL_AUTO) {
        // (AUTO prefers the two-agent wave kernel where it applies: decided below; the tuple kernel takes what that one cannot)
        TuplePlan tp = plan_tuple(c, run);
        if (tp.ok && per_game_logs) { tp.ok = false; snprintf(tp.why, sizeof(tp.why), "per-game logs requested"); }
        if (tp.ok && (b->sweep_eps_end || b->sweep_eps_step) && !b->sweep_eps) {
            // a game's epsilon must survive from launch to launch: it lives in sweep_eps
            tp.ok = false; snprintf(tp.why, sizeof(tp.why), "epsilon-schedule sweep without a per-game epsilon array (sweep_eps)");
        }
        tp.a.sweep = any_sweep ? 1 : 0;
        if (k == THRL_KERNEL_TUPLE) {
            if (!tp.ok) return fail(THRL_ERR_UNSUPPORTED, "tuple kernel cannot run this config: %s", tp.why);
            return run_tuple(c, b, run, tp, (hipStream_t)stream);
        }
        if (tp.ok && !plan_wave(c, run, injected).ok) return run_tuple(c, b, run, tp, (hipStream_t)stream);
    }
    const int force_variant = k == THRL_KERNEL_WAVE_PLAIN ? 1 : (k == THRL_KERNEL_WAVE_GREEDY ? 2 : 0);
    if (force_variant) k = THRL_KERNEL_WAVE;
    if (k != THRL_KERNEL_GENERIC) {
        WavePlan p = plan_wave(c, run, injected);
        if (p.ok && per_game_logs) { p.ok = false; snprintf(p.why, sizeof(p.why), "per-game logs requested"); }
        if (p.ok && (p.epk > 1 || p.replay_from > 0) && (b->sweep_gamma || b->sweep_alpha || b->sweep_eps_end || b->sweep_eps_step || b->sweep_eps || b->sweep_noise_prob)) {
            p.ok = false; snprintf(p.why, sizeof(p.why), "per-game sweeps with a multi-episode training cycle or a truncated deque");
        }
        if (p.ok) return run_wave(c, b, run, p, force_variant, (hipStream_t)stream);
        if (k == THRL_KERNEL_WAVE) return fail(THRL_ERR_UNSUPPORTED, "wave kernel cannot run this config: %s", p.why);
    }
    return run_generic(c, b, run, (hipStream_t)stream);
}

int thrl_play_greedy(const thrl_cfg* c, const void* q, const double* state0, int32_t iters, uint64_t seed,
                     uint64_t game_offset, double* mean_reward, double* mean_action, void* stream) {
    int rc = validate(c);
    if (rc) return rc;
    if (!q || !mean_reward || !mean_action) return fail(THRL_ERR_NULL, "q/mean_reward/mean_action is NULL");
    if (iters < 0) return fail(THRL_ERR_BAD_CONFIG, "iters < 0");
    if (iters == 0) return THRL_OK;
    PlayArgs a;
    memset(&a, 0, sizeof(a));
    a.G = c->n_games; a.N = c->n_agents; a.T = c->max_steps; a.iters = iters;
    a.stride = (int64_t)thrl_table_stride(c);
    fill_agents(c, a.ag, &a.env);
    a.q = q; a.state0 = state0; a.mean_reward = mean_reward; a.mean_action = mean_action;
    a.seed = seed; a.game_offset = game_offset;
    const int e = launch_play(a, c->q_dtype, (hipStream_t)stream);
    return e ? hip_fail(e, "k_play_greedy launch") : THRL_OK;
}

static int op_common(const thrl_cfg* c, OpArgs* a) {
    int rc = validate(c);
    if (rc) return rc;
    memset(a, 0, sizeof(*a));
    a->G = c->n_games; a->N = c->n_agents; a->stride = (int64_t)thrl_table_stride(c);
    fill_agents(c, a->ag, &a->env);
    return THRL_OK;
}

int thrl_op_sample_action(const thrl_cfg* c, int agent, const void* q, const double* price, double eps,
                          const double* u, const int8_t* choice, int encode32, int32_t* action_out,
                          void* stream) {
    OpArgs a;
    int rc = op_common(c, &a);
    if (rc) return rc;
    if (agent < 0 || agent >= c->n_agents) return fail(THRL_ERR_BAD_CONFIG, "agent %d out of range", agent);
    if (!q || !price || !action_out) return fail(THRL_ERR_NULL, "q/price/action_out is NULL");
    if (u && !choice) return fail(THRL_ERR_NULL, "u given without choice");
    a.agent = agent; a.q = const_cast<void*>(q); a.price = price; a.eps = eps; a.u = u; a.choice = choice;
    a.encode32 = encode32; a.action_out = action_out;
    const int e = launch_op_sample(a, c->q_dtype, (hipStream_t)stream);
    return e ? hip_fail(e, "k_op_sample launch") : THRL_OK;
}

int thrl_op_encode(const thrl_cfg* c, int agent, const double* price, int as_float32, int32_t* row_out,
                   void* stream) {
    OpArgs a;
    int rc = op_common(c, &a);
    if (rc) return rc;
    if (agent < 0 || agent >= c->n_agents) return fail(THRL_ERR_BAD_CONFIG, "agent %d out of range", agent);
    if (!price || !row_out) return fail(THRL_ERR_NULL, "price/row_out is NULL");
    a.agent = agent; a.price = price; a.encode32 = as_float32; a.action_out = row_out;
    const int e = launch_op_encode(a, (hipStream_t)stream);
    return e ? hip_fail(e, "k_op_encode launch") : THRL_OK;
}

int thrl_op_scale(const thrl_cfg* c, int agent, const int32_t* action, double* scaled_out, void* stream) {
    OpArgs a;
    int rc = op_common(c, &a);
    if (rc) return rc;
    if (agent < 0 || agent >= c->n_agents) return fail(THRL_ERR_BAD_CONFIG, "agent %d out of range", agent);
    if (!action || !scaled_out) return fail(THRL_ERR_NULL, "action/scaled_out is NULL");
    a.agent = agent; a.action = action; a.scaled_out = scaled_out;
    const int e = launch_op_scale(a, (hipStream_t)stream);
    return e ? hip_fail(e, "k_op_scale launch") : THRL_OK;
}

int thrl_op_env_step(const thrl_cfg* c, const double* scaled, const double* noise_u, const double* noise_a,
                     double* price_out, double* reward_out, void* stream) {
    OpArgs a;
    int rc = op_common(c, &a);
    if (rc) return rc;
    if (!scaled || !price_out || !reward_out) return fail(THRL_ERR_NULL, "scaled/price_out/reward_out is NULL");
    if (noise_u && !noise_a) return fail(THRL_ERR_NULL, "noise_u given without noise_a");
    a.scaled = scaled; a.noise_u = noise_u; a.noise_a = noise_a; a.price_out = price_out;
    a.reward_out = reward_out;
    const int e = launch_op_env_step(a, (hipStream_t)stream);
    return e ? hip_fail(e, "k_op_env_step launch") : THRL_OK;
}

int thrl_op_td_update(const thrl_cfg* c, int agent, void* q, int32_t* counter, int32_t n, const double* price,
                      const int32_t* action, const double* reward, const double* next_price, double* scratch,
                      void* stream) {
    OpArgs a;
    int rc = op_common(c, &a);
    if (rc) return rc;
    if (agent < 0 || agent >= c->n_agents) return fail(THRL_ERR_BAD_CONFIG, "agent %d out of range", agent);
    if (n < 0) return fail(THRL_ERR_BAD_CONFIG, "n < 0");
    if (n == 0) return THRL_OK;
    if (!q || !price || !action || !reward || !next_price || !scratch)
        return fail(THRL_ERR_NULL, "a required pointer is NULL");
    a.agent = agent; a.q = q; a.counter = counter; a.n = n; a.price = price; a.action = action;
    a.reward = reward; a.next_price = next_price; a.reward_out = scratch;
    const int e = launch_op_td(a, c->q_dtype, (hipStream_t)stream);
    return e ? hip_fail(e, "k_op_td launch") : THRL_OK;
}

size_t thrl_nn_param_count(int n_actions) {
    if (n_actions < 2 || n_actions > 32) return 0;
    return (size_t)(2 * THRL_NN_HIDDEN + n_actions * THRL_NN_HIDDEN + n_actions);
}

static int nn_check(int n_games, int n_actions) {
    if (n_games < 1) return fail(THRL_ERR_BAD_CONFIG, "n_games=%d must be >= 1", n_games);
    if (n_actions < 2 || n_actions > 32) return fail(THRL_ERR_BAD_CONFIG, "neural agent: actions=%d out of [2,32]", n_actions);
    return THRL_OK;
}

int thrl_nn_init(int n_games, int n_actions, float* params, uint64_t seed, uint64_t game_offset, int agent,
                 void* stream) {
    int rc = nn_check(n_games, n_actions);
    if (rc) return rc;
    if (!params) return fail(THRL_ERR_NULL, "params is NULL");
    const int e = launch_nn_init(n_games, n_actions, params, seed, game_offset, agent, 0, (hipStream_t)stream);
    return e ? hip_fail(e, "k_nn_init launch") : THRL_OK;
}

size_t thrl_ac_param_count(int n_actions) {
    const size_t p = thrl_nn_param_count(n_actions);
    return p ? p + THRL_NN_HIDDEN + 1 : 0;
}

int thrl_ac_init(int n_games, int n_actions, float* params, uint64_t seed, uint64_t game_offset, int agent,
                 void* stream) {
    int rc = nn_check(n_games, n_actions);
    if (rc) return rc;
    if (!params) return fail(THRL_ERR_NULL, "params is NULL");
    const int e = launch_nn_init(n_games, n_actions, params, seed, game_offset, agent, 1, (hipStream_t)stream);
    return e ? hip_fail(e, "k_nn_init launch") : THRL_OK;
}

int thrl_ac_act(int n_games, int n_actions, const float* params, const double* price, const double* u,
                int32_t* action_out, float* prob_out, void* stream) {
    int rc = nn_check(n_games, n_actions);
    if (rc) return rc;
    if (!params || !price || !action_out) return fail(THRL_ERR_NULL, "params/price/action_out is NULL");
    const int e = launch_nn_act(n_games, n_actions, params, (int)thrl_ac_param_count(n_actions), price, u, action_out,
                                prob_out, (hipStream_t)stream);
    return e ? hip_fail(e, "k_nn_act launch") : THRL_OK;
}

int thrl_ac_train(int n_games, int n_actions, float* params, float* adam_m, float* adam_v, int32_t step, int32_t n,
                  int32_t ld, const double* price, const int32_t* action, const double* reward, const double* next_price,
                  double gamma, double entropy_coef, double lr, const double* sweep_gamma, const double* sweep_entropy,
                  float* grad_out, void* stream) {
    int rc = nn_check(n_games, n_actions);
    if (rc) return rc;
    if (!params || !adam_m || !adam_v || !price || !action || !reward || !next_price)
        return fail(THRL_ERR_NULL, "a required pointer is NULL");
    if (n < 2 || n > THRL_NN_MAX_TRANSITIONS)
        return fail(THRL_ERR_UNSUPPORTED, "n=%d transitions out of [2,%d]", n, THRL_NN_MAX_TRANSITIONS);
    if (step < 0) return fail(THRL_ERR_BAD_CONFIG, "step < 0");
    if (nn_train_lds_bytes(n_actions, n, 1) > 160 * 1024) return fail(THRL_ERR_UNSUPPORTED, "transition buffer does not fit LDS");
    if (ld < n) return fail(THRL_ERR_BAD_CONFIG, "ld=%d < n=%d", ld, n);
    const int e = launch_nn_train(n_games, n_actions, params, adam_m, adam_v, step, n, ld, price, action, reward, next_price,
                                  (float)gamma, (float)entropy_coef, (float)lr, sweep_gamma, sweep_entropy, grad_out, nullptr,
                                  (hipStream_t)stream);
    return e ? hip_fail(e, "k_nn_reinforce_train<AC> launch") : THRL_OK;
}

int thrl_nn_act(int n_games, int n_actions, const float* params, const double* price, const double* u,
                int32_t* action_out, float* prob_out, void* stream) {
    int rc = nn_check(n_games, n_actions);
    if (rc) return rc;
    if (!params || !price || !action_out) return fail(THRL_ERR_NULL, "params/price/action_out is NULL");
    const int e = launch_nn_act(n_games, n_actions, params, (int)thrl_nn_param_count(n_actions), price, u, action_out,
                                prob_out, (hipStream_t)stream);
    return e ? hip_fail(e, "k_nn_act launch") : THRL_OK;
}

int thrl_nn_reinforce_train(int n_games, int n_actions, float* params, float* adam_m, float* adam_v, int32_t step,
                            int32_t n, int32_t ld, const double* price, const int32_t* action, const double* reward,
                            double gamma, double entropy_coef, double lr, const double* sweep_gamma,
                            const double* sweep_entropy, float* grad_out, float* returns_scratch, void* stream) {
    int rc = nn_check(n_games, n_actions);
    if (rc) return rc;
    if (!params || !adam_m || !adam_v || !price || !action || !reward) return fail(THRL_ERR_NULL, "a required pointer is NULL");
    if (n < 2 || n > THRL_NN_MAX_TRANSITIONS)
        return fail(THRL_ERR_UNSUPPORTED, "n=%d transitions out of [2,%d]", n, THRL_NN_MAX_TRANSITIONS);
    if (step < 0) return fail(THRL_ERR_BAD_CONFIG, "step < 0");
    if (nn_train_lds_bytes(n_actions, n, 0) > 160 * 1024) return fail(THRL_ERR_UNSUPPORTED, "transition buffer does not fit LDS");
    if (ld < n) return fail(THRL_ERR_BAD_CONFIG, "ld=%d < n=%d", ld, n);
    const int e = launch_nn_train(n_games, n_actions, params, adam_m, adam_v, step, n, ld, price, action, reward, nullptr,
                                  (float)gamma, (float)entropy_coef, (float)lr, sweep_gamma, sweep_entropy, grad_out, returns_scratch,
                                  (hipStream_t)stream);
    return e ? hip_fail(e, "k_nn_reinforce_train launch") : THRL_OK;
}

// ---- the tuple-chain kernel for two-agent games with discrete neural policies (thrl_ptuple.hip)
struct PTuplePlan { bool ok; char why[160]; PTupleArgs a; int waves_per_block, blocks_per_cu; size_t scratch_bytes; };
constexpr size_t kPTupleLutRegion = 64 * 1024;     // head of thrl_mixed.policy_tab: LUT image + the launch's work counter (last 64 bytes)

static double h_scale_kind(int k, const thrl_cfg* c, int i, int kind) {
    if (kind == 0) return h_scale(k, c, i);
    double x = (double)k / (double)c->n_actions[i];             // Reinforce.scale (agents.py:153-157): action / actions
    x = x * (c->act_hi[i] - c->act_lo[i]);
    return x + c->act_lo[i];
}

static PTuplePlan plan_ptuple(const thrl_cfg* c, const thrl_mixed* mx) {
    PTuplePlan p;
    memset(&p, 0, sizeof(p));
#define NO(msg) do { snprintf(p.why, sizeof(p.why), "%s", msg); return p; } while (0)
    PTupleArgs& a = p.a;
    if (c->n_agents != 2) NO("not a two-agent game");
    const int T = c->max_steps;
    if (T > 256) NO("more than 256 steps per episode");
    a.qi = -1; a.n_r = 0;
    for (int i = 0; i < 2; i++) {
        a.kind[i] = mx->kind[i];
        if (mx->kind[i] == 0) {
            if (a.qi >= 0) NO("no neural agent");
            a.qi = i;
            if (c->n_actions[i] > 64) NO("QTable agent with more than 64 actions");
            // train_net trains -- and empties the ring -- after every episode (agents.py:60,77)
            if (!(mx->buf_len[i] > 0 && mx->min_memory[i] <= T && T <= mx->buf_len[i] && mx->count[i] == 0)) NO("QTable replay buffer does not train once per episode");
        } else if (mx->kind[i] == 1 || mx->kind[i] == 2) {
            if (c->n_actions[i] > 32) NO("neural agent with more than 32 actions");
            if (mx->buf_len[i] < T) NO("a neural agent's replay ring is shorter than an episode");
            a.ri[a.n_r++] = i;
        } else NO("continuous agent");
    }
    if (a.n_r == 0) NO("no neural agent");
    // per-game sweeps: the QTable agent's rows and noise_prob need the kernel's sweep variant (rows of neural agents only reach
    // their update kernels); a game's epsilon must survive the launch in sweep_eps
    const bool q_sweep = mx->sweep_gamma || mx->sweep_alpha || mx->sweep_eps_end || mx->sweep_eps_step || mx->sweep_eps;
    if (a.qi < 0 && mx->sweep_noise_prob) NO("noise_prob sweep without a QTable agent in the game");
    if (a.qi >= 0 && (mx->sweep_eps_end || mx->sweep_eps_step) && !mx->sweep_eps) NO("epsilon-schedule sweep without a per-game epsilon array (sweep_eps)");
    a.sweep = (a.qi >= 0 && (q_sweep || mx->sweep_noise_prob)) ? 1 : 0;
    const long tuples = (long)c->n_actions[0] * c->n_actions[1];
    if (tuples > kTupMaxTuples) NO("more than 4,096 action pairs");
    a.tuples = (int)tuples; a.T = T;
    // distinct float32 prices, in order of first occurrence (the device repeats this enumeration: k_ptuple_lut), and the
    // QTable agent's row window
    static thread_local float seen[kTupMaxTuples];
    int npid = 0, lo = 1 << 30, hi = -1;
    const double ratio = c->env_a / c->env_b;
    for (int a0 = 0; a0 < c->n_actions[0]; a0++)
        for (int a1 = 0; a1 < c->n_actions[1]; a1++) {
            double Q = 0.0;
            Q = Q + ratio * h_scale_kind(a0, c, 0, mx->kind[0]);
            Q = Q + ratio * h_scale_kind(a1, c, 1, mx->kind[1]);
            double price = c->env_a - c->env_b * Q;
            if (!(price > 0.0)) price = 0.0;
            const float x = (float)price;
            int j = 0;
            while (j < npid && memcmp(&seen[j], &x, 4) != 0) j++;
            if (j == npid) seen[npid++] = x;
            if (a.qi >= 0) {
                // (with env noise the intercept ranges over [0.7a, a]: both encodes are monotonic in the price)
                const double intercepts[2] = {c->env_a, c->env_a * 0.7};
                for (int v = 0; v < (c->noise_prob > 0.0 ? 2 : 1); v++) {
                    double pr = intercepts[v] - c->env_b * Q;
                    if (!(pr > 0.0)) pr = 0.0;
                    const int r64 = h_encode64(pr, c, a.qi), r32 = h_encode32(pr, c, a.qi);
                    if (r64 < 0 || r64 > c->n_states[a.qi] || r32 < 0 || r32 > c->n_states[a.qi]) NO("price outside the table on the action grid");
                    lo = r64 < lo ? r64 : lo; lo = r32 < lo ? r32 : lo;
                    hi = r64 > hi ? r64 : hi; hi = r32 > hi ? r32 : hi;
                }
            }
        }
    if (npid > 2048) NO("more than 2,048 distinct prices");
    a.npid = npid;
    int amax = 0;
    for (int r = 0; r < a.n_r; r++) amax = c->n_actions[a.ri[r]] > amax ? c->n_actions[a.ri[r]] : amax;
    const int apad = amax <= 24 ? 24 : 32;
    const int esz = c->q_dtype == 1 ? 8 : 4;
    size_t off = 0;
    if (a.qi >= 0) {
        a.row_lo = lo; a.win_rows = hi - lo + 1;
        if (a.win_rows + 2 > 256) NO("reachable row window > 254 rows");
        const int cells = (a.win_rows + 2) * c->n_actions[a.qi];
        off = align_up((size_t)cells * esz, 16);
        a.am_off = (int)off; off = align_up(off + a.win_rows + 2, 16);
        a.g_off = (int)off; off = align_up(off + (size_t)tuples + 1, 16);
        a.hist_off = (int)off; a.hist_dwords = (cells + 1) / 2; off = align_up(off + 4 * (size_t)a.hist_dwords, 16);
    }
    a.cdf_off = (int)off;
    if (a.n_r == 2) {
        const size_t cdf = (size_t)2 * (npid + 1) * apad * 4;
        if (cdf > 24 * 1024) NO("price grid too large for in-LDS policy tables");
        off = align_up(off + cdf, 16);
    }
    a.logs_off = (int)off; off += 64 * 4 * 8;
    a.game_lds_bytes = (int)align_up(off, 16);
    a.qrows_off = (int)align_up((size_t)tuples * 2, 16);
    a.xf_off = (int)align_up((size_t)a.qrows_off + (size_t)tuples * 2, 16);
    a.aq_off = (int)align_up((size_t)a.xf_off + (size_t)npid * 4, 16);
    a.lut_lds_bytes = a.aq_off + 2 * 128 * 8;
    a.price_off = a.lut_lds_bytes;
    a.qsum_off = a.price_off + 8 * (int)tuples;
    if ((size_t)a.qsum_off + 8 * (size_t)tuples > kPTupleLutRegion - 64) NO("LUT image too large");
    p.scratch_bytes = kPTupleLutRegion + (a.n_r == 1 ? (size_t)c->n_games * (size_t)(npid + 1) * apad * 4 : 0);
    const DevInfo dv = dev_info();
    const int cap_waves = a.n_r == 1 ? 12 : 16;                     // compiled for 3 / 4 waves per SIMD
    int best_w = 0, best_total = 0, best_b = 0;
    for (int w = 1; w <= 8; w++) {
        const int lds = a.lut_lds_bytes + w * a.game_lds_bytes;
        if (lds > dv.lds_per_cu) break;
        int b = dv.lds_per_cu / (((lds + 511) / 512) * 512);
        if (b * w > cap_waves) b = cap_waves / w;
        if (b < 1) continue;
        const int total = b * w;
        const bool even = (w & 3) == 0, best_even = best_w > 0 && (best_w & 3) == 0;
        if (total > best_total || (total == best_total && ((even && !best_even) || (even == best_even && w < best_w)))) {
            best_total = total; best_w = w; best_b = b;
        }
    }
    if (best_w == 0) NO("one game does not fit LDS");
    p.waves_per_block = best_w; p.blocks_per_cu = best_b;
    p.ok = true;
    return p;
#undef NO
}

int thrl_mixed_episodes(const thrl_cfg* c, thrl_mixed* mx, void* q, int32_t* counter, double* state, thrl_run* run,
                        double* game_reward_log, double* game_action_log, void* stream) {
    int rc = validate(c);
    if (rc) return rc;
    if (!mx || !q || !state || !run || !game_reward_log || !game_action_log)
        return fail(THRL_ERR_NULL, "a required pointer is NULL");
    if (run->n_episodes < 0) return fail(THRL_ERR_BAD_CONFIG, "n_episodes < 0");
    if (run->n_episodes == 0) return THRL_OK;
    run->kernel_used = 0;                          // THRL_KERNEL_TUPLE when the tuple-chain kernel ran
    MixedArgs a;
    memset(&a, 0, sizeof(a));
    a.G = c->n_games; a.N = c->n_agents; a.T = c->max_steps; a.n_episodes = run->n_episodes;
    a.stride = (int64_t)thrl_table_stride(c);
    fill_agents(c, a.ag, &a.env);
    a.q = q; a.counter = counter; a.state = state;
    a.game_reward_log = game_reward_log; a.game_action_log = game_action_log;
    a.seed = run->seed; a.game_offset = run->game_offset; a.first_episode = run->first_episode;
    for (int i = 0; i < c->n_agents; i++) {
        if (mx->kind[i] < 0 || mx->kind[i] > 3) return fail(THRL_ERR_BAD_CONFIG, "agent %d: unknown kind %d", i, mx->kind[i]);
        if (mx->kind[i] != 0 && (!mx->nn_params[i] || (mx->kind[i] != 3 && c->n_actions[i] > 32)))
            return fail(THRL_ERR_NULL, "agent %d: a neural agent needs nn_params and actions <= 32", i);
        a.nn_stride[i] = mx->kind[i] == 2 ? (int32_t)thrl_ac_param_count(c->n_actions[i])
                                          : (int32_t)thrl_nn_param_count(c->n_actions[i]);
        if (mx->buf_len[i] > 0 && (!mx->buf_price[i] || !mx->buf_action[i] || !mx->buf_reward[i] || !mx->buf_nprice[i]))
            return fail(THRL_ERR_NULL, "agent %d: replay buffer pointers are NULL", i);
        if (mx->kind[i] == 0 && mx->buf_len[i] > 0 && !mx->buf_scratch[i])
            return fail(THRL_ERR_NULL, "agent %d: QTable agents need buf_scratch", i);
        a.kind[i] = mx->kind[i]; a.nn_params[i] = mx->nn_params[i];
        a.buf_price[i] = mx->buf_price[i]; a.buf_action[i] = mx->buf_action[i]; a.buf_reward[i] = mx->buf_reward[i];
        a.buf_nprice[i] = mx->buf_nprice[i]; a.buf_ov[i] = mx->buf_scratch[i];
        a.buf_len[i] = mx->buf_len[i]; a.min_memory[i] = mx->min_memory[i]; a.count0[i] = mx->count[i];
        a.eps0[i] = run->eps[i];
    }
    if ((mx->sweep_eps_end || mx->sweep_eps_step) && !mx->sweep_eps)
        return fail(THRL_ERR_NULL, "sweep_eps_end / sweep_eps_step need the per-game epsilon state sweep_eps");
    if (mx->sweep_noise_prob && !(c->noise_prob > 0.0))
        return fail(THRL_ERR_BAD_CONFIG, "sweep_noise_prob needs cfg.noise_prob > 0 (it switches the noise draws on)");
    a.sw_gamma = mx->sweep_gamma; a.sw_alpha = mx->sweep_alpha; a.sw_eps_end = mx->sweep_eps_end;
    a.sw_eps_step = mx->sweep_eps_step; a.sw_eps = mx->sweep_eps; a.sw_noise_prob = mx->sweep_noise_prob;
    // ---- two-agent games with discrete policies, noise-free: the tuple-chain kernel (thrl_ptuple.hip), same results
    {
        PTuplePlan tp = plan_ptuple(c, mx);
        if (tp.ok && mx->policy_tab && mx->policy_tab_bytes >= tp.scratch_bytes && !(mx->flags & THRL_MIXED_NO_TUPLE_KERNEL)) {
            PTupleArgs& t = tp.a;
            t.G = c->n_games; t.n_episodes = run->n_episodes; t.stride = a.stride;
            t.env = a.env;
            for (int i = 0; i < 2; i++) {
                t.ag[i] = a.ag[i];
                t.nn_params[i] = a.nn_params[i]; t.nn_stride[i] = a.nn_stride[i];
                t.buf_price[i] = a.buf_price[i]; t.buf_action[i] = a.buf_action[i]; t.buf_reward[i] = a.buf_reward[i];
                t.buf_nprice[i] = a.buf_nprice[i]; t.buf_len[i] = a.buf_len[i]; t.count0[i] = a.count0[i];
                t.eps0[i] = a.eps0[i];
            }
            t.q = q; t.counter = counter; t.state = state;
            t.sw_gamma = mx->sweep_gamma; t.sw_alpha = mx->sweep_alpha; t.sw_eps_end = mx->sweep_eps_end;
            t.sw_eps_step = mx->sweep_eps_step; t.sw_eps = mx->sweep_eps; t.sw_noise_prob = mx->sweep_noise_prob;
            unsigned char* base = (unsigned char*)mx->policy_tab;
            t.lut = base;
            t.next_game = (int32_t*)(base + kPTupleLutRegion - 64);
            t.policy_tab = (float*)(base + kPTupleLutRegion);
            t.game_reward_log = game_reward_log; t.game_action_log = game_action_log;
            t.seed = run->seed; t.game_offset = run->game_offset; t.first_episode = run->first_episode;
            t.waves_per_block = tp.waves_per_block;
            int grid = (c->n_games + tp.waves_per_block - 1) / tp.waves_per_block;
            const int max_grid = dev_info().cus * tp.blocks_per_cu;
            if (grid > max_grid) grid = max_grid;
            const size_t lds = (size_t)t.lut_lds_bytes + (size_t)tp.waves_per_block * t.game_lds_bytes;
            hipStream_t s = (hipStream_t)stream;
            int e = launch_ptuple_lut(t, base, s);
            if (e) return hip_fail(e, "k_ptuple_lut launch");
            if (hipMemsetAsync(t.next_game, 0, sizeof(int32_t), s) != hipSuccess) return hip_fail((int)hipGetLastError(), "hipMemsetAsync");
            e = launch_ptuple(t, c->q_dtype, grid, tp.waves_per_block * 64, lds, s);
            if (e) return hip_fail(e, "k_ptuple_episodes launch");
            run->kernel_used = THRL_KERNEL_TUPLE;
            goto bookkeeping;
        }
    }
    a.policy_tab = mx->policy_tab; a.policy_tab_bytes = mx->policy_tab ? mx->policy_tab_bytes : 0;
    {
    const char* why = "";
    if (plan_mixed(a, c->q_dtype, &why)) return fail(THRL_ERR_UNSUPPORTED, "thrl_mixed_episodes: %s", why);
    const int e = launch_mixed(a, c->q_dtype, (hipStream_t)stream);
    if (e) return hip_fail(e, "k_mixed_episodes launch");
    }
bookkeeping:
    // host mirror of the bookkeeping that is identical for every game
    for (int ep = 0; ep < run->n_episodes; ep++)
        for (int i = 0; i < c->n_agents; i++) {
            const int cap = mx->buf_len[i];
            if (cap > 0)
                for (int t = 0; t < c->max_steps; t++) {
                    mx->count[i] += 1;
                    if (mx->count[i] >= 2 * cap) mx->count[i] -= cap;
                }
            if (mx->kind[i] == 0) {
                const int len = mx->count[i] < cap ? mx->count[i] : cap;
                if (cap > 0 && len >= mx->min_memory[i]) mx->count[i] = 0;
                run->eps[i] = c->eps_end[i] + (run->eps[i] - c->eps_end[i]) * c->eps_step[i];
            }
        }
    return THRL_OK;
}

size_t thrl_mixed_policy_table_bytes(const thrl_cfg* c, const thrl_mixed* mx) {
    if (validate(c) != THRL_OK || !mx) return 0;
    MixedArgs a;
    memset(&a, 0, sizeof(a));
    a.G = c->n_games; a.N = c->n_agents; a.T = c->max_steps;
    fill_agents(c, a.ag, &a.env);
    for (int i = 0; i < c->n_agents; i++) {
        if (mx->kind[i] < 0 || mx->kind[i] > 3) return 0;
        a.kind[i] = mx->kind[i];
    }
    const char* why = "";
    size_t need = 0;
    if (plan_mixed(a, c->q_dtype, &why) == 0 && a.ptab_tuples > 0) need = a.ptab_need_bytes;
    // the tuple-chain kernel keeps its LUT image (and, against a QTable, its CDF rows) in the same scratch; buffer
    // state (count) does not change the size
    thrl_mixed probe = *mx;
    for (int i = 0; i < c->n_agents && i < THRL_MAXA; i++) probe.count[i] = 0;
    const PTuplePlan tp = plan_ptuple(c, &probe);
    if (tp.ok && tp.scratch_bytes > need) need = tp.scratch_bytes;
    return need;
}

int thrl_cac_init(int n_games, float* params, uint64_t seed, uint64_t game_offset, int agent, void* stream) {
    if (n_games < 1) return fail(THRL_ERR_BAD_CONFIG, "n_games=%d must be >= 1", n_games);
    if (!params) return fail(THRL_ERR_NULL, "params is NULL");
    const int e = launch_cac_init(n_games, params, seed, game_offset, agent, (hipStream_t)stream);
    return e ? hip_fail(e, "k_cac_init launch") : THRL_OK;
}

int thrl_cac_act(int n_games, const float* params, const double* price, const double* u1, const double* u2,
                 float* action_out, float* mu_out, float* std_out, float* v_out, void* stream) {
    if (n_games < 1) return fail(THRL_ERR_BAD_CONFIG, "n_games=%d must be >= 1", n_games);
    if (!params || !price || !action_out) return fail(THRL_ERR_NULL, "params/price/action_out is NULL");
    if ((u1 == nullptr) != (u2 == nullptr)) return fail(THRL_ERR_NULL, "u1 and u2 must both be given or both NULL");
    const int e = launch_cac_act(n_games, params, price, u1, u2, action_out, mu_out, std_out, v_out, (hipStream_t)stream);
    return e ? hip_fail(e, "k_cac_act launch") : THRL_OK;
}

int thrl_cac_train(int n_games, float* params, float* adam_m, float* adam_v, int32_t step, int32_t n, int32_t ld,
                   const double* price, const float* action, const double* reward, const double* next_price,
                   double gamma, double entropy_coef, double lr, const double* sweep_gamma, const double* sweep_entropy,
                   float* grad_out, void* stream) {
    if (n_games < 1) return fail(THRL_ERR_BAD_CONFIG, "n_games=%d must be >= 1", n_games);
    if (!params || !adam_m || !adam_v || !price || !action || !reward || !next_price)
        return fail(THRL_ERR_NULL, "a required pointer is NULL");
    if (n < 2 || cac_train_lds_bytes(n) > 160 * 1024)
        return fail(THRL_ERR_UNSUPPORTED, "n=%d transitions: need 2 <= n and the batch in LDS (n <= ~5600)", n);
    if (step < 0) return fail(THRL_ERR_BAD_CONFIG, "step < 0");
    if (ld < n) return fail(THRL_ERR_BAD_CONFIG, "ld=%d < n=%d", ld, n);
    const int e = launch_cac_train(n_games, params, adam_m, adam_v, step, n, ld, price, action, reward, next_price,
                                   (float)gamma, (float)entropy_coef, (float)lr, sweep_gamma, sweep_entropy, grad_out,
                                   (hipStream_t)stream);
    return e ? hip_fail(e, "k_cac_train launch") : THRL_OK;
}

int thrl_op_draws(const thrl_cfg* c, uint64_t seed, uint64_t game_offset, uint64_t episode, int32_t step,
                  double* u_out, int8_t* choice_out, double* u2_out, double* noise_u_out, double* noise_a_out,
                  void* stream) {
    int rc = validate(c);
    if (rc) return rc;
    if (!u_out || !choice_out) return fail(THRL_ERR_NULL, "u_out/choice_out is NULL");
    if ((noise_u_out == nullptr) != (noise_a_out == nullptr)) return fail(THRL_ERR_NULL, "noise outputs must both be given or both NULL");
    int32_t nA[THRL_MAXA];
    for (int i = 0; i < THRL_MAXA; i++) nA[i] = i < c->n_agents ? c->n_actions[i] : 1;
    const int e = launch_op_draws(c->n_games, c->n_agents, seed, game_offset, (uint32_t)episode, (uint32_t)step,
                                  c->env_a, c->env_a * 0.7, nA, u_out, choice_out, u2_out, noise_u_out, noise_a_out,
                                  (hipStream_t)stream);
    return e ? hip_fail(e, "k_op_draws launch") : THRL_OK;
}

}  // extern "C"
