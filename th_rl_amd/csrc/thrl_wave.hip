// thrl_wave.hip -- host side of the fused one-wavefront-per-game episode kernel
// (thrl_wave_kernel.h): the payoff-LUT builder, the log reduction and the dispatch over the
// compiled variants.  The variants themselves are instantiated in thrl_wave_f32.hip,
// thrl_wave_f32n.hip and thrl_wave_f64*.hip (separate translation units: they compile in parallel).
#include <cstdlib>
#include "thrl_kernels.h"
#include "thrl_wave_lut.h"

namespace thrl {

// builds the payoff LUT image in HBM (copied to LDS by every block)
__global__ void __launch_bounds__(256) k_wave_lut(const WaveArgs a, unsigned char* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int A = a.A;
    const WaveLut L = wave_lut_layout(A);
    if (idx < A * A) {
        const int a0 = idx / A, a1 = idx - a0 * A;
        double scaled[2] = {scale_action(a0, a.ag[0]), scale_action(a1, a.ag[1])};
        double rew[2];
        const double price = env_step<2>(a.env, 2, scaled, a.env.a, rew);
        // next-state row per action pair, window-local: play row (float32 encode, trainer.py:53)
        // in the low byte, train row (float64 encode, agents.py:62,66) in the high byte
        reinterpret_cast<unsigned short*>(out + L.ns_off)[idx] =
            (unsigned short)((encode32(price, a.ag[0]) - a.row_lo) | ((encode64(price, a.ag[0]) - a.row_lo) << 8));
    }
    if (idx < 2 * A) {
        const int i = idx / A, k = idx - i * A;
        const double sc = scale_action(k, a.ag[i]);
        reinterpret_cast<double*>(out + L.aq_off)[idx] = __dmul_rn(a.env.ratio, sc);
        reinterpret_cast<double*>(out + L.sct_off)[idx] = __ddiv_rn(sc, (double)a.T);
    }
}

// reduction of the per-wave fixed-point partials -> mean logs [E][2].  One block per accumulator
// slot j = e*4+k.  Integer sums: exact, so the result does not depend on which wave played which
// game nor on the launch geometry.
__global__ void __launch_bounds__(256) k_wave_reduce(const long long* partial, double scale_r, double scale_a,
                                                     int total_waves, int n_episodes,
                                                     double G, double* reward_log, double* action_log) {
    __shared__ long long red[256];
    const int j = blockIdx.x;
    const int e = j >> 2, k = j & 3;
    if (e >= n_episodes) return;
    long long s = 0;
    for (int w = threadIdx.x; w < total_waves; w += 256) s += partial[(size_t)w * 128 + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double m = __ddiv_rn(__ddiv_rn((double)red[0], k < 2 ? scale_r : scale_a), G);
        if (k < 2) { if (reward_log) reward_log[e * 2 + k] = m; }
        else { if (action_log) action_log[e * 2 + (k - 2)] = m; }
    }
}

int launch_wave_lut(const WaveArgs& a, unsigned char* out, hipStream_t s) {
    const int n = a.A * a.A > 2 * a.A ? a.A * a.A : 2 * a.A;
    hipLaunchKernelGGL(k_wave_lut, dim3((n + 255) / 256), dim3(256), 0, s, a, out);
    return (int)hipGetLastError();
}

// (THRL_GREEDY_EPS overrides the threshold for measurements; read once per process)
static double greedy_eps() {
    static const double v = [] { const char* e = getenv("THRL_GREEDY_EPS"); return e ? atof(e) : 0.05; }();
    return v;
}

int launch_wave(const WaveArgs& a, int q_dtype, int grid, int block, size_t lds, hipStream_t s) {
    const bool sweep = a.sw_gamma || a.sw_alpha || a.sw_eps_end || a.sw_eps_step || a.sw_eps || a.sw_noise_prob;
    const int variant = sweep ? 2 : (a.env.noise_prob > 0.0 ? 1 : 0);     // sweep: noise code present, taken per game
    const bool cycle = a.epk > 1 || a.replay_from > 0;                    // (never together with sweeps: thrl_api.hip)
    // Once both agents explore in fewer than ~5 % of their steps the variant that skips the table build of all-greedy
    // groups of four steps and runs cyclic segments as register recurrences is the faster one: measured on trained
    // tables along a run (262,144 games, round 3, profiles/exp_greedy_threshold.py), plain vs GREEDY: 3.05 vs 2.97e10 at
    // epsilon 0.069, 3.01 vs 3.01 at 0.054, 2.98 vs 3.06 at 0.042, 2.93 vs 3.15 at 0.026, 2.81 vs 3.44 at 0.0065 -- its
    // per-group tests cost while the agents still explore.  Same results either way.
    const bool greedy_exists = !cycle && variant == 0;
    if (a.force_variant == 2 && !greedy_exists) return -2;               // (thrl_api.hip turns it into THRL_ERR_UNSUPPORTED)
    const bool greedy = greedy_exists && a.force_variant != 1 &&
                        (a.force_variant == 2 || (a.eps[0][0] <= greedy_eps() && a.eps[0][1] <= greedy_eps()));
    if (q_dtype == 1) {
        switch (variant) {
            case 0: return cycle ? launch_wave_f64_plain_cycle(a, grid, block, lds, s)
                          : greedy ? launch_wave_f64_plain_greedy(a, grid, block, lds, s) : launch_wave_f64_plain(a, grid, block, lds, s);
            case 1: return cycle ? launch_wave_f64_noise_cycle(a, grid, block, lds, s) : launch_wave_f64_noise(a, grid, block, lds, s);
            default: return launch_wave_f64_sweep(a, grid, block, lds, s);
        }
    }
    switch (variant) {
        case 0: return cycle ? launch_wave_f32_plain_cycle(a, grid, block, lds, s)
                      : greedy ? launch_wave_f32_plain_greedy(a, grid, block, lds, s) : launch_wave_f32_plain(a, grid, block, lds, s);
        case 1: return cycle ? launch_wave_f32_noise_cycle(a, grid, block, lds, s) : launch_wave_f32_noise(a, grid, block, lds, s);
        default: return launch_wave_f32_sweep(a, grid, block, lds, s);
    }
}

int launch_wave_reduce(const long long* partial, const double* log_scale, int total_waves, int n_episodes, int G, double* reward_log,
                       double* action_log, hipStream_t s) {
    hipLaunchKernelGGL(k_wave_reduce, dim3(4 * kWaveMaxEpisodes), dim3(256), 0, s, partial, log_scale[0], log_scale[1],
                       total_waves, n_episodes, (double)G,
                       reward_log, action_log);
    return (int)hipGetLastError();
}

}  // namespace thrl
