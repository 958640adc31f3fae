// thrl_wave.hip -- fused episode kernel for gfx950: ONE WAVEFRONT PER GAME.
//
// The performance path for the reference's headline shape (2 QTable agents on
// one NoisyPriceState, noise off, float32 tables).  Same semantics as
// thrl_generic.hip (and bit-identical results to it and to the float32 oracle):
// trainer.train_one's loop (th_rl/trainer.py:46-70) with QTable.sample_action
// (agents.py:80-89), scale (:51-57), NoisyPriceState.step (environments.py:25-39),
// ReplayBuffer append/replay/empty (buffers.py) and QTable.train_net
// (agents.py:59-78) fused, for `n_episodes` episodes per launch.
//
// MI355X mapping (DESIGN.md "wave kernel"):
//   * a wave owns one game for the whole launch; its two agents' Q-table WINDOWS
//     (only the rows the payoff grid can reach, + 2 spill rows for an arbitrary
//     initial state) are streamed coalesced HBM -> LDS once, stay resident for
//     all episodes of the launch, and are streamed back once;
//   * lanes 0-31 serve agent 0, lanes 32-63 agent 1; lane l&31 is action column
//     l&31, so a row max is one ds_read_b32 + 5 DPP v_max steps;
//   * the discretised action grid (next-state row + price per action pair) is a
//     LUT staged in LDS once per block ("payoff LUT");
//   * everything that is not on the serial state->action->state chain is done
//     lane-parallel over the T steps of an episode (lane = step): Philox draws,
//     reward / old-value gathers, log sums, visit-counter atomics;
//   * the serial chains (play: s -> argmax row s -> LUT -> s'; replay: live row
//     max -> TD write) touch only SGPRs, LDS and a few VALU ops per step.
#include "thrl_kernels.h"
#include "thrl_wave_lut.h"

namespace thrl {

typedef unsigned int v2u __attribute__((ext_vector_type(2)));

// max over each 32-lane half; the result is valid in the UPPER 16-lane row of each half
// (lanes 16-31 and 48-63), which is where the replay loop's writer lanes (16, 48) sit.
// Four single-instruction DPP max steps give every lane its 16-lane row max; row_bcast:15
// then folds row 0 into row 1 and row 2 into row 3 (row_mask 0xA).
// The s_nop 1 before each DPP op are the 2 wait states a DPP read of a just-written
// VGPR needs (hipcc does not look inside asm statements).
__device__ __forceinline__ float half_max_upper_row(float v) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return v;
}

// lanes<32 of the result: lanes 0-31 of a ; lanes>=32: lanes 0-31 of b   (.x)
// and the same for the upper halves (.y): one v_permlane32_swap.
__device__ __forceinline__ v2u pack_halves(unsigned a, unsigned b) {
    return __builtin_amdgcn_permlane32_swap(a, b, false, false);
}

// LDS access by 32-bit LDS address (address space 3): no generic-pointer arithmetic
typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ float lds_load_f32(unsigned addr) { return *(const lds_f32*)(uintptr_t)addr; }
__device__ __forceinline__ void lds_store_f32(unsigned addr, float v) { *(lds_f32*)(uintptr_t)addr = v; }

__device__ __forceinline__ unsigned bperm(unsigned byte_sel, unsigned v) {
    return (unsigned)__builtin_amdgcn_ds_bpermute((int)byte_sel, (int)v);
}

// DPP move of a double (two dword moves)
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false));
}
// Sum FOUR per-lane doubles over the 64 lanes in one pass ("transpose" reduction):
// returns, in every lane L, the wave total of quantity (L & 3) where the quantities are
// ordered (q0, q1, q2, q3).  7 double adds instead of 24, no LDS traffic.
__device__ __forceinline__ double wave_sum4(double q0, double q1, double q2, double q3, int lane) {
    // step 1 (partner lane^1): even lanes keep (q0,q2), odd lanes keep (q1,q3)
    const bool odd = lane & 1;
    const double k0 = odd ? q1 : q0, k1 = odd ? q3 : q2;      // kept
    const double s0 = odd ? q0 : q1, s1 = odd ? q2 : q3;      // what the partner keeps
    const double a0 = k0 + dpp_mov64<0xB1>(s0);               // quad_perm [1,0,3,2]
    const double a1 = k1 + dpp_mov64<0xB1>(s1);
    // step 2 (partner lane^2): bit1 == 0 keeps the first, bit1 == 1 keeps the second
    const bool b1 = lane & 2;
    const double kk = b1 ? a1 : a0, ss = b1 ? a0 : a1;
    double v = kk + dpp_mov64<0x4E>(ss);                      // quad_perm [2,3,0,1]
    // now lane L holds quantity (L&3) summed over its quad; rotate-add within the 16-lane row
    v = v + dpp_mov64<0x124>(v);                              // row_ror:4
    v = v + dpp_mov64<0x128>(v);                              // row_ror:8
    // across the four rows: swap-add with v_permlane16_swap / v_permlane32_swap
    {
        unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        unsigned lo2 = lo, hi2 = hi;
        asm("" : "+v"(lo2), "+v"(hi2));
        const v2u rl = __builtin_amdgcn_permlane16_swap(lo, lo2, false, false);
        const v2u rh = __builtin_amdgcn_permlane16_swap(hi, hi2, false, false);
        const unsigned rlx = rl.x, rly = rl.y, rhx = rh.x, rhy = rh.y;
        v = __hiloint2double((int)rhx, (int)rlx) + __hiloint2double((int)rhy, (int)rly);
    }
    {
        unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        unsigned lo2 = lo, hi2 = hi;
        asm("" : "+v"(lo2), "+v"(hi2));
        const v2u rl = __builtin_amdgcn_permlane32_swap(lo, lo2, false, false);
        const v2u rh = __builtin_amdgcn_permlane32_swap(hi, hi2, false, false);
        const unsigned rlx = rl.x, rly = rl.y, rhx = rh.x, rhy = rh.y;
        v = __hiloint2double((int)rhx, (int)rlx) + __hiloint2double((int)rhy, (int)rly);
    }
    return v;
}

__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// v = lane `lane` of `old` replaced by the (uniform) value `val`: one v_writelane_b32 with the
// lane select in M0 (two different SGPR operands would break the gfx9 constant-bus limit)
__device__ __forceinline__ uint32_t writelane_u(uint32_t old, uint32_t val, int lane) {
    asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(val), "s"(lane));
    return old;
}

// value of a lane-indexed-by-row register pair at (uniform) row s
template <int NRSEG>
__device__ __forceinline__ uint32_t read_row(const uint32_t (&r)[NRSEG], int s) {
    uint32_t v = readlane_u(r[0], s & 63);
    if (NRSEG > 1) {
        const uint32_t h = readlane_u(r[NRSEG - 1], s & 63);
        if (s >= 64) v = h;
    }
    return v;
}
// per-lane gather from a lane-indexed-by-row register pair (row differs per lane)
template <int NRSEG>
__device__ __forceinline__ uint32_t gather_row(const uint32_t (&r)[NRSEG], uint32_t row) {
    uint32_t v = bperm((row & 63u) << 2, r[0]);
    if (NRSEG > 1) {
        const uint32_t h = bperm((row & 63u) << 2, r[NRSEG - 1]);
        if (row >= 64u) v = h;
    }
    return v;
}

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
        reinterpret_cast<double*>(out + L.price_off)[idx] = price;
    }
    if (idx < 2 * A) {
        const int i = idx / A, k = idx - i * A;
        const double sc = scale_action(k, a.ag[i]);
        reinterpret_cast<double*>(out + L.aq_off)[idx] = __dmul_rn(a.env.ratio, sc);
        reinterpret_cast<double*>(out + L.sct_off)[idx] = __ddiv_rn(sc, (double)a.T);
    }
}

// LDS capacity allows 20 resident waves per CU for the headline window, i.e. 5 per
// SIMD: keep the register allocation at <= 96 VGPRs there (NSEG <= 2).
// NOISE: environment noise (environments.py:28-31) handled per step (price not on the LUT);
// its larger row window leaves room for fewer waves, so the register budget is relaxed.
// SWEEP: per-game hyper-parameter arrays (thrl_buffers.sweep_*); compiled only together with NOISE
// so the headline variant carries none of that state.
template <int NSEG, int NRSEG, bool NOISE, bool SWEEP>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(NOISE ? 4 : (NSEG <= 2 ? 5 : 4))))
k_wave_episodes(const WaveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int A = a.A, W = a.win_rows, T = a.T, lo = a.row_lo;
    const WaveLut L = wave_lut_layout(A);

    {   // stage the payoff LUT once per block
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.lut_ns);
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem);
        for (int k = threadIdx.x; k < (a.lut_bytes >> 2); k += blockDim.x) dst[k] = src[k];
    }
    __syncthreads();
    const unsigned short* lut_ns = reinterpret_cast<const unsigned short*>(smem + L.ns_off);   // play | train<<8
    const double* __restrict__ lut_price = reinterpret_cast<const double*>(a.lut_ns + L.price_off);   // HBM/L2
    const double* lut_aq = reinterpret_cast<const double*>(smem + L.aq_off);
    const double* lut_sct = reinterpret_cast<const double*>(smem + L.sct_off);

    float* tab0 = reinterpret_cast<float*>(smem + a.lut_bytes + (size_t)wib * a.game_lds_bytes);
    float* tab1 = tab0 + (W + 2) * A;
    const int half = lane >> 5;
    const int col = min(lane & 31, A - 1);
    float* tabh_col = (half ? tab1 : tab0) + col;
    const unsigned tab0_off = lds_addr(tab0);        // absolute LDS byte addresses
    const unsigned tab1_off = lds_addr(tab1);
    const unsigned tabh_col_lds = lds_addr(tabh_col);
    const unsigned sel_base = (unsigned)(lane & 32) << 2;      // bpermute byte index of this half's lane 0
    const bool writer = (lane & 31) == 16;          // first lane of the row that holds the half max

    const AgentParams& p0 = a.ag[0];
    const AgentParams& p1 = a.ag[1];
    const double inv_T_den = (double)T;

    // lane ((e&15)*4+k): sum over this wave's games of episode-e log value k (e < 16 / e >= 16), as
    // fixed-point integers: the games a wave gets are not deterministic, integer sums do not care
    long long acc = 0, acc_hi = 0;
    const double log_scale = (lane & 2) ? a.log_scale[1] : a.log_scale[0];
    const int wave_gid = blockIdx.x * a.waves_per_block + wib;

    // Games are handed out dynamically (one atomic per game): waves on less crowded CUs simply take
    // more games, which measured 7-13 % faster than the static grid-stride assignment.
    for (;;) {
        int g = 0;
        if (lane == 0) g = atomicAdd(a.next_game, 1);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= a.G) break;
        const uint64_t gid = a.game_offset + (uint64_t)g;
        float* __restrict__ q0 = a.q + (int64_t)g * a.stride + p0.table_off;
        float* __restrict__ q1 = a.q + (int64_t)g * a.stride + p1.table_off;

        // ---- per-game hyper-parameters (sweeps) or the config's scalars
        float gamma_h = half ? p1.gamma_f : p0.gamma_f;
        float alpha_h = half ? p1.alpha_f : p0.alpha_f;
        float oma0 = p0.one_minus_alpha_f, oma1 = p1.one_minus_alpha_f;
        if (SWEEP && a.sw_gamma) gamma_h = (float)a.sw_gamma[(size_t)half * a.G + g];
        if (SWEEP && a.sw_alpha) {
            alpha_h = (float)a.sw_alpha[(size_t)half * a.G + g];
            oma0 = (float)__dsub_rn(1.0, a.sw_alpha[g]);
            oma1 = (float)__dsub_rn(1.0, a.sw_alpha[(size_t)a.G + g]);
        }
        double epsg0 = 0.0, epsg1 = 0.0;               // per-game epsilon (sweep mode)
        const bool sw_eps_on = SWEEP && a.sw_eps != nullptr;
        if (sw_eps_on) { epsg0 = a.sw_eps[g]; epsg1 = a.sw_eps[(size_t)a.G + g]; }
        const float ag_h = __fmul_rn(alpha_h, gamma_h);  // the only coefficient on the replay chain (float32 TD form, thrl_device.h)
        const double eend0 = (SWEEP && a.sw_eps_end) ? a.sw_eps_end[g] : p0.eps_end;
        const double eend1 = (SWEEP && a.sw_eps_end) ? a.sw_eps_end[(size_t)a.G + g] : p1.eps_end;
        const double estep0 = (SWEEP && a.sw_eps_step) ? a.sw_eps_step[g] : p0.eps_step;
        const double estep1 = (SWEEP && a.sw_eps_step) ? a.sw_eps_step[(size_t)a.G + g] : p1.eps_step;
        const double noise_prob_g = (SWEEP && a.sw_noise_prob) ? a.sw_noise_prob[g] : a.env.noise_prob;

        // ---- initial state -> local rows (window or spill)
        const double price0 = a.state[g];
        int sp = __builtin_amdgcn_readfirstlane(encode32(price0, p0));
        int st = __builtin_amdgcn_readfirstlane(encode64(price0, p0));
        sp = min(max(sp, 0), a.rows - 1);
        st = min(max(st, 0), a.rows - 1);
        int spill0 = -1, spill1 = -1, sp_l, st_l;
        if (sp >= lo && sp < lo + W) sp_l = sp - lo; else { spill0 = sp; sp_l = W; }
        if (st == sp) st_l = sp_l;
        else if (st >= lo && st < lo + W) st_l = st - lo;
        else { spill1 = st; st_l = W + 1; }

        // ---- stream the table windows HBM -> LDS (contiguous, coalesced)
        {
            const float* s0 = q0 + lo * A;
            const float* s1 = q1 + lo * A;
            const int n = W * A;
            // 8 loads per agent in flight before the first LDS write (one HBM latency per
            // batch instead of one per 64 floats)
            for (int k0 = 0; k0 < n; k0 += 512) {
                float v0[8], v1[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = min(k0 + j * 64 + lane, n - 1);
                    v0[j] = s0[k]; v1[j] = s1[k];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = k0 + j * 64 + lane;
                    if (k < n) { tab0[k] = v0[j]; tab1[k] = v1[j]; }
                }
            }
            if (spill0 >= 0 && lane < A) {
                tab0[W * A + lane] = q0[spill0 * A + lane];
                tab1[W * A + lane] = q1[spill0 * A + lane];
            }
            if (spill1 >= 0 && lane < A) {
                tab0[(W + 1) * A + lane] = q0[spill1 * A + lane];
                tab1[(W + 1) * A + lane] = q1[spill1 * A + lane];
            }
        }
        __builtin_amdgcn_wave_barrier();

        int s = sp_l | (st_l << 8);          // current state: play row | train row << 8
        double last_price = price0;
        for (int e = 0; e < a.n_episodes; e++) {
            const uint32_t eg = (uint32_t)(a.first_episode + (uint64_t)e);
            const double eps0 = sw_eps_on ? epsg0 : a.eps[e][0], eps1 = sw_eps_on ? epsg1 : a.eps[e][1];

            // ---- (a) greedy action of every local row, lane = row (the table is frozen
            //          during play: agents.py only writes it in train_net), and the
            //          greedy-greedy successor row of every row
            uint32_t am0[NRSEG], am1[NRSEG], am0A[NRSEG], grow[NRSEG];
            uint32_t am0A2[NRSEG], am1x2[NRSEG];      // byte offsets into the u16 LUT (x2), for the play loop
#pragma unroll
            for (int k = 0; k < NRSEG; k++) {
                const int row = min(lane + 64 * k, W + 1);
                const float* r0 = tab0 + row * A;
                const float* r1 = tab1 + row * A;
                float b0 = r0[0], b1 = r1[0];
                uint32_t i0 = 0, i1 = 0;
#pragma unroll 4
                for (int j = 1; j < A; j++) {
                    const float v0 = r0[j], v1 = r1[j];
                    if (v0 > b0) { b0 = v0; i0 = j; }
                    if (v1 > b1) { b1 = v1; i1 = j; }
                }
                am0[k] = i0; am1[k] = i1; am0A[k] = i0 * (uint32_t)A;
                am0A2[k] = am0A[k] * 2u; am1x2[k] = i1 * 2u;
                grow[k] = lut_ns[i0 * (uint32_t)A + i1];
            }

            // ---- (b,c) play: lane-parallel Philox, then the serial state chain.
            //      seq[seg] lane t = row in which step t was played.
            uint32_t seq[NSEG], rwv[NSEG];
            uint32_t kwv[NSEG];            // per step: flags | K << 8, K = (f0 ? c0*A : 0) + (f1 ? c1 : 0)
            double nav[NSEG];              // NOISE: the uniform(0.7a, a) draw of a noisy step (lane = step)
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                const int n = min(64, T - seg * 64);
                uint32_t rw;
                nav[seg] = 0.0;
                if (a.inj_u) {
                    // parity mode: the reference's recorded draws, [E][T][2][G] (agents.py:81-82)
                    const int tt = min(seg * 64 + lane, T - 1);
                    const size_t k = (((size_t)e * T + tt) * 2) * (size_t)a.G + (size_t)g;
                    const uint32_t ex0 = a.inj_u[k] < eps0 ? 1u : 0u;
                    const uint32_t ex1 = a.inj_u[k + a.G] < eps1 ? 2u : 0u;
                    const uint32_t c0 = min((uint32_t)(uint8_t)a.inj_choice[k], (uint32_t)(A - 1));
                    const uint32_t c1 = min((uint32_t)(uint8_t)a.inj_choice[k + a.G], (uint32_t)(A - 1));
                    rw = ex0 | ex1 | (c0 << 8) | (c1 << 16);
                } else {
                    const u32x4 x = draw(a.seed, gid, eg, (uint32_t)(seg * 64 + lane), 0u);
                    const uint32_t ex0 = u01_32(x.x) < eps0 ? 1u : 0u;
                    const uint32_t ex1 = u01_32(x.z) < eps1 ? 2u : 0u;
                    rw = ex0 | ex1 | (__umulhi(x.y, (uint32_t)A) << 8) | (__umulhi(x.w, (uint32_t)A) << 16);
                }
                if (NOISE) {               // environments.py:28-29, bit 2 of rw = noisy step
                    double nu, na;
                    if (a.inj_u) {
                        const int tt = min(seg * 64 + lane, T - 1);
                        const size_t k = ((size_t)e * T + tt) * (size_t)a.G + (size_t)g;
                        nu = a.inj_noise_u[k]; na = a.inj_noise_a[k];
                    } else {
                        const u32x4 xn = draw(a.seed, gid, eg, (uint32_t)(seg * 64 + lane), kStreamNoise);
                        nu = u01_32(xn.x);
                        na = __dadd_rn(a.env.noise_lo, __dmul_rn(__dsub_rn(a.env.a, a.env.noise_lo), u01_32(xn.y)));
                    }
                    if (nu < noise_prob_g) rw |= 4u;
                    nav[seg] = na;
                }
                rwv[seg] = rw;
                const uint32_t kw = (rw & 7u) |
                    ((((rw & 1u) ? ((rw >> 8) & 0xFFu) * (uint32_t)A : 0u) + ((rw & 2u) ? ((rw >> 16) & 0xFFu) : 0u)) << 8);
                kwv[seg] = kw;
                uint32_t sq = 0;
                // Steps are taken 4 at a time.  Phase 1 (off the serial chain, lane = ROW):
                // for every row r the next row if step t were played in r,
                //   nsr_t[r] = LUT[a0][a1],  a_i = explore_i(t) ? choice_i(t) : argmax_i[r];
                // it does not depend on the current state, so its LDS gathers overlap.
                // Phase 2 (the chain): s <- nsr_t[s], one v_readlane per step.
                for (int t0 = 0; t0 < n; t0 += 4) {
                    uint32_t nsr[4][NRSEG];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t w = readlane_u(NOISE ? rw : kw, min(t0 + j, 63));
                        if (NOISE && (w & 4u)) {
                            // noisy step: the price is not on the LUT; evaluate it for every row
                            const int tl = min(t0 + j, 63);
                            const double na = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(nav[seg]), tl),
                                                               __builtin_amdgcn_readlane(__double2loint(nav[seg]), tl));
                            const uint32_t c0 = (w >> 8) & 0xFFu, c1 = (w >> 16) & 0xFFu;
#pragma unroll
                            for (int k = 0; k < NRSEG; k++) {
                                const uint32_t a0r = (w & 1u) ? c0 : am0[k], a1r = (w & 2u) ? c1 : am1[k];
                                const double Q = __dadd_rn(lut_aq[a0r], lut_aq[A + a1r]);
                                double pr = __dsub_rn(na, __dmul_rn(a.env.b, Q));
                                if (!(pr > 0.0)) pr = 0.0;
                                const int r32 = min(max(encode32(pr, p0) - lo, 0), W - 1);
                                const int r64 = min(max(encode64(pr, p0) - lo, 0), W - 1);
                                nsr[j][k] = (uint32_t)(r32 | (r64 << 8));
                            }
                        } else if ((w & 3u) == 0u) {
#pragma unroll
                            for (int k = 0; k < NRSEG; k++) nsr[j][k] = grow[k];
                        } else {
                            if (NOISE) {
                                const uint32_t c0A = ((w >> 8) & 0xFFu) * (uint32_t)A, c1 = (w >> 16) & 0xFFu;
#pragma unroll
                                for (int k = 0; k < NRSEG; k++) {
                                    const uint32_t idx = ((w & 1u) ? c0A : am0A[k]) + ((w & 2u) ? c1 : am1[k]);
                                    nsr[j][k] = lut_ns[idx];
                                }
                            } else {
                                // idx = greedy part (masked by the not-exploring flags) + precomputed K
                                const uint32_t nf0 = (w & 1u) ^ 1u, nf1 = ((w >> 1) & 1u) ^ 1u, K2 = (w >> 8) << 1;
#pragma unroll
                                for (int k = 0; k < NRSEG; k++) {
                                    const uint32_t off = __umul24(am0A2[k], nf0) + __umul24(am1x2[k], nf1) + K2;
                                    nsr[j][k] = *reinterpret_cast<const unsigned short*>(
                                        reinterpret_cast<const unsigned char*>(lut_ns) + off);
                                }
                            }
                        }
                    }
                    if (t0 + 4 <= n) {          // full group: no per-step bound checks on the chain
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            sq = writelane_u(sq, (uint32_t)s, t0 + j);
                            s = (int)read_row<NRSEG>(nsr[j], s & 0xFF);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (t0 + j < n) {
                                sq = writelane_u(sq, (uint32_t)s, t0 + j);
                                s = (int)read_row<NRSEG>(nsr[j], s & 0xFF);
                            }
                        }
                    }
                }
                seq[seg] = sq;
            }
            const int s_end = s;

            // ---- (d1) lane-parallel (lane = step): actions, old-value snapshot
            //      (agents.py:67) for ALL steps before any TD write
            uint32_t act[NSEG];            // a0 | a1<<8 | train_row<<16 | next_row<<24
            v2u t4q[NSEG];                 // (1-alpha)*old_value, halves packed per 32 steps
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                const int tt = seg * 64 + lane;
                const bool valid = tt < T;
                const uint32_t my_s = seq[seg] & 0xFFu;              // row the step was played in
                const uint32_t my_train = (seq[seg] >> 8) & 0xFFu;   // row train_net sees for that state
                const uint32_t rw = rwv[seg];
                // gather with ALL lanes active (a bpermute reads only from active lanes, and
                // the source lane here is a table row, unrelated to this lane's step), then select
                const uint32_t g0 = gather_row<NRSEG>(am0, my_s);
                const uint32_t g1 = gather_row<NRSEG>(am1, my_s);
                uint32_t a0 = (rw & 1u) ? ((rw >> 8) & 0xFFu) : g0;
                uint32_t a1 = (rw & 2u) ? ((rw >> 16) & 0xFFu) : g1;
                uint32_t nxt = (uint32_t)__shfl_down((int)seq[seg], 1, 64);
                if (seg + 1 < NSEG) { if (lane == 63) nxt = readlane_u(seq[seg + 1 < NSEG ? seg + 1 : seg], 0); }
                const uint32_t ns = (tt + 1 < T) ? ((nxt >> 8) & 0xFFu) : ((uint32_t)s_end >> 8);
                uint32_t srow = my_train;
                if (!valid) { a0 = 0; a1 = 0; srow = 0; }
                const float ov0 = tab0[srow * A + a0];
                const float ov1 = tab1[srow * A + a1];
                t4q[seg] = pack_halves(__builtin_bit_cast(unsigned, __fmul_rn(oma0, ov0)),
                                       __builtin_bit_cast(unsigned, __fmul_rn(oma1, ov1)));
                act[seg] = a0 | (a1 << 8) | (srow << 16) | (ns << 24);
            }
            __builtin_amdgcn_wave_barrier();

            double lr0 = 0.0, lr1 = 0.0, la0 = 0.0, la1 = 0.0;
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                // ---- (d2) rewards, LDS write addresses, logs, visit counters of this segment
                const int tt = seg * 64 + lane;
                const bool valid = tt < T;
                const uint32_t a0 = act[seg] & 0xFFu, a1 = (act[seg] >> 8) & 0xFFu;
                const uint32_t srow = (act[seg] >> 16) & 0xFFu, ns = act[seg] >> 24;
                double price = lut_price[a0 * (uint32_t)A + a1];
                if (NOISE) {
                    double pn = __dsub_rn(nav[seg], __dmul_rn(a.env.b, __dadd_rn(lut_aq[a0], lut_aq[A + a1])));
                    if (!(pn > 0.0)) pn = 0.0;
                    if (rwv[seg] & 4u) price = pn;
                }
                const double r0d = __dmul_rn(price, lut_aq[a0]);
                const double r1d = __dmul_rn(price, lut_aq[A + a1]);
                if (seg == NSEG - 1) {
                    const int ll = T - 1 - seg * 64;       // lane of the episode's last step
                    last_price = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(price), ll),
                                                  __builtin_amdgcn_readlane(__double2loint(price), ll));
                }
                const v2u req = pack_halves(__builtin_bit_cast(unsigned, (float)r0d),
                                            __builtin_bit_cast(unsigned, (float)r1d));
                const v2u woq = pack_halves(tab0_off + (srow * A + a0) * 4u, tab1_off + (srow * A + a1) * 4u);
                // next-state row offsets of steps (t, t+1) packed in lane t: one v_readlane per loop
                // iteration below, the halves are split on the scalar unit
                const uint32_t nsoff1 = ns * (uint32_t)A * 4u;
                const uint32_t nsoff = nsoff1 | ((uint32_t)__shfl_down((int)nsoff1, 1, 64) << 16);
                if (valid) {
                    lr0 += r0d; lr1 += r1d;          // divided by T once per episode below
                    la0 += lut_sct[a0]; la1 += lut_sct[A + a1];
                }
                // visit counters (agents.py:76): the packed transition word goes to this wave's
                // log (coalesced, L2-resident); the counts are built per game below
                if (a.counter)
                    a.tlog[(((size_t)wave_gid * kWaveMaxEpisodes + e) * NSEG + seg) * 64 + lane] =
                        valid ? act[seg] : 0xFFFFFFFFu;

                // ---- (e) replay chain (agents.py:68-76): live next_max, sequential writes.
                //      Per step: 2 bpermutes fetch this half's next_max-independent part of the
                //      target and the write address, one ds_read of the next-state row, the half
                //      max, ONE fma, one masked ds_write.
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int nsub = __builtin_amdgcn_readfirstlane(min(32, T - seg * 64 - k * 32));
                    // b = fma(alpha, reward, (1-alpha)*old_value): everything of the target that does
                    // not depend on the live next_max, lane-parallel (lane = step, halves = agents)
                    const unsigned c1_k = __builtin_bit_cast(unsigned, __fmaf_rn(alpha_h,
                        __builtin_bit_cast(float, k ? req.y : req.x), __builtin_bit_cast(float, k ? t4q[seg].y : t4q[seg].x)));
                    const unsigned wo_k = k ? woq.y : woq.x;
                    // Operands of step t+1 are fetched while step t's max chain runs: the row
                    // read is issued FIRST (LDS returns in order, so the wait before the max is
                    // lgkmcnt(2), not 0).  Two steps per iteration so no register rotation.
                    unsigned sel = sel_base;
                    unsigned c1A = bperm(sel, c1_k), woA = bperm(sel, wo_k);
                    unsigned c1B = 0, woB = 0;
                    const int tb = k * 32;
                    for (int t = 0; t < nsub; t += 2) {
                        const uint32_t off2 = readlane_u(nsoff, tb + t);
                        {
                            const float row_v = lds_load_f32(tabh_col_lds + (off2 & 0xFFFFu));
                            __builtin_amdgcn_sched_barrier(0);
                            sel += 4u;
                            c1B = bperm(sel, c1_k); woB = bperm(sel, wo_k);
                            const float nm = half_max_upper_row(row_v);
                            const float val = __fmaf_rn(ag_h, nm, __builtin_bit_cast(float, c1A));
                            if (writer) lds_store_f32(woA, val);
                            __builtin_amdgcn_wave_barrier();
                        }
                        if (t + 1 < nsub) {
                            const float row_v = lds_load_f32(tabh_col_lds + (off2 >> 16));
                            __builtin_amdgcn_sched_barrier(0);
                            sel += 4u;
                            c1A = bperm(sel, c1_k); woA = bperm(sel, wo_k);
                            const float nm = half_max_upper_row(row_v);
                            const float val = __fmaf_rn(ag_h, nm, __builtin_bit_cast(float, c1B));
                            if (writer) lds_store_f32(woB, val);
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
            }

            // ---- (f) per-episode log sums of this game into the wave accumulator:
            //      lane L gets the wave total of quantity L&3 = (reward0, reward1, action0, action1)
            {
                double v = wave_sum4(lr0, lr1, la0, la1, lane);
                if ((lane & 3) < 2) v = __ddiv_rn(v, inv_T_den);
                const long long vq = __double2ll_rn(__dmul_rn(v, log_scale));
                if ((lane >> 2) == (e & 15)) { if (e < 16) acc += vq; else acc_hi += vq; }
            }
            // epsilon decays after every train_net call (agents.py:78)
            if (SWEEP) {
                epsg0 = __dadd_rn(eend0, __dmul_rn(__dsub_rn(epsg0, eend0), estep0));
                epsg1 = __dadd_rn(eend1, __dmul_rn(__dsub_rn(epsg1, eend1), estep1));
            }
        }

        // ---- stream the windows back LDS -> HBM, store the env state
        {
            float* d0 = q0 + lo * A;
            float* d1 = q1 + lo * A;
            const int n = W * A;
            for (int k = lane; k < n; k += 64) { d0[k] = tab0[k]; d1[k] = tab1[k]; }
            if (spill0 >= 0 && lane < A) {
                q0[spill0 * A + lane] = tab0[W * A + lane];
                q1[spill0 * A + lane] = tab1[W * A + lane];
            }
            if (spill1 >= 0 && lane < A) {
                q0[spill1 * A + lane] = tab0[(W + 1) * A + lane];
                q1[spill1 * A + lane] = tab1[(W + 1) * A + lane];
            }
            if (lane == 0 && a.n_episodes > 0) a.state[g] = last_price;
            if (lane == 0 && sw_eps_on) { a.sw_eps[g] = epsg0; a.sw_eps[(size_t)a.G + g] = epsg1; }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- visit counters of this game (agents.py:76).  The tables are back in HBM, so
        //      the wave's LDS region is free: build the launch's visit histogram there
        //      (u16 pairs in dwords, ds_add_u32; E*T <= 32*256 < 65536 so no carry) from the
        //      transition log, then apply it to the counter window with plain coalesced
        //      read-add-write -- this game's counters belong to this wave alone, so no
        //      global atomics are needed (2e9 scattered atomics per launch were a 70 ms floor).
        if (a.counter) {
            const int cells = (W + 2) * A;                       // per agent
            const int hw = (cells + 1) >> 1;                     // dwords per agent
            // may_alias: the histogram overlays the float tables (no type-based reordering)
            typedef unsigned __attribute__((may_alias)) hist_u32;
            hist_u32* hist = reinterpret_cast<hist_u32*>(tab0);   // 2*hw dwords <= 2*cells floats
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            for (int k = lane; k < 2 * hw; k += 64) hist[k] = 0u;
            __builtin_amdgcn_wave_barrier();
            // log read-back: 4 episodes' loads in flight at a time (sc1 = L2-served: the wave
            // reads what it stored itself)
            for (int e0 = 0; e0 < a.n_episodes; e0 += 4) {
                unsigned w[4][NSEG];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int seg = 0; seg < NSEG; seg++)
                        w[j][seg] = __hip_atomic_load(
                            &a.tlog[(((size_t)wave_gid * kWaveMaxEpisodes + min(e0 + j, a.n_episodes - 1)) * NSEG + seg) * 64 + lane],
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int seg = 0; seg < NSEG; seg++) {
                        const unsigned ww = w[j][seg];
                        if (e0 + j < a.n_episodes && ww != 0xFFFFFFFFu) {
                            const unsigned srow = (ww >> 16) & 0xFFu;
                            const unsigned c0 = srow * (unsigned)A + (ww & 0xFFu);
                            const unsigned c1 = srow * (unsigned)A + ((ww >> 8) & 0xFFu);
                            __hip_atomic_fetch_add(&hist[c0 >> 1], 1u << ((c0 & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            __hip_atomic_fetch_add(&hist[hw + (c1 >> 1)], 1u << ((c1 & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
            }
            __builtin_amdgcn_wave_barrier();
            // apply: window rows are contiguous in HBM, so cell k of the window is element
            // lo*A + k of the agent's table (no row/column split); spill rows separately
            int32_t* cw0 = a.counter + (int64_t)g * a.stride + p0.table_off;
            int32_t* cw1 = a.counter + (int64_t)g * a.stride + p1.table_off;
            const int nwin = W * A;
            for (int k = lane; k < nwin; k += 64) {
                const unsigned n0 = (hist[k >> 1] >> ((k & 1) << 4)) & 0xFFFFu;
                const unsigned n1 = (hist[hw + (k >> 1)] >> ((k & 1) << 4)) & 0xFFFFu;
                if (n0) cw0[lo * A + k] += (int32_t)n0;
                if (n1) cw1[lo * A + k] += (int32_t)n1;
            }
            if (lane < 2 * A) {
                const int which = lane >= A, col = lane - which * A;
                const int grow_ = which ? spill1 : spill0;
                const int k = nwin + lane;
                if (grow_ >= 0) {
                    const unsigned n0 = (hist[k >> 1] >> ((k & 1) << 4)) & 0xFFFFu;
                    const unsigned n1 = (hist[hw + (k >> 1)] >> ((k & 1) << 4)) & 0xFFFFu;
                    if (n0) cw0[grow_ * A + col] += (int32_t)n0;
                    if (n1) cw1[grow_ * A + col] += (int32_t)n1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    a.partial[(size_t)wave_gid * 128 + lane] = acc;
    a.partial[(size_t)wave_gid * 128 + 64 + lane] = acc_hi;
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

template <int NSEG, int NRSEG, bool NOISE, bool SWEEP>
static int launch_wave_t(const WaveArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((k_wave_episodes<NSEG, NRSEG, NOISE, SWEEP>), dim3(grid), dim3(block), lds, s, a);
    return (int)hipGetLastError();
}

int launch_wave_lut(const WaveArgs& a, unsigned char* out, hipStream_t s) {
    const int n = a.A * a.A > 2 * a.A ? a.A * a.A : 2 * a.A;
    hipLaunchKernelGGL(k_wave_lut, dim3((n + 255) / 256), dim3(256), 0, s, a, out);
    return (int)hipGetLastError();
}

template <bool NOISE, bool SWEEP>
static int launch_wave_n(const WaveArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    const int nseg = (a.T + 63) / 64;
    const int nrseg = (a.win_rows + 2 + 63) / 64;
    if (nrseg == 1) {
        switch (nseg) {
            case 1: return launch_wave_t<1, 1, NOISE, SWEEP>(a, grid, block, lds, s);
            case 2: return launch_wave_t<2, 1, NOISE, SWEEP>(a, grid, block, lds, s);
            case 3: return launch_wave_t<3, 1, NOISE, SWEEP>(a, grid, block, lds, s);
            case 4: return launch_wave_t<4, 1, NOISE, SWEEP>(a, grid, block, lds, s);
        }
    } else if (nrseg == 2) {
        switch (nseg) {
            case 1: return launch_wave_t<1, 2, NOISE, SWEEP>(a, grid, block, lds, s);
            case 2: return launch_wave_t<2, 2, NOISE, SWEEP>(a, grid, block, lds, s);
            case 3: return launch_wave_t<3, 2, NOISE, SWEEP>(a, grid, block, lds, s);
            case 4: return launch_wave_t<4, 2, NOISE, SWEEP>(a, grid, block, lds, s);
        }
    }
    return -1;
}

int launch_wave(const WaveArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    const bool sweep = a.sw_gamma || a.sw_alpha || a.sw_eps_end || a.sw_eps_step || a.sw_eps || a.sw_noise_prob;
    if (sweep) return launch_wave_n<true, true>(a, grid, block, lds, s);      // noise code present, taken per game
    return a.env.noise_prob > 0.0 ? launch_wave_n<true, false>(a, grid, block, lds, s)
                                  : launch_wave_n<false, false>(a, grid, block, lds, s);
}

int launch_wave_reduce(const long long* partial, const double* log_scale, int total_waves, int n_episodes, int G, double* reward_log,
                       double* action_log, hipStream_t s) {
    hipLaunchKernelGGL(k_wave_reduce, dim3(4 * kWaveMaxEpisodes), dim3(256), 0, s, partial, log_scale[0], log_scale[1],
                       total_waves, n_episodes, (double)G,
                       reward_log, action_log);
    return (int)hipGetLastError();
}

}  // namespace thrl
