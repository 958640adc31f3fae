// thrl_tuple_kernel.h -- fused episode kernel for 1-4 QTable agents with INDIVIDUAL grids: ONE WAVEFRONT PER GAME,
// tables in LDS ("tuple-chain" kernel; template body, instantiated in thrl_tuple_f32.hip / thrl_tuple_f64.hip).
//
// The LDS-resident path for what the two-agent wave kernel (thrl_wave_kernel.h) does not take: other than two
// agents, per-agent state / action grids, per-agent max_state (the reference allows any nplayers and any QTable
// per agent: th_rl/trainer.py:21-23, agents.py:13-28).  Same semantics as thrl_generic.hip and bit-identical results
// to it and to the oracle: trainer.train_one's loop (trainer.py:46-70) with QTable.sample_action (agents.py:80-89),
// scale (:51-57), NoisyPriceState.step (environments.py:25-39), ReplayBuffer append / replay / empty
// and QTable.train_net (agents.py:59-78) for `n_episodes` episodes per launch.
//
// Idea: without noise the state after a step is a function of that step's ACTION TUPLE, so a game's state is a
// small integer tau = ((a0*A1 + a1)*A2 + a2)*A3 + a3 (< 4,096) and everything the loop needs is a table over it:
//   * per block, staged in LDS once: prow[tau] / trow[tau] = the window-local rows of the price after tuple tau, byte i =
//     agent i (play rows: float32 encode, trainer.py:53; train rows: float64 encode, agents.py:62,66 -- both kept);
//     per-action quantities (a/b)*scale_i(k) and scale_i(k)/T; the price per tuple stays in HBM (L2), it is only
//     gathered lane-parallel;
//   * per game and episode: the greedy action of every table row (lane = row; tables are frozen during play), composed
//     into G[tau] = the agents' greedy actions in state tau as one u16 action word (bit fields, below);
//   * play = a chain of T steps on the SCALAR unit that carries the LDS address of G[tau]: word = (G[tau] & the greedy
//     agents' fields) | the explorers' choices (both masks lane-precomputed per step), next address = gbase + sum_i a_i * 2 w_i
//     (one s_bfe_u32 + s_mul_i32 per agent, the sums as a tree; the last agent's term is its masked field) --
//     one LDS read, two v_readlane, one v_writelane and ~10 scalar instructions per step;
//   * everything else is lane-parallel over the steps (lane = step): Philox draws, rows, prices, rewards, the
//     old-value snapshot (agents.py:67), the log sums;
//   * replay = train_net's serial loop (agents.py:68-76) for ALL agents at once: agent i owns lanes 16i..16i+15, a
//     lane reads ceil(A_i/16) columns of the next-state row, four DPP steps give the row maximum, lane 16i stores
//     the TD value; step operands reach the 16-lane rows by ds_bpermute per 16 steps and row_newbcast DPP per step.
//     Strictly one transition at a time per agent, in order: no hazard analysis needed.  The float32 step is hand-written
//     (replay_step_f32: 11-15 vector instructions, no branch).
//   * visit counters: every step's cells (u16 per agent) go to a per-wave log in HBM / L2 (coalesced, lane = step); after a
//     game's tables are written back the log is folded into a u16 histogram that OVERLAYS the table region and is applied
//     to the int32 counters once per launch (as the wave kernel does: no LDS is spent on the histogram).
//   * action words: the agents' actions as bit fields (ceil(log2 A_i) bits each, laid out from bit 1 with the LAST agent
//     first, 16 bits in all at most: TupleArgs.act_sh / act_bits), so G is u16.
//   * env noise (template NOISE, noise_prob > 0): a step whose intercept was redrawn (environments.py:29-31) leaves the
//     action grid.  The state after it is carried as explicit rows (both encodes of its price, computed once on the
//     chain from qsum[tau] = the tuple's total quantity); the step played in it reads the agents' greedy actions from
//     the per-row bytes instead of G[tau]; every other step is the plain chain step.  The row windows cover the
//     prices of every tuple for every intercept in [0.7a, a].
//   * per-game sweeps (template SWEEP, compiled with NOISE): gamma / alpha / epsilon schedule / noise_prob per game.
#pragma once
#include <type_traits>
#include "thrl_kernels.h"

// THRL_TUP_ABLATE: timing-only diagnostic builds (python -m th_rl_amd.build --ablate-tuple MASK --out ..., profiles/ablate_tuple.py):
// a phase is skipped, the results are wrong by construction; thrl_ablate_mask() reports it and bench.py refuses such a library.
//   1 replay  2 play chain  4 G table  8 draws  16 per-row argmax  32 visit log + counters  64 log sums
#ifndef THRL_TUP_ABLATE
#define THRL_TUP_ABLATE 0
#endif

namespace thrl {
namespace tup {
constexpr int kTupAblate = THRL_TUP_ABLATE;

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
template <typename QT> __device__ __forceinline__ QT lds_load(unsigned addr) {
    typedef __attribute__((address_space(3))) QT lds_t;
    return *(const lds_t*)(uintptr_t)addr;
}
template <typename QT> __device__ __forceinline__ void lds_store(unsigned addr, QT v) {
    typedef __attribute__((address_space(3))) QT lds_t;
    *(lds_t*)(uintptr_t)addr = v;
}
__device__ __forceinline__ unsigned bperm(unsigned byte_sel, unsigned v) {
    return (unsigned)__builtin_amdgcn_ds_bpermute((int)byte_sel, (int)v);
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ float dppf(float v) { return __builtin_bit_cast(float, dpp32<CTRL>(__builtin_bit_cast(uint32_t, v))); }
template <int CTRL> __device__ __forceinline__ double dppd(double v) {
    return __hiloint2double((int)dpp32<CTRL>((uint32_t)__double2hiint(v)), (int)dpp32<CTRL>((uint32_t)__double2loint(v)));
}
// maximum over each aligned 16-lane row, result in every lane of the row
__device__ __forceinline__ float row16_allmax(float v) {
    v = fmaxf(v, dppf<0xB1>(v));          // quad_perm [1,0,3,2]
    v = fmaxf(v, dppf<0x4E>(v));          // quad_perm [2,3,0,1]
    v = fmaxf(v, dppf<0x141>(v));         // row_half_mirror
    v = fmaxf(v, dppf<0x140>(v));         // row_mirror
    return v;
}
__device__ __forceinline__ double row16_allmax(double v) {
    v = fmax(v, dppd<0xB1>(v));
    v = fmax(v, dppd<0x4E>(v));
    v = fmax(v, dppd<0x141>(v));
    v = fmax(v, dppd<0x140>(v));
    return v;
}
typedef unsigned v2u __attribute__((ext_vector_type(2)));
// Sum FOUR per-lane doubles over the 64 lanes in one pass ("transpose" reduction): returns, in every lane L, the wave
// total of quantity (L & 3).  DPP moves and two permlane swaps, no LDS traffic (a __shfl_xor tree is 12 ds_bpermute
// per value: 24 LDS cycles each).
__device__ __forceinline__ double wave_sum4(double q0, double q1, double q2, double q3, int lane) {
    const bool odd = lane & 1;
    const double k0 = odd ? q1 : q0, k1 = odd ? q3 : q2;
    const double s0 = odd ? q0 : q1, s1 = odd ? q2 : q3;
    const double a0 = k0 + dppd<0xB1>(s0);                    // quad_perm [1,0,3,2]
    const double a1 = k1 + dppd<0xB1>(s1);
    const bool b1 = lane & 2;
    const double kk = b1 ? a1 : a0, ss = b1 ? a0 : a1;
    double v = kk + dppd<0x4E>(ss);                           // quad_perm [2,3,0,1]
    v = v + dppd<0x124>(v);                                   // row_ror:4
    v = v + dppd<0x128>(v);                                   // row_ror:8
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
__device__ __forceinline__ float neg_inf(float) { return -INFINITY; }
__device__ __forceinline__ double neg_inf(double) { return -(double)INFINITY; }

// operands of a transition that do not depend on the live next_max (thrl_device.h td_value):
// float : c1 = fma(alpha, reward, (1-alpha)*old_value);  double: reward and (1-alpha)*old_value
template <typename QT> struct Ops;
template <> struct Ops<float> {
    uint32_t c1;
    __device__ __forceinline__ void set(float ov, double re, const TdCoef& c) {
        c1 = __builtin_bit_cast(uint32_t, __fmaf_rn(c.alpha_f, (float)re, __fmul_rn(c.one_minus_alpha_f, ov)));
    }
    __device__ __forceinline__ void gather(unsigned sel, const Ops<float>& src, bool mine) { const uint32_t v = bperm(sel, src.c1); if (mine) c1 = v; }
    template <int J> __device__ __forceinline__ Ops<float> bcast() const { Ops<float> o; o.c1 = dpp32<0x150 + J>(c1); return o; }
    __device__ __forceinline__ float value(float nm, float alpha_gamma, float, float) const {
        return __fmaf_rn(alpha_gamma, nm, __builtin_bit_cast(float, c1));
    }
};
template <> struct Ops<double> {
    uint32_t r_lo, r_hi, t_lo, t_hi;
    __device__ __forceinline__ void set(double ov, double re, const TdCoef& c) {
        const double t4 = __dmul_rn(c.one_minus_alpha, ov);
        r_lo = (uint32_t)__double2loint(re); r_hi = (uint32_t)__double2hiint(re);
        t_lo = (uint32_t)__double2loint(t4); t_hi = (uint32_t)__double2hiint(t4);
    }
    __device__ __forceinline__ void gather(unsigned sel, const Ops<double>& s, bool mine) {
        const uint32_t a = bperm(sel, s.r_lo), b = bperm(sel, s.r_hi), c = bperm(sel, s.t_lo), d = bperm(sel, s.t_hi);
        if (mine) { r_lo = a; r_hi = b; t_lo = c; t_hi = d; }
    }
    template <int J> __device__ __forceinline__ Ops<double> bcast() const {
        Ops<double> o;
        o.r_lo = dpp32<0x150 + J>(r_lo); o.r_hi = dpp32<0x150 + J>(r_hi); o.t_lo = dpp32<0x150 + J>(t_lo); o.t_hi = dpp32<0x150 + J>(t_hi);
        return o;
    }
    __device__ __forceinline__ double value(double nm, double, double alpha, double gamma) const {
        const double re = __hiloint2double((int)r_hi, (int)r_lo), t4 = __hiloint2double((int)t_hi, (int)t_lo);
        return __dadd_rn(t4, __dmul_rn(alpha, __dadd_rn(re, __dmul_rn(gamma, nm))));      // agents.py:72-74
    }
};

// One transition of every agent (agents.py:68-76): J = position inside the 16-step block (row_newbcast lane).
// xr / xc: byte offsets of the next-state row and of the rewritten cell inside the agent's table, lane l16 = step l16 of the
// block; NC = columns per lane (ceil(max A / 16)); tc0..tc3 = tab_me + the lane's column offsets.
// float: hand-written.  hipcc neither folds a DPP move into its user nor drops the canonicalising v_max of fmaxf (35 vector
// instructions per transition as C++); here the row_newbcast broadcasts ride on the address adds, the four steps of the
// 16-lane maximum are single v_max_f32_dpp, and the store swaps EXEC instead of branching: 11-15 vector instructions.  The
// s_mov / s_nop pairs are the two wait states a DPP read of a just-written VGPR needs (hipcc does not look inside asm, and it
// may have copied an operand right before it).  Same arithmetic: first-max select over the lane's columns, maximum over the
// row, fma(alpha * gamma, max, c1).
template <int J, int NC>
__device__ __forceinline__ void replay_step_f32(uint32_t xr, uint32_t xc, uint32_t xc1, unsigned tab_me, unsigned tc0, unsigned tc1, unsigned tc2,
                                                unsigned tc3, float alpha_gamma, unsigned long long smask) {
    unsigned t0, t1, t2, t3, sa;
    float m, v1, v2, v3, c;
    unsigned long long save;
    if (NC == 1) {
        asm volatile(
            "s_mov_b64 %[save], exec\n\ts_nop 0\n\t"
            "v_add_u32_dpp %[t0], %[xr], %[tc0] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %[sa], %[xc], %[tab] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "ds_read_b32 %[m], %[t0]\n\t"
            "v_mov_b32_dpp %[c], %[xc1] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] row_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_mov_b64 exec, %[mask]\n\t"
            "v_fmac_f32 %[c], %[ag], %[m]\n\t"
            "ds_write_b32 %[sa], %[c]\n\t"
            "s_mov_b64 exec, %[save]"
            : [save] "=&s"(save), [t0] "=&v"(t0), [sa] "=&v"(sa), [m] "=&v"(m), [c] "=&v"(c)
            : [xr] "v"(xr), [xc] "v"(xc), [xc1] "v"(xc1), [tab] "v"(tab_me), [tc0] "v"(tc0), [ag] "v"(alpha_gamma), [mask] "s"(smask), [j] "n"(J)
            : "memory");
    } else if (NC == 2) {
        asm volatile(
            "s_mov_b64 %[save], exec\n\ts_nop 0\n\t"
            "v_add_u32_dpp %[t0], %[xr], %[tc0] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %[t1], %[xr], %[tc1] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "ds_read_b32 %[m], %[t0]\n\t"
            "ds_read_b32 %[v1], %[t1]\n\t"
            "v_add_u32_dpp %[sa], %[xc], %[tab] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %[c], %[xc1] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_gt_f32 vcc, %[v1], %[m]\n\t"
            "v_cndmask_b32 %[m], %[m], %[v1], vcc\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] row_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_mov_b64 exec, %[mask]\n\t"
            "v_fmac_f32 %[c], %[ag], %[m]\n\t"
            "ds_write_b32 %[sa], %[c]\n\t"
            "s_mov_b64 exec, %[save]"
            : [save] "=&s"(save), [t0] "=&v"(t0), [t1] "=&v"(t1), [sa] "=&v"(sa), [m] "=&v"(m), [v1] "=&v"(v1), [c] "=&v"(c)
            : [xr] "v"(xr), [xc] "v"(xc), [xc1] "v"(xc1), [tab] "v"(tab_me), [tc0] "v"(tc0), [tc1] "v"(tc1), [ag] "v"(alpha_gamma),
              [mask] "s"(smask), [j] "n"(J)
            : "memory", "vcc");
    } else {
        asm volatile(
            "s_mov_b64 %[save], exec\n\ts_nop 0\n\t"
            "v_add_u32_dpp %[t0], %[xr], %[tc0] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %[t1], %[xr], %[tc1] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %[t2], %[xr], %[tc2] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %[t3], %[xr], %[tc3] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "ds_read_b32 %[m], %[t0]\n\t"
            "ds_read_b32 %[v1], %[t1]\n\t"
            "ds_read_b32 %[v2], %[t2]\n\t"
            "ds_read_b32 %[v3], %[t3]\n\t"
            "v_add_u32_dpp %[sa], %[xc], %[tab] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %[c], %[xc1] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_gt_f32 vcc, %[v1], %[m]\n\t"
            "v_cndmask_b32 %[m], %[m], %[v1], vcc\n\t"
            "v_cmp_gt_f32 vcc, %[v2], %[m]\n\t"
            "v_cndmask_b32 %[m], %[m], %[v2], vcc\n\t"
            "v_cmp_gt_f32 vcc, %[v3], %[m]\n\t"
            "v_cndmask_b32 %[m], %[m], %[v3], vcc\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_f32_dpp %[m], %[m], %[m] row_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_mov_b64 exec, %[mask]\n\t"
            "v_fmac_f32 %[c], %[ag], %[m]\n\t"
            "ds_write_b32 %[sa], %[c]\n\t"
            "s_mov_b64 exec, %[save]"
            : [save] "=&s"(save), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sa] "=&v"(sa), [m] "=&v"(m), [v1] "=&v"(v1),
              [v2] "=&v"(v2), [v3] "=&v"(v3), [c] "=&v"(c)
            : [xr] "v"(xr), [xc] "v"(xc), [xc1] "v"(xc1), [tab] "v"(tab_me), [tc0] "v"(tc0), [tc1] "v"(tc1), [tc2] "v"(tc2), [tc3] "v"(tc3),
              [ag] "v"(alpha_gamma), [mask] "s"(smask), [j] "n"(J)
            : "memory", "vcc");
    }
}

// the same in C++ for float64 tables (the reference's four separately rounded operations per TD value)
template <int J, int NC>
__device__ __forceinline__ void replay_step_f64(uint32_t xr, uint32_t xc, const Ops<double>& xo, unsigned tab_me, unsigned tc0, unsigned tc1, unsigned tc2,
                                                unsigned tc3, bool storer, double alpha, double gamma) {
    const uint32_t roff = dpp32<0x150 + J>(xr), coff = dpp32<0x150 + J>(xc);
    const Ops<double> o = xo.template bcast<J>();
    double m = lds_load<double>(tc0 + roff);
    if (NC > 1) { const double v = lds_load<double>(tc1 + roff); m = v > m ? v : m; }
    if (NC > 2) { const double v = lds_load<double>(tc2 + roff); m = v > m ? v : m; }
    if (NC > 2) { const double v = lds_load<double>(tc3 + roff); m = v > m ? v : m; }
    m = row16_allmax(m);
    const double val = o.value(m, 0.0, alpha, gamma);
    if (storer) lds_store<double>(tab_me + coff, val);
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+s"(x)); return x; }
// scalar bit-field extract with a prepared operand (offset | width << 16): one s_bfe_u32 (hipcc emits a shift and an AND)
__device__ __forceinline__ uint32_t sbfe(uint32_t x, uint32_t op) { uint32_t r; asm("s_bfe_u32 %0, %1, %2" : "=s"(r) : "s"(x), "s"(op) : "scc"); return r; }
// the u16 at a wave-uniform LDS address, as a scalar (ds_read_u16 zero-extends: no masking afterwards)
__device__ __forceinline__ uint32_t lds_u16_uniform(uint32_t addr) {
    uint32_t v, r;
    asm volatile("ds_read_u16 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, %0" : "=&v"(v), "=s"(r) : "v"(addr) : "memory");
    return r;
}
// sixteen (FULL) or nb < 16 consecutive transitions
template <typename QT, int NC, bool FULL>
__device__ __forceinline__ void replay_block(int nb, uint32_t xr, uint32_t xc, const Ops<QT>& xo, unsigned tab_me, unsigned tc0, unsigned tc1, unsigned tc2,
                                             unsigned tc3, bool storer, unsigned long long smask, QT alpha_gamma, QT alpha, QT gamma) {
#define THRL_TUP_STEP(J)                                                                                                         \
    if (FULL || (J) < opaque(nb)) {      /* (a scalar compare per step, not sixteen hoisted-and-spilled booleans) */             \
        if constexpr (std::is_same<QT, float>::value) replay_step_f32<J, NC>(xr, xc, xo.c1, tab_me, tc0, tc1, tc2, tc3, alpha_gamma, smask); \
        else replay_step_f64<J, NC>(xr, xc, xo, tab_me, tc0, tc1, tc2, tc3, storer, alpha, gamma);                              \
    }
    THRL_TUP_STEP(0) THRL_TUP_STEP(1) THRL_TUP_STEP(2) THRL_TUP_STEP(3) THRL_TUP_STEP(4) THRL_TUP_STEP(5) THRL_TUP_STEP(6) THRL_TUP_STEP(7)
    THRL_TUP_STEP(8) THRL_TUP_STEP(9) THRL_TUP_STEP(10) THRL_TUP_STEP(11) THRL_TUP_STEP(12) THRL_TUP_STEP(13) THRL_TUP_STEP(14) THRL_TUP_STEP(15)
#undef THRL_TUP_STEP
}

template <typename QT, int N, int NSEG, bool NOISE, bool SWEEP>
__global__ void __launch_bounds__(1024) k_tuple_episodes(const TupleArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int T = a.T, tuples = a.tuples;
    {   // stage the LUT (rows per tuple and agent, per-action quantities) once per block
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.lut);
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem);
        for (int k = threadIdx.x; k < (a.lut_lds_bytes >> 2); k += blockDim.x) dst[k] = src[k];
    }
    __syncthreads();
    const uint32_t* prow = reinterpret_cast<const uint32_t*>(smem);                                  // [tuples] play rows, byte i = agent i
    const uint32_t* trow = prow + tuples;                                                            // [tuples] train rows
    const double* lut_aq = reinterpret_cast<const double*>(smem + a.aq_off);                        // [N][64]
    const double* lut_sct = lut_aq + N * 64;                                                         // [N][64]
    const double* price_lut = reinterpret_cast<const double*>(a.lut + a.price_off);                 // [tuples], HBM / L2
    const double* qsum_lut = reinterpret_cast<const double*>(a.lut + a.qsum_off);                   // [tuples], HBM / L2 (NOISE)
    unsigned char* game = smem + a.lut_lds_bytes + (size_t)wib * a.game_lds_bytes;
    QT* const tabs = reinterpret_cast<QT*>(game);
    unsigned char* const am = game + a.am_off;
    unsigned short* const gt = reinterpret_cast<unsigned short*>(game + a.g_off);                    // [tuples + 1]
    typedef unsigned __attribute__((may_alias)) hist_u32;
    hist_u32* const hist = reinterpret_cast<hist_u32*>(game);             // overlays the tables once they are back in HBM
    // visit log of this wave: one word of N u16 cells per step
    typedef typename std::conditional<(N <= 2), uint32_t, uint64_t>::type VW;
    VW* const vlog = reinterpret_cast<VW*>(reinterpret_cast<char*>(a.vlog) + ((int64_t)blockIdx.x * a.waves_per_block + wib) * a.vlog_wave_bytes);
    // action words: agent i's action = bits [sh_i, sh_i + nb_i)
    int sh[N];
    uint32_t fm[N], fmask[N];
#pragma unroll
    for (int i = 0; i < N; i++) { sh[i] = a.act_sh[i]; fm[i] = (1u << a.act_bits[i]) - 1u; fmask[i] = fm[i] << sh[i]; }
    // the chain carries the LDS byte address of G[tau]: gbase + 2 tau = gbase + sum_i a_i * gstr[i] (gstr = 2 x the mixed-radix
    // weight).  The last agent's weight is 1 and its field sits at bit 1, so its term is the masked word itself.
    const uint32_t gbase = lds_addr(gt);
    uint32_t gstr[N], bfeop[N];
    {
        uint32_t wgt = 2u;
#pragma unroll
        for (int i = N - 1; i >= 0; i--) { gstr[i] = wgt; wgt *= (uint32_t)a.ag[i].n_actions; bfeop[i] = (uint32_t)sh[i] | ((uint32_t)a.act_bits[i] << 16); }
    }
#define THRL_FLD(x, i) (((uint32_t)(x) >> sh[i]) & fm[i])

    // replay ("exec") layout: agent my_ag owns the 16-lane row `lane >> 4`
    const int my_ag = lane >> 4, l16 = lane & 15;
    const bool ag_ok = my_ag < N;
    const AgentParams& pme = a.ag[ag_ok ? my_ag : 0];
    const int A_me = pme.n_actions;
    const unsigned a_bytes = (unsigned)A_me * (unsigned)sizeof(QT);
    const unsigned tab_me = lds_addr(tabs + a.tab_off[ag_ok ? my_ag : 0]);
    const unsigned col_b0 = (unsigned)min(l16, A_me - 1) * (unsigned)sizeof(QT), col_b1 = (unsigned)min(l16 + 16, A_me - 1) * (unsigned)sizeof(QT);
    const unsigned col_b2 = (unsigned)min(l16 + 32, A_me - 1) * (unsigned)sizeof(QT), col_b3 = (unsigned)min(l16 + 48, A_me - 1) * (unsigned)sizeof(QT);
    int amax = 1;
#pragma unroll
    for (int i = 0; i < N; i++) amax = max(amax, a.ag[i].n_actions);
    const int ncol = (amax + 15) >> 4;
    const bool storer = ag_ok && l16 == 0;
    const unsigned long long smask = __ballot(storer);            // the lanes that store a transition's TD value: lane 16 i, i < N
    const unsigned tc0 = tab_me + col_b0, tc1 = tab_me + col_b1, tc2 = tab_me + col_b2, tc3 = tab_me + col_b3;
    const TdCoef tc_me = td_coef(pme);
    const QT alpha_me = std::is_same<QT, float>::value ? (QT)tc_me.alpha_f : (QT)tc_me.alpha;
    const QT gamma_me = (QT)pme.gamma;
    const QT ag_me = std::is_same<QT, float>::value ? (QT)tc_me.alpha_gamma_f : (QT)0;

    // per-wave log sums: slot = episode * 8 + k (k < N: reward of agent k; 4 <= k < 4 + N: action), lane = slot & 63
    double accd[4] = {0.0, 0.0, 0.0, 0.0};

    int g_claim = 0;
    if (lane == 0) g_claim = atomicAdd(a.next_game, 1);
    for (;;) {
        const int g = __builtin_amdgcn_readfirstlane(g_claim);
        if (g >= a.G) break;
        if (lane == 0) g_claim = atomicAdd(a.next_game, 1);
        const uint64_t gid = a.game_offset + (uint64_t)g;
        QT* __restrict__ qg = reinterpret_cast<QT*>(a.q) + (int64_t)g * a.stride;
        const double price0 = a.state[g];
        // SWEEP: this game's own gamma / alpha / epsilon schedule / noise_prob (thrl_buffers.sweep_*: the reference's research
        // loop is configs x runs; here a config is a game), derived exactly as fill_agents() derives the scalars
        TdCoef tcs[N];
        double epsg[N];
        double np_g = a.env.noise_prob;
        QT alpha_g = alpha_me, gamma_g = gamma_me, ag_g = ag_me;
#pragma unroll
        for (int i = 0; i < N; i++) {
            tcs[i] = td_coef(a.ag[i]);
            epsg[i] = 0.0;
            if (SWEEP) {
                const size_t k = (size_t)i * (size_t)a.G + (size_t)g;
                const double al = a.sw_alpha ? a.sw_alpha[k] : a.ag[i].alpha, ga = a.sw_gamma ? a.sw_gamma[k] : a.ag[i].gamma;
                tcs[i] = td_coef(al, ga);
                epsg[i] = a.sw_eps ? a.sw_eps[k] : a.eps[0][i];
                if (my_ag == i) {
                    alpha_g = std::is_same<QT, float>::value ? (QT)tcs[i].alpha_f : (QT)tcs[i].alpha;
                    gamma_g = (QT)ga;
                    ag_g = std::is_same<QT, float>::value ? (QT)tcs[i].alpha_gamma_f : (QT)0;
                }
            }
        }
        if (SWEEP && a.sw_noise_prob && a.env.noise_prob > 0.0) np_g = a.sw_noise_prob[g];     // (as the generic kernel: a sweep of a noisy env)

        // ---- tables -> LDS (window rows are contiguous in HBM), initial state -> local rows (window or spill)
        int init_play[N], init_train[N], spill_p[N], spill_t[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            const AgentParams& p = a.ag[i];
            const int A = p.n_actions, W = a.win_rows[i], lo = a.row_lo[i];
            QT* t = tabs + a.tab_off[i];
            const QT* src = qg + p.table_off + lo * A;
            for (int k = lane; k < W * A; k += 64) t[k] = src[k];
            const int sp = __builtin_amdgcn_readfirstlane(encode32(price0, p)), st = __builtin_amdgcn_readfirstlane(encode64(price0, p));
            spill_p[i] = spill_t[i] = -1;
            if (sp >= lo && sp < lo + W) init_play[i] = sp - lo; else { spill_p[i] = sp; init_play[i] = W; }
            if (st == sp) init_train[i] = init_play[i];
            else if (st >= lo && st < lo + W) init_train[i] = st - lo;
            else { spill_t[i] = st; init_train[i] = W + 1; }
            if (lane < A) {
                // (both spill rows are always filled: a row nobody addresses is never read)
                t[W * A + lane] = qg[p.table_off + (spill_p[i] >= 0 ? spill_p[i] : 0) * A + lane];
                t[(W + 1) * A + lane] = qg[p.table_off + (spill_t[i] >= 0 ? spill_t[i] : 0) * A + lane];
            }
        }
        __builtin_amdgcn_wave_barrier();

        int tau = tuples;                    // current state: action tuple of the last step; `tuples` = the launch's initial state
        uint32_t ga = gbase + 2u * (uint32_t)tuples;          // ... carried as the address of G[tau]
        // NOISE: a step whose intercept was redrawn (environments.py:29-31) leaves the action grid -- the state after it is
        // "off": its rows are carried explicitly (byte i = agent i's window-local play / train row) instead of a tuple.
        // The launch's initial state is handled the same way.
        bool off = NOISE;
        uint32_t offp = 0u, offt = 0u;
        double p_off = price0;
        uint32_t init_pw = 0u, init_tw = 0u;          // the initial state's rows, packed as prow / trow are
#pragma unroll
        for (int i = 0; i < N; i++) { init_pw |= (uint32_t)init_play[i] << (8 * i); init_tw |= (uint32_t)init_train[i] << (8 * i); }
        if (NOISE) { tau = 0; ga = gbase; offp = init_pw; offt = init_tw; }
        for (int e = 0; e < a.n_episodes; e++) {
            const uint32_t eg = (uint32_t)(a.first_episode + (uint64_t)e);

            // ---- (a) greedy action of every local row (first max wins, numpy.argmax: agents.py:85), lane = row
#pragma unroll
            for (int i = 0; i < N; i++) {
                if (kTupAblate & 16) break;
                const int A = a.ag[i].n_actions, R = a.win_rows[i] + 2;
                const QT* t = tabs + a.tab_off[i];
                for (int base = 0; base < R; base += 64) {
                    const int row = min(base + lane, R - 1);
                    const QT* r = t + row * A;
                    QT b = r[0];
                    int bi = 0;
                    // (eight loads in flight per round trip instead of one: the row reads do not depend on the compares)
                    int j = 1;
                    for (; j + 8 <= A; j += 8) {
                        QT v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = r[j + u];
#pragma unroll
                        for (int u = 0; u < 8; u++) if (v[u] > b) { b = v[u]; bi = j + u; }
                    }
                    for (; j < A; j++) { const QT v = r[j]; if (v > b) { b = v; bi = j; } }
                    if (base + lane < R) am[a.am_off_i[i] + row] = (unsigned char)bi;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- (b) G[tau] = the agents' greedy actions in state tau, one action word (entry `tuples`: the initial state)
            for (int base = 0; base <= tuples && !(kTupAblate & 4); base += 256) {         // four batches of 64 tuples in flight
                uint32_t pw[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int tq = min(base + u * 64 + lane, tuples);
                    pw[u] = prow[tq];                           // (entry `tuples` reads the next table: replaced below)
                    if (tq == tuples) pw[u] = init_pw;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    uint32_t packed = 0u;
#pragma unroll
                    for (int i = 0; i < N; i++) packed |= (uint32_t)am[a.am_off_i[i] + (int)((pw[u] >> (8 * i)) & 0xFFu)] << sh[i];
                    if (base + u * 64 + lane <= tuples) gt[base + u * 64 + lane] = (unsigned short)packed;
                }
            }
            __builtin_amdgcn_wave_barrier();

            // ---- (c) draws, lane = step: Mw field i = all ones where agent i explores, Cw field i = its random choice
            uint32_t Mw[NSEG], Cw[NSEG];
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                const int tt = min(seg * 64 + lane, T - 1);
                uint32_t mw = 0u, cw = 0u;
#pragma unroll
                for (int i = 0; i < N; i++) {
                    const AgentParams& p = a.ag[i];
                    double u; uint32_t ch;
                    if (kTupAblate & 8) { u = 1.0; ch = 0u; } else
                    if (a.inj_u) {           // parity mode: the reference's recorded draws [E][T][N][G] (agents.py:81-82)
                        const size_t k = (((size_t)e * T + tt) * N + i) * (size_t)a.G + (size_t)g;
                        u = a.inj_u[k]; ch = min((uint32_t)(uint8_t)a.inj_choice[k], (uint32_t)(p.n_actions - 1));
                    } else {
                        const u32x4 x = draw(a.seed, gid, eg, (uint32_t)tt, (uint32_t)(i >> 1));
                        u = u01_32((i & 1) ? x.z : x.x);
                        ch = __umulhi((i & 1) ? x.w : x.y, (uint32_t)p.n_actions);
                    }
                    if (u < ((SWEEP && a.sw_eps) ? epsg[i] : a.eps[e][i])) mw |= fmask[i];
                    cw |= ch << sh[i];
                }
                Mw[seg] = mw; Cw[seg] = cw;
            }
            // NOISE: the env's draw of every step (environments.py:28-31): nzm bit t = intercept redrawn, NA = its value
            uint64_t nzm[NSEG];
            double NA[NSEG];
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                nzm[seg] = 0ull; NA[seg] = a.env.a;
                if (NOISE) {
                    const int tt = min(seg * 64 + lane, T - 1);
                    double nu, na;
                    if (a.inj_u) {
                        const size_t k = ((size_t)e * T + tt) * (size_t)a.G + (size_t)g;
                        nu = a.inj_noise_u[k]; na = a.inj_noise_a[k];
                    } else {
                        const u32x4 xn = draw(a.seed, gid, eg, (uint32_t)tt, kStreamNoise);
                        nu = u01_32(xn.x);
                        na = __dadd_rn(a.env.noise_lo, __dmul_rn(__dsub_rn(a.env.a, a.env.noise_lo), u01_32(xn.y)));
                    }
                    nzm[seg] = __ballot(seg * 64 + lane < T && nu < (SWEEP ? np_g : a.env.noise_prob));
                    NA[seg] = na;
                }
            }

            // ---- (d) play: the serial chain, on the scalar unit.  seq[seg] lane t = state (tuple) step t was played in
            // (NOISE: the packed train rows where that state is off the grid; acts[seg] lane t = the actions of step t)
            uint32_t seq[NSEG], acts[NSEG];
            const bool off0 = off;
            // is the state step (seg, lane) is played in off the grid (NOISE: the step before it redrew the intercept)?
            auto state_off = [&](int seg) -> bool {
                if (!NOISE) return false;
                return lane > 0 ? (bool)((nzm[seg] >> (lane - 1)) & 1ull) : (seg > 0 ? (bool)((nzm[seg > 0 ? seg - 1 : 0] >> 63) & 1ull) : off0);
            };
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                seq[seg] = 0u; acts[seg] = 0u;
                const int n = (kTupAblate & 2) ? 0 : min(64, T - seg * 64);
                const uint32_t CM = Cw[seg] & Mw[seg], NM = ~Mw[seg] & 0xFFFFu;         // per step: the explorers' choices; the greedy agents' fields
                for (int tl = 0; tl < n; tl++) {
                    uint32_t w;
                    if (NOISE && off) {
                        uint32_t wv = 0u;
#pragma unroll
                        for (int i = 0; i < N; i++) wv |= (uint32_t)am[a.am_off_i[i] + (int)((offp >> (8 * i)) & 0xFFu)] << sh[i];
                        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)wv);
                    } else {
                        w = lds_u16_uniform(ga);
                    }
                    const uint32_t cm = (uint32_t)__builtin_amdgcn_readlane((int)CM, tl), nm = (uint32_t)__builtin_amdgcn_readlane((int)NM, tl);
                    const uint32_t ap = (w & nm) | cm;
                    const uint32_t rec = (NOISE && off) ? offt : ga;         // (addresses become tuples below, lane-parallel)
                    asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(seq[seg]) : "s"(rec), "s"(tl));
                    if (NOISE) asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(acts[seg]) : "s"(ap), "s"(tl));
                    // address of G[tau'], tau' = mixed radix of the actions: the products are independent, the sums a tree
                    uint32_t t0 = gbase + (ap & fmask[N - 1]), t1 = 0u;
                    if (N >= 2) t1 = sbfe(ap, bfeop[N >= 2 ? N - 2 : 0]) * gstr[N >= 2 ? N - 2 : 0];
                    if (N >= 3) t0 += sbfe(ap, bfeop[N >= 3 ? N - 3 : 0]) * gstr[N >= 3 ? N - 3 : 0];
                    if (N >= 4) t1 += sbfe(ap, bfeop[0]) * gstr[0];
                    ga = t0 + t1;
                    if (NOISE) {
                        off = (nzm[seg] >> tl) & 1ull;
                        if (off) {           // environments.py:29-33 with the redrawn intercept; rows by both encodes
                            const double na = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(NA[seg]), tl),
                                                               __builtin_amdgcn_readlane(__double2loint(NA[seg]), tl));
                            double pr = __dsub_rn(na, __dmul_rn(a.env.b, qsum_lut[(ga - gbase) >> 1]));
                            if (!(pr > 0.0)) pr = 0.0;
                            uint32_t op = 0u, ot = 0u;
#pragma unroll
                            for (int i = 0; i < N; i++) {
                                const int W = a.win_rows[i];
                                op |= (uint32_t)min(max(encode32(pr, a.ag[i]) - a.row_lo[i], 0), W - 1) << (8 * i);
                                ot |= (uint32_t)min(max(encode64(pr, a.ag[i]) - a.row_lo[i], 0), W - 1) << (8 * i);
                            }
                            offp = (uint32_t)__builtin_amdgcn_readfirstlane((int)op);
                            offt = (uint32_t)__builtin_amdgcn_readfirstlane((int)ot);
                            p_off = pr;
                        }
                    }
                }
                if (!state_off(seg)) seq[seg] = (seq[seg] - gbase) >> 1;           // on-grid states: address of G[tau] -> tau
                if (kTupAblate & 2) seq[seg] = 0u;                                  // (chain skipped: keep the indices in range)
            }
            tau = (int)((ga - gbase) >> 1);
            const int tau_end = tau;

            // ---- (e) lane-parallel over the steps: actions, rows, prices, rewards, old-value snapshot (agents.py:67), logs
            uint32_t wr[NSEG][N], wc[NSEG][N];           // byte offsets inside the agent's table: next-state row, rewritten cell
            Ops<QT> ops[NSEG][N];
            double lr[N], la[N];
#pragma unroll
            for (int i = 0; i < N; i++) { lr[i] = 0.0; la[i] = 0.0; }
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                const int tt = seg * 64 + lane;
                const bool valid = tt < T;
                const uint32_t end_word = (NOISE && off) ? offt : (uint32_t)tau_end;
                uint32_t tq = valid ? seq[seg] : end_word;
                uint32_t nxt = (uint32_t)__shfl_down((int)seq[seg], 1, 64);
                if (seg + 1 < NSEG) { if (lane == 63) nxt = (uint32_t)__builtin_amdgcn_readlane((int)seq[seg + 1 < NSEG ? seg + 1 : seg], 0); }
                if (tt + 1 >= T) nxt = end_word;
                // NOISE: is my state / the state after my step off the grid (its word = packed train rows, not a tuple)?
                bool s_off = false, n_off = false;
                if (NOISE) {
                    n_off = valid && ((nzm[seg] >> lane) & 1ull);
                    s_off = state_off(seg);
                    if (!valid) { s_off = false; tq = 0u; n_off = false; nxt = 0u; }
                }
                uint32_t ap;
                double price;
                VW vw = 0;
                if (NOISE) {
                    ap = acts[seg];
                    int nt = 0;
#pragma unroll
                    for (int i = 0; i < N; i++) nt = nt * a.ag[i].n_actions + (int)THRL_FLD(ap, i);
                    if (!valid) nt = 0;
                    price = price_lut[nt];
                    if (n_off) {
                        price = __dsub_rn(NA[seg], __dmul_rn(a.env.b, qsum_lut[nt]));
                        if (!(price > 0.0)) price = 0.0;
                    }
                } else {
                    const uint32_t w = gt[tq];
                    ap = (Cw[seg] & Mw[seg]) | (w & ~Mw[seg]);
                    price = price_lut[nxt];
                }
                // train rows of my state and of the state after my step, byte i = agent i
                uint32_t sw, nw;
                if (NOISE) {
                    sw = trow[s_off ? 0u : tq]; if (s_off) sw = tq;
                    nw = trow[n_off ? 0u : nxt]; if (n_off) nw = nxt;
                } else {
                    sw = trow[tq]; if (tq == (uint32_t)tuples) sw = init_tw;       // (entry `tuples` reads the start of aq: replaced)
                    nw = trow[nxt];
                }
#pragma unroll
                for (int i = 0; i < N; i++) {
                    const AgentParams& p = a.ag[i];
                    const int A = p.n_actions;
                    const uint32_t act = valid ? THRL_FLD(ap, i) : 0u;
                    const uint32_t srow = (sw >> (8 * i)) & 0xFFu, ns = (nw >> (8 * i)) & 0xFFu;
                    const uint32_t cell = valid ? srow * (uint32_t)A + act : 0u;
                    const double aq = lut_aq[i * 64 + act];
                    const double re = __dmul_rn(price, aq);                       // environments.py:33
                    const QT ov = tabs[a.tab_off[i] + cell];
                    ops[seg][i].set(ov, re, SWEEP ? tcs[i] : td_coef(p));
                    wr[seg][i] = ns * (uint32_t)A * (uint32_t)sizeof(QT);
                    wc[seg][i] = cell * (uint32_t)sizeof(QT);
                    if (valid) { lr[i] += re; la[i] += lut_sct[i * 64 + act]; }
                    vw |= (VW)cell << (16 * i);            // visit counter of the transition (agents.py:76): logged, counted after the launch's last episode
                }
                if (valid && a.counter && !(kTupAblate & 32)) vlog[e * T + tt] = vw;
            }
            __builtin_amdgcn_wave_barrier();

            // ---- (f) replay = train_net's loop (agents.py:68-76), every agent at once, one transition per agent at a time
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int base_t = seg * 64 + b * 16;
                    if (base_t >= T) break;
                    if (kTupAblate & 1) {          // keep the operands alive
#pragma unroll
                        for (int i = 0; i < N; i++) asm volatile("" :: "v"(wr[seg][i]), "v"(wc[seg][i]), "v"(*reinterpret_cast<const uint32_t*>(&ops[seg][i])));
                        break;
                    }
                    const unsigned sel = (unsigned)(b * 16 + l16) << 2;
                    uint32_t xr = 0u, xc = 0u;
                    Ops<QT> xo = ops[seg][0];
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        const uint32_t vr = bperm(sel, wr[seg][i]), vc = bperm(sel, wc[seg][i]);
                        if (my_ag == i) { xr = vr; xc = vc; }
                        xo.gather(sel, ops[seg][i], my_ag == i);
                    }
                    const int nb = min(16, T - base_t);
                    const QT agx = SWEEP ? ag_g : ag_me, alx = SWEEP ? alpha_g : alpha_me, gax = SWEEP ? gamma_g : gamma_me;
#define THRL_TUP_BLOCK(NC)                                                                                                      \
                    if (nb == 16) replay_block<QT, NC, true>(nb, xr, xc, xo, tab_me, tc0, tc1, tc2, tc3, storer, smask, agx, alx, gax); \
                    else replay_block<QT, NC, false>(nb, xr, xc, xo, tab_me, tc0, tc1, tc2, tc3, storer, smask, agx, alx, gax);
                    if (ncol == 1) { THRL_TUP_BLOCK(1) } else if (ncol == 2) { THRL_TUP_BLOCK(2) } else { THRL_TUP_BLOCK(4) }
#undef THRL_TUP_BLOCK
                }
            }

            // ---- (g) log sums of this game and episode into the wave accumulators (mean over games: host / finalize)
            if (kTupAblate & 64) {
#pragma unroll
                for (int i = 0; i < N; i++) asm volatile("" :: "v"(lr[i]), "v"(la[i]));
            } else {
                // lane L of tr: reward total of agent L & 3; of ta: action total (already divided by T per step)
                const double tr = __ddiv_rn(wave_sum4(lr[0], N > 1 ? lr[N > 1 ? 1 : 0] : 0.0, N > 2 ? lr[N > 2 ? 2 : 0] : 0.0,
                                                      N > 3 ? lr[N > 3 ? 3 : 0] : 0.0, lane), (double)T);
                const double ta = wave_sum4(la[0], N > 1 ? la[N > 1 ? 1 : 0] : 0.0, N > 2 ? la[N > 2 ? 2 : 0] : 0.0,
                                            N > 3 ? la[N > 3 ? 3 : 0] : 0.0, lane);
                // slot e*8 + k lives in lane (e*8 + k) & 63 of accd[(e*8 + k) >> 6]: lanes with (lane >> 3) == (e & 7)
                const int k = lane & 7;
                const double v = k < 4 ? tr : ta;              // (lane & 3) == (k & 3): the right agent's total
                if ((lane >> 3) == (e & 7) && (k & 3) < N) {
#pragma unroll
                    for (int r = 0; r < 4; r++) if ((e >> 3) == r) accd[r] += v;
                }
            }
            if (SWEEP) {                      // epsilon decays after every train_net call (agents.py:78), per game
#pragma unroll
                for (int i = 0; i < N; i++) {
                    const size_t k = (size_t)i * (size_t)a.G + (size_t)g;
                    const double eend = a.sw_eps_end ? a.sw_eps_end[k] : a.ag[i].eps_end;
                    const double estep = a.sw_eps_step ? a.sw_eps_step[k] : a.ag[i].eps_step;
                    epsg[i] = __dadd_rn(eend, __dmul_rn(__dsub_rn(epsg[i], eend), estep));
                }
            }
        }
        if (SWEEP && a.sw_eps && lane == 0) {
#pragma unroll
            for (int i = 0; i < N; i++) a.sw_eps[(size_t)i * (size_t)a.G + (size_t)g] = epsg[i];
        }

        // ---- tables back to HBM, env state
#pragma unroll
        for (int i = 0; i < N; i++) {
            const AgentParams& p = a.ag[i];
            const int A = p.n_actions, W = a.win_rows[i], lo = a.row_lo[i];
            const QT* t = tabs + a.tab_off[i];
            QT* dst = qg + p.table_off + lo * A;
            for (int k = lane; k < W * A; k += 64) dst[k] = t[k];
            if (lane < A) {
                if (spill_p[i] >= 0) qg[p.table_off + spill_p[i] * A + lane] = t[W * A + lane];
                if (spill_t[i] >= 0) qg[p.table_off + spill_t[i] * A + lane] = t[(W + 1) * A + lane];
            }
        }
        // ---- visit counters (agents.py:76).  The tables are back in HBM, so their LDS region is free: fold the launch's visit
        //      log into a u16 histogram there (E*T <= 32*256 < 65536: no carry) and apply it with plain coalesced read-add-write
        //      (a game's counters belong to this wave alone: no global atomics)
        if (a.counter && !(kTupAblate & 32)) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            for (int k = lane; k < a.hist_dwords; k += 64) hist[k] = 0u;
            __builtin_amdgcn_wave_barrier();
            const int total = a.n_episodes * T;
            for (int k0 = 0; k0 < total; k0 += 512) {          // eight loads in flight (L2-served: the wave reads what it stored itself)
                VW w[8];
#pragma unroll
                for (int j = 0; j < 8; j++)
                    w[j] = __hip_atomic_load(&vlog[min(k0 + j * 64 + lane, total - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (k0 + j * 64 + lane < total) {
#pragma unroll
                        for (int i = 0; i < N; i++) {
                            const unsigned cell = (unsigned)(w[j] >> (16 * i)) & 0xFFFFu;
                            __hip_atomic_fetch_add(&hist[a.hist_off_i[i] + (cell >> 1)], 1u << ((cell & 1u) << 4), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < N; i++) {
                const AgentParams& p = a.ag[i];
                const int A = p.n_actions, W = a.win_rows[i], lo = a.row_lo[i];
                int32_t* cg = a.counter + (int64_t)g * a.stride + p.table_off;
                const hist_u32* h = hist + a.hist_off_i[i];
                for (int k = lane; k < W * A; k += 64) {
                    const unsigned n = (h[k >> 1] >> ((k & 1) << 4)) & 0xFFFFu;
                    if (n) cg[lo * A + k] += (int32_t)n;
                }
                if (lane < 2 * A) {
                    const int which = lane >= A, col = lane - which * A, k = (W + which) * A + col;
                    const int grow = which ? spill_t[i] : spill_p[i];
                    const unsigned n = (h[k >> 1] >> ((k & 1) << 4)) & 0xFFFFu;
                    if (grow >= 0 && n) cg[grow * A + col] += (int32_t)n;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        if (lane == 0 && a.n_episodes > 0) a.state[g] = (NOISE && off) ? p_off : price_lut[tau];
        __builtin_amdgcn_wave_barrier();
    }

    // ---- the wave's log sums -> the launch's [E][N] sums (float64 atomics: the mean logs are compared to 1e-12)
    if (a.sum_reward) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int slot = r * 64 + lane, e = slot >> 3, k = slot & 7;
            if (e < a.n_episodes && (k & 3) < N && accd[r] != 0.0) {
                if (k < 4) atomicAdd(&a.sum_reward[(size_t)e * N + k], accd[r]);
                else atomicAdd(&a.sum_action[(size_t)e * N + (k - 4)], accd[r]);
            }
        }
    }
}

template <typename QT, int N, bool NOISE, bool SWEEP>
static int launch_tuple_n(const TupleArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    const int nseg = (a.T + 63) / 64;
#define THRL_TUP_LAUNCH(NS)                                                                                          \
    {                                                                                                                \
        auto kern = k_tuple_episodes<QT, N, NS, NOISE, SWEEP>;                                                       \
        if (lds > 64 * 1024) {                                                                                       \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                            \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
            if (e != hipSuccess) return (int)e;                                                                      \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, a);                                                \
        return (int)hipGetLastError();                                                                               \
    }
    if (nseg <= 1) THRL_TUP_LAUNCH(1)
    if (nseg == 2) THRL_TUP_LAUNCH(2)
    THRL_TUP_LAUNCH(4)
#undef THRL_TUP_LAUNCH
}
#undef THRL_FLD

template <typename QT, bool NOISE, bool SWEEP>
static int launch_tuple_t(const TupleArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    switch (a.N) {
        case 1: return launch_tuple_n<QT, 1, NOISE, SWEEP>(a, grid, block, lds, s);
        case 2: return launch_tuple_n<QT, 2, NOISE, SWEEP>(a, grid, block, lds, s);
        case 3: return launch_tuple_n<QT, 3, NOISE, SWEEP>(a, grid, block, lds, s);
        case 4: return launch_tuple_n<QT, 4, NOISE, SWEEP>(a, grid, block, lds, s);
    }
    return -1;
}

}  // namespace tup
}  // namespace thrl
