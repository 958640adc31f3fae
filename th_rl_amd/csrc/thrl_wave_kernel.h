// thrl_wave_kernel.h -- fused episode kernel for gfx950: ONE WAVEFRONT PER GAME (template body).
//
// The performance path for the reference's headline shape: 2 QTable agents with the same grid
// sizes on one NoisyPriceState.  Table type QT = float (the metric's dtype) or double (the
// reference's own numerics, agents.py:29).  Same semantics as thrl_generic.hip and bit-identical
// results to it and to the oracle in the same dtype: trainer.train_one's loop (th_rl/trainer.py:46-70)
// with QTable.sample_action (agents.py:80-89), scale (:51-57), NoisyPriceState.step
// (environments.py:25-39), ReplayBuffer append/replay/empty (buffers.py) and QTable.train_net
// (agents.py:59-78) fused, for `n_episodes` episodes per launch.
//
// MI355X mapping:
//   * a wave owns one game for the whole launch; its two agents' Q-table WINDOWS (only the rows the
//     payoff grid can reach, + 2 spill rows for an arbitrary initial state) are streamed coalesced
//     HBM -> LDS once, stay resident for all episodes of the launch, and are streamed back once;
//   * lanes 0-31 serve agent 0, lanes 32-63 agent 1;
//   * the discretised action grid (next-state row + price per action pair) is a LUT staged in LDS
//     once per block ("payoff LUT");
//   * everything that is not on a serial chain is done lane-parallel over the T steps of an episode
//     (lane = step) or over the table rows (lane = row): Philox draws, per-row greedy actions, reward /
//     old-value gathers, log sums, the replay schedule;
//   * play chain: s <- next_row_t[s], one v_readlane per step;
//   * replay (train_net's serial loop, agents.py:68-76) runs FOUR transitions per pass: each 32-lane
//     half is split into four 8-lane groups, group k reads the whole next-state row of transition
//     4g+k (3 columns per lane), one v_max3 + three DPP steps give every lane of the group that row's
//     max, and the lane holding transition 4g+k's operands computes the TD value and stores it.
//     A pass is legal when no transition in it reads a row, or rewrites a cell, that an earlier
//     transition of the same pass writes; the schedule (where a group of four must be cut into
//     several passes) is computed lane-parallel before the loop, so results are exactly those of the
//     serial loop;
//   * the kernel is bound by the number of instructions a wave issues, so the hot loops are
//     straight-line code: the play loop is unrolled over a segment's 16 groups of four steps and the replay
//     loop over a block's 8 groups (step / lane numbers are immediates), with copies for full segments and
//     full blocks that carry no bounds tests;
//   * GREEDY variants (launched once epsilon is small): groups of four steps in which nobody explores read
//     their next rows from per-episode composed tables, and a segment that cycles through 1-4 transitions
//     (a converged game) is replayed as a recurrence in registers (cyclic_segment).
#pragma once
#include <type_traits>
#include "thrl_kernels.h"
#include "thrl_wave_lut.h"

// THRL_ABLATE: timing-only diagnostic builds (python -m th_rl_amd.build --ablate MASK, profiles/ablate.py).
// A set bit removes one phase of the kernel -- results are wrong by construction; the product library is
// always built with 0.  1 replay passes, 2 play chain, 4 play tables, 8 Philox, 16 per-row argmax,
// 32 visit counters (log + histogram), 64 log sums, 128 replay schedule.
#ifndef THRL_ABLATE
#define THRL_ABLATE 0
#endif

namespace thrl {

constexpr int kAblate = THRL_ABLATE;

typedef unsigned int v2u __attribute__((ext_vector_type(2)));

// lanes<32 of the result: lanes 0-31 of a ; lanes>=32: lanes 0-31 of b   (.x)
// and the same for the upper halves (.y): one v_permlane32_swap.
__device__ __forceinline__ v2u pack_halves(unsigned a, unsigned b) {
    return __builtin_amdgcn_permlane32_swap(a, b, false, false);
}

// LDS access by 32-bit LDS address (address space 3): no generic-pointer arithmetic
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
template <typename QT>
__device__ __forceinline__ QT lds_load(unsigned addr) {
    typedef __attribute__((address_space(3))) QT lds_t;
    return *(const lds_t*)(uintptr_t)addr;
}
template <typename QT>
__device__ __forceinline__ void lds_store(unsigned addr, QT v) {
    typedef __attribute__((address_space(3))) QT lds_t;
    *(lds_t*)(uintptr_t)addr = v;
}

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
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

// max over each aligned group of 8 lanes, result in every lane of the group.
// float: three single-instruction DPP max steps (xor 1, xor 2 inside the quad, then the mirror of
// the 8-lane half row).  The s_nop 1 before each are the 2 wait states a DPP read of a just-written
// VGPR needs (hipcc does not look inside asm statements).
__device__ __forceinline__ float group8_allmax(float v) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ double group8_allmax(double v) {
    v = fmax(v, dpp_mov64<0xB1>(v));       // quad_perm [1,0,3,2]
    v = fmax(v, dpp_mov64<0x4E>(v));       // quad_perm [2,3,0,1]
    v = fmax(v, dpp_mov64<0x141>(v));      // row_half_mirror
    return v;
}
// LDS store by the lanes of `mask` only, as straight-line code: hipcc turns `if (cond) store` into
// saveexec + skip branch + store + branch back; the mask is never empty here, so EXEC is simply swapped around
// the store.
__device__ __forceinline__ void store_where(unsigned long long mask, unsigned addr, float v) {
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void store_where(unsigned long long mask, unsigned addr, double v) {
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b64 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(addr), "v"(v) : "memory");
}
// the value of the lane 8 further on in the same 16-lane row (row_ror:8): swaps the two 8-lane groups of a row
__device__ __forceinline__ float dpp_ror8(float v) { return __builtin_bit_cast(float, dpp_mov32<0x128>(__builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ double dpp_ror8(double v) { return dpp_mov64<0x128>(v); }
__device__ __forceinline__ float max_of(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double max_of(double a, double b);
// max over each 32-lane half, result in every lane of the half: the 8-lane step above, the mirror of the 16-lane
// row, then the two rows of the half exchanged by v_permlane16_swap
__device__ __forceinline__ float half32_allmax(float v) {
    v = group8_allmax(v);
    v = fmaxf(v, __builtin_bit_cast(float, dpp_mov32<0x140>(__builtin_bit_cast(uint32_t, v))));      // row_mirror
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm("" : "+v"(b));
    const v2u r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const unsigned rx = r.x, ry = r.y;
    return fmaxf(__builtin_bit_cast(float, rx), __builtin_bit_cast(float, ry));
}
__device__ __forceinline__ double half32_allmax(double v) {
    v = group8_allmax(v);
    v = fmax(v, dpp_mov64<0x140>(v));
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    unsigned lo2 = lo, hi2 = hi;
    asm("" : "+v"(lo2), "+v"(hi2));
    const v2u rl = __builtin_amdgcn_permlane16_swap(lo, lo2, false, false);
    const v2u rh = __builtin_amdgcn_permlane16_swap(hi, hi2, false, false);
    const unsigned rlx = rl.x, rly = rl.y, rhx = rh.x, rhy = rh.y;
    return fmax(__hiloint2double((int)rhx, (int)rlx), __hiloint2double((int)rhy, (int)rly));
}
__device__ __forceinline__ double max_of(double a, double b) { return fmax(a, b); }

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

// lo16(x) + hi16(x) in one VALU instruction (sub-dword operand selects)
__device__ __forceinline__ uint32_t halves_sum(uint32_t x) {
    uint32_t y;
    asm("v_add_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(y) : "v"(x));
    return y;
}

// Four steps of the play chain for one-register row tables: s <- nsr_j[s & 63] for j = 0..3, and the
// four states the steps were played in recorded TWO PER LANE (lane t0: s_t0 | s_t0+1 << 16, lane t0+2:
// s_t0+2 | s_t0+3 << 16; the caller unpacks lane-parallel) -- 6 vector instructions per group instead of 8.
// Hand-placed wait states (hipcc does not look inside asm): a v_readlane whose lane select was written by
// a VALU (the previous v_readlane) needs 4 wait states; the s_pack and the v_writelane (lane number as an
// immediate: the step numbers are compile-time constants) fill them where they can, s_nop pads the rest.
template <int T0>
__device__ __forceinline__ void chain_group(uint32_t& sq, int& s, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n3) {
    int s1, s2, s3, s4, pk;
    asm volatile(
        "s_nop 3\n\t"
        "v_readlane_b32 %1, %7, %6\n\t"
        "s_pack_ll_b32_b16 %5, %6, %1\n\t"
        "v_writelane_b32 %0, %5, %11\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 %2, %8, %1\n\t"
        "s_nop 3\n\t"
        "v_readlane_b32 %3, %9, %2\n\t"
        "s_pack_ll_b32_b16 %5, %2, %3\n\t"
        "v_writelane_b32 %0, %5, %12\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 %4, %10, %3"
        : "+v"(sq), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(pk)
        : "s"(s), "v"(n0), "v"(n1), "v"(n2), "v"(n3), "n"(T0), "n"(T0 + 2)
        : "scc");
    s = s4;
}

// The same for four steps in which nobody explores: with G = "next row of a greedy step" composed with itself
// (g1 = G, g2 = G.G, g3 = G.G.G, g4 = G^4, built once per episode) the four reads all use the group's entry
// state, so only the first waits for it.
template <int T0>
__device__ __forceinline__ void chain_group_greedy(uint32_t& sq, int& s, uint32_t g1, uint32_t g2, uint32_t g3, uint32_t g4) {
    int s1, s2, s3, s4, pk;
    asm volatile(
        "s_nop 3\n\t"
        "v_readlane_b32 %1, %7, %6\n\t"
        "v_readlane_b32 %2, %8, %6\n\t"
        "v_readlane_b32 %3, %9, %6\n\t"
        "v_readlane_b32 %4, %10, %6\n\t"
        "s_pack_ll_b32_b16 %5, %6, %1\n\t"
        "v_writelane_b32 %0, %5, %11\n\t"
        "s_pack_ll_b32_b16 %5, %2, %3\n\t"
        "v_writelane_b32 %0, %5, %12"
        : "+v"(sq), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(pk)
        : "s"(s), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "n"(T0), "n"(T0 + 2)
        : "scc");
    s = s4;
}

template <bool ON> struct GreedyTables {            // (nothing in the variants that do not use them)
    __device__ __forceinline__ void build(unsigned) {}
    template <int T0> __device__ __forceinline__ void chain(uint32_t&, int&) const {}
};
template <> struct GreedyTables<true> {
    uint32_t g1, g2, g3, g4;
    __device__ __forceinline__ void build(unsigned lut_addr) {
        g1 = (uint32_t)lds_load<unsigned short>(lut_addr);
        g2 = bperm((g1 & 63u) << 2, g1);
        g3 = bperm((g2 & 63u) << 2, g1);
        g4 = bperm((g2 & 63u) << 2, g2);
    }
    template <int T0> __device__ __forceinline__ void chain(uint32_t& sq, int& s) const { chain_group_greedy<T0>(sq, s, g1, g2, g3, g4); }
};

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

// ---- table-type dependent pieces of train_net's arithmetic (thrl_device.h td_value) ------------
// Operands of one 32-transition block in "step layout": lane 32h + t = agent h, transition t.
// float : c1 = fma(alpha, reward, (1-alpha)*old_value) -- everything of the target that does not
//         depend on the live next_max; the pass then executes ONE fma (thrl_device.h, float32 form).
// double: the reference's four separately rounded operations need reward and (1-alpha)*old_value.
template <typename QT> struct BlockOps;
template <> struct BlockOps<float> {
    unsigned c1;
    __device__ __forceinline__ void permute(unsigned sel) { c1 = bperm(sel, c1); }
    __device__ __forceinline__ void keep() const { asm volatile("" :: "v"(c1)); }
    __device__ __forceinline__ float value(float nm, float alpha_gamma, float, float) const {
        return __fmaf_rn(alpha_gamma, nm, __builtin_bit_cast(float, c1));
    }
};
template <> struct BlockOps<double> {
    unsigned r_lo, r_hi, t4_lo, t4_hi;
    __device__ __forceinline__ void keep() const { asm volatile("" :: "v"(r_lo), "v"(r_hi), "v"(t4_lo), "v"(t4_hi)); }
    __device__ __forceinline__ void permute(unsigned sel) {
        r_lo = bperm(sel, r_lo); r_hi = bperm(sel, r_hi); t4_lo = bperm(sel, t4_lo); t4_hi = bperm(sel, t4_hi);
    }
    __device__ __forceinline__ double value(double nm, double, double alpha, double gamma) const {
        const double re = __hiloint2double((int)r_hi, (int)r_lo), t4 = __hiloint2double((int)t4_hi, (int)t4_lo);
        const double t2 = __dadd_rn(re, __dmul_rn(gamma, nm));                 // agents.py:73
        return __dadd_rn(t4, __dmul_rn(alpha, t2));                           // agents.py:72,74
    }
};
// (1-alpha)*old_value of both agents for the 64 transitions of a segment, halves packed per 32
template <typename QT> struct Snapshot;
template <> struct Snapshot<float> {
    v2u t4;
    __device__ __forceinline__ void set(float oma0, float ov0, float oma1, float ov1) {
        t4 = pack_halves(__builtin_bit_cast(unsigned, __fmul_rn(oma0, ov0)), __builtin_bit_cast(unsigned, __fmul_rn(oma1, ov1)));
    }
};
template <> struct Snapshot<double> {
    v2u lo, hi;
    __device__ __forceinline__ void set(double oma0, double ov0, double oma1, double ov1) {
        const double x0 = __dmul_rn(oma0, ov0), x1 = __dmul_rn(oma1, ov1);
        lo = pack_halves((unsigned)__double2loint(x0), (unsigned)__double2loint(x1));
        hi = pack_halves((unsigned)__double2hiint(x0), (unsigned)__double2hiint(x1));
    }
};
__device__ __forceinline__ BlockOps<float> make_ops(const Snapshot<float>& s, int k, double r0d, double r1d, float alpha_h) {
    const v2u req = pack_halves(__builtin_bit_cast(unsigned, (float)r0d), __builtin_bit_cast(unsigned, (float)r1d));
    BlockOps<float> o;
    o.c1 = __builtin_bit_cast(unsigned, __fmaf_rn(alpha_h, __builtin_bit_cast(float, k ? req.y : req.x),
                                                  __builtin_bit_cast(float, k ? s.t4.y : s.t4.x)));
    return o;
}
__device__ __forceinline__ BlockOps<double> make_ops(const Snapshot<double>& s, int k, double r0d, double r1d, double) {
    const v2u rl = pack_halves((unsigned)__double2loint(r0d), (unsigned)__double2loint(r1d));
    const v2u rh = pack_halves((unsigned)__double2hiint(r0d), (unsigned)__double2hiint(r1d));
    BlockOps<double> o;
    o.r_lo = k ? rl.y : rl.x; o.r_hi = k ? rh.y : rh.x;
    o.t4_lo = k ? s.lo.y : s.lo.x; o.t4_hi = k ? s.hi.y : s.hi.x;
    return o;
}
// per-agent hyper-parameters in the table's arithmetic type
template <typename QT> struct HP;
template <> struct HP<float> {
    static __device__ __forceinline__ float gamma(const AgentParams& p) { return p.gamma_f; }
    static __device__ __forceinline__ float alpha(const AgentParams& p) { return p.alpha_f; }
    static __device__ __forceinline__ float one_minus_alpha(const AgentParams& p) { return p.one_minus_alpha_f; }
    static __device__ __forceinline__ float from_double(double x) { return (float)x; }
    static __device__ __forceinline__ float product(float a, float g) { return __fmul_rn(a, g); }
};
template <> struct HP<double> {
    static __device__ __forceinline__ double gamma(const AgentParams& p) { return p.gamma; }
    static __device__ __forceinline__ double alpha(const AgentParams& p) { return p.alpha; }
    static __device__ __forceinline__ double one_minus_alpha(const AgentParams& p) { return p.one_minus_alpha; }
    static __device__ __forceinline__ double from_double(double x) { return x; }
    static __device__ __forceinline__ double product(double, double) { return 0.0; }    // unused in float64 mode
};

// One 32-transition block of train_net's loop (agents.py:68-76), four transitions per pass.
// RDN = row reads per lane: 2 (A = 2), 3 (A <= 24) or 4; a lane reads RDN CONSECUTIVE columns of its
// transition's next-state row (one address: the loads differ by an immediate offset).
//   P        (step layout, lane 4g of the segment): the four next-state rows of group g, 7 bits each,
//            | cut bits << 28 (bit j-1: a new pass starts before transition j of the group)
//            | bit 31: the four transitions are IDENTICAL and stay in their row (a converged game sits in a
//              fixed point: same state, same greedy actions, same reward, step after step).  Serially that
//              is four dependent (row max, TD value) pairs on one cell; here the row is read once, split
//              into "the cell" and "max of the rest", and the four pairs run in registers: one store.
//   ops, wo  operands / LDS store address of transition 4*(lane&7) + ((lane>>3)&3) ("exec layout")
template <typename QT, int RDN, bool FIXED_POINTS, bool PER2>
__device__ __forceinline__ void replay_group(int gi, uint32_t P, int lane_base, int nsub, const BlockOps<QT>& ops, unsigned wo,
                                             unsigned rd_base, unsigned row_shift, unsigned my_step,
                                             unsigned row_bytes, QT alpha_gamma, QT alpha, QT gamma, bool upper_half,
                                             int block_step0, int replay_from) {
    // transitions that had already dropped out of the deque when the buffer trained are skipped
    const int first = replay_from - (block_step0 + gi * 4);      // <= 0: the whole group trains
    if (first >= 4) return;
    const uint32_t sP = readlane_u(P, lane_base + gi * 4);
    const int nv = min(4, nsub - gi * 4);
    unsigned ad0;                                           // row * row_bytes + this lane's first column: one v_mad_u32_u24
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ad0) : "v"((sP >> row_shift) & 0x7Fu), "s"(row_bytes), "v"(rd_base));
    unsigned ad[RDN];                                       // this lane's columns of its transition's next-state row
#pragma unroll
    for (int i = 0; i < RDN; i++) ad[i] = ad0 + (unsigned)(i * sizeof(QT));
    if (FIXED_POINTS && PER2 && (sP >> 31) && nv == 4 && first <= 0 && !(kAblate & 256) && (sP & 0x7Fu) != ((sP >> 7) & 0x7Fu)) {
        // PERIOD-2 CYCLE (A, B, A, B): A is played in row sA and leads to row sB, B leads back.  Serially that is
        // four dependent (row max, TD value) pairs: each transition reads the row its predecessor just wrote.
        // Here groups 0/2 (the A's) read row sB once and split it into "the cell B writes" and the rest, groups
        // 1/3 do the same for row sA and A's cell, and the four values run in registers, each handed to the
        // neighbouring 8-lane group by one DPP rotate: y0 = A(max row sB), y1 = B(max(rest sA, y0)),
        // y2 = A(max(rest sB, y1)), y3 = B(max(rest sA, y2)); the cells end as y2 / y3.
        // exec layout: transition 4*gi + i sits in lanes 8*i + gi / 32 + 8*i + gi
        const unsigned ca0 = readlane_u(wo, gi), ca1 = readlane_u(wo, 32 + gi);
        const unsigned cb0 = readlane_u(wo, 8 + gi), cb1 = readlane_u(wo, 40 + gi);
        const bool odd = (my_step & 1u) != 0u;                  // my_step & 3 = this lane's 8-lane group
        const unsigned cell = odd ? (upper_half ? ca1 : ca0) : (upper_half ? cb1 : cb0);
        QT rest = -(QT)INFINITY, m = -(QT)INFINITY;
#pragma unroll
        for (int i = 0; i < RDN; i++) {
            const QT e = lds_load<QT>(ad[i]);
            m = max_of(m, e);
            rest = max_of(rest, ad[i] == cell ? -(QT)INFINITY : e);
        }
        rest = group8_allmax(rest);
        m = group8_allmax(m);
        const QT y0 = ops.value(m, alpha_gamma, alpha, gamma);
        const QT y1 = ops.value(max_of(rest, dpp_ror8(y0)), alpha_gamma, alpha, gamma);
        const QT y2 = ops.value(max_of(rest, dpp_ror8(y1)), alpha_gamma, alpha, gamma);
        const QT y3 = ops.value(max_of(rest, dpp_ror8(y2)), alpha_gamma, alpha, gamma);
        store_where(__ballot((unsigned)(my_step - (unsigned)(gi * 4 + 2)) < 2u), wo, odd ? y3 : y2);
        __builtin_amdgcn_wave_barrier();
        return;
    }
    if (FIXED_POINTS && (sP >> 31) && nv == 4 && first <= 0 && !(kAblate & 256)) {
        // the rewritten cell, per agent: lanes gi / 32+gi hold transition 4*gi's store address
        const unsigned c0 = readlane_u(wo, gi), c1 = readlane_u(wo, 32 + gi);
        const unsigned cell = upper_half ? c1 : c0;
        QT rest = -(QT)INFINITY, cur = -(QT)INFINITY;
#pragma unroll
        for (int i = 0; i < RDN; i++) {
            const QT e = lds_load<QT>(ad[i]);
            rest = max_of(rest, ad[i] == cell ? -(QT)INFINITY : e);
            cur = max_of(cur, ad[i] == cell ? e : -(QT)INFINITY);
        }
        rest = group8_allmax(rest);
        cur = group8_allmax(cur);
#pragma unroll
        for (int k = 0; k < 4; k++) cur = ops.value(max_of(rest, cur), alpha_gamma, alpha, gamma);
        if (my_step == (unsigned)(gi * 4 + 3)) lds_store<QT>(wo, cur);
        __builtin_amdgcn_wave_barrier();
        return;
    }
    const uint32_t cuts = (((sP >> 28) & 7u) << 1) | (1u << nv);       // bit j: a pass ends before transition j
    int lo = max(first, 0);
    if (lo >= nv) return;
    const unsigned d_in_group = my_step - (unsigned)(gi * 4);            // 0..3 in the lanes that hold this group's transitions
    do {
        const int hi = lo + 1 + __builtin_ctz(cuts >> (lo + 1));
        QT m = lds_load<QT>(ad[0]);
#pragma unroll
        for (int i = 1; i < RDN; i++) m = max_of(m, lds_load<QT>(ad[i]));
        m = group8_allmax(m);
        const QT val = ops.value(m, alpha_gamma, alpha, gamma);
        store_where(__ballot((unsigned)(d_in_group - (unsigned)lo) < (unsigned)(hi - lo)), wo, val);
        __builtin_amdgcn_wave_barrier();
        lo = hi;
    } while (lo < nv);
}
// UNROLL: the block's 8 groups as straight-line code (group numbers become immediates, no loop control):
// every variant with <= 128 steps per cycle except float64 + noise (hipcc 7.2 fails on that one: "illegal VGPR to SGPR copy").
template <typename QT, int RDN, bool FIXED_POINTS, bool UNROLL, bool PER2>
__device__ __forceinline__ void replay_block(uint32_t P, int lane_base, int nsub, const BlockOps<QT>& ops, unsigned wo,
                                             unsigned rd_base, unsigned row_shift, unsigned my_step,
                                             unsigned row_bytes, QT alpha_gamma, QT alpha, QT gamma, bool upper_half,
                                             int block_step0, int replay_from) {
    if (kAblate & 1) { ops.keep(); asm volatile("" :: "v"(wo), "v"(P)); return; }
    if (UNROLL) {
        if (nsub == 32) {
            // a full block (three of the four blocks of a 100-step episode): every "is this group / transition
            // inside the block" test folds away
#pragma unroll
            for (int gi = 0; gi < 8; gi++)
                replay_group<QT, RDN, FIXED_POINTS, PER2>(gi, P, lane_base, 32, ops, wo, rd_base, row_shift, my_step, row_bytes,
                                                    alpha_gamma, alpha, gamma, upper_half, block_step0, replay_from);
            return;
        }
#pragma unroll
        for (int gi = 0; gi < 8; gi++) {
            if (gi * 4 >= nsub) break;
            replay_group<QT, RDN, FIXED_POINTS, PER2>(gi, P, lane_base, nsub, ops, wo, rd_base, row_shift, my_step, row_bytes,
                                                alpha_gamma, alpha, gamma, upper_half, block_step0, replay_from);
        }
    } else {
        for (int gi = 0; gi * 4 < nsub; gi++)
            replay_group<QT, RDN, FIXED_POINTS, PER2>(gi, P, lane_base, nsub, ops, wo, rd_base, row_shift, my_step, row_bytes,
                                                alpha_gamma, alpha, gamma, upper_half, block_step0, replay_from);
    }
}

// A segment whose transitions repeat with period P (P different rows; transition j of the cycle is played in row
// s_j, leads to row s_(j+1) and rewrites cell c_j of row s_j): train_net's serial loop (agents.py:68-76) as a
// recurrence in registers.  cur[j] = live value of c_j, rest[j] = max of row s_j without c_j (the other cells of
// the row do not change during the segment), so transition t = j (mod P) is
//     cur[j] <- TD value(max(rest[j+1], cur[j+1]))        -- the same arithmetic, in the same order, as the passes.
// ops0 / woq_x are in step layout (lane 32h + t = agent h, transition t < 32); W[j] = transition word of step j.
template <typename QT, int P>
__device__ __forceinline__ void cyclic_segment(const BlockOps<QT>& ops0, unsigned woq_x, const uint32_t (&W)[4], int nseg, int lane,
                                               unsigned tabh, int A, unsigned row_bytes, QT alpha_gamma, QT alpha, QT gamma) {
    BlockOps<QT> ops[P];
    unsigned cell[P];
    QT rest[P], cur[P];
    const unsigned colb = (unsigned)min(lane & 31, A - 1) * (unsigned)sizeof(QT);
#pragma unroll
    for (int j = 0; j < P; j++) {
        const unsigned sel = (unsigned)((lane & 32) + j) << 2;
        ops[j] = ops0;
        ops[j].permute(sel);
        cell[j] = bperm(sel, woq_x);
        const unsigned ad = tabh + ((W[j] >> 16) & 0xFFu) * row_bytes + colb;
        const QT v = lds_load<QT>(ad);
        cur[j] = lds_load<QT>(cell[j]);
        rest[j] = half32_allmax(ad == cell[j] ? -(QT)INFINITY : v);
    }
    int t = 0;
    for (; t + P <= nseg; t += P) {
#pragma unroll
        for (int j = 0; j < P; j++) cur[j] = ops[j].value(max_of(rest[(j + 1) % P], cur[(j + 1) % P]), alpha_gamma, alpha, gamma);
    }
#pragma unroll
    for (int j = 0; j < P; j++)
        if (t + j < nseg) cur[j] = ops[j].value(max_of(rest[(j + 1) % P], cur[(j + 1) % P]), alpha_gamma, alpha, gamma);
    if ((lane & 31) == 0) {
#pragma unroll
        for (int j = 0; j < P; j++) lds_store<QT>(cell[j], cur[j]);
    }
}

// LDS capacity allows 20 resident waves per CU for the headline window in float32, i.e. 5 per
// SIMD: keep the register allocation at <= 96 VGPRs there (NSEG <= 2).  float64 tables are twice
// the size (11 games per CU), so the register budget is relaxed there.
// NOISE: environment noise (environments.py:28-31) handled per step (price not on the LUT);
// its larger row window leaves room for fewer waves, so the register budget is relaxed.
// SWEEP: per-game hyper-parameter arrays (thrl_buffers.sweep_*); compiled only together with NOISE
// so the headline variant carries none of that state.
template <typename QT, int NSEG, int NRSEG, bool NOISE, bool SWEEP, bool CYCLE, bool GREEDY = false>
__global__ void __launch_bounds__(1024)
__attribute__((amdgpu_waves_per_eu(sizeof(QT) == 8 ? 3 : (NOISE ? 4 : (NSEG <= 2 ? 5 : 4)))))
k_wave_episodes(const WaveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool kUnrollReplay = NSEG <= 2 && !(NOISE && sizeof(QT) == 8);     // (hipcc 7.2 cannot compile the unrolled float64 noise variant)
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // A training cycle = epk episodes played against frozen tables (the replay buffer reaches min_memory
    // every epk-th episode), then ONE train_net over the transitions still in the deque.  T = transitions
    // per cycle; for the reference's example configs epk = 1 and a cycle is an episode.
    // CYCLE = false: the variant for epk == 1 with nothing dropped from the deque (min_memory <= T <= capacity),
    // compiled without any of the cycle bookkeeping.
    const int A = a.A, W = a.win_rows, Tenv = a.T, epk = CYCLE ? a.epk : 1, T = a.T * epk, lo = a.row_lo;
    const int replay_from = CYCLE ? a.replay_from : 0;
    const WaveLut L = wave_lut_layout(A);

    {   // stage the payoff LUT once per block
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.lut_ns);
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem);
        for (int k = threadIdx.x; k < (a.lut_bytes >> 2); k += blockDim.x) dst[k] = src[k];
    }
    __syncthreads();
    const unsigned lut_ns_lds = lds_addr(smem + L.ns_off);                            // u16 [A*A]: play row | train row << 8
    const double* lut_aq = reinterpret_cast<const double*>(smem + L.aq_off);
    const double* lut_sct = reinterpret_cast<const double*>(smem + L.sct_off);

    QT* tab0 = reinterpret_cast<QT*>(smem + a.lut_bytes + (size_t)wib * a.game_lds_bytes);
    QT* tab1 = tab0 + (W + 2) * A;
    const int half = lane >> 5;
    const unsigned tab0_off = lds_addr(tab0);        // absolute LDS byte addresses
    const unsigned tab1_off = lds_addr(tab1);
    // replay ("exec") layout of a 32-transition block: lane 32h + 8k + j  <->  agent h, transition 4j + k
    const int kk = (lane >> 3) & 3, jj = lane & 7;
    const unsigned perm_sel = (unsigned)((lane & 32) + 4 * jj + kk) << 2;      // bpermute source of this lane
    const unsigned my_step = (unsigned)(4 * jj + kk);
    const unsigned row_shift = (unsigned)(7 * kk);
    const unsigned row_bytes = (unsigned)A * (unsigned)sizeof(QT);
    // replay row reads: lane jj of an 8-lane group reads columns rdn*jj .. rdn*jj + rdn-1 (the last lanes
    // re-read the row's last rdn columns), rdn = 2 / 3 / 4 for A = 2 / <= 24 / <= 32
    const int rdn = A < 3 ? 2 : (A <= 24 ? 3 : 4);
    const unsigned rd_base = (half ? tab1_off : tab0_off) + (unsigned)sizeof(QT) * (unsigned)min(rdn * jj, A - rdn);
    // 256 bytes behind the wave's tables (float32, noise-free configurations: thrl_api.hip plan_wave): per-step words of a segment
    constexpr bool kLdsMK = sizeof(QT) == 4 && !NOISE && !GREEDY && NRSEG == 1 && !(kAblate & 6);
    const unsigned mk_addr = tab0_off + 2u * (unsigned)((W + 2) * A) * (unsigned)sizeof(QT);

    const AgentParams& p0 = a.ag[0];
    const AgentParams& p1 = a.ag[1];
    const double inv_T_den = (double)Tenv;

    // lane ((e&15)*4+k): sum over this wave's games of episode-e log value k (e < 16 / e >= 16), as
    // fixed-point integers: the games a wave gets are not deterministic, integer sums do not care
    long long acc = 0, acc_hi = 0;
    const double log_scale = (lane & 2) ? a.log_scale[1] : a.log_scale[0];
    const int wave_gid = blockIdx.x * a.waves_per_block + wib;

    // Games are handed out dynamically (one atomic per game): waves on less crowded CUs simply take
    // more games, which measured 7-13 % faster than the static grid-stride assignment.
    // The NEXT game's id is claimed when a game starts, so the atomic's round trip is off the path.
    int g_claim = 0;
    if (lane == 0) g_claim = atomicAdd(a.next_game, 1);
    for (;;) {
        const int g = __builtin_amdgcn_readfirstlane(g_claim);
        if (g >= a.G) break;
        if (lane == 0) g_claim = atomicAdd(a.next_game, 1);
        const uint64_t gid = a.game_offset + (uint64_t)g;
        QT* __restrict__ q0 = reinterpret_cast<QT*>(a.q) + (int64_t)g * a.stride + p0.table_off;
        QT* __restrict__ q1 = reinterpret_cast<QT*>(a.q) + (int64_t)g * a.stride + p1.table_off;
        const double price0 = a.state[g];              // issued ahead of the table stream, used after it

        // ---- per-game hyper-parameters (sweeps) or the config's scalars
        QT gamma_h = half ? HP<QT>::gamma(p1) : HP<QT>::gamma(p0);
        QT alpha_h = half ? HP<QT>::alpha(p1) : HP<QT>::alpha(p0);
        QT oma0 = HP<QT>::one_minus_alpha(p0), oma1 = HP<QT>::one_minus_alpha(p1);
        if (SWEEP && a.sw_gamma) gamma_h = HP<QT>::from_double(a.sw_gamma[(size_t)half * a.G + g]);
        if (SWEEP && a.sw_alpha) {
            alpha_h = HP<QT>::from_double(a.sw_alpha[(size_t)half * a.G + g]);
            oma0 = HP<QT>::from_double(__dsub_rn(1.0, a.sw_alpha[g]));
            oma1 = HP<QT>::from_double(__dsub_rn(1.0, a.sw_alpha[(size_t)a.G + g]));
        }
        double epsg0 = 0.0, epsg1 = 0.0;               // per-game epsilon (sweep mode)
        const bool sw_eps_on = SWEEP && a.sw_eps != nullptr;
        if (sw_eps_on) { epsg0 = a.sw_eps[g]; epsg1 = a.sw_eps[(size_t)a.G + g]; }
        const QT ag_h = HP<QT>::product(alpha_h, gamma_h);   // float32: the only coefficient on the replay chain
        const double eend0 = (SWEEP && a.sw_eps_end) ? a.sw_eps_end[g] : p0.eps_end;
        const double eend1 = (SWEEP && a.sw_eps_end) ? a.sw_eps_end[(size_t)a.G + g] : p1.eps_end;
        const double estep0 = (SWEEP && a.sw_eps_step) ? a.sw_eps_step[g] : p0.eps_step;
        const double estep1 = (SWEEP && a.sw_eps_step) ? a.sw_eps_step[(size_t)a.G + g] : p1.eps_step;
        const double noise_prob_g = (SWEEP && a.sw_noise_prob) ? a.sw_noise_prob[g] : a.env.noise_prob;

        // ---- stream the table windows HBM -> LDS (contiguous, coalesced)
        {
            const QT* s0 = q0 + lo * A;
            const QT* s1 = q1 + lo * A;
            const int n = W * A;
            // 8 loads per agent in flight before the first LDS write (one HBM latency per
            // batch instead of one per 64 elements)
            for (int k0 = 0; k0 < n; k0 += 512) {
                QT v0[8], v1[8];
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
        }
        // ---- initial state -> local rows (window or spill)
        int sp = __builtin_amdgcn_readfirstlane(encode32(price0, p0));
        int st = __builtin_amdgcn_readfirstlane(encode64(price0, p0));
        sp = min(max(sp, 0), a.rows - 1);
        st = min(max(st, 0), a.rows - 1);
        int spill0 = -1, spill1 = -1, sp_l, st_l;
        if (sp >= lo && sp < lo + W) sp_l = sp - lo; else { spill0 = sp; sp_l = W; }
        if (st == sp) st_l = sp_l;
        else if (st >= lo && st < lo + W) st_l = st - lo;
        else { spill1 = st; st_l = W + 1; }

        {
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
        for (int e = 0; e < a.n_episodes; e += epk) {
            const uint32_t eg = (uint32_t)(a.first_episode + (uint64_t)e);

            // ---- (a) greedy action of every local row, lane = row (the table is frozen
            //          during play: agents.py only writes it in train_net).  AM = a0 | a1 << 8; R packs the
            //          two greedy byte offsets into the u16 payoff LUT: agent 0's (a0*A*2) in the low
            //          half, agent 1's (a1*2) in the high half.
            uint32_t AM[NRSEG], R[NRSEG];
#pragma unroll
            for (int k = 0; k < NRSEG; k++) {
                const int row = min(lane + 64 * k, W + 1);
                const QT* r0 = tab0 + row * A;
                const QT* r1 = tab1 + row * A;
                QT b0 = r0[0], b1 = r1[0];
                uint32_t i0 = 0, i1 = 0;
                if (kAblate & 16) {
                } else if (A == 21) {
                    // the reference's example configs: constant trip count, so the column numbers are inline
                    // constants of the selects (no index register, no loop control): 7 instructions per column pair
#pragma unroll
                    for (int j = 1; j < 21; j++) {
                        const QT v0 = r0[j], v1 = r1[j];
                        if (v0 > b0) { b0 = v0; i0 = j; }
                        if (v1 > b1) { b1 = v1; i1 = j; }
                    }
                } else {
#pragma unroll 4
                    for (int j = 1; j < A; j++) {
                        const QT v0 = r0[j], v1 = r1[j];
                        if (v0 > b0) { b0 = v0; i0 = j; }
                        if (v1 > b1) { b1 = v1; i1 = j; }
                    }
                }
                if (kAblate & 16) { i0 = (uint32_t)(lane * 5) % (uint32_t)A; i1 = (uint32_t)(lane * 3) % (uint32_t)A; }
                AM[k] = i0 | (i1 << 8);
                R[k] = (i0 * (uint32_t)A * 2u) | (i1 << 17);
            }
            // "next row if a step in which NOBODY explores is played in row r": one table for the whole episode
            // (the tables are frozen during play).  A group of four such steps needs no per-step table at all --
            // late in training that is nearly every group.
            GreedyTables<GREEDY && NRSEG == 1> gt;          // + G composed with itself: rows after 2, 3, 4 greedy steps
            gt.build(halves_sum(R[0]) + lut_ns_lds);

            // ---- (b,c) play: lane-parallel Philox, then the serial state chain.
            //      seq[seg] lane t = row in which step t was played.
            uint32_t seq[NSEG], rwv[NSEG];
            unsigned explored_segs = 0u;   // GREEDY: bit seg = somebody explored in that segment (it cannot be one cycle then)
            double nav[NSEG];              // NOISE: the uniform(0.7a, a) draw of a noisy step (lane = step)
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                int n = min(64, T - seg * 64);
                // opaque to the optimiser: the unrolled play loop's "is this group inside the segment" tests stay
                // scalar compares here instead of being hoisted out of the episode loop as spilled booleans
                asm volatile("" : "+s"(n));
                uint32_t rw;
                nav[seg] = 0.0;
                // this lane's step: episode ep of the cycle, step st of that episode; its epsilon
                const int tcl = min(seg * 64 + lane, T - 1);
                const int ep = epk == 1 ? 0 : tcl / Tenv;
                const uint32_t st_in_ep = (uint32_t)(tcl - ep * Tenv);
                double eps0, eps1;
                if (sw_eps_on) { eps0 = epsg0; eps1 = epsg1; }                       // (sweeps: epk == 1)
                else if (epk == 1) { eps0 = a.eps[e][0]; eps1 = a.eps[e][1]; }
                else { eps0 = a.eps[e + ep][0]; eps1 = a.eps[e + ep][1]; }            // per-lane read of the launch's table
                if (a.inj_u) {
                    // parity mode: the reference's recorded draws, [E][T][2][G] (agents.py:81-82)
                    const int tt = tcl;
                    const size_t k = (((size_t)e * Tenv + tt) * 2) * (size_t)a.G + (size_t)g;
                    const uint32_t ex0 = a.inj_u[k] < eps0 ? 1u : 0u;
                    const uint32_t ex1 = a.inj_u[k + a.G] < eps1 ? 2u : 0u;
                    const uint32_t c0 = min((uint32_t)(uint8_t)a.inj_choice[k], (uint32_t)(A - 1));
                    const uint32_t c1 = min((uint32_t)(uint8_t)a.inj_choice[k + a.G], (uint32_t)(A - 1));
                    rw = ex0 | ex1 | (c0 << 8) | (c1 << 16);
                } else if (kAblate & 8) {
                    const uint32_t h = (uint32_t)(lane * 2654435761u) ^ eg;
                    rw = (h & 3u) | (((h >> 4) % (uint32_t)A) << 8) | (((h >> 12) % (uint32_t)A) << 16);
                } else {
                    // the ten round keys are seed + r * constant: recomputed here by scalar adds (opaque seed) --
                    // hoisted out of the episode loop they were twenty spilled SGPRs, a v_readlane each per use
                    uint32_t seed_lo = (uint32_t)a.seed, seed_hi = (uint32_t)(a.seed >> 32);
                    asm volatile("" : "+s"(seed_lo), "+s"(seed_hi));
                    const uint64_t seed_here = ((uint64_t)seed_hi << 32) | seed_lo;
                    const u32x4 x = draw(seed_here, gid, eg + (uint32_t)ep, st_in_ep, 0u);
                    const uint32_t ex0 = u01_32(x.x) < eps0 ? 1u : 0u;
                    const uint32_t ex1 = u01_32(x.z) < eps1 ? 2u : 0u;
                    rw = ex0 | ex1 | (__umulhi(x.y, (uint32_t)A) << 8) | (__umulhi(x.w, (uint32_t)A) << 16);
                }
                if (NOISE) {               // environments.py:28-29, bit 2 of rw = noisy step
                    double nu, na;
                    if (a.inj_u) {
                        const size_t k = ((size_t)e * Tenv + tcl) * (size_t)a.G + (size_t)g;
                        nu = a.inj_noise_u[k]; na = a.inj_noise_a[k];
                    } else {
                        const u32x4 xn = draw(a.seed, gid, eg + (uint32_t)ep, st_in_ep, kStreamNoise);
                        nu = u01_32(xn.x);
                        na = __dadd_rn(a.env.noise_lo, __dmul_rn(__dsub_rn(a.env.a, a.env.noise_lo), u01_32(xn.y)));
                    }
                    if (nu < noise_prob_g) rw |= 4u;
                    nav[seg] = na;
                }
                rwv[seg] = rw;
                // Per step (lane = step), what turns the rows' greedy offsets R into the LUT address of
                // "the next row if this step were played in row r":
                //   Mv keeps agent i's half of R iff agent i does NOT explore in this step,
                //   Kv = LUT base + 2 * (explore0 ? c0*A : 0) + 2 * (explore1 ? c1 : 0)
                // so address(r) = lo16(R[r] & Mv) + hi16(R[r] & Mv) + Kv: no scalar control per step.
                // (Deciding the category -- who explores -- on the scalar unit instead needs only 2 vector
                // instructions per step, but its branches cost more than the 3 it saves: measured 13 % slower.)
                const uint32_t Mv = ((rw & 1u) ? 0u : 0xFFFFu) | ((rw & 2u) ? 0u : 0xFFFF0000u);
                const uint32_t Kv = lut_ns_lds + 2u * (((rw & 1u) ? ((rw >> 8) & 0xFFu) * (uint32_t)A : 0u) +
                                                      ((rw & 2u) ? ((rw >> 16) & 0xFFu) : 0u));
                const unsigned long long noisy_steps = NOISE ? __ballot((rw & 4u) != 0u) : 0ull;   // scalar: no per-step readlane
                // bit 4g: group g (steps 4g..4g+3) has a step that needs its own table (somebody explores, the step is
                // noisy, or it lies beyond the segment's end)
                unsigned long long busy_groups = 0ull;
                if (GREEDY && __ballot((rw & 3u) != 0u && seg * 64 + lane < T) != 0ull) explored_segs |= 1u << seg;
                if (GREEDY && NRSEG == 1) {
                    busy_groups = __ballot((rw & (NOISE ? 7u : 3u)) != 0u || seg * 64 + lane >= T);
                    busy_groups |= busy_groups >> 1;
                    busy_groups |= busy_groups >> 2;
                }
                // GREEDY variants (launched once epsilon is small, thrl_wave.hip): the play loop looks at those bits,
                // two scalar instructions per group, and skips the table build of the all-greedy groups
                constexpr bool greedy_copy = GREEDY && NRSEG == 1 && !(kAblate & 6);
                // Phase 1 (off the serial chain, lane = ROW): nsr_t[r] for one step; its LDS gather does
                // not depend on the current state.
                auto build = [&](int t, uint32_t (&out)[NRSEG], bool may_be_noisy) {
                    const int tl = t;            // < 64: only groups that start below n <= 64 are built
                    if (NOISE && may_be_noisy) {
                        if ((noisy_steps >> tl) & 1ull) {
                            // noisy step: the price is not on the LUT; evaluate it for every row
                            const uint32_t w = readlane_u(rw, tl);
                            const double na = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(nav[seg]), tl),
                                                               __builtin_amdgcn_readlane(__double2loint(nav[seg]), tl));
                            const uint32_t c0 = (w >> 8) & 0xFFu, c1 = (w >> 16) & 0xFFu;
#pragma unroll
                            for (int k = 0; k < NRSEG; k++) {
                                const uint32_t a0r = (w & 1u) ? c0 : (AM[k] & 0xFFu), a1r = (w & 2u) ? c1 : (AM[k] >> 8);
                                const double Q = __dadd_rn(lut_aq[a0r], lut_aq[A + a1r]);
                                double pr = __dsub_rn(na, __dmul_rn(a.env.b, Q));
                                if (!(pr > 0.0)) pr = 0.0;
                                const int r32 = min(max(encode32_fast(pr, p0) - lo, 0), W - 1);
                                const int r64 = min(max(encode64_fast(pr, p0) - lo, 0), W - 1);
                                out[k] = (uint32_t)(r32 | (r64 << 8));
                            }
                            return;
                        }
                    }
                    if (kAblate & 4) {
#pragma unroll
                        for (int k = 0; k < NRSEG; k++) out[k] = (R[k] >> 17) + (uint32_t)(t & 3);
                        return;
                    }
                    const uint32_t sM = readlane_u(Mv, tl), sK = readlane_u(Kv, tl);
#pragma unroll
                    for (int k = 0; k < NRSEG; k++) out[k] = lds_load<unsigned short>(halves_sum(R[k] & sM) + sK);
                };
                // Phase 2 (the chain): s <- nsr_t[s], one v_readlane per step; the state each step was
                // played in is recorded in lane t of sq.
                uint32_t sq = 0, sq_tail = 0;          // sq: two states per lane (full groups); sq_tail: one per lane
                auto chain4 = [&](auto t0c, const int n, const uint32_t (&tab)[4][NRSEG], const bool all_greedy = false) {   // n: steps in the segment
                    constexpr int t0 = decltype(t0c)::value;
                    if (kAblate & 2) {
                        asm volatile("" :: "v"(tab[0][0]), "v"(tab[1][0]), "v"(tab[2][0]), "v"(tab[3][0]));
                        sq_tail = (uint32_t)min(lane, W - 1) * 0x101u;
                        return;
                    }
                    if (NRSEG == 1 && t0 + 4 <= n) {            // full group, one-register tables: hand-scheduled steps
                        s = __builtin_amdgcn_readfirstlane(s);      // "s" operands must be provably uniform
                        if (GREEDY && all_greedy) gt.template chain<t0>(sq, s);
                        else chain_group<t0>(sq, s, tab[0][0], tab[1][0], tab[2][0], tab[3][0]);
                        return;
                    }
                    s = __builtin_amdgcn_readfirstlane(s);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (t0 + j < n) {
                            sq_tail = writelane_u(sq_tail, (uint32_t)s, t0 + j);
                            s = (int)read_row<NRSEG>(tab[j], s & 0xFF);
                        }
                    }
                };
                // the tables of group g+1 are built (their gathers in flight) while group g's chain runs
                uint32_t ta[4][NRSEG], tb[4][NRSEG];
                // float32 plain variants: the two per-step words reach the lanes as VGPRs through a uniform-address LDS read
                // issued ONE GROUP AHEAD, so the table build runs in the fast issue class (all-VGPR v_and / v_add: 2.3 cycles per
                // instruction per SIMD against 4.1 with an SGPR operand, and no v_readlane: profiles/r03_ubench_issue.md)
                v2u mkr[4] = {v2u{0u, 0u}, v2u{0u, 0u}, v2u{0u, 0u}, v2u{0u, 0u}};
                // (256 bytes per wave hold 32 steps: lanes 0-31 write theirs before the first group, lanes 32-63 overwrite them
                //  when the build reaches step 28 -- every read of steps 0-31 has been issued by then and the LDS runs a wave's
                //  operations in order)
                if (kLdsMK) {
                    if (lane < 32) lds_store<v2u>(mk_addr + 8u * (unsigned)lane, v2u{Mv, Kv});
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int j = 0; j < 4; j++) mkr[j] = lds_load<v2u>(mk_addr + 8u * (unsigned)j);
                }
                auto build4 = [&](int t0, uint32_t (&tab)[4][NRSEG]) {
                    if (kLdsMK) {
                        unsigned ad[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) ad[j] = halves_sum(R[0] & mkr[j].x) + mkr[j].y;
#pragma unroll
                        for (int j = 0; j < 4; j++) tab[j][0] = lds_load<unsigned short>(ad[j]);
                        if (t0 == 28) {
                            if (lane >= 32) lds_store<v2u>(mk_addr + 8u * (unsigned)(lane - 32), v2u{Mv, Kv});
                            __builtin_amdgcn_wave_barrier();
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) mkr[j] = lds_load<v2u>(mk_addr + 8u * (unsigned)(min(t0 + 4 + j, 63) & 31));
                        return;
                    }
                    // one scalar test per group of four steps: a group without a noisy step (81 % of them at
                    // noise_prob 0.05) takes the straight-line path
                    if (NOISE && ((noisy_steps >> t0) & 0xFull)) {
#pragma unroll
                        for (int j = 0; j < 4; j++) build(t0 + j, tab[j], true);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++) build(t0 + j, tab[j], false);
                    }
                };
                // Fully unrolled over the segment's 16 groups of four steps (step numbers become immediates: no
                // index arithmetic or loop control per step; a group past the segment's end costs one scalar test).
#define THRL_PLAY2(G, N_)                                                                    \
                if ((G) * 4 < (N_)) {                                                        \
                    if ((G) * 4 + 4 < (N_)) build4((G) * 4 + 4, tb);                         \
                    chain4(std::integral_constant<int, (G) * 4>(), (N_), ta);                \
                    if ((G) * 4 + 4 < (N_)) {                                                \
                        if ((G) * 4 + 8 < (N_)) build4((G) * 4 + 8, ta);                     \
                        chain4(std::integral_constant<int, (G) * 4 + 4>(), (N_), tb);        \
                    }                                                                        \
                }
#define THRL_PLAY16(N_)                                                                      \
                THRL_PLAY2(0, N_) THRL_PLAY2(2, N_) THRL_PLAY2(4, N_) THRL_PLAY2(6, N_)      \
                THRL_PLAY2(8, N_) THRL_PLAY2(10, N_) THRL_PLAY2(12, N_) THRL_PLAY2(14, N_)
                // the same with the all-greedy groups skipping their table build (g0: the group the chain runs next)
#define THRL_PLAY2G(G, N_)                                                                   \
                if ((G) * 4 + 8 <= (N_) && g0 &&                                             \
                    !(busy_groups & (((G) * 4 + 8 < 64 ? 0x110ull : 0x10ull) << ((G) * 4)))) {   \
                    /* this group, the next one and the one after it are all-greedy: two chains, no build, one test */ \
                    chain4(std::integral_constant<int, (G) * 4>(), (N_), ta, true);          \
                    chain4(std::integral_constant<int, (G) * 4 + 4>(), (N_), tb, true);      \
                } else if ((G) * 4 < (N_)) {                                                 \
                    const bool g1 = !((busy_groups >> ((G) * 4 + 4)) & 1ull);                \
                    if ((G) * 4 + 4 < (N_) && !g1) build4((G) * 4 + 4, tb);                  \
                    chain4(std::integral_constant<int, (G) * 4>(), (N_), ta, g0);            \
                    if ((G) * 4 + 4 < (N_)) {                                                \
                        g0 = (G) * 4 + 8 < 64 && !((busy_groups >> (((G) * 4 + 8) & 63)) & 1ull);   \
                        if ((G) * 4 + 8 < (N_) && !g0) build4((G) * 4 + 8, ta);              \
                        chain4(std::integral_constant<int, (G) * 4 + 4>(), (N_), tb, g1);    \
                    }                                                                        \
                }
#define THRL_PLAY16G(N_)                                                                         \
                THRL_PLAY2G(0, N_) THRL_PLAY2G(2, N_) THRL_PLAY2G(4, N_) THRL_PLAY2G(6, N_)      \
                THRL_PLAY2G(8, N_) THRL_PLAY2G(10, N_) THRL_PLAY2G(12, N_) THRL_PLAY2G(14, N_)
                // a full segment (the first of a 100-step episode) runs a copy without any of the tests
                if (greedy_copy) {
                    bool g0 = !(busy_groups & 1ull);
                    if (!g0) build4(0, ta);
                    if (n == 64) { THRL_PLAY16G(64) } else if (n == 36) { THRL_PLAY16G(36) } else { THRL_PLAY16G(n) }
                } else {
                    build4(0, ta);
                    // (36 = the second segment of the reference's 100-step episodes: also without tests)
                    if (n == 64) { THRL_PLAY16(64) } else if (n == 36) { THRL_PLAY16(36) } else { THRL_PLAY16(n) }
                }
#undef THRL_PLAY16G
#undef THRL_PLAY2G
#undef THRL_PLAY16
#undef THRL_PLAY2
                {   // unpack: lanes below the last full group of the hand-scheduled path hold two states per even lane
                    const int packed_end = (NRSEG == 1 && !(kAblate & 2)) ? (n & ~3) : 0;
                    const uint32_t prev = dpp_mov32<0x111>(sq);                 // row_shr:1 -- lane t-1 (t odd: same row)
                    const uint32_t un = (lane & 1) ? (prev >> 16) : (sq & 0xFFFFu);
                    seq[seg] = lane < packed_end ? un : sq_tail;
                }
            }
            const int s_end = s;

            // ---- (d1) lane-parallel (lane = step): actions, old-value snapshot
            //      (agents.py:67) for ALL steps before any TD write
            uint32_t act[NSEG];            // a0 | a1<<8 | train_row<<16 | next_row<<24
            Snapshot<QT> snap[NSEG];       // (1-alpha)*old_value, halves packed per 32 steps
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                const int tt = seg * 64 + lane;
                const bool valid = tt < T;
                const uint32_t my_s = seq[seg] & 0xFFu;              // row the step was played in
                const uint32_t my_train = (seq[seg] >> 8) & 0xFFu;   // row train_net sees for that state
                const uint32_t rw = rwv[seg];
                // gather with ALL lanes active (a bpermute reads only from active lanes, and
                // the source lane here is a table row, unrelated to this lane's step), then select
                const uint32_t gam = gather_row<NRSEG>(AM, my_s);
                uint32_t a0 = (rw & 1u) ? ((rw >> 8) & 0xFFu) : (gam & 0xFFu);
                uint32_t a1 = (rw & 2u) ? ((rw >> 16) & 0xFFu) : (gam >> 8);
                uint32_t nxt = (uint32_t)__shfl_down((int)seq[seg], 1, 64);
                if (seg + 1 < NSEG) { if (lane == 63) nxt = readlane_u(seq[seg + 1 < NSEG ? seg + 1 : seg], 0); }
                const uint32_t ns = (tt + 1 < T) ? ((nxt >> 8) & 0xFFu) : ((uint32_t)s_end >> 8);
                uint32_t srow = my_train;
                if (!valid) { a0 = 0; a1 = 0; srow = 0; }
                snap[seg].set(oma0, tab0[srow * A + a0], oma1, tab1[srow * A + a1]);
                act[seg] = a0 | (a1 << 8) | (srow << 16) | (ns << 24);
            }
            __builtin_amdgcn_wave_barrier();

            double lr0 = 0.0, lr1 = 0.0, la0 = 0.0, la1 = 0.0;
            auto log_into = [&](int epi, double q0, double q1, double q2, double q3) {
                double v = (kAblate & 64) ? q0 + q1 + q2 + q3 : wave_sum4(q0, q1, q2, q3, lane);
                if ((lane & 3) < 2) v = __ddiv_rn(v, inv_T_den);
                const long long vq = __double2ll_rn(__dmul_rn(v, log_scale));
                if ((lane >> 2) == (epi & 15)) { if (epi < 16) acc += vq; else acc_hi += vq; }
            };
#pragma unroll
            for (int seg = 0; seg < NSEG; seg++) {
                // ---- (d2) rewards, LDS write addresses, logs, visit counters of this segment
                const int tt = seg * 64 + lane;
                const bool valid = tt < T;
                const uint32_t a0 = act[seg] & 0xFFu, a1 = (act[seg] >> 8) & 0xFFu;
                const uint32_t srow = (act[seg] >> 16) & 0xFFu;
                // NoisyPriceState.step (environments.py:25-39) from the staged quantities A_i = (a/b)*scaled_i:
                // three float64 operations per 64 steps instead of a gather from the L2-resident price table
                const double aq0 = lut_aq[a0], aq1 = lut_aq[A + a1];
                const double a_eff = (NOISE && (rwv[seg] & 4u)) ? nav[seg] : a.env.a;
                double price = __dsub_rn(a_eff, __dmul_rn(a.env.b, __dadd_rn(aq0, aq1)));
                if (!(price > 0.0)) price = 0.0;
                const double r0d = __dmul_rn(price, aq0);
                const double r1d = __dmul_rn(price, aq1);
                if (seg == NSEG - 1) {
                    const int ll = T - 1 - seg * 64;       // lane of the episode's last step
                    last_price = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(price), ll),
                                                  __builtin_amdgcn_readlane(__double2loint(price), ll));
                }
                const v2u woq = pack_halves(tab0_off + (srow * A + a0) * (unsigned)sizeof(QT),
                                            tab1_off + (srow * A + a1) * (unsigned)sizeof(QT));
                if (epk == 1) {
                    if (valid) {
                        lr0 += r0d; lr1 += r1d;          // divided by T once per episode below
                        la0 += lut_sct[a0]; la1 += lut_sct[A + a1];
                    }
                } else {
                    // several episodes per cycle: this segment's part of each episode's sums goes to that
                    // episode's accumulator slot (fixed point: exact whatever the split)
                    const int ep_l = min(tt, T - 1) / Tenv;
                    const int j0 = (seg * 64) / Tenv, j1 = (min(T, seg * 64 + 64) - 1) / Tenv;
                    const double s0 = lut_sct[a0], s1 = lut_sct[A + a1];
                    for (int j = j0; j <= j1; j++) {
                        const bool m = valid && ep_l == j;
                        log_into(e + j, m ? r0d : 0.0, m ? r1d : 0.0, m ? s0 : 0.0, m ? s1 : 0.0);
                    }
                }
                // visit counters (agents.py:76): the packed transition word goes to this wave's
                // log (coalesced, L2-resident); the counts are built per game below
                if (a.counter && !(kAblate & 32))
                    a.tlog[(((size_t)wave_gid * kWaveMaxEpisodes + e) * NSEG + seg) * 64 + lane] =
                        valid ? act[seg] : 0xFFFFFFFFu;

                // ---- replay schedule of this segment's 16 groups of four transitions, lane-parallel.
                //      Transition j may share a pass with an earlier transition i of its group unless it
                //      reads the row i writes (live next_max, agents.py:71) or rewrites i's cell (:75).
                uint32_t P;
                if (kAblate & 128) P = (act[seg] >> 24) * 0x204081u;
                else {
                    const uint32_t w = act[seg];
                    const uint32_t b0 = dpp_mov32<0x00>(w), b1 = dpp_mov32<0x55>(w), b2 = dpp_mov32<0xAA>(w),
                                   b3 = dpp_mov32<0xFF>(w);                       // quad broadcasts of positions 0..3
                    const uint32_t qp = (uint32_t)lane & 3u;
                    auto hz = [&](uint32_t bi, uint32_t i) -> uint32_t {
                        const bool raw = (w >> 24) == ((bi >> 16) & 0xFFu);            // my next-state row == its written row
                        const bool waw = ((w >> 16) & 0xFFu) == ((bi >> 16) & 0xFFu) &&
                                         ((w & 0xFFu) == (bi & 0xFFu) || ((w >> 8) & 0xFFu) == ((bi >> 8) & 0xFFu));
                        return (qp > i && (raw || waw)) ? 1u : 0u;
                    };
                    const uint32_t hzw = hz(b0, 0u) | (hz(b1, 1u) << 1) | (hz(b2, 2u) << 2);
                    const uint32_t h1 = dpp_mov32<0x55>(hzw), h2 = dpp_mov32<0xAA>(hzw), h3 = dpp_mov32<0xFF>(hzw);
                    const uint32_t c1 = h1 & 1u;
                    const uint32_t c2 = ((h2 >> 1) & 1u) | ((c1 ^ 1u) & h2 & 1u);
                    const uint32_t c3 = ((h3 >> 2) & 1u) | ((c2 ^ 1u) & (((h3 >> 1) & 1u) | ((c1 ^ 1u) & h3 & 1u)));
                    // bit 31: the group runs in registers (replay_group): four identical transitions that stay in
                    // their row, or a period-2 cycle A, B, A, B between two different rows (told apart by rows 0 / 1)
                    // (the period-2 path only in the GREEDY variants: while the agents explore there are no such groups)
                    // NOISE: a noisy step that stays in its row has the SAME word as its neighbours but another reward
                    // (environments.py:29-33: the intercept is drawn), so its group must take the ordinary passes
                    bool quad_quiet = true;
                    if (NOISE) quad_quiet = ((__ballot((rwv[seg] & 4u) != 0u) >> (lane & 60)) & 0xFull) == 0ull;
                    const uint32_t same4 = ((quad_quiet && b0 == b1 && b1 == b2 && b2 == b3 && (b0 >> 24) == ((b0 >> 16) & 0xFFu)) ||
                                            (GREEDY && b0 == b2 && b1 == b3 && (b0 >> 24) == ((b1 >> 16) & 0xFFu) &&
                                             (b1 >> 24) == ((b0 >> 16) & 0xFFu) && (b0 >> 24) != (b1 >> 24))) ? 1u : 0u;
                    P = (b0 >> 24) | ((b1 >> 24) << 7) | ((b2 >> 24) << 14) | ((b3 >> 24) << 21) |
                        ((c1 | (c2 << 1) | (c3 << 2)) << 28) | (same4 << 31);
                }

                // ---- (e) replay (agents.py:68-76): live next_max, writes in transition order
                // GREEDY variants: a converged game repeats ONE transition all segment long (a fixed point of the
                // greedy play) or cycles through TWO, THREE or FOUR.  Then train_net's loop is a recurrence on one cell
                // per cycle position and agent: the rows are read once, split into "the cell" and "the max of the rest",
                // and the segment's transitions run in registers -- two vector instructions each (cyclic_segment).
                bool seg_done = false;
                if (GREEDY && !(kAblate & 257) && !((explored_segs >> seg) & 1u)) {
                    const int nseg = min(64, T - seg * 64);
                    const uint32_t w = act[seg];
                    // smallest period p <= 4 of the segment's transition words (lane t against lane t - p)
                    const uint32_t w1 = dpp_mov32<0x138>(w), w2 = dpp_mov32<0x138>(w1);           // wave_shr:1, twice
                    const uint32_t w3 = dpp_mov32<0x138>(w2), w4 = dpp_mov32<0x138>(w3);
                    const bool in_seg = lane < nseg;
                    const int period = __ballot(in_seg && lane >= 1 && w != w1) == 0ull ? 1
                                     : __ballot(in_seg && lane >= 2 && w != w2) == 0ull ? 2
                                     : __ballot(in_seg && lane >= 3 && w != w3) == 0ull ? 3
                                     : __ballot(in_seg && lane >= 4 && w != w4) == 0ull ? 4 : 0;
                    if (period > 0 && nseg >= period) {
                        uint32_t W[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) W[j] = readlane_u(w, j);
                        // the cycle must close (last -> first) and visit `period` DIFFERENT rows, so that every row
                        // holds exactly one of the rewritten cells
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (j < period) {
                                const uint32_t nxt_row = (W[(j + 1 < period) ? j + 1 : 0] >> 16) & 0xFFu;
                                ok = ok && (W[j] >> 24) == nxt_row;
#pragma unroll
                                for (int i = 0; i < j; i++) ok = ok && ((W[i] >> 16) & 0xFFu) != ((W[j] >> 16) & 0xFFu);
                            }
                        if (ok) {
                            seg_done = true;
                            const BlockOps<QT> ops0 = make_ops(snap[seg], 0, r0d, r1d, alpha_h);      // step layout: lane 32h + t
                            const unsigned tabh = half ? tab1_off : tab0_off;
                            if (period == 1) cyclic_segment<QT, 1>(ops0, woq.x, W, nseg, lane, tabh, A, row_bytes, ag_h, alpha_h, gamma_h);
                            else if (period == 2) cyclic_segment<QT, 2>(ops0, woq.x, W, nseg, lane, tabh, A, row_bytes, ag_h, alpha_h, gamma_h);
                            else if (period == 3) cyclic_segment<QT, 3>(ops0, woq.x, W, nseg, lane, tabh, A, row_bytes, ag_h, alpha_h, gamma_h);
                            else cyclic_segment<QT, 4>(ops0, woq.x, W, nseg, lane, tabh, A, row_bytes, ag_h, alpha_h, gamma_h);
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
                const bool any_fixed = __ballot((P >> 31) != 0u) != 0ull;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (GREEDY && seg_done) break;
                    int nsub = __builtin_amdgcn_readfirstlane(min(32, T - seg * 64 - k * 32));
                    asm volatile("" : "+s"(nsub));          // (as for the play loop: keeps the per-group tests scalar compares)
                    if (nsub <= 0) break;
                    BlockOps<QT> ops = make_ops(snap[seg], k, r0d, r1d, alpha_h);
                    ops.permute(perm_sel);                                   // step layout -> exec layout
                    const unsigned wo = bperm(perm_sel, k ? woq.y : woq.x);
                    // 3 row reads per lane cover 24 columns (fewer columns: clamped duplicates), 4 cover 32.
                    // The fixed-point path is compiled into a second copy of the loop, entered only when the
                    // segment has such a group: the common exploring-regime loop stays as tight as without it.
#define THRL_REPLAY(RDN_, FP_) replay_block<QT, RDN_, FP_, kUnrollReplay && RDN_ == 3, GREEDY>(P, k * 32, nsub, ops, wo, rd_base, row_shift, my_step, row_bytes, \
                                                          ag_h, alpha_h, gamma_h, half != 0, seg * 64 + k * 32, replay_from)
                    if (A > 24)      { if (any_fixed) THRL_REPLAY(4, true); else THRL_REPLAY(4, false); }
                    else if (A >= 3) { if (any_fixed) THRL_REPLAY(3, true); else THRL_REPLAY(3, false); }
                    else             { if (any_fixed) THRL_REPLAY(2, true); else THRL_REPLAY(2, false); }
#undef THRL_REPLAY
                }
            }

            // ---- (f) per-episode log sums of this game into the wave accumulator:
            //      lane L gets the wave total of quantity L&3 = (reward0, reward1, action0, action1)
            if (epk == 1) log_into(e, lr0, lr1, la0, la1);
            // epsilon decays after every train_net call (agents.py:78)
            if (SWEEP) {
                epsg0 = __dadd_rn(eend0, __dmul_rn(__dsub_rn(epsg0, eend0), estep0));
                epsg1 = __dadd_rn(eend1, __dmul_rn(__dsub_rn(epsg1, eend1), estep1));
            }
        }

        // ---- stream the windows back LDS -> HBM, store the env state
        {
            QT* d0 = q0 + lo * A;
            QT* d1 = q1 + lo * A;
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
        if (a.counter && !(kAblate & 32)) {
            const int cells = (W + 2) * A;                       // per agent
            const int hw = (cells + 1) >> 1;                     // dwords per agent
            // may_alias: the histogram overlays the tables (no type-based reordering)
            typedef unsigned __attribute__((may_alias)) hist_u32;
            hist_u32* hist = reinterpret_cast<hist_u32*>(tab0);   // 2*hw dwords <= 2*cells table elements
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            for (int k = lane; k < 2 * hw; k += 64) hist[k] = 0u;
            __builtin_amdgcn_wave_barrier();
            // log read-back: 4 episodes' loads in flight at a time (sc1 = L2-served: the wave
            // reads what it stored itself)
            const int n_cycles = a.n_episodes / epk;
            for (int c0 = 0; c0 < n_cycles; c0 += 4) {
                unsigned w[4][NSEG];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int seg = 0; seg < NSEG; seg++)
                        w[j][seg] = __hip_atomic_load(
                            &a.tlog[(((size_t)wave_gid * kWaveMaxEpisodes + min(c0 + j, n_cycles - 1) * epk) * NSEG + seg) * 64 + lane],
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int seg = 0; seg < NSEG; seg++) {
                        const unsigned ww = w[j][seg];
                        // only transitions still in the deque when it trained were counted (agents.py:76)
                        if (c0 + j < n_cycles && ww != 0xFFFFFFFFu && seg * 64 + lane >= replay_from) {
                            const unsigned srow = (ww >> 16) & 0xFFu;
                            const unsigned c0_ = srow * (unsigned)A + (ww & 0xFFu);
                            const unsigned c1_ = srow * (unsigned)A + ((ww >> 8) & 0xFFu);
                            __hip_atomic_fetch_add(&hist[c0_ >> 1], 1u << ((c0_ & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            __hip_atomic_fetch_add(&hist[hw + (c1_ >> 1)], 1u << ((c1_ & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
            }
            __builtin_amdgcn_wave_barrier();
            // apply: window rows are contiguous in HBM, so cell k of the window is element
            // lo*A + k of the agent's table (no row/column split); spill rows separately
            int32_t* cw0 = a.counter + (int64_t)g * a.stride + p0.table_off;
            int32_t* cw1 = a.counter + (int64_t)g * a.stride + p1.table_off;
            const int nwin = W * A;
            // all counter loads of a batch in flight before the first store (one HBM round trip per 512
            // cells instead of one per 64)
            for (int k0 = 0; k0 < nwin; k0 += 512) {
                int32_t c0v[8], c1v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = min(k0 + j * 64 + lane, nwin - 1);
                    c0v[j] = cw0[lo * A + k]; c1v[j] = cw1[lo * A + k];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = k0 + j * 64 + lane;
                    if (k < nwin) {
                        const unsigned n0 = (hist[k >> 1] >> ((k & 1) << 4)) & 0xFFFFu;
                        const unsigned n1 = (hist[hw + (k >> 1)] >> ((k & 1) << 4)) & 0xFFFFu;
                        if (n0) cw0[lo * A + k] = c0v[j] + (int32_t)n0;
                        if (n1) cw1[lo * A + k] = c1v[j] + (int32_t)n1;
                    }
                }
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

template <typename QT, int NSEG, int NRSEG, bool NOISE, bool SWEEP, bool CYCLE, bool GREEDY = false>
static int launch_wave_t(const WaveArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    auto kern = k_wave_episodes<QT, NSEG, NRSEG, NOISE, SWEEP, CYCLE, GREEDY>;
    if (lds > 64 * 1024) {                       // beyond the default dynamic-LDS limit (float64 tables: one block per CU)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, a);
    return (int)hipGetLastError();
}

template <typename QT, bool NOISE, bool SWEEP, bool CYCLE, bool GREEDY = false>
static int launch_wave_n(const WaveArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    const int nseg = (a.T * a.epk + 63) / 64;
    const int nrseg = (a.win_rows + 2 + 63) / 64;
    if (nrseg == 1) {
        switch (nseg) {
            case 1: return launch_wave_t<QT, 1, 1, NOISE, SWEEP, CYCLE, GREEDY>(a, grid, block, lds, s);
            case 2: return launch_wave_t<QT, 2, 1, NOISE, SWEEP, CYCLE, GREEDY>(a, grid, block, lds, s);
            case 3: return launch_wave_t<QT, 3, 1, NOISE, SWEEP, CYCLE, GREEDY>(a, grid, block, lds, s);
            case 4: return launch_wave_t<QT, 4, 1, NOISE, SWEEP, CYCLE, GREEDY>(a, grid, block, lds, s);
        }
    } else if (nrseg == 2) {
        switch (nseg) {
            case 1: return launch_wave_t<QT, 1, 2, NOISE, SWEEP, CYCLE, GREEDY>(a, grid, block, lds, s);
            case 2: return launch_wave_t<QT, 2, 2, NOISE, SWEEP, CYCLE, GREEDY>(a, grid, block, lds, s);
            case 3: return launch_wave_t<QT, 3, 2, NOISE, SWEEP, CYCLE, GREEDY>(a, grid, block, lds, s);
            case 4: return launch_wave_t<QT, 4, 2, NOISE, SWEEP, CYCLE, GREEDY>(a, grid, block, lds, s);
        }
    }
    return -1;
}

}  // namespace thrl
