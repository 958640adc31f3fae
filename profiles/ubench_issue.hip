// ubench_issue.hip -- gfx950 instruction-issue microbenchmark (round 3, verdict item 1a).
//
// What does one SIMD of an MI355X CU sustain, in cycles per wave64 instruction, for the instruction classes the
// fused episode kernel (th_rl_amd/csrc/thrl_wave_kernel.h) is made of, at 1..8 resident waves per SIMD?
// The answer prices the kernel's SQ instruction counters (profiles/r03_*_pmc_summary.csv) into "issue cycles used"
// and settles the 2-cycle-vs-4-cycle question of the round-2 verdict with a measurement instead of a reading of
// SQ_ACTIVE_INST_VALU.
//
// Method: every wave runs ITERS x 128 independent instructions of ONE class (8 register chains in rotation, so no
// instruction depends on a result younger than 8 instructions: far past every VALU/DPP/readlane hazard window),
// stamped with s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop.  Blocks are 4 waves (one per
// SIMD) and dynamic LDS is sized so that exactly W blocks fit a CU: W waves per SIMD, checked from HW_ID.
// cycles per instruction per SIMD = median over waves of (dt_shader / (ITERS * 128)) / W.
//
//   hipcc --offload-arch=gfx950 -O2 -o ubench_issue profiles/ubench_issue.hip && ./ubench_issue > r03_ubench_issue.json
//
// Standalone (links /opt/rocm's HIP runtime; not part of libthrl_hip.so).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Cls { V_AND, V_AND_E64, V_AND_SGPR, V_ADD_U32, V_MOV, V_LSHL, V_ADD_F32, V_MUL_F32, V_FMAC_F32, V_PK_FMA_F32, V_ADD3, V_CNDMASK_E64, V_CNDMASK_E64_VCC, V_CNDMASK_NEWDST, V_CNDMASK_AFTER_CMP, V_MUL_LO_U32, V_MUL_HI_U32, V_XOR, V_CVT_F64_U32, V_CMP_LT_F64, CHAIN4,
           V_ADD_SDWA, V_MAD_U24, V_FMA_F32, V_MAX_DPP, V_MOV_DPP_SHR, V_CNDMASK, V_READLANE, V_WRITELANE, V_PERMLANE32,
           V_FMA_F64, V_ADD_F64, V_MUL_F64, V_MAX3_F32, V_CMP_F32, S_ADD, S_NOP0, S_PACK, DS_READ_U16, DS_READ_B32, DS_READ2_B32, DS_BPERMUTE,
           MIX_VALU_SALU, MIX_VALU_LDS, MIX_CND_AND_1_1, MIX_CND_AND_1_3, N_CLS };
static const char* kNames[N_CLS] = {"v_and_b32", "v_and_b32_e64 (VOP3 encoding)", "v_and_b32 (SGPR src0)", "v_add_u32", "v_mov_b32", "v_lshlrev_b32",
                                    "v_add_f32", "v_mul_f32", "v_fmac_f32 (VOP2)", "v_pk_fma_f32", "v_add3_u32", "v_cndmask_b32_e64 (SGPR-pair mask)", "v_cndmask_b32_e64 (mask = vcc)", "v_cndmask_b32 vcc, dst != src", "v_cmp_gt_f32 vcc + 7 v_cndmask_b32 vcc",
                                    "v_mul_lo_u32", "v_mul_hi_u32", "v_xor_b32", "v_cvt_f64_u32", "v_cmp_lt_f64(->vcc)",
                                    "play chain group: 4 dependent v_readlane + 2 s_pack + 2 v_writelane + s_nop pads (12 instructions = 4 steps)",
                                    "v_add_u32_sdwa", "v_mad_u32_u24", "v_fma_f32", "v_max_f32_dpp(quad_perm)", "v_mov_b32_dpp(row_shr:1)",
                                    "v_cndmask_b32", "v_readlane_b32", "v_writelane_b32", "v_permlane32_swap_b32", "v_fma_f64", "v_add_f64", "v_mul_f64",
                                    "v_max3_f32", "v_cmp_gt_f32(->vcc)", "s_add_u32", "s_nop 0", "s_pack_ll_b32_b16", "ds_read_u16", "ds_read_b32",
                                    "ds_read2_b32", "ds_bpermute_b32", "mix: v_and_b32 + s_add_u32 (1:1)", "mix: 3 v_and_b32 + 1 ds_read_b32",
                                    "mix: v_cndmask_b32 (vcc) + v_and_b32 (1:1)", "mix: v_cndmask_b32 (vcc) + 3 v_and_b32"};

// 8 instructions of class C on 8 independent registers
#define V8(OP)                                                                                              \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                            \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                 : "v"(x), "s"(sx), "v"(ldsa) : "vcc", "scc", "memory")
#define D8(OP)                                                                                              \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                            \
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) \
                 : "v"(dx) : "memory")
#define S8(OP)                                                                                              \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                            \
                 : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]), "+s"(s[4]), "+s"(s[5]), "+s"(s[6]), "+s"(s[7]) \
                 : "s"(sx), "v"(x) : "scc", "memory")
// (operand numbering: 0-7 the chains, 8 = x / dx / sx, 9 = sx / x, 10 = ldsa)
#define I_AND(k)   "v_and_b32 %" #k ", %8, %" #k "\n\t"
#define I_ANDE64(k) "v_and_b32_e64 %" #k ", %8, %" #k "\n\t"
#define I_ANDS(k)  "v_and_b32 %" #k ", %9, %" #k "\n\t"
#define I_ADDU(k)  "v_add_u32 %" #k ", %8, %" #k "\n\t"
#define I_MOV(k)   "v_mov_b32 %" #k ", %8\n\t"
#define I_LSHL(k)  "v_lshlrev_b32 %" #k ", 1, %" #k "\n\t"
#define I_ADDF(k)  "v_add_f32 %" #k ", %8, %" #k "\n\t"
#define I_MULF(k)  "v_mul_f32 %" #k ", %8, %" #k "\n\t"
#define I_FMAC(k)  "v_fmac_f32 %" #k ", %8, %8\n\t"
#define I_ADD3(k)  "v_add3_u32 %" #k ", %" #k ", %8, %8\n\t"
#define I_CNDE64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n\t"
#define I_PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %8\n\t"
#define I_CNDE64V(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, vcc\n\t"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n\t"
#define I_MULHI(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n\t"
#define I_XOR(k)   "v_xor_b32 %" #k ", %8, %" #k "\n\t"
#define I_CMP64(k) "v_cmp_lt_f64 vcc, %" #k ", %8\n\t"
#define I_SDWA(k)  "v_add_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
#define I_MAD(k)   "v_mad_u32_u24 %" #k ", %" #k ", %8, %8\n\t"
#define I_FMA(k)   "v_fma_f32 %" #k ", %" #k ", %8, %8\n\t"
#define I_DPPMAX(k) "v_max_f32_dpp %" #k ", %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPMOV(k) "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_CND(k)   "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define I_WL(k)    "v_writelane_b32 %" #k ", %9, 5\n\t"
#define I_PL32(k)  "v_permlane32_swap_b32 %" #k ", %" #k "\n\t"
#define I_MAX3(k)  "v_max3_f32 %" #k ", %" #k ", %8, %8\n\t"
#define I_CMP(k)   "v_cmp_gt_f32 vcc, %" #k ", %8\n\t"
#define I_FMA64(k) "v_fma_f64 %" #k ", %" #k ", %8, %8\n\t"
#define I_ADD64(k) "v_add_f64 %" #k ", %" #k ", %8\n\t"
#define I_MUL64(k) "v_mul_f64 %" #k ", %" #k ", %8\n\t"
#define I_RL(k)    "v_readlane_b32 %" #k ", %9, 5\n\t"
#define I_SADD(k)  "s_add_u32 %" #k ", %" #k ", %8\n\t"
#define I_SNOP(k)  "s_nop 0\n\t"
#define I_SPACK(k) "s_pack_ll_b32_b16 %" #k ", %" #k ", %8\n\t"
#define I_DSU16(k) "ds_read_u16 %" #k ", %10\n\t"
#define I_DSB32(k) "ds_read_b32 %" #k ", %10\n\t"
#define I_DSBP(k)  "ds_bpermute_b32 %" #k ", %10, %8\n\t"
#define I_MIXVS(k) "v_and_b32 %" #k ", %9, %" #k "\n\ts_add_u32 %8, %8, 1\n\t"

template <int C>
__global__ void __launch_bounds__(256) k_issue(int iters, uint64_t* out) {
    extern __shared__ unsigned lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    unsigned v[8];
    double d[8];
    unsigned s[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = lane * 17 + k; d[k] = 1.0 + 1e-9 * (lane + k); s[k] = k + blockIdx.x; }
    unsigned x = 0x00FF00FFu ^ lane;
    double dx = 1.0000001;
    unsigned sx = __builtin_amdgcn_readfirstlane((int)blockIdx.x) | 3u;
    unsigned ldsa = (unsigned)(lane * 4) & 0xFFCu;          // conflict-free: consecutive dwords
    unsigned v2[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v2[k] = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {            // 16 x 8 = 128 instructions per iteration
            if (C == V_AND) V8(I_AND);
            else if (C == V_AND_E64) V8(I_ANDE64);
            else if (C == V_AND_SGPR) V8(I_ANDS);
            else if (C == V_ADD_U32) V8(I_ADDU);
            else if (C == V_MOV) V8(I_MOV);
            else if (C == V_LSHL) V8(I_LSHL);
            else if (C == V_ADD_F32) V8(I_ADDF);
            else if (C == V_MUL_F32) V8(I_MULF);
            else if (C == V_FMAC_F32) V8(I_FMAC);
            else if (C == V_ADD3) V8(I_ADD3);
            else if (C == V_CNDMASK_E64) {
                asm volatile("s_mov_b64 s[20:21], 0x5555\n\t" I_CNDE64(0) I_CNDE64(1) I_CNDE64(2) I_CNDE64(3) I_CNDE64(4) I_CNDE64(5) I_CNDE64(6) I_CNDE64(7)
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                             : "v"(x) : "s20", "s21", "memory");
            }
            else if (C == V_CNDMASK_E64_VCC) V8(I_CNDE64V);
            else if (C == V_CNDMASK_NEWDST) {
                asm volatile("v_cndmask_b32 %0, %4, %8, vcc\n\tv_cndmask_b32 %1, %5, %8, vcc\n\tv_cndmask_b32 %2, %6, %8, vcc\n\tv_cndmask_b32 %3, %7, %8, vcc\n\t"
                             "v_cndmask_b32 %4, %0, %8, vcc\n\tv_cndmask_b32 %5, %1, %8, vcc\n\tv_cndmask_b32 %6, %2, %8, vcc\n\tv_cndmask_b32 %7, %3, %8, vcc\n\t"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(x) : "vcc");
            }
            else if (C == V_CNDMASK_AFTER_CMP) {
                asm volatile("v_cmp_gt_f32 vcc, %0, %8\n\t" I_CND(1) I_CND(2) I_CND(3) I_CND(4) I_CND(5) I_CND(6) I_CND(7)
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(x) : "vcc");
            }
            else if (C == V_MUL_LO_U32) V8(I_MULLO);
            else if (C == V_MUL_HI_U32) V8(I_MULHI);
            else if (C == V_XOR) V8(I_XOR);
            else if (C == V_CVT_F64_U32) {
                asm volatile("v_cvt_f64_u32 %0, %4\n\tv_cvt_f64_u32 %1, %5\n\tv_cvt_f64_u32 %2, %6\n\tv_cvt_f64_u32 %3, %7\n\t"
                             "v_cvt_f64_u32 %0, %5\n\tv_cvt_f64_u32 %1, %6\n\tv_cvt_f64_u32 %2, %7\n\tv_cvt_f64_u32 %3, %4\n\t"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
            }
            else if (C == V_CMP_LT_F64) {
                asm volatile(I_CMP64(0) I_CMP64(1) I_CMP64(2) I_CMP64(3) I_CMP64(4) I_CMP64(5) I_CMP64(6) I_CMP64(7)
                             :: "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(dx) : "vcc");
            }
            else if (C == V_PK_FMA_F32) D8(I_PKFMA);
            else if (C == CHAIN4) {
                // the hand-scheduled play-chain group of thrl_wave_kernel.h (chain_group): s <- table_j[s], four steps,
                // the states recorded two per lane; 12 instructions
                int s1, s2, s3, s4, pk;
                int st = (int)(s[0] & 63u);
                asm volatile("s_nop 3\n\tv_readlane_b32 %1, %7, %6\n\ts_pack_ll_b32_b16 %5, %6, %1\n\tv_writelane_b32 %0, %5, 4\n\ts_nop 1\n\t"
                             "v_readlane_b32 %2, %8, %1\n\ts_nop 3\n\tv_readlane_b32 %3, %9, %2\n\ts_pack_ll_b32_b16 %5, %2, %3\n\t"
                             "v_writelane_b32 %0, %5, 6\n\ts_nop 1\n\tv_readlane_b32 %4, %10, %3"
                             : "+v"(v[7]), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(pk)
                             : "s"(st), "v"(v[0] & 63u), "v"(v[1] & 63u), "v"(v[2] & 63u), "v"(v[3] & 63u) : "scc");
                s[0] = (unsigned)s4;
            }
            else if (C == V_ADD_SDWA) V8(I_SDWA);
            else if (C == V_MAD_U24) V8(I_MAD);
            else if (C == V_FMA_F32) V8(I_FMA);
            else if (C == V_MAX_DPP) V8(I_DPPMAX);
            else if (C == V_MOV_DPP_SHR) V8(I_DPPMOV);
            else if (C == V_CNDMASK) V8(I_CND);
            else if (C == V_WRITELANE) V8(I_WL);
            else if (C == V_PERMLANE32) {
                asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                             "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            }
            else if (C == V_MAX3_F32) V8(I_MAX3);
            else if (C == V_CMP_F32) V8(I_CMP);
            else if (C == V_FMA_F64) D8(I_FMA64);
            else if (C == V_ADD_F64) D8(I_ADD64);
            else if (C == V_MUL_F64) D8(I_MUL64);
            else if (C == V_READLANE) S8(I_RL);
            else if (C == S_ADD) S8(I_SADD);
            else if (C == S_NOP0) S8(I_SNOP);
            else if (C == S_PACK) S8(I_SPACK);
            else if (C == DS_READ_U16) { V8(I_DSU16); }
            else if (C == DS_READ_B32) { V8(I_DSB32); }
            else if (C == DS_READ2_B32) {
                asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset0:2 offset1:3\n\t"
                             "ds_read2_b32 %2, %4 offset0:4 offset1:5\n\tds_read2_b32 %3, %4 offset0:6 offset1:7\n\t"
                             "ds_read2_b32 %0, %4 offset0:8 offset1:9\n\tds_read2_b32 %1, %4 offset0:10 offset1:11\n\t"
                             "ds_read2_b32 %2, %4 offset0:12 offset1:13\n\tds_read2_b32 %3, %4 offset0:14 offset1:15\n\t"
                             : "+v"(*(uint64_t*)&v[0]), "+v"(*(uint64_t*)&v[2]), "+v"(*(uint64_t*)&v[4]), "+v"(*(uint64_t*)&v[6]) : "v"(ldsa) : "memory");
            }
            else if (C == DS_BPERMUTE) { V8(I_DSBP); }
            else if (C == MIX_VALU_SALU) {
                asm volatile(I_MIXVS(0) I_MIXVS(1) I_MIXVS(2) I_MIXVS(3)
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+s"(s[0])
                             : "v"(x) : "scc", "memory");
            }
            else if (C == MIX_CND_AND_1_1) {
                asm volatile(I_CND(0) I_AND(1) I_CND(2) I_AND(3) I_CND(4) I_AND(5) I_CND(6) I_AND(7)
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(x) : "vcc");
            }
            else if (C == MIX_CND_AND_1_3) {
                asm volatile(I_CND(0) I_AND(1) I_AND(2) I_AND(3) I_CND(4) I_AND(5) I_AND(6) I_AND(7)
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(x) : "vcc");
            }
            else if (C == MIX_VALU_LDS) {
                asm volatile("v_and_b32 %0, %8, %0\n\tv_and_b32 %1, %8, %1\n\tv_and_b32 %2, %8, %2\n\tds_read_b32 %3, %9\n\t"
                             "v_and_b32 %4, %8, %4\n\tv_and_b32 %5, %8, %5\n\tv_and_b32 %6, %8, %6\n\tds_read_b32 %7, %9\n\t"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v2[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v2[7])
                             : "v"(x), "v"(ldsa) : "memory");
            }
        }
        if (C >= DS_READ_U16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // once per 128: keeps the LDS queue bounded
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    double dacc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { acc ^= v[k] ^ s[k] ^ v2[k]; dacc += d[k]; }
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lane == 0) {
        const size_t w = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
        out[w] = t1 - t0; out[w + 1] = r1 - r0; out[w + 2] = ((uint64_t)xcc << 32) | hwid;
        out[w + 3] = acc + (uint64_t)dacc;
    }
}

typedef void (*kern_t)(int, uint64_t*);
template <int C> struct Tab { static void fill(kern_t* t) { t[C] = k_issue<C>; Tab<C + 1>::fill(t); } };
template <> struct Tab<N_CLS> { static void fill(kern_t*) {} };

// instructions each wave executes per loop iteration
static int per_iter(int c) {
    if (c == MIX_VALU_SALU) return 16 * 8;          // 4 VALU + 4 SALU per block
    if (c == CHAIN4) return 16 * 12;                // 16 groups of 12 instructions (= 64 steps of the chain)
    return 128;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int lds_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    kern_t tab[N_CLS];
    Tab<0>::fill(tab);
    const int Ws[] = {1, 2, 3, 4, 5, 6, 8};
    uint64_t* out;
    CK(hipMalloc(&out, sizeof(uint64_t) * 4 * 4 * cus * 8));
    std::vector<uint64_t> h((size_t)4 * 4 * cus * 8);
    printf("{\"device\": \"%s\", \"cus\": %d, \"lds_per_cu\": %d, \"iters\": %d, \"instructions_per_wave\": %d,\n \"unit\": \"shader cycles per wave64 instruction, per SIMD (median over waves of dt / n_inst / waves_per_simd)\",\n \"rows\": [\n",
           prop.gcnArchName, cus, lds_cu, iters, iters * 128);
    bool first = true;
    for (int c = 0; c < N_CLS; c++) {
        for (int wi = 0; wi < (int)(sizeof(Ws) / sizeof(Ws[0])); wi++) {
            const int W = Ws[wi];
            // exactly W blocks of 4 waves per CU: LDS per block just under lds_cu / W, more than lds_cu / (W + 1)
            size_t lds = (size_t)(lds_cu / W) - 1024;
            lds = lds / 512 * 512;
            if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)tab[c], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * W;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(tab[c], dim3(grid), dim3(256), lds, 0, iters / 8 + 1, out);        // warm-up (clocks, icache)
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(tab[c], dim3(grid), dim3(256), lds, 0, iters, out);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const int nw = grid * 4;
            CK(hipMemcpy(h.data(), out, sizeof(uint64_t) * 4 * nw, hipMemcpyDeviceToHost));
            std::vector<double> cyc(nw), clk(nw);
            std::map<uint64_t, int> per_simd;
            for (int w = 0; w < nw; w++) {
                cyc[w] = (double)h[4 * w];
                clk[w] = (double)h[4 * w] / (double)h[4 * w + 1] * 100.0;           // MHz
                const uint64_t id = h[4 * w + 2];
                // gfx9 HW_ID: simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
                const uint64_t key = ((id >> 32) & 0xF) << 16 | ((id >> 4) & 0x3) | (((id >> 8) & 0xFF) << 2);
                per_simd[key]++;
            }
            std::sort(cyc.begin(), cyc.end());
            std::sort(clk.begin(), clk.end());
            int wmin = 1 << 30, wmax = 0;
            for (auto& kv : per_simd) { wmin = std::min(wmin, kv.second); wmax = std::max(wmax, kv.second); }
            const double n_inst = (double)iters * per_iter(c);
            const double med = cyc[nw / 2], p10 = cyc[nw / 10], p90 = cyc[nw * 9 / 10];
            const double mhz = clk[nw / 2];
            // wall-clock cross-check: the whole grid's instructions / (SIMDs x wall x clock)
            const double wall_cyc_per_inst_simd = (ms * 1e-3 * mhz * 1e6) / (n_inst * W);
            printf("%s  {\"class\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_per_simd\": %.3f, \"p10\": %.3f, \"p90\": %.3f, "
                   "\"one_wave_cycles_per_inst\": %.3f, \"wall_cycles_per_inst_per_simd\": %.3f, \"shader_mhz\": %.0f, \"simds_seen\": %d, "
                   "\"waves_per_simd_seen\": [%d, %d], \"kernel_ms\": %.3f}",
                   first ? "" : ",\n", kNames[c], W, med / n_inst / W, p10 / n_inst / W, p90 / n_inst / W, med / n_inst,
                   wall_cyc_per_inst_simd, mhz, (int)per_simd.size(), wmin, wmax, ms);
            first = false;
            fflush(stdout);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
    }
    printf("\n ]}\n");
    CK(hipFree(out));
    return 0;
}
