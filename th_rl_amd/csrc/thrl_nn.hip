// thrl_nn.hip -- the reference's `Reinforce` agent (agents.py:119-220) for G games:
// one 256-thread block per game; a 1 -> 256 -> A MLP, float32, no MFMA (BASELINE
// config #4: per-game independent weights make every product a tiny GEMV).
#include <math.h>

#include "thrl_policy.h"
#include <type_traits>
#include "thrl_kernels.h"

namespace thrl {

constexpr uint32_t kStreamNnInit = 0x90u;

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

// torch.nn.Linear.reset_parameters: weight ~ kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +),
// bias ~ U(-1/sqrt(fan_in), +)
// value_head: ActorCritic's fc_v (256 -> 1) appended, bias filled with 1000 (agents.py:243-244)
__global__ void __launch_bounds__(256) k_nn_init(int G, int A, float* params, uint64_t seed,
                                                  uint64_t game_offset, int agent, int value_head) {
    const int Pp = 2 * kH + A * kH + A;
    const int P = Pp + (value_head ? kH + 1 : 0);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)G * P) return;
    const int g = (int)(idx / P), j = (int)(idx - (int64_t)g * P);
    const u32x4 x = draw(seed, game_offset + (uint64_t)g, (uint32_t)agent, (uint32_t)(j >> 2), kStreamNnInit);
    const uint32_t r = (j & 3) == 0 ? x.x : ((j & 3) == 1 ? x.y : ((j & 3) == 2 ? x.z : x.w));
    const float u = (float)((double)r * 0x1p-32);                      // [0,1)
    const float bound = j < 2 * kH ? 1.0f : 1.0f / sqrtf((float)kH);   // fan_in = 1 for fc1, 256 for fc_pi / fc_v
    params[idx] = j == Pp + kH ? 1000.0f : (2.0f * u - 1.0f) * bound;
}

// pi() + categorical sample / argmax: one game per wavefront (thrl_policy.h), 4 games per block
template <int APAD>
__global__ void __launch_bounds__(256) k_nn_act(int G, int A, const float* __restrict__ params, int P,
                                                 const double* __restrict__ price, const double* __restrict__ u,
                                                 int32_t* __restrict__ action_out, float* __restrict__ prob_out) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    PolicyRegs<APAD> r;
    policy_load(r, params + (int64_t)g * P, A, lane);
    float prob;
    const int a = policy_act(r, A, (float)price[g], u != nullptr, u ? (float)u[g] : 0.0f, lane, &prob);
    if (lane == 0) action_out[g] = a;
    if (prob_out && !(lane & 1) && (lane >> 1) < A) prob_out[(int64_t)g * A + (lane >> 1)] = prob;
}

// train_net for one game per block (agents.py:171-193; ActorCritic: agents.py:280-305), 256 threads.  Two paths:
//   * the FOLDED path (the usual one): the policy has one input, so it is piecewise linear in the price; over the batch's
//     sorted distinct states its forward and backward passes are running sums -- O((states + 256) A) operations, exact
//     integer accumulation (see the comment at `if (U > 0)`).  Taken for up to 448 distinct states found by an LDS hash table
//     (Reinforce and ActorCritic), and for Reinforce also with one state per transition (continuous prices: env noise) up to
//     1,024 transitions;
//   * the PLAIN path for everything else.  The n transitions are processed in chunks of kChunk = 256:
//   pass A1  thread = (4 transitions) x (6 actions): logits.  Each fc_pi weight read from LDS
//            feeds 4 FMAs (packed v_pk_fma_f32), so the pass is VALU-bound, not LDS-bound.
//   pass A2  thread = transition: softmax, entropy, d loss / d logits (in place over the logits).
//   pass B   thread = (hidden units j, j+128) x (every other transition): that unit pair's column
//            of every gradient; one 24-float dz row read feeds 84 FMAs.  The two transition
//            halves are added through LDS after the last chunk.
// Rows of fc_pi^T and of dz are padded to kPad = 24 or 32 floats (zero) for aligned vector LDS reads.
constexpr int kChunk = 256;
constexpr int kUmax = 64;          // state folding: distinct states handled per chunk
constexpr int kUfold = 1024;       // ... and per update: every transition may have its own state (env noise: continuous prices)
constexpr int kUhash = 448;        // distinct states the hash table folds (21 x 21 action pairs = 441 prices); beyond: one state per transition
constexpr int kHash = 1024;        // slots of the LDS table that finds them (overlaid on dz: dead before the passes start)
constexpr int kXu = kUfold + 64;   // the distinct states, zero padded to whole chunks
constexpr int kXuAc = 512;         // ... of an ActorCritic batch (states and next states together; hash fold only)
__host__ __device__ constexpr size_t train_ac_fold_bytes(int pad) {      // sga [kUmax][pad] i64 | vstate [kUhash] f32 | gvfix [kUhash] i64
    return (size_t)kUmax * pad * 8 + (size_t)kUhash * 4 + (size_t)kUhash * 8;
}
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// round(v * 2^40) as int64 for |v| < 2^22: two 32-bit conversions of the rounded product instead of the library's emulated
// double -> int64 (a dozen of these per thread and chunk in the folded update)
__device__ __forceinline__ long long fix40(double v) {
    const double xr = rint(v * 1099511627776.0);                        // integer valued
    const double hi = floor(xr * 0x1p-32);
    const unsigned lo = (unsigned)__builtin_fma(-hi, 4294967296.0, xr);   // in [0, 2^32), exact
    return (long long)(((unsigned long long)(unsigned)(int)hi << 32) | (unsigned long long)lo);
}

__device__ __forceinline__ int train_xs_len(int N) { return (N + kChunk - 1) / kChunk * kChunk; }

// AC = true: ActorCritic.train_net (agents.py:280-305) as the reference executes it -- `rewards` [N]
// against v / v_prime [N,1] broadcasts the advantage to [N,N], adv[i,j] = r_j + gamma*v'_i - v_i;
// its row sums drive the critic (d loss/d v_i = -(2/N^2)(R + N c_i), c_i = gamma*v'_i - v_i, v' not
// detached) and its column sums the actor (weight r_j + C/N in place of Reinforce's return).
template <int kPad, bool AC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_nn_reinforce_train(int G, int A, float* __restrict__ params,
        float* __restrict__ adam_m, float* __restrict__ adam_v, int step, int N, int ld,
        const double* __restrict__ price, const int32_t* __restrict__ action, const double* __restrict__ reward,
        const double* __restrict__ nprice, float gamma, float ent_coef, float lr,
        const double* __restrict__ gamma_g, const double* __restrict__ ent_g, float* __restrict__ grad_out,
        const float* __restrict__ returns) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_nn[];
    if (gamma_g) gamma = (float)gamma_g[blockIdx.x];        // per-game sweeps (main.py:13-21 as one batch)
    if (ent_g) ent_coef = (float)ent_g[blockIdx.x];
    const int NX = train_xs_len(N);
    float* W2t = reinterpret_cast<float*>(smem_nn);         // [kH][kPad]  fc_pi.weight transposed (AC: column A = fc_v.weight)
    float* dz = W2t + kH * kPad;                            // [kChunk][kPad]
    // Reinforce:   xs | Gs | w1s | b1s | b2s | red | uid | xu | ucnt | redi | sga
    // ActorCritic: w1s | b1s | xs | Gs | xps | gvs | wvs | (pad) | uid | uidp | xu | ucnt | redi | b2s | red, and the folded path's
    //              accumulators sga | vstate | gvfix OVER w1s .. wvs (+ pad), which are dead by then (train_ac_fold_bytes)
    float* xs; float* Gs; float* w1s; float* b1s; float* b2s; float* red; float* xps; float* gvs; float* wvs;
    unsigned short* uid; unsigned short* uidp; float* xu; int* ucnt; int* redi; long long* sga;
    float* vstate = nullptr;                                // [kUhash] v(x) of every distinct state            (AC fold)
    long long* gvfix = nullptr;                             // [kUhash] d loss / d v summed per state, fixed point (AC fold)
    if (!AC) {
        xs = dz + kChunk * kPad; Gs = xs + NX; w1s = Gs + NX; b1s = w1s + kH; b2s = b1s + kH; red = b2s + kMaxA;
        xps = red + 8; gvs = xps; wvs = gvs;                // (unused)
        uid = reinterpret_cast<unsigned short*>(red + 8);   // [NX]  index of the transition's distinct state
        uidp = uid;
        xu = reinterpret_cast<float*>(uid + NX);            // [kXu] the distinct states in ascending order, zero padded
        ucnt = reinterpret_cast<int*>(xu + kXu);            // [kUmax] transitions per state of the current chunk
        redi = ucnt + kUmax;                                // [8]
        sga = reinterpret_cast<long long*>(redi + 8);       // [kUmax][kPad] returns by (state, action) of the chunk, 2^-40 fixed point
    } else {
        w1s = dz + kChunk * kPad; b1s = w1s + kH;
        xs = b1s + kH; Gs = xs + NX;
        xps = Gs + NX;                                      // [NX]  next states
        gvs = xps + NX;                                     // [NX]  c_i, then d loss/d v_i (plain path)
        wvs = gvs + NX;                                     // [kH]  fc_v.weight
        const size_t region = (size_t)(reinterpret_cast<unsigned char*>(wvs + kH) - reinterpret_cast<unsigned char*>(w1s));
        const size_t need = train_ac_fold_bytes(kPad);
        uid = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(w1s) + (region > need ? region : need));   // [NX] index of the transition's state ...
        uidp = uid + NX;                                    // [NX]  ... and of its next state
        xu = reinterpret_cast<float*>(uidp + NX);           // [kXuAc]
        ucnt = reinterpret_cast<int*>(xu + kXuAc);
        redi = ucnt + kUmax;
        b2s = reinterpret_cast<float*>(redi + 8);           // (behind everything the gradient staging area overlays: the norm's
        red = b2s + kMaxA;                                  //  reduction scratch is used while the staged gradient is live)
        sga = reinterpret_cast<long long*>(w1s);
        vstate = reinterpret_cast<float*>(sga + kUmax * kPad);
        gvfix = reinterpret_cast<long long*>(vstate + kUhash);
    }
    // the hash table lives inside dz (8 KB in; the first 2 KB hold the packed keys): nothing else uses dz until the passes
    unsigned* hkeys = reinterpret_cast<unsigned*>(dz) + 2048;  // [kHash] hash table of the distinct states (float bits)
    unsigned short* hrank = reinterpret_cast<unsigned short*>(hkeys + kHash);   // [kHash] slot -> index of the state
    static_assert((2048 + kHash) * 4 + kHash * 2 <= kChunk * 24 * 4 && kUhash <= 512 && kUfold <= 1024, "hash table fits inside dz");
    // (dz words 0-1023: packed keys; words 1024-1279: their ranks as u16)
    const int g = blockIdx.x, tid = threadIdx.x;
    const int Pp = 2 * kH + A * kH + A;                     // policy part; fc_v follows it
    const int P = Pp + (AC ? kH + 1 : 0);
    float* w = params + (int64_t)g * P;
    // the replayed transitions of game g: one contiguous row [ld] per array (game-major rings, ABI v3)
    price += (size_t)g * ld; reward += (size_t)g * ld; action += (size_t)g * ld;
    if (AC) nprice += (size_t)g * ld;
    const bool have_returns = !AC && returns != nullptr;          // discounted returns computed by k_nn_returns (same bits)
    if (have_returns) returns += (size_t)g * ld;

    // The weights are staged behind the transitions: for Reinforce the serial return recurrence (one thread, ~25 % of the
    // block's time) needs only the rewards, so waves 1-3 transpose the weights into LDS WHILE thread 0 runs it.
    auto stage_weights = [&](int r0, int stride) {
        for (int r = r0; r < kH; r += stride) {
            for (int k = 0; k < A; k++) W2t[r * kPad + k] = w[2 * kH + k * kH + r];
            for (int k = A; k < kPad; k++) W2t[r * kPad + k] = (AC && k == A) ? w[Pp + r] : 0.0f;     // (the folded path's value column)
            w1s[r] = w[r]; b1s[r] = w[kH + r];
            if (r < kMaxA) b2s[r] = r < A ? w[2 * kH + A * kH + r] : 0.0f;
            if (AC) wvs[r] = w[Pp + r];
        }
    };
    for (int n = tid; n < NX; n += 256) {
        xs[n] = n < N ? (float)price[n] : 0.0f;
        Gs[n] = n < N ? (have_returns ? returns[n] : (float)reward[n]) : 0.0f;
        if (AC) xps[n] = n < N ? (float)nprice[n] : 0.0f;
    }
    if (AC || have_returns) stage_weights(tid, 256);
    __syncthreads();
    float wva = 0.0f, wvb = 0.0f, gbv = 0.0f;
    int U = 0;                                              // > 0: passes run over U distinct states
    if (!AC) {
        // discounted return: the reference's serial recurrence, last to first (agents.py:178-181)
        // One thread, the same operations in the same order; the chain runs in registers on aligned quads, so the
        // LDS reads of the next quads are in flight while the current one is computed (+1.7 % on 2 x Reinforce).
        if (have_returns) {
            // (nothing: Gs already holds the returns)
        } else if (tid == 0) {
            float carry = Gs[N - 1];
            int n = N - 2;
            for (; n >= 0 && ((n + 1) & 3) != 0; n--) { carry = __fadd_rn(Gs[n], __fmul_rn(gamma, carry)); Gs[n] = carry; }
            auto quad = [&](f4& q) {
                q.w = __fadd_rn(q.w, __fmul_rn(gamma, carry));
                q.z = __fadd_rn(q.z, __fmul_rn(gamma, q.w));
                q.y = __fadd_rn(q.y, __fmul_rn(gamma, q.z));
                q.x = __fadd_rn(q.x, __fmul_rn(gamma, q.y));
                carry = q.x;
            };
            // sixteen values per round trip: the four LDS reads are issued before the dependent chain starts
            for (; n >= 15; n -= 16) {
                f4 q0 = *reinterpret_cast<const f4*>(Gs + n - 3), q1 = *reinterpret_cast<const f4*>(Gs + n - 7);
                f4 q2 = *reinterpret_cast<const f4*>(Gs + n - 11), q3 = *reinterpret_cast<const f4*>(Gs + n - 15);
                quad(q0); quad(q1); quad(q2); quad(q3);
                *reinterpret_cast<f4*>(Gs + n - 3) = q0; *reinterpret_cast<f4*>(Gs + n - 7) = q1;
                *reinterpret_cast<f4*>(Gs + n - 11) = q2; *reinterpret_cast<f4*>(Gs + n - 15) = q3;
            }
            for (; n >= 3; n -= 4) {
                f4 q = *reinterpret_cast<const f4*>(Gs + n - 3);
                quad(q);
                *reinterpret_cast<f4*>(Gs + n - 3) = q;
            }
        } else if (tid >= 64) {
            stage_weights(tid - 64, 192);
        }
        __syncthreads();
        float part = 0.0f;
        for (int n = tid; n < N; n += 256) part += Gs[n];
        const float mean = block_sum(part, red) / (float)N;
        part = 0.0f;
        for (int n = tid; n < N; n += 256) { const float d = Gs[n] - mean; part += d * d; }
        const float sd = sqrtf(block_sum(part, red) / (float)(N - 1));      // torch.std: unbiased
        for (int n = tid; n < N; n += 256) Gs[n] = (Gs[n] - mean) / sd;
        __syncthreads();
    }
    {
        // ---- State dedupe.  In a noise-free game the price takes one value per pair of actions, so
        // the n transitions visit few DISTINCT states (two agents on the same grid: 41).  The forward
        // pass depends on the state only, and every gradient is linear in d loss/d logits, so the
        // transitions of a state are folded first:
        //   sum_{n in u} dz_n[k] = (p_u[k]*SG_u - SGA_u[k] + cnt_u*ent*p_u[k]*(log p_u[k] + H_u)) / N
        // with SG_u = sum of returns, SGA_u[k] = sum of returns of the transitions that took action k.
        // Passes A and B then run over the U distinct states instead of the n transitions.  The sums
        // are accumulated in 2^-40 fixed point with integer LDS atomics: order independent, so the
        // update stays deterministic.  The states are taken kUmax at a time; more than kUfold distinct
        // states: the plain path.
        // ActorCritic: the states AND the next states of the batch are folded together (v is needed at both).
        {
            constexpr int kOwn1 = (THRL_NN_MAX_TRANSITIONS + 255) / 256;
            constexpr int kOwn = AC ? 2 * kOwn1 : kOwn1;        // values per thread: x of transition tid + 256 q, then x'
            float xq[kOwn];
            unsigned open_mask = 0u;
#pragma unroll
            for (int q2 = 0; q2 < kOwn; q2++) {
                const int n = tid + 256 * (q2 % kOwn1);
                xq[q2] = n < N ? (q2 < kOwn1 ? xs[n] : xps[n]) : 0.0f;
                if (n < N) open_mask |= 1u << q2;
            }
            for (int k = tid; k < (AC ? kXuAc : kXu); k += 256) xu[k] = 0.0f;
            // distinct states by open addressing in a kHash-slot LDS table (key = the float32 state's
            // bits), then numbered by ascending key so the numbering -- and with it the order of
            // every later sum -- does not depend on which thread won which slot
            constexpr unsigned kEmpty = 0xFFFFFFFFu;
            constexpr int kParts = kHash / 256;
#pragma unroll
            for (int r = 0; r < kParts; r++) hkeys[tid + 256 * r] = kEmpty;
            if (tid < 8) redi[tid] = 0;
            __syncthreads();
            int myslot[kOwn];
            bool lost = false;
#pragma unroll
            for (int q2 = 0; q2 < kOwn; q2++) {
                myslot[q2] = 0;
                if ((open_mask >> q2) & 1u) {
                    const unsigned bits = __float_as_uint(xq[q2]);
                    unsigned h = (bits * 2654435761u) >> 22;
                    int probe = 0;
                    for (; probe < kHash; probe++) {
                        if (redi[5] > kUhash) { probe = kHash; break; }      // too many states already: no folding
                        const unsigned old = atomicCAS(&hkeys[h], kEmpty, bits);
                        if (old == kEmpty) atomicAdd(&redi[5], 1);
                        if (old == kEmpty || old == bits) break;
                        h = (h + 1) & (kHash - 1);
                    }
                    lost |= probe == kHash;
                    myslot[q2] = (int)h;
                }
            }
            __syncthreads();
            unsigned key[kParts];
            unsigned long long occ[kParts];
#pragma unroll
            for (int r = 0; r < kParts; r++) {
                key[r] = hkeys[tid + 256 * r];
                occ[r] = __ballot(key[r] != kEmpty);
                if ((tid & 63) == 0) ucnt[4 * r + (tid >> 6)] = __popcll(occ[r]);
            }
            if (lost) redi[4] = 1;
            __syncthreads();
            // exclusive prefix of the 4 kParts per-wave counts: lane l < 16 holds count l, scanned across the lanes
            int n_states, before[kParts];
            {
                const int lane = tid & 63;
                int c = lane < 4 * kParts ? ucnt[lane] : 0, inc = c;
#pragma unroll
                for (int d = 1; d < 4 * kParts; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
                n_states = __shfl(inc, 4 * kParts - 1, 64);
#pragma unroll
                for (int r = 0; r < kParts; r++) before[r] = __shfl(inc - c, 4 * r + (tid >> 6), 64);
            }
            // (the folded update costs O((states + 256) * A): it always pays)
            U = (redi[4] != 0 || n_states > kUhash || (AC && (A >= kPad || NX > 1024))) ? 0 : n_states;   // (AC: the value head takes column A; the accumulators overlay Gs beyond 1,024 transitions)
            if (U > 0) {
                // keys packed densely (in slot order), then ranked by value over the U of them
                unsigned* dense = reinterpret_cast<unsigned*>(dz);
                const unsigned long long lt = (1ull << (tid & 63)) - 1ull;
                unsigned short* rankd = reinterpret_cast<unsigned short*>(dense + 1024);      // [kUfold] rank of packed key i
                int di[kParts];
#pragma unroll
                for (int r = 0; r < kParts; r++) {
                    di[r] = before[r] + __popcll(occ[r] & lt);
                    if (key[r] != kEmpty) dense[di[r]] = key[r];
                }
                {
                    // rank of packed key i = the number of keys below it: thread i (and i + 256) counts over the U keys, every read a
                    // broadcast, eight in flight.  (A bitonic sort in LDS was three times slower at 441 keys: 45 dependent
                    // LDS round trips; with 41 keys only the first wave has work.)
                    __syncthreads();
                    const bool two = tid + 256 < U;
                    const unsigned mine0 = tid < U ? dense[tid] : 0u, mine1 = two ? dense[tid + 256] : 0u;
                    int rank0 = 0, rank1 = 0;
                    if (tid < U) {
                        int j2 = 0;
                        for (; j2 + 8 <= U; j2 += 8) {
                            unsigned kj[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) kj[u] = dense[j2 + u];
#pragma unroll
                            for (int u = 0; u < 8; u++) { rank0 += kj[u] < mine0 ? 1 : 0; rank1 += kj[u] < mine1 ? 1 : 0; }
                        }
                        for (; j2 < U; j2++) { const unsigned kj = dense[j2]; rank0 += kj < mine0 ? 1 : 0; rank1 += kj < mine1 ? 1 : 0; }
                        rankd[tid] = (unsigned short)rank0;
                        xu[rank0] = __uint_as_float(mine0);
                        if (two) { rankd[tid + 256] = (unsigned short)rank1; xu[rank1] = __uint_as_float(mine1); }
                    }
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < kParts; r++)
                        if (key[r] != kEmpty) hrank[tid + 256 * r] = rankd[di[r]];
                }
                __syncthreads();
#pragma unroll
                for (int q2 = 0; q2 < kOwn; q2++)
                    if ((open_mask >> q2) & 1u) (q2 < kOwn1 ? uid : uidp)[tid + 256 * (q2 % kOwn1)] = hrank[myslot[q2]];
            }
            if (!AC && U == 0 && N <= kUfold) {
                // More distinct states than the table folds (a game with env noise: continuous prices).  The piecewise-linear
                // update does not need folding, only the order: every transition becomes its own state, ranked by (price, index).
                __syncthreads();                            // (the hash table inside dz is dead)
                unsigned* dense = reinterpret_cast<unsigned*>(dz);
#pragma unroll
                for (int q2 = 0; q2 < kOwn; q2++)
                    if ((open_mask >> q2) & 1u) dense[tid + 256 * q2] = __float_as_uint(xq[q2]);
                __syncthreads();
                int rank[kOwn];
#pragma unroll
                for (int q2 = 0; q2 < kOwn; q2++) rank[q2] = 0;
                for (int j2 = 0; j2 < N; j2 += 8) {
                    unsigned kj[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) kj[u] = dense[min(j2 + u, N - 1)];
#pragma unroll
                    for (int u = 0; u < 8; u++)
#pragma unroll
                        for (int q2 = 0; q2 < kOwn; q2++) {
                            const unsigned mine = __float_as_uint(xq[q2]);
                            rank[q2] += (j2 + u < N && (kj[u] < mine || (kj[u] == mine && j2 + u < tid + 256 * q2))) ? 1 : 0;
                        }
                }
#pragma unroll
                for (int q2 = 0; q2 < kOwn; q2++)
                    if ((open_mask >> q2) & 1u) { uid[tid + 256 * q2] = (unsigned short)rank[q2]; xu[rank[q2]] = xq[q2]; }
                U = N;
            }
        }
    }
    if (AC && U == 0) {
        // v(s), v(s') per transition (agents.py:287-288), c_i = gamma*v'_i - v_i
        const float bv = w[Pp + kH];
        float cpart = 0.0f, rpart = 0.0f;
        for (int n = tid; n < N; n += 256) {
            const float x = xs[n], xp = xps[n];
            float v = bv, vp = bv;
            for (int j = 0; j < kH; j++) {
                const float w1 = w1s[j], b1 = b1s[j], wv = wvs[j];
                v = __fmaf_rn(wv, fmaxf(__fmaf_rn(w1, x, b1), 0.0f), v);
                vp = __fmaf_rn(wv, fmaxf(__fmaf_rn(w1, xp, b1), 0.0f), vp);
            }
            const float c = gamma * vp - v;
            gvs[n] = c; cpart += c; rpart += Gs[n];
        }
        const float C = block_sum(cpart, red), R = block_sum(rpart, red);
        const float fN = (float)N;
        float gpart = 0.0f;
        for (int n = tid; n < N; n += 256) {
            Gs[n] = Gs[n] + C / fN;                                       // column sums of adv / N
            const float gv = -(2.0f / (fN * fN)) * (R + fN * gvs[n]);      // - row sums of adv * 2/N^2
            gvs[n] = gv; gpart += gv;
        }
        gbv = (1.0f - gamma) * block_sum(gpart, red);                     // sum_i (gv_i + gv'_i), gv' = -gamma*gv
        wva = wvs[tid & 127]; wvb = wvs[(tid & 127) + 128];
    }
    __syncthreads();

    const float invN = 1.0f / (float)N;
    float* comb = reinterpret_cast<float*>(smem_nn);        // [128][kPad * 2 + 4], over W2t / dz (once they are no longer needed)
    constexpr int kRow = kPad * 2 + 6;
    float* gl = comb + 128 * kRow + 256;                    // [P] unscaled gradient for the Adam sweep (behind the combine scratch)
    float sq = 0.0f;                                        // this thread's share of |gradient|^2
    if (U > 0) {
        // ---- Folded update on the PIECEWISE-LINEAR form of the network.  The policy has ONE input (the price, agents.py:127-133:
        // Linear(1, 256) -> relu -> Linear(256, A)), so unit j is active on a half line of prices: with the distinct states
        // sorted (xu ascending) its active states are a suffix (w1 > 0) or a prefix (w1 < 0) starting / ending at a threshold
        // index t_j -- found by binary search with the SAME test the direct evaluation uses (fma(w1, x, b1) > 0, monotonic in x).
        //   forward   z_k(x_s) = b2_k + x_s * A_k(s) + B_k(s),  A_k(s) = sum over the units active at s of W2[k][j] * w1_j,
        //             B_k likewise with b1_j: every unit adds its 2 x A terms ONCE, at its threshold, into per-state buckets;
        //             a running sum over the sorted states gives A, B per state.
        //   backward  d W2[k][j] = w1_j * S1_j[k] + b1_j * S0_j[k],  d w1_j = sum_k W2[k][j] S1_j[k],  d b1_j = sum_k W2[k][j] S0_j[k],
        //             S0_j[k] = sum of d_s[k] over j's active states, S1_j[k] = the same of d_s[k] * x_s: range sums of two
        //             prefix arrays over the sorted states.
        // O((U + 256) * A) operations instead of O(U * 256 * A).  All sums are integers in 2^-40 fixed point (LDS integer
        // atomics: order independent, deterministic; range sums are exact differences), combined in float64 at the end.
        // ActorCritic (agents.py:222-305): the value head fc_v is one more output column (column A of W2t), v(x) is piecewise linear
        // in the price like the logits, and its "d loss / d logits" per state is the sum of d loss / d v_i over the transitions
        // that start there plus d loss / d v'_i over those that end there -- the rest is the same machinery.
        constexpr double kUnfix = 0x1p-40;
        typedef unsigned long long u64;
        const int Acols = A + (AC ? 1 : 0);
        long long* EA = reinterpret_cast<long long*>(dz);   // [kUmax][kPad]  buckets -> A per state -> d -> prefix of d
        long long* EB = EA + kUmax * kPad;                  // [kUmax][kPad]  ... B ... d * x
        static_assert(kChunk * kPad * sizeof(float) == 2 * kUmax * kPad * sizeof(long long), "EA / EB overlay dz exactly");
        const float w1 = w1s[tid], b1 = b1s[tid];           // this thread's hidden unit: j = tid
        const float* wr = W2t + tid * kPad;                 // this unit's row of fc_pi^T (stays in LDS: the staging area starts behind it)
        const bool pos = !(w1 < 0.0f);                      // active states: [t, U) if pos, [0, t) otherwise
        int t;
        {
            int lo = 0, hi = U;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const bool act = __fmaf_rn(w1, xu[mid], b1) > 0.0f;
                if (act == pos) hi = mid; else lo = mid + 1;
            }
            t = lo;
        }
        constexpr int kBaseParts = 256 / kPad;
        int* in_base = reinterpret_cast<int*>(xs);          // [kH] (the per-transition states are no longer needed: uid has their index)
        in_base[tid] = (pos && t == 0) || (!pos && t > 0);
        // The units that are active from the first state on (suffix units below the range, every prefix unit that is active at
        // all) start the running sums: about half of all units, so they are summed by (action, part) threads -- in a fixed
        // order in float64: deterministic, one conversion -- instead of 256 x 2A atomics on the same 2A words.  (Done here,
        // before the per-unit accumulators exist: the thirteen-wide load batches need the registers.)
        long long base_a = 0, base_b = 0;
        __syncthreads();
        if (tid < kBaseParts * kPad) {
            const int k = tid % kPad, part = tid / kPad;
            double sa = 0.0, sb = 0.0;
            if (k < Acols) {
                constexpr int kIt = (kH + kBaseParts - 1) / kBaseParts;
#pragma unroll
                for (int i0 = 0; i0 < kIt; i0 += 13) {              // thirteen units' operands in flight per round trip
                    float wv[13], w1v[13], b1v[13];
#pragma unroll
                    for (int u = 0; u < 13; u++) {
                        const int j = min(part + (i0 + u) * kBaseParts, kH - 1);
                        const bool on = i0 + u < kIt && part + (i0 + u) * kBaseParts < kH && in_base[j] != 0;
                        wv[u] = on ? W2t[j * kPad + k] : 0.0f; w1v[u] = w1s[j]; b1v[u] = b1s[j];
                    }
#pragma unroll
                    for (int u = 0; u < 13; u++) { sa = fma((double)wv[u], (double)w1v[u], sa); sb = fma((double)wv[u], (double)b1v[u], sb); }
                }
            }
            base_a = fix40(sa); base_b = fix40(sb);
        }
        long long S0[kPad], S1[kPad];
#pragma unroll
        for (int k = 0; k < kPad; k++) { S0[k] = 0; S1[k] = 0; }
        long long run = 0, gb2i = 0;                        // threads < 2 kPad: running A_k / B_k; threads < kPad: sum of d_s[k]
        constexpr int kQ = kPad / 4;
        auto quad_max = [](float v) {
            v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
            return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
        };
        auto quad_sum = [](float v) {
            v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
            return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
        };
        // this thread's transitions: state index, action, return (ActorCritic: weight r + C / N) in fixed point -- read once,
        // used by every chunk
        constexpr int kOwnT = (THRL_NN_MAX_TRANSITIONS + 255) / 256;
        int t_uid[kOwnT], t_act[kOwnT];
        long long t_ret[kOwnT];
#pragma unroll
        for (int q2 = 0; q2 < kOwnT; q2++) {
            const int n = tid + 256 * q2;
            t_uid[q2] = n < N ? (int)uid[n] : -1;
            t_act[q2] = n < N ? action[n] : 0;
            t_ret[q2] = (!AC && n < N) ? fix40((double)Gs[n]) : 0;
        }
        if (AC) {
            // v(x) of every distinct state first (c_i, C, R -- and with them every transition's weight -- need all of them): the
            // running sums of the value column alone, all states at once
            long long* EvA = reinterpret_cast<long long*>(dz);      // [512]  buckets -> A_v per state
            long long* EvB = EvA + 512;                             // [512]
            for (int k = tid; k < 1024; k += 256) EvA[k] = 0;
            __syncthreads();
            if (t > 0 && t < U) {
                const long long ia = fix40((double)wr[A] * (double)w1), ib = fix40((double)wr[A] * (double)b1);
                atomicAdd(reinterpret_cast<u64*>(&EvA[t]), (u64)(pos ? ia : -ia));
                atomicAdd(reinterpret_cast<u64*>(&EvB[t]), (u64)(pos ? ib : -ib));
            }
            if (tid < kBaseParts * kPad && tid % kPad == A) {
                atomicAdd(reinterpret_cast<u64*>(&EvA[0]), (u64)base_a);
                atomicAdd(reinterpret_cast<u64*>(&EvB[0]), (u64)base_b);
            }
            __syncthreads();
            if (tid < 64) {                                         // lane l: states 8 l .. 8 l + 7
                const float bv = w[Pp + kH];
                long long va[8], vb[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { va[u] = EvA[8 * tid + u]; vb[u] = EvB[8 * tid + u]; }
#pragma unroll
                for (int u = 1; u < 8; u++) { va[u] += va[u - 1]; vb[u] += vb[u - 1]; }
                long long ia = va[7], ib = vb[7];                   // inclusive scan of the lanes' totals
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const long long oa = __shfl_up(ia, d, 64), ob = __shfl_up(ib, d, 64);
                    if (tid >= d) { ia += oa; ib += ob; }
                }
                const long long offa = ia - va[7], offb = ib - vb[7];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int sidx = 8 * tid + u;
                    if (sidx < U)
                        vstate[sidx] = (float)(fma((double)(va[u] + offa) * kUnfix, (double)xu[sidx], (double)(vb[u] + offb) * kUnfix) + (double)bv);
                }
            }
            for (int k = tid; k < kUhash; k += 256) gvfix[k] = 0;
            __syncthreads();
            // c_i = gamma v'_i - v_i, C, R; weight r_j + C / N; d loss / d v_i = -(2 / N^2)(R + N c_i), d loss / d v'_i = -gamma times it
            float c_i[kOwnT], r_i[kOwnT];
            float cpart = 0.0f, rpart = 0.0f;
#pragma unroll
            for (int q2 = 0; q2 < kOwnT; q2++) {
                const int n = tid + 256 * q2;
                c_i[q2] = 0.0f; r_i[q2] = 0.0f;
                if (n < N) {
                    const float v = vstate[uid[n]], vp = vstate[uidp[n]];
                    c_i[q2] = gamma * vp - v; r_i[q2] = Gs[n];
                    cpart += c_i[q2]; rpart += r_i[q2];
                }
            }
            const float C = block_sum(cpart, red), R = block_sum(rpart, red);
            const float fN = (float)N;
#pragma unroll
            for (int q2 = 0; q2 < kOwnT; q2++) {
                const int n = tid + 256 * q2;
                if (n < N) {
                    const float wt = r_i[q2] + C / fN;                                 // column sums of adv / N
                    const float gv = -(2.0f / (fN * fN)) * (R + fN * c_i[q2]);          // - row sums of adv * 2/N^2
                    const float gvp = -gamma * gv;
                    t_ret[q2] = fix40((double)wt);
                    atomicAdd(reinterpret_cast<u64*>(&gvfix[uid[n]]), (u64)fix40((double)gv));
                    atomicAdd(reinterpret_cast<u64*>(&gvfix[uidp[n]]), (u64)fix40((double)gvp));
                }
            }
        }
        __syncthreads();                                    // (ActorCritic: the accumulators below overlay arrays read above)
        for (int c0 = 0; c0 < U; c0 += kUmax) {
            const int cn = min(kUmax, U - c0);
            for (int k = tid; k < 2 * kUmax * kPad; k += 256) EA[k] = 0;
            for (int k = tid; k < kUmax * kPad; k += 256) sga[k] = 0;
            if (tid < kUmax) ucnt[tid] = 0;
            __syncthreads();
            {   // a unit whose threshold lies strictly inside the state range enters (or leaves) the running sums there: few
                // units per bucket.  (The operands pass through an empty asm so that the 2 x A conversions are not hoisted out of
                // the chunk loop as loop invariants: 96 registers.)
                const int rel = t - c0;
                if (t > 0 && t < U && rel >= 0 && rel < kUmax) {
                    float w1v = w1, b1v = b1;
                    asm volatile("" : "+v"(w1v), "+v"(b1v));
#pragma unroll
                    for (int k = 0; k < kPad; k++)
                        if (k < Acols) {
                            const long long ia = fix40((double)wr[k] * (double)w1v), ib = fix40((double)wr[k] * (double)b1v);
                            atomicAdd(reinterpret_cast<u64*>(&EA[rel * kPad + k]), (u64)(pos ? ia : -ia));
                            atomicAdd(reinterpret_cast<u64*>(&EB[rel * kPad + k]), (u64)(pos ? ib : -ib));
                        }
                }
            }
            if (c0 == 0 && tid < kBaseParts * kPad) {           // the units active from the first state on (summed above)
                atomicAdd(reinterpret_cast<u64*>(&EA[tid % kPad]), (u64)base_a);
                atomicAdd(reinterpret_cast<u64*>(&EB[tid % kPad]), (u64)base_b);
            }
            // returns of this chunk's states by (state, action), and their transition counts
#pragma unroll
            for (int q2 = 0; q2 < kOwnT; q2++) {
                const int u = t_uid[q2] - c0;
                if (t_uid[q2] >= 0 && u >= 0 && u < cn) {
                    atomicAdd(reinterpret_cast<u64*>(&sga[u * kPad + t_act[q2]]), (u64)t_ret[q2]);
                    atomicAdd(&ucnt[u], 1);
                }
            }
            __syncthreads();
            // running sums over the sorted states, A_k(s) and B_k(s) in place: FOUR threads per column, a quarter of the states
            // each (one batch of 16 loads in flight instead of four in sequence), quarter totals exchanged inside the quad
            auto column_prefix = [&](long long carry_in, long long& carry_out) {
                const int col = tid >> 2, qtr = tid & 3;            // col < 2 kPad: EA columns, then EB columns
                if (col < 2 * kPad) {
                    long long* arr = (col < kPad ? EA : EB) + (col < kPad ? col : col - kPad) + qtr * 16 * kPad;
                    long long v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) v[u] = arr[u * kPad];
#pragma unroll
                    for (int u = 1; u < 16; u++) v[u] += v[u - 1];
                    const long long tot = v[15];
                    const long long t1 = __shfl_up(tot, 1, 4), t2 = __shfl_up(tot, 2, 4), t3 = __shfl_up(tot, 3, 4);
                    const long long off = carry_in + (qtr > 0 ? t1 : 0) + (qtr > 1 ? t2 : 0) + (qtr > 2 ? t3 : 0);
#pragma unroll
                    for (int u = 0; u < 16; u++) arr[u * kPad] = v[u] + off;
                    carry_out = __shfl(v[15] + off, 3, 4);          // the column's last value, in all four threads
                }
            };
            column_prefix(run, run);
            __syncthreads();
            // softmax, entropy, d loss / d logits folded over the transitions of each state: four threads per state
            if ((tid >> 6) * 16 >= cn) {
                // (this wave's sixteen states are all beyond the chunk: their rows enter the prefix as zeros)
                const int st = tid >> 2, part = tid & 3;
#pragma unroll
                for (int i = 0; i < kQ; i++) { EA[st * kPad + kQ * part + i] = 0; EB[st * kPad + kQ * part + i] = 0; }
            } else {
                const int st = tid >> 2, part = tid & 3;
                const bool live = st < cn;
                const double x = (double)xu[c0 + st];
                float zz[kQ], lp[kQ];
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < kQ; i++) {
                    const int k = kQ * part + i;
                    zz[i] = 0.0f;
                    if (k < A) {
                        zz[i] = (float)(fma((double)EA[st * kPad + k] * kUnfix, x, (double)EB[st * kPad + k] * kUnfix) + (double)b2s[k]);
                        m = fmaxf(m, zz[i]);
                    }
                }
                m = quad_max(m);
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < kQ; i++) if (kQ * part + i < A) { zz[i] = expf(zz[i] - m); sum += zz[i]; }
                sum = quad_sum(sum);
                float Hn = 0.0f;
#pragma unroll
                for (int i = 0; i < kQ; i++) {
                    lp[i] = 0.0f;
                    if (kQ * part + i < A) {
                        zz[i] = zz[i] / sum;
                        lp[i] = logf(fminf(fmaxf(zz[i], 1.1920929e-07f), 1.0f - 1.1920929e-07f));
                        Hn -= zz[i] * lp[i];
                    }
                }
                Hn = quad_sum(Hn);
                const long long* row = sga + st * kPad + kQ * part;
                long long rk[kQ], sg = 0;
#pragma unroll
                for (int i = 0; i < kQ; i++) { rk[i] = kQ * part + i < A ? row[i] : 0; sg += rk[i]; }
                sg += __shfl_xor(sg, 1, 64);
                sg += __shfl_xor(sg, 2, 64);
                const float SG = (float)((double)sg * kUnfix), cnt = (float)ucnt[st];
#pragma unroll
                for (int i = 0; i < kQ; i++) {
                    const int k = kQ * part + i;
                    const float d = (k < A && live) ? (zz[i] * SG - (float)((double)rk[i] * kUnfix) + cnt * (ent_coef * zz[i] * (lp[i] + Hn))) * invN : 0.0f;
                    long long d0 = fix40((double)d), d1 = fix40((double)d * x);
                    if (AC && k == A) {                                 // the value column: d loss / d v summed over the state's transitions
                        d0 = live ? gvfix[c0 + st] : 0;
                        d1 = fix40((double)d0 * kUnfix * x);
                    }
                    EA[st * kPad + k] = d0;
                    EB[st * kPad + k] = d1;
                }
            }
            __syncthreads();
            {   // inclusive prefix of d_s[k] (EA) and d_s[k] * x_s (EB) over the chunk (rows >= cn hold zeros)
                long long last = 0;
                column_prefix(0, last);
                if ((tid & 3) == 0 && (tid >> 2) < kPad) gb2i += last;      // sum of d_s[k] over the chunk: d b2
            }
            __syncthreads();
            {   // this unit's range sums over the chunk
                const int lo_s = pos ? max(t - c0, 0) : 0, hi_s = pos ? cn : min(t - c0, cn);
                if (lo_s < hi_s) {
                    const long long* h0 = EA + (hi_s - 1) * kPad;
                    const long long* h1 = EB + (hi_s - 1) * kPad;
                    const long long* l0 = EA + (lo_s > 0 ? lo_s - 1 : 0) * kPad;
                    const long long* l1 = EB + (lo_s > 0 ? lo_s - 1 : 0) * kPad;
#pragma unroll
                    for (int k = 0; k < kPad; k++)
                        if (k < Acols) {
                            S0[k] += h0[k] - (lo_s > 0 ? l0[k] : 0);
                            S1[k] += h1[k] - (lo_s > 0 ? l1[k] : 0);
                        }
                }
            }
            __syncthreads();
        }
        // the gradient of this unit's parameters, straight into the sweep's staging area (EA / EB and W2t are dead: synced above)
        double dw1 = 0.0, db1 = 0.0;
#pragma unroll
        for (int k = 0; k < kPad; k++)
            if (k < Acols) {
                const double s0 = (double)S0[k] * kUnfix, s1 = (double)S1[k] * kUnfix;
                const float gw = (float)((double)w1 * s1 + (double)b1 * s0);
                if (k < A) gl[2 * kH + k * kH + tid] = gw;              // fc_pi.weight[k][j]
                else gl[Pp + tid] = gw;                                 // fc_v.weight[j]
                sq += gw * gw;
                dw1 += (double)wr[k] * s1; db1 += (double)wr[k] * s0;
            }
        const float gw1 = (float)dw1, gb1 = (float)db1;
        gl[tid] = gw1; gl[kH + tid] = gb1;
        sq += gw1 * gw1 + gb1 * gb1;
        if ((tid & 3) == 0 && (tid >> 2) < Acols) {
            const int col = tid >> 2;
            const float gb2 = (float)((double)gb2i * kUnfix);
            if (col < A) gl[2 * kH + A * kH + col] = gb2;               // fc_pi.bias
            else gl[Pp + kH] = gb2;                                     // fc_v.bias: sum_i (d loss / d v_i + d loss / d v'_i)
            sq += gb2 * gb2;
        }
    } else {
        // pass B ownership: hidden units ja = tid & 127 and jb = ja + 128, transitions of parity tid >> 7
        const int ja = tid & 127, jb = ja + 128, half = tid >> 7;
        f2 gWa[kPad / 2], gWb[kPad / 2], ca[kPad / 2], cb[kPad / 2];
#pragma unroll
        for (int p = 0; p < kPad / 2; p++) {
            gWa[p] = f2{0.0f, 0.0f}; gWb[p] = f2{0.0f, 0.0f};
            ca[p] = *reinterpret_cast<const f2*>(W2t + ja * kPad + 2 * p);
            cb[p] = *reinterpret_cast<const f2*>(W2t + jb * kPad + 2 * p);
        }
        float gw1a = 0.0f, gb1a = 0.0f, gw1b = 0.0f, gb1b = 0.0f, gb2 = 0.0f, gwva = 0.0f, gwvb = 0.0f;
        const float w1a = w1s[ja], b1a = b1s[ja], w1b = w1s[jb], b1b = b1s[jb];
        // pass A1 ownership: transitions 4q..4q+3 of the chunk, actions kPad/4 * kg .. (6 or 8 of them)
        constexpr int kGp = kPad / 8;                           // action pairs per thread
        const int q = tid >> 2, kg = tid & 3;

        const float* xa = xs;
        for (int c0 = 0; c0 < N; c0 += kChunk) {                // 256 transitions per chunk
            const int cn = min(kChunk, N - c0);
            {   // ---- pass A1: logits (without bias) into dz: thread = (transition quad tid>>2 of 64, action group tid&3), all hidden units
                const f4 x4 = *reinterpret_cast<const f4*>(xa + c0 + 4 * q);
                f2 za[kGp], zb[kGp], zc[kGp], zd[kGp];          // transitions 0..3 of the quad, kGp action pairs each
#pragma unroll
                for (int p = 0; p < kGp; p++) { za[p] = zb[p] = zc[p] = zd[p] = f2{0.0f, 0.0f}; }
                const float* wrow = W2t + 2 * kGp * kg;
#pragma unroll 4
                for (int j = 0; j < kH; j++) {
                    const float w1 = w1s[j], b1 = b1s[j];
                    const float h0 = fmaxf(__fmaf_rn(w1, x4.x, b1), 0.0f), h1 = fmaxf(__fmaf_rn(w1, x4.y, b1), 0.0f);
                    const float h2 = fmaxf(__fmaf_rn(w1, x4.z, b1), 0.0f), h3 = fmaxf(__fmaf_rn(w1, x4.w, b1), 0.0f);
#pragma unroll
                    for (int p = 0; p < kGp; p++) {
                        const f2 wv = *reinterpret_cast<const f2*>(wrow + j * kPad + 2 * p);
                        za[p] = pk_fma(wv, f2{h0, h0}, za[p]); zb[p] = pk_fma(wv, f2{h1, h1}, zb[p]);
                        zc[p] = pk_fma(wv, f2{h2, h2}, zc[p]); zd[p] = pk_fma(wv, f2{h3, h3}, zd[p]);
                    }
                }
                float* o = dz + (4 * q) * kPad + 2 * kGp * kg;
#pragma unroll
                for (int p = 0; p < kGp; p++) {
                    *reinterpret_cast<f2*>(o + 2 * p) = za[p];
                    *reinterpret_cast<f2*>(o + kPad + 2 * p) = zb[p];
                    *reinterpret_cast<f2*>(o + 2 * kPad + 2 * p) = zc[p];
                    *reinterpret_cast<f2*>(o + 3 * kPad + 2 * p) = zd[p];
                }
            }
            __syncthreads();
            // ---- pass A2 (thread = transition): softmax, entropy, d loss / d logits (in place over the logits)
            if (tid < cn) {
                const int n = c0 + tid;
                float zz[kPad];
#pragma unroll
                for (int k4 = 0; k4 < kPad / 4; k4++) {
                    const f4 v = *reinterpret_cast<const f4*>(dz + tid * kPad + 4 * k4);
                    zz[4 * k4] = v.x; zz[4 * k4 + 1] = v.y; zz[4 * k4 + 2] = v.z; zz[4 * k4 + 3] = v.w;
                }
                float m = -INFINITY;
#pragma unroll
                for (int k = 0; k < kPad; k++) if (k < A) { zz[k] += b2s[k]; m = fmaxf(m, zz[k]); }
                float sum = 0.0f;
#pragma unroll
                for (int k = 0; k < kPad; k++) if (k < A) { zz[k] = expf(zz[k] - m); sum += zz[k]; }
                float Hn = 0.0f;
                float lp[kPad];
#pragma unroll
                for (int k = 0; k < kPad; k++)
                    if (k < A) {
                        zz[k] = zz[k] / sum;
                        lp[k] = logf(fminf(fmaxf(zz[k], 1.1920929e-07f), 1.0f - 1.1920929e-07f));
                        Hn -= zz[k] * lp[k];
                    }
                const int a_n = action[n];
                const float Gn = Gs[n];
#pragma unroll
                for (int k = 0; k < kPad; k++)
                    zz[k] = k < A ? (Gn * (zz[k] - (k == a_n ? 1.0f : 0.0f)) + ent_coef * zz[k] * (lp[k] + Hn)) * invN : 0.0f;
#pragma unroll
                for (int k4 = 0; k4 < kPad / 4; k4++)
                    *reinterpret_cast<f4*>(dz + tid * kPad + 4 * k4) = f4{zz[4 * k4], zz[4 * k4 + 1], zz[4 * k4 + 2], zz[4 * k4 + 3]};
            }
            __syncthreads();
            // ---- pass B: fc_pi.weight[:, j], fc1.weight[j], fc1.bias[j] for j in {ja, jb}
            for (int i = half; i < cn; i += 2) {
                const float x = xa[c0 + i];
                const float pa = __fmaf_rn(w1a, x, b1a), pb = __fmaf_rn(w1b, x, b1b);
                const float ha = fmaxf(pa, 0.0f), hb = fmaxf(pb, 0.0f);
                f2 d[kPad / 2];
#pragma unroll
                for (int k4 = 0; k4 < kPad / 4; k4++) {
                    const f4 v = *reinterpret_cast<const f4*>(dz + i * kPad + 4 * k4);
                    d[2 * k4] = f2{v.x, v.y}; d[2 * k4 + 1] = f2{v.z, v.w};
                }
                f2 da = f2{0.0f, 0.0f}, db = f2{0.0f, 0.0f};
#pragma unroll
                for (int p = 0; p < kPad / 2; p++) {
                    gWa[p] = pk_fma(d[p], f2{ha, ha}, gWa[p]);
                    gWb[p] = pk_fma(d[p], f2{hb, hb}, gWb[p]);
                    da = pk_fma(ca[p], d[p], da);
                    db = pk_fma(cb[p], d[p], db);
                }
                float dha = da.x + da.y, dhb = db.x + db.y;
                if (AC) {
                    const float gv = gvs[c0 + i], gvp = -gamma * gv, xp = xps[c0 + i];
                    const float qa = __fmaf_rn(w1a, xp, b1a), qb = __fmaf_rn(w1b, xp, b1b);
                    gwva = __fmaf_rn(gv, ha, gwva); gwva = __fmaf_rn(gvp, fmaxf(qa, 0.0f), gwva);
                    gwvb = __fmaf_rn(gv, hb, gwvb); gwvb = __fmaf_rn(gvp, fmaxf(qb, 0.0f), gwvb);
                    dha = __fmaf_rn(gv, wva, dha); dhb = __fmaf_rn(gv, wvb, dhb);
                    const float ea = gvp * wva, eb = gvp * wvb;               // d loss / d h' through fc_v
                    if (qa > 0.0f) { gw1a = __fmaf_rn(ea, xp, gw1a); gb1a += ea; }
                    if (qb > 0.0f) { gw1b = __fmaf_rn(eb, xp, gw1b); gb1b += eb; }
                }
                if (pa > 0.0f) { gw1a = __fmaf_rn(dha, x, gw1a); gb1a += dha; }
                if (pb > 0.0f) { gw1b = __fmaf_rn(dhb, x, gw1b); gb1b += dhb; }
            }
            {   // fc_pi.bias: thread (k = tid & 31, part = tid >> 5) sums every 8th row of column k
                const int k = tid & 31;
                if (k < A) for (int i = tid >> 5; i < cn; i += 8) gb2 += dz[i * kPad + k];
            }
            __syncthreads();
        }

        // ---- add the two transition halves (and the 8 parts of gb2) through LDS
        if (half == 1) {
            float* o = comb + ja * kRow;
#pragma unroll
            for (int p = 0; p < kPad / 2; p++) {
                *reinterpret_cast<f2*>(o + 2 * p) = gWa[p];
                *reinterpret_cast<f2*>(o + kPad + 2 * p) = gWb[p];
            }
            o[2 * kPad] = gw1a; o[2 * kPad + 1] = gb1a; o[2 * kPad + 2] = gw1b; o[2 * kPad + 3] = gb1b;
            o[2 * kPad + 4] = gwva; o[2 * kPad + 5] = gwvb;
        }
        float* gb2s = comb + 128 * kRow;                        // [8][32]
        gb2s[tid] = gb2;
        __syncthreads();
        if (half == 0) {
            const float* o = comb + ja * kRow;
#pragma unroll
            for (int p = 0; p < kPad / 2; p++) {
                gWa[p] += *reinterpret_cast<const f2*>(o + 2 * p);
                gWb[p] += *reinterpret_cast<const f2*>(o + kPad + 2 * p);
            }
            gw1a += o[2 * kPad]; gb1a += o[2 * kPad + 1]; gw1b += o[2 * kPad + 2]; gb1b += o[2 * kPad + 3];
            gwva += o[2 * kPad + 4]; gwvb += o[2 * kPad + 5];
        }
        if (tid < A) {
            gb2 = 0.0f;
            for (int part = 0; part < 8; part++) gb2 += gb2s[part * 32 + tid];
        }
        // clip_grad_norm_(1.0): this thread's share (agents.py:192)
        if (half == 0) {
            sq = gw1a * gw1a + gb1a * gb1a + gw1b * gw1b + gb1b * gb1b + gwva * gwva + gwvb * gwvb;
#pragma unroll
            for (int p = 0; p < kPad / 2; p++) sq += gWa[p].x * gWa[p].x + gWa[p].y * gWa[p].y + gWb[p].x * gWb[p].x + gWb[p].y * gWb[p].y;
        }
        if (tid < A) sq += gb2 * gb2;
        if (AC && tid == 255) sq += gbv * gbv;
        __syncthreads();                                        // (the combine scratch has been read: the staging area overlays it)
        if (half == 0) {
            gl[ja] = gw1a; gl[jb] = gw1b; gl[kH + ja] = gb1a; gl[kH + jb] = gb1b;
#pragma unroll
            for (int p = 0; p < kPad / 2; p++) {
                if (2 * p < A) { gl[2 * kH + (2 * p) * kH + ja] = gWa[p].x; gl[2 * kH + (2 * p) * kH + jb] = gWb[p].x; }
                if (2 * p + 1 < A) { gl[2 * kH + (2 * p + 1) * kH + ja] = gWa[p].y; gl[2 * kH + (2 * p + 1) * kH + jb] = gWb[p].y; }
            }
            if (AC) { gl[Pp + ja] = gwva; gl[Pp + jb] = gwvb; }
        }
        if (tid < A) gl[2 * kH + A * kH + tid] = gb2;
        if (AC && tid == 255) gl[Pp + kH] = gbv;
    }

    // Adam state of the first sweep iteration: requested here, so the HBM round trip overlaps the norm reduction and the
    // staging of the gradient below
    constexpr int kB = 12;
    constexpr bool kAdamPrefetch = true;
    float* mg = adam_m + (int64_t)g * P;
    float* vg = adam_v + (int64_t)g * P;
    float mm[kB], vv[kB], ww[kB];
    auto load_state = [&](int i0) {
#pragma unroll
        for (int b = 0; b < kB; b++) {
            const int idx = min(i0 + 256 * b, P - 1);
            mm[b] = mg[idx]; vv[b] = vg[idx]; ww[b] = w[idx];
        }
    };
    if (kAdamPrefetch) load_state(tid);

    // clip_grad_norm_(1.0) (agents.py:192)
    const float norm = sqrtf(block_sum(sq, red));
    const float coef = fminf(1.0f, 1.0f / (norm + 1e-6f));

    // Adam (torch.optim.Adam defaults, lr from the caller).  The gradient sits in the registers of the pass-B owners
    // (128 threads x 46 parameters); updating from there is 46 dependent HBM round trips per block -- the phase was
    // ~80 % of the kernel.  So it is staged in LDS and all 256 threads sweep the parameter vector in order:
    // coalesced loads of m, v, w, twelve per thread in flight, two round trips for the 5,909 parameters.
    const float t = (float)(step + 1);
    const float bc1 = 1.0f - powf(0.9f, t), bc2s = sqrtf(1.0f - powf(0.999f, t));
    const float step_size = lr / bc1;
    __syncthreads();
    for (int i0 = tid; i0 < P; i0 += 256 * kB) {
        float gg[kB];
        if (!kAdamPrefetch || i0 != tid) load_state(i0);
#pragma unroll
        for (int b = 0; b < kB; b++) gg[b] = gl[min(i0 + 256 * b, P - 1)];
#pragma unroll
        for (int b = 0; b < kB; b++) {
            const int idx = i0 + 256 * b;
            if (idx < P) {
                const float grad = gg[b] * coef;
                if (grad_out) grad_out[(int64_t)g * P + idx] = grad;
                const float m = 0.9f * mm[b] + 0.1f * grad;
                const float v = 0.999f * vv[b] + 0.001f * grad * grad;
                mg[idx] = m; vg[idx] = v;
                w[idx] = ww[b] - step_size * (m / (sqrtf(v) / bc2s + 1e-8f));
            }
        }
    }
}

// Discounted returns of Reinforce.train_net (agents.py:178-181) for G games, ONE LANE PER GAME: the recurrence
// R_n = r_n + gamma * R_(n+1) must round its product and its sum separately, as the reference does, so it is a serial chain of
// 2 N dependent operations per game -- a quarter of the update kernel's block time when one thread of a 256-thread block runs
// it (profiles/exp_train_stamps.py), nothing when 64 games run it side by side.  A wave takes 64 games; tiles of 64
// transitions are read coalesced (a game's ring row is contiguous), transposed through LDS, chained lane = game from the last
// tile to the first, and written back coalesced as float32.  Same operations in the same order as the in-kernel form.
__global__ void __launch_bounds__(64) k_nn_returns(int G, int N, int ld, const double* __restrict__ reward, float gamma,
                                                    const double* __restrict__ gamma_g, float* __restrict__ out) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x, g0 = blockIdx.x * 64;
    const int gme = min(g0 + lane, G - 1);
    const float gam = gamma_g ? (float)gamma_g[gme] : gamma;
    float carry = 0.0f;
    bool first = true;
    for (int n0 = (N - 1) / 64 * 64; n0 >= 0; n0 -= 64) {
        const int cols = min(64, N - n0);
        for (int r0 = 0; r0 < 64; r0 += 16) {               // row r = game g0 + r, lane = transition n0 + lane
            double v[16];                                   // sixteen row loads in flight per round trip
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = reward[(size_t)min(g0 + r0 + u, G - 1) * ld + n0 + min(lane, cols - 1)];
#pragma unroll
            for (int u = 0; u < 16; u++) tile[r0 + u][lane] = (float)v[u];
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        for (int c = cols - 1; c >= 0; c--) {               // lane = game
            float v = tile[lane][c];
            if (!first) v = __fadd_rn(v, __fmul_rn(gam, carry));
            first = false;
            carry = v;
            tile[lane][c] = v;
        }
        __builtin_amdgcn_wave_barrier();
        for (int r = 0; r < 64; r++)
            if (g0 + r < G && lane < cols) out[(size_t)(g0 + r) * ld + n0 + lane] = tile[r][lane];
        __builtin_amdgcn_wave_barrier();
    }
}

// Philox draws of one lockstep step (same counters as the episode kernels)
__global__ void __launch_bounds__(256) k_op_draws(int G, int N, uint64_t seed, uint64_t game_offset,
        uint32_t episode, uint32_t step, double env_a, double noise_lo, int32_t nA0, int32_t nA1, int32_t nA2,
        int32_t nA3, int32_t nA4, int32_t nA5, int32_t nA6, int32_t nA7, double* u_out, int8_t* choice_out,
        double* u2_out, double* noise_u_out, double* noise_a_out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int32_t nA[8] = {nA0, nA1, nA2, nA3, nA4, nA5, nA6, nA7};
    const uint64_t gid = game_offset + (uint64_t)g;
    u32x4 x = {0, 0, 0, 0};
    for (int i = 0; i < N; i++) {
        if ((i & 1) == 0) x = draw(seed, gid, episode, step, (uint32_t)(i >> 1));
        const uint32_t xu = (i & 1) ? x.z : x.x, xc = (i & 1) ? x.w : x.y;
        u_out[(size_t)i * G + g] = u01_32(xu);
        choice_out[(size_t)i * G + g] = (int8_t)__umulhi(xc, (uint32_t)nA[i]);
        if (u2_out) u2_out[(size_t)i * G + g] = u01_32(xc);      // the second word as a uniform (CAC's Box-Muller)
    }
    if (noise_u_out) {
        const u32x4 xn = draw(seed, gid, episode, step, kStreamNoise);
        noise_u_out[g] = u01_32(xn.x);
        noise_a_out[g] = __dadd_rn(noise_lo, __dmul_rn(__dsub_rn(env_a, noise_lo), u01_32(xn.y)));
    }
}

int launch_nn_init(int G, int A, float* params, uint64_t seed, uint64_t off, int agent, int value_head, hipStream_t s) {
    const int64_t n = (int64_t)G * (2 * kH + A * kH + A + (value_head ? kH + 1 : 0));
    hipLaunchKernelGGL(k_nn_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, G, A, params, seed, off, agent,
                       value_head);
    return (int)hipGetLastError();
}
int launch_nn_act(int G, int A, const float* params, int P, const double* price, const double* u, int32_t* act,
                  float* prob, hipStream_t s) {
    const dim3 grid((unsigned)((G + 3) / 4)), block(256);
    if (A <= 8) hipLaunchKernelGGL(k_nn_act<8>, grid, block, 0, s, G, A, params, P, price, u, act, prob);
    else if (A <= 24) hipLaunchKernelGGL(k_nn_act<24>, grid, block, 0, s, G, A, params, P, price, u, act, prob);
    else hipLaunchKernelGGL(k_nn_act<32>, grid, block, 0, s, G, A, params, P, price, u, act, prob);
    return (int)hipGetLastError();
}
size_t nn_train_lds_bytes(int A, int N, int value_head) {
    const size_t pad = A <= 24 ? 24 : 32;
    const size_t nx = ((size_t)N + kChunk - 1) / kChunk * kChunk;
    size_t work;
    if (!value_head) {
        // W2t | dz | xs | Gs | w1s | b1s | b2s | red | uid (u16) | xu | ucnt | redi | sga (i64)
        work = sizeof(float) * ((size_t)kH * pad + (size_t)kChunk * pad + 2 * nx + 2 * kH + kMaxA + 8 +
                                nx / 2 + kXu + kUmax + 8 + 2 * (size_t)kUmax * pad);
    } else {
        // W2t | dz | [w1s | b1s | xs | Gs | xps | gvs | wvs, padded to the folded path's accumulators] | uid | uidp | xu | ucnt | redi | b2s | red
        const size_t region = sizeof(float) * (3 * (size_t)kH + 4 * nx), need = train_ac_fold_bytes((int)pad);
        work = sizeof(float) * ((size_t)kH * pad + (size_t)kChunk * pad + kMaxA + 8) + (region > need ? region : need) +
               2 * nx * sizeof(unsigned short) + sizeof(float) * (kXuAc + kUmax + 8);
    }
    // the Adam sweep stages the gradient [P] behind the combine scratch [128][2 pad + 6] + [256]
    const size_t stage = sizeof(float) * (128 * (2 * pad + 6) + 256 + (size_t)(2 * kH + A * kH + A + (value_head ? kH + 1 : 0)));
    return work > stage ? work : stage;
}
int launch_nn_train(int G, int A, float* params, float* m, float* v, int step, int N, int ld, const double* price,
                    const int32_t* action, const double* reward, const double* nprice, float gamma, float ent, float lr,
                    const double* gamma_g, const double* ent_g, float* grad, float* returns_scratch, hipStream_t s) {
    const size_t lds = nn_train_lds_bytes(A, N, nprice != nullptr);
    if (returns_scratch && !nprice) {
        hipLaunchKernelGGL(k_nn_returns, dim3((G + 63) / 64), dim3(64), 0, s, G, N, ld, reward, gamma, gamma_g, returns_scratch);
        const hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return (int)e0;
    }
    auto kern = nprice ? (A <= 24 ? k_nn_reinforce_train<24, true> : k_nn_reinforce_train<32, true>)
                       : (A <= 24 ? k_nn_reinforce_train<24, false> : k_nn_reinforce_train<32, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, s, G, A, params, m, v, step, N, ld, price, action, reward, nprice,
                       gamma, ent, lr, gamma_g, ent_g, grad, nprice ? nullptr : returns_scratch);
    return (int)hipGetLastError();
}
int launch_op_draws(int G, int N, uint64_t seed, uint64_t off, uint32_t episode, uint32_t step, double env_a,
                    double noise_lo, const int32_t* nA, double* u, int8_t* ch, double* u2, double* nu, double* na,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_op_draws, dim3((G + 255) / 256), dim3(256), 0, s, G, N, seed, off, episode, step, env_a,
                       noise_lo, nA[0], nA[1], nA[2], nA[3], nA[4], nA[5], nA[6], nA[7], u, ch, u2, nu, na);
    return (int)hipGetLastError();
}

}  // namespace thrl
