// thrl_tuple.hip -- host side of the tuple-chain kernel (thrl_tuple_kernel.h): the LUT builder and the dispatch
// over the table type.  The kernel variants are instantiated in thrl_tuple_f32.hip / thrl_tuple_f64.hip.
#include "thrl_kernels.h"

namespace thrl {

// LUT image: prow [tuples] u32 (window-local play rows, byte i = agent i), trow [tuples] u32 (train rows), aq [N][64], sct [N][64], price [tuples],
// qsum [tuples] (total quantity of the tuple: the price under a redrawn intercept is max(0, a' - b * qsum))
__global__ void __launch_bounds__(256) k_tuple_lut(const TupleArgs a, unsigned char* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.N;
    if (idx < a.tuples) {
        int digit[kTupMaxN];
        int rest = idx;
        for (int i = N - 1; i >= 0; i--) { digit[i] = rest % a.ag[i].n_actions; rest /= a.ag[i].n_actions; }
        double scaled[kTupMaxN], rew[kTupMaxN];
        for (int i = 0; i < N; i++) scaled[i] = scale_action(digit[i], a.ag[i]);
        const double price = env_step<kTupMaxN>(a.env, N, scaled, a.env.a, rew);          // environments.py:25-39
        reinterpret_cast<double*>(out + a.price_off)[idx] = price;
        double Q = 0.0;
        for (int i = 0; i < N; i++) Q = __dadd_rn(Q, __dmul_rn(a.env.ratio, scaled[i]));          // as env_step sums it
        reinterpret_cast<double*>(out + a.qsum_off)[idx] = Q;
        uint32_t pw = 0u, tw = 0u;
        for (int i = 0; i < N; i++) {
            const int W = a.win_rows[i];
            const int rp = min(max(encode32(price, a.ag[i]) - a.row_lo[i], 0), W - 1);     // play row (trainer.py:53)
            const int rt = min(max(encode64(price, a.ag[i]) - a.row_lo[i], 0), W - 1);     // train row (agents.py:62,66)
            pw |= (uint32_t)rp << (8 * i); tw |= (uint32_t)rt << (8 * i);
        }
        reinterpret_cast<uint32_t*>(out)[idx] = pw;
        reinterpret_cast<uint32_t*>(out)[a.tuples + idx] = tw;
    }
    if (idx < N * 64) {
        const int i = idx >> 6, k = idx & 63;
        const double sc = scale_action(min(k, a.ag[i].n_actions - 1), a.ag[i]);
        reinterpret_cast<double*>(out + a.aq_off)[idx] = __dmul_rn(a.env.ratio, sc);
        reinterpret_cast<double*>(out + a.aq_off)[N * 64 + idx] = __ddiv_rn(sc, (double)a.T);
    }
}

int launch_tuple_lut(const TupleArgs& a, unsigned char* out, hipStream_t s) {
    const int n = a.tuples > a.N * 64 ? a.tuples : a.N * 64;
    hipLaunchKernelGGL(k_tuple_lut, dim3((n + 255) / 256), dim3(256), 0, s, a, out);
    return (int)hipGetLastError();
}

int launch_tuple(const TupleArgs& a, int q_dtype, int grid, int block, size_t lds, hipStream_t s) {
    if (a.sweep) return q_dtype == 1 ? launch_tuple_f64_sweep(a, grid, block, lds, s) : launch_tuple_f32_sweep(a, grid, block, lds, s);
    if (a.env.noise_prob > 0.0) return q_dtype == 1 ? launch_tuple_f64_noise(a, grid, block, lds, s) : launch_tuple_f32_noise(a, grid, block, lds, s);
    return q_dtype == 1 ? launch_tuple_f64(a, grid, block, lds, s) : launch_tuple_f32(a, grid, block, lds, s);
}

}  // namespace thrl
