// thrl_wave_lut.h -- byte layout of the wave kernel's payoff LUT image
// (built in HBM by k_wave_lut, copied verbatim into LDS by every block).
#pragma once
#include <hip/hip_runtime.h>

namespace thrl {

struct WaveLut {
    int ns_off;     // uint16 [A*A]  window-local next-state row per action pair (a0*A+a1):
                    //               play row (float32 encode) | train row (float64 encode) << 8
    int aq_off;     // double [2][A] (a/b)*scale_i(k): quantity of agent i at action k
    int sct_off;    // double [2][A] scale_i(k)/T: per-step contribution to actions_log
    int lds_bytes;  // all of it is staged in LDS
    int bytes;      // total, multiple of 16
};

__host__ __device__ inline WaveLut wave_lut_layout(int A) {
    WaveLut l;
    l.ns_off = 0;
    l.aq_off = (2 * A * A + 15) & ~15;
    l.sct_off = l.aq_off + 16 * A;
    l.lds_bytes = (l.sct_off + 16 * A + 15) & ~15;
    l.bytes = l.lds_bytes;
    return l;
}

}  // namespace thrl
