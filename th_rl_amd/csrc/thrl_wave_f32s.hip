// thrl_wave_f32s.hip -- instantiates k_wave_episodes<float, *, *, NOISE=true, SWEEP=true, CYCLE=false> (thrl_wave_kernel.h)
#include "thrl_wave_kernel.h"

namespace thrl {

int launch_wave_f32_sweep(const WaveArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    return launch_wave_n<float, true, true, false>(a, grid, block, lds, s);
}

}  // namespace thrl
