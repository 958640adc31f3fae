// thrl_wave_f32nc.hip -- instantiates k_wave_episodes<float, *, *, NOISE=true, SWEEP=false, CYCLE=true> (thrl_wave_kernel.h)
#include "thrl_wave_kernel.h"

namespace thrl {

int launch_wave_f32_noise_cycle(const WaveArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    return launch_wave_n<float, true, false, true>(a, grid, block, lds, s);
}

}  // namespace thrl
