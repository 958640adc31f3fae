// thrl_wave_f64.hip -- instantiates k_wave_episodes<double, *, *, NOISE=false, SWEEP=false, CYCLE=false> (thrl_wave_kernel.h)
#include "thrl_wave_kernel.h"

namespace thrl {

int launch_wave_f64_plain(const WaveArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    return launch_wave_n<double, false, false, false>(a, grid, block, lds, s);
}

}  // namespace thrl
