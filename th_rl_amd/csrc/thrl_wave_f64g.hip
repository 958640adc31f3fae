// thrl_wave_f64g.hip -- instantiates k_wave_episodes<double, *, *, NOISE=false, SWEEP=false, CYCLE=false, GREEDY=true> (thrl_wave_kernel.h)
#include "thrl_wave_kernel.h"

namespace thrl {

int launch_wave_f64_plain_greedy(const WaveArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    return launch_wave_n<double, false, false, false, true>(a, grid, block, lds, s);
}

}  // namespace thrl
