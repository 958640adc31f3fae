// thrl_tuple_f32_sweep.hip -- instantiates k_tuple_episodes<float, N, NSEG, true, true>: per-game sweeps (thrl_tuple_kernel.h)
#include "thrl_tuple_kernel.h"

namespace thrl {

int launch_tuple_f32_sweep(const TupleArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    return tup::launch_tuple_t<float, true, true>(a, grid, block, lds, s);
}

}  // namespace thrl
