// thrl_tuple_f64_noise.hip -- instantiates k_tuple_episodes<double, N, NSEG, true, false>: games with env noise (thrl_tuple_kernel.h)
#include "thrl_tuple_kernel.h"

namespace thrl {

int launch_tuple_f64_noise(const TupleArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    return tup::launch_tuple_t<double, true, false>(a, grid, block, lds, s);
}

}  // namespace thrl
