// thrl_tuple_f64.hip -- instantiates k_tuple_episodes<double, N, NSEG, false, false> (thrl_tuple_kernel.h)
#include "thrl_tuple_kernel.h"

namespace thrl {

int launch_tuple_f64(const TupleArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    return tup::launch_tuple_t<double, false, false>(a, grid, block, lds, s);
}

}  // namespace thrl
