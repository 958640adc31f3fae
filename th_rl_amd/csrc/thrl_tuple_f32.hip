// thrl_tuple_f32.hip -- instantiates k_tuple_episodes<float, N, NSEG, false, false> (thrl_tuple_kernel.h)
#include "thrl_tuple_kernel.h"

namespace thrl {

int launch_tuple_f32(const TupleArgs& a, int grid, int block, size_t lds, hipStream_t s) {
    return tup::launch_tuple_t<float, false, false>(a, grid, block, lds, s);
}

}  // namespace thrl
