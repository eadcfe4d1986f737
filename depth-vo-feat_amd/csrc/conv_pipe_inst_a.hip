// Explicit instantiations of the pipelined convolution kernel (split over several units to build in parallel).
#include "conv_pipe.h"
namespace dvfp {
DVF_PIPE_FAMILY(1, 1, 1)
DVF_PIPE_FAMILY(1, 2, 1)
}  // namespace dvfp
