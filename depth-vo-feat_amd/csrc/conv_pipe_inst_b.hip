// Explicit instantiations of the pipelined convolution kernel (split over several units to build in parallel).
#include "conv_pipe.h"
namespace dvfp {
DVF_PIPE_FAMILY(2, 1, 1)
}  // namespace dvfp
