// step_fast_sums.hip -- the one-wavefront kernels of step_fast.hip once more, as the instantiations that sum every member's outputs
// inside the launch (fastBody<..., Sums>: stepFastSumsKernel): what sipnet_batch_run_sums launches for batches the shape policy
// gives the one-wavefront kernel (more than four 64-member chunks per compute unit, or that kernel forced).  A translation unit
// of its own so that step_fast.o stays the object that was profiled and the two compile side by side.
#define SIPNET_FAST_SUMS_TU 1
#include "step_fast.hip"
