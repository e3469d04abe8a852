// step_coop_sums.hip -- the cooperative kernels of step_coop.hip once more, as the instantiations that sum every member's outputs
// inside the launch (coopBody<..., Sums>) for what step_coop.hip's own launcher does not carry: fp32-mixed batches on every
// layout (one, two and four chunks per workgroup, the optional-physics and nitrogen-cycle families) and fp64 batches on the
// four-chunk layout.  A translation unit of its own so that step_coop.o stays the object whose code placement was measured
// (tests/test_code_placement.py) and the two compile side by side.  sipnet_batch_run_sums launches these through
// sums2::launchStepCoopSums.
#define SIPNET_COOP_SUMS_TU 1
#include "step_coop.hip"
