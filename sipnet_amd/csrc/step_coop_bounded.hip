// step_coop_bounded.hip -- the cooperative kernels of step_coop.hip with BOUNDED hand-over waits (see "hand-over
// waits" there): the same source compiled a second time under -DSIPNET_COOP_BOUNDED into sipnet::bounded, lean
// instantiations only.  SIPNET_KOPT_BOUNDED_WAITS launches these; a wait that exhausts its budget of polls ends the
// launch and sipnet_batch_run reports SIPNET_ERR_INTERNAL with the wait's number and step.  Never the shape policy's
// choice: the polls' exit path costs the step ~10 %.
#define SIPNET_COOP_BOUNDED 1
#include "step_coop.hip"
