// armada_sched_aux.hip — second translation unit of libarmada_sched.so: the device code of armada_sched.hip compiled into its own
// code object with a single kernel, k_control_aux, which runs the submit-check commands (asched_submit_check; SURVEY 8f-2,
// internal/scheduler/submitcheck.go:342-371).  See the ASCHED_AUX_TU section at the end of armada_sched.hip for why this is a
// separate code object rather than one more case in the round kernel's command switch.
#define ASCHED_AUX_TU 1
#include "armada_sched.hip"
