// armada_sched_ft.hip — fourth translation unit of libarmada_sched.so: the round kernel once more, WITH the fair-share threshold table (round_ft.h), as k_control_ft in a
// code object of its own.  The table answers the fair-share half of a preempting job's question without a wide pass (BASELINE configs[4]); its call sites inside the default
// round kernel cost the headline 2-3 % by code placement alone (profiles/r03f_headline_regression_bisect_ab.txt), so the default kernel is built without it and the host
// launches this one for handles that carry a table — on request only (ASCHED_FT=1; asched_host.inc ensureFt has the round-5 measurement: no gain on configs[4] until the
// gate / urgency sweep is an index lookup too).  Device code only: the grid-wide kernels and the C ABI live in armada_sched.hip.
#define ASCHED_FT_TU 1
#define ASCHED_WITH_FT 1
#include "armada_sched.hip"
