// armada_sched_wk.hip — fifth translation unit of libarmada_sched.so: the round kernel once more, for handles whose ORDER KEY TAKES TWO WORDS (k_control_wk, k_bulk_wk).
// The reference's index key is one 8-byte word per indexed resource plus the node index (internal/scheduler/nodedb/encoding.go:22-54), unbounded; the packed key of the
// default kernels is one 64-bit word.  When the fields of a pool need more (asched_host.inc layoutKeys: fine resolutions, large nodes, a fifth indexed resource, a million
// nodes) the key is (high word, low word), the handle runs on the generic path — the reference statement by statement, round_ctl.h / round_run.h — and node selection is two
// plane passes (armada_sched.hip wgFirstFitKey).  WIDE_KEYS() is a compile-time `true` here and `false` in every other code object, so the one-word kernels carry nothing
// of this (their ISA hash is unchanged: profiles/r06z2_*).  Device code only: the C ABI lives in armada_sched.hip.
#define ASCHED_WK_TU 1
#include "armada_sched.hip"
