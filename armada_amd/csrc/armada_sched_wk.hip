// armada_sched_wk.hip — fifth translation unit of libarmada_sched.so: the round kernel once more, for the handles the default kernels do not serve (k_control_wk, k_bulk_wk,
// k_fit_batch_wk), so that nothing of what follows moves an instruction of k_control (tools/kcontrol_isa_hash.sh: 220bc4c6... before and after):
//  * ORDER KEYS OF TWO WORDS.  The reference's index key is one 8-byte word per indexed resource plus the node index (internal/scheduler/nodedb/encoding.go:22-54), unbounded; the
//    packed key of the default kernels is one 64-bit word.  When a pool needs more (asched_host.inc layoutKeys: fine resolutions, large nodes, a fifth indexed resource, a million
//    nodes) the key is one 128-bit integer stored as (high word, low word); the handle runs on the generic path — the reference statement by statement, round_ctl.h / round_run.h —
//    and node selection is two plane passes (armada_sched.hip wgFirstFitKey).  WIDE_KEYS() is a run-time test here (dev.h), a compile-time `false` in every other code object.
//  * SHARDED WIDE PASSES: one pool's round on several GPUs, exact (asched_shard_round / asched_shard_peers; dev.h SHARD_ON, armada_sched.hip shardReduce): the plane scan and the
//    fair-share evaluation look at this replica's share of the node words and exchange their two result words — through the host proxy or GPU-to-GPU.  One-word handles keep
//    their fast path here (the whole round kernel is compiled in).
//  * every control command of such a handle, the auxiliary ones and market-driven rounds included (this code object also carries round_mkt.h).
// Device code only: the C ABI lives in armada_sched.hip.
#define ASCHED_WK_TU 1
#include "armada_sched.hip"
