"""One pool on several GPUs, exactly: the NODES of a NodeDb partitioned across ranks for the wide, read-only node-selection queries.

SURVEY 8e: a query is a min-reduction over nodes, so rows [g*N/G, (g+1)*N/G) of the node matrices can live on GPU g while the (small) job and
queue state is replicated; the collective is ONE all-reduce MIN per batch of queries over a 64-bit word per query that orders like the
reference's index key — (rounded allocatable on the indexed columns ..., global node index) — so `ReduceOp.MIN` on int64 IS the reference's
"first node in index order that fits" across shards (nodedb.go:840-879, encoding.go:37-54).  Whoever holds the winning row would then apply the
bind locally; the operations served here do not bind:

  * fit_select_batch  — BASELINE configs[1], n independent selectNodeForPodAtPriority calls against a fixed state;
  * submit check, individual units — SubmitChecker on a pristine NodeDb (submitcheck.go:272-290): the same question per scheduling key.

Each rank owns an ordinary library handle over ITS rows (asched_nodes_upsert of the shard, running jobs of other shards masked out), answers the
whole query batch against them with one k_fit_batch launch, and the library packs (order key in the layout all shards share, global rank) per query ON THE
DEVICE into a buffer the caller owns (asched_fit_select_batch_global: armada_sched_mgpu.hip k_mgpu_pack); `torch.distributed.all_reduce(MIN)` on that
tensor, in place (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" on host memory in the CPU tests), folds the shards.  Traffic: 8 bytes per query per rank — latency
bound, which is why it is only used for batched passes.  The sequential round (every placement reads what all earlier placements wrote) does
not shard this way without a collective per job; pools remain the unit of parallelism for rounds (multipool.py).
"""
from __future__ import annotations

import copy
from typing import Optional, Sequence

import numpy as np

from . import workloads as W
from .binding import EVICTED_PRIORITY, Scheduler

NO_NODE_WORD = np.int64(2 ** 63 - 1)   # ASCHED_NO_NODE_WORD


def shard_bounds(n_nodes: int, world: int):
    return [n_nodes * g // world for g in range(world + 1)]


def shard_workload(wl: W.Workload, rank: int, world: int) -> W.Workload:
    """rows [lo, hi) of the node tables; running jobs on other shards become unbound (their resources are on rows this rank does not hold)"""
    lo, hi = shard_bounds(wl.num_nodes, world)[rank:rank + 2]
    s = copy.copy(wl)
    s.node_total = wl.node_total[lo:hi]
    s.node_allocatable = None if wl.node_allocatable is None else wl.node_allocatable[lo:hi]
    s.node_taints = None if wl.node_taints is None else wl.node_taints[lo:hi]
    s.node_labels = None if wl.node_labels is None else wl.node_labels[lo:hi]
    s.node_id_rank = None if wl.node_id_rank is None else np.argsort(np.argsort(wl.node_id_rank[lo:hi])).astype(np.int32)
    node = wl.job_node.copy()
    mine = (node >= lo) & (node < hi)
    node[~mine] = -1
    node[mine] -= lo
    s.job_node = node.astype(np.int32)
    s.meta = dict(wl.meta, shard=(rank, world, lo, hi))
    return s


class ShardedFit:
    """rank-local view of one NodeDb partitioned by node rows; collective calls must be made by every rank with the same arguments"""

    def __init__(self, lib, wl: W.Workload, rank: int, world: int, dist=None, device: Optional[str] = None):
        self.wl, self.rank, self.world, self.dist, self.device = wl, rank, world, dist, device
        self.lo, self.hi = shard_bounds(wl.num_nodes, world)[rank:rank + 2]
        self.local = shard_workload(wl, rank, world)
        self.in_library = False
        self.s: Scheduler = W.load(lib, self.local)
        cfg = wl.config
        self.idx_col = list(cfg.indexed_col)
        self.idx_res = [int(r) for r in cfg.indexed_resolution]
        # width of every key field from the GLOBAL capacity, so that all ranks pack identically (the library's own packed key is sized per shard)
        cap = wl.node_total if wl.node_allocatable is None else wl.node_allocatable
        self.width = [max(1, int(int(cap[:, c].max(initial=0)) // r + 1).bit_length()) for c, r in zip(self.idx_col, self.idx_res)]
        self.row_bits = max(1, int(wl.num_nodes - 1).bit_length())
        if sum(self.width) + self.row_bits > 62:
            raise ValueError("order key does not fit 62 bits: use two reductions (fields, then row)")

    def comm_init(self, transport: str = "rccl"):
        """give the handle its own communicator: "rccl" (ncclCommInitRank inside the library; the process group only carries the unique id) or "external"
        (the library calls back into torch.distributed: gloo in the CPU tests).  fit_select_batch then runs the collective inside the library."""
        from . import comm
        if transport == "rccl":
            comm.init_rccl(self.s, self.dist, device=self.device)
        else:
            comm.init_external(self.s, self.dist, device_memory=self.device is not None)
        self.in_library = True

    def _sync_device(self):
        if self.device is not None:
            import torch
            torch.cuda.synchronize()

    def prepare(self):
        """bind the running jobs of this shard (populateNodeDb); the queue side is not needed for read-only queries"""
        W.prepare(self.s, self.local)

    def _word_buffer(self, n: int):
        """the int64 tensor the library fills and the collective reduces IN PLACE: on the handle's GPU when there is one (RCCL), on the host otherwise (gloo)"""
        import torch
        return torch.empty(max(n, 1), dtype=torch.int64, device=self.device or "cpu")

    def fit_select_batch(self, jobs: Sequence[int], priority: int = EVICTED_PRIORITY) -> np.ndarray:
        """first feasible node (GLOBAL row, -1 = none) per job at `priority` against the current state — collective.

        The library runs k_fit_batch over this shard's rows and packs, on the device, one word per query that orders like the reference's index key with the
        field widths all shards share (asched_fit_select_batch_global); ONE all_reduce(MIN) over that buffer folds the shards; the low bits of the winning
        word are the node's global rank (= its row: shards are contiguous row ranges in index order)."""
        n = len(jobs)
        if self.in_library:
            # the product path: k_fit_batch + pack + ncclAllReduce(MIN) + unpack as ONE stream-ordered sequence inside the library (asched_fit_select_batch_sharded)
            return self.s.fit_select_batch_sharded(np.asarray(jobs, dtype=np.int32), priority, self.width, self.row_bits, rank_offset=self.lo)
        t = self._word_buffer(n)
        self.s.fit_select_batch_global(np.asarray(jobs, dtype=np.int32), priority, self.width, self.row_bits, t.data_ptr(), rank_offset=self.lo)
        if self.dist is not None and self.world > 1:
            self._sync_device()   # the library wrote the words on ITS stream (synchronised on return); the collective runs on torch's: order them explicitly
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
            self._sync_device()
        best = t[:n].cpu().numpy()
        out = (best & ((1 << self.row_bits) - 1)).astype(np.int32)
        out[best == NO_NODE_WORD] = -1
        return out

    def close(self):
        self.s.close()
