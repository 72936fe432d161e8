"""User-sharded BPR-MF training across the GPUs of one node.

The reference is single-process / single-device (`config['gpu']` is one id,
AbstractRecommender.py:99); this is the data-parallel form of the same step:

* users — and with them their interactions and their rows of P — are split into
  contiguous ranges, one per rank: P traffic is rank-local, no collective;
* Q (item table) is replicated; each rank forms a partial item gradient gQ from its own
  samples and the ranks SUM-all-reduce it (RCCL over xGMI) — the one real exchange;
* the three sums of squares under the non-squared Frobenius regulariser
  (MFRecommender.py:88-89,94-95) and the loss are batch-wide, so the 7 batch sums are
  all-reduced (56 bytes) before any update.

The union of the per-rank batches is one global batch: the result equals the single-GPU
step on that union up to fp32 summation order.

Per step and rank (compute stream | collective):
    forward -> stats[0:7]              | all_reduce(stats[0:7]) async --+  (56 B, latency bound)
    item_grad_data -> gQ (no norms)    |   overlapped                 <+
    wait; finalize; item_grad_reg      |
                                       | all_reduce(gQ) async  --+
    user_sgd (reads Q pre-step)        |   overlapped           <+
    wait; item_sgd_apply(dense)        |
(the reproducible item modes form data term and regulariser in one kernel, so there the first
 all-reduce is waited for before item_grad)
"""
from __future__ import annotations

import torch.distributed as dist

from . import _native as N


def user_range(user_num: int, world_size: int, rank: int):
    """Contiguous, near-equal split of user ids: rank r owns [lo, hi)."""
    base, rem = divmod(int(user_num), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_triples(triples, user_num: int, world_size: int, rank: int):
    """Rows of an int32 [N,3] (user,pos,neg) array that belong to `rank`'s user range."""
    lo, hi = user_range(user_num, world_size, rank)
    u = triples[:, 0]
    return triples[(u >= lo) & (u < hi)]


class UserShardedBprTrainer:
    """One instance per rank.  `ctx` is the step backend: a `daisyrec_amd.ops.BprContext`
    (the HIP path) in production; the CPU gloo tests inject an oracle-backed stand-in with
    the same methods.  `P_local` holds rows [lo,hi) of the user table, `Q` the full item
    table; both are updated in place."""

    def __init__(self, ctx, P_local, Q, user_lo, lr, reg_1, reg_2, loss_type=N.LOSS_BPR,
                 gamma=1e-10, item_mode=N.ITEM_CHUNKED, group=None, overlap=True):
        self.ctx, self.P, self.Q = ctx, P_local, Q
        self.user_lo = int(user_lo)
        self.lr, self.reg_1, self.reg_2 = float(lr), float(reg_1), float(reg_2)
        self.loss_type, self.gamma, self.item_mode = int(loss_type), float(gamma), int(item_mode)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.overlap = bool(overlap)

    def _all_reduce(self, t, async_op=False):
        if self.world == 1:
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def step_from_triples(self, triples, idx=None, start=0, B=None):
        """One global step; `triples` are this rank's rows (global user ids)."""
        c = self.ctx
        c.set_batch_from_triples(triples, idx=idx, start=start, B=B, user_base=self.user_lo)
        return self._step()

    def step_from_plan(self, plan, k):
        """One global step on batch k of this rank's epoch plan."""
        self.ctx.set_batch_from_plan(plan, k)
        return self._step()

    def step(self, u_local, i, j):
        self.ctx.set_batch(u_local, i, j)
        return self._step()

    def _step(self):
        c = self.ctx
        c.forward(self.P, self.Q, self.loss_type, self.gamma)
        split = self.item_mode in (N.ITEM_CHUNKED, N.ITEM_FUSED) and hasattr(c, "item_grad_data")
        if split:
            w0 = self._all_reduce(c.stats[:7], async_op=self.overlap)
            c.item_grad_data(self.P, self.Q, self.item_mode)      # overlaps the 56-byte all-reduce
            if w0 is not None:
                w0.wait()
            c.finalize(self.reg_1, self.reg_2)             # every rank: the GLOBAL loss and norms
            c.item_grad_reg(self.Q, self.reg_1, self.reg_2)
        else:
            self._all_reduce(c.stats[:7])
            c.finalize(self.reg_1, self.reg_2)
            c.item_grad(self.P, self.Q, self.reg_1, self.reg_2, self.item_mode)
        work = self._all_reduce(c.gQ, async_op=self.overlap)
        c.user_sgd(self.P, self.Q, self.lr, self.reg_1, self.reg_2)   # overlaps the all-reduce
        if work is not None:
            work.wait()
        c.item_sgd_apply(self.Q, self.lr, dense=self.world > 1)
        return c.stats
