"""User-sharded BPR-MF training across the GPUs of one node.

The reference is single-process / single-device (`config['gpu']` is one id,
AbstractRecommender.py:99); this is the data-parallel form of the same step:

* users — and with them their interactions and their rows of P — are split into
  contiguous ranges, one per rank: P traffic is rank-local, no collective;
* Q (item table) is replicated; each rank forms a partial item gradient from its own
  samples — the one real exchange;
* the batch-wide sums under the non-squared Frobenius regulariser
  (MFRecommender.py:88-89,94-95) and the loss are all-reduced (a few doubles) before
  any row that depends on them moves.

The union of the per-rank batches is one global batch: the result equals the single-GPU
step on that union up to fp32 summation order.

STAGED protocol (item_mode 'fused', the default; kernels of csrc/bpr_staged.hip).  The item
exchange costs I/N rows of apply work per rank instead of I, and its wire volume is the
reduce-scatter + all-gather pair a ring all-reduce is made of:

    prenorm                  sum_b |P[u_b]|^2 from the row-norm cache   | all_reduce (8 B)
    staged_user              forward + user rows of P in place          |
                             stats[0:7] = the batch sums                | all_reduce (56 B) --+
    staged_item -> gQ, cnt   data term of dL/dQ + per-item entry counts |   overlapped        <+
    finalize                 the GLOBAL loss and norms                  |
                                                                        | reduce_scatter(gQ), reduce_scatter(cnt)
    item_apply_counts        rank r: SGD on its rows [r*I/N,(r+1)*I/N)  |
                             of Q, regulariser from the GLOBAL counts   | all_gather(Q rows), in place
    (gQ, cnt are re-zeroed behind the reduce-scatter)

Per step and rank at d=64: 2*(N-1)/N * I * 264 B on the wire (I=1M, N=8: 2 x 231 MB), against
B_local interactions of compute; DESIGN.md section 5 has the budget.

`slices=S` (> 1; 'auto' = `auto_exchange_slices`, the default of `MF.fit` and `bench.py` since round 5): the item rows are cut into S ranges, the item pass runs range by range
(daisy_bpr_staged_item_slice; the entries are sorted by item) and the exchange of a finished range - reduce-scatter,
owner update, all-gather - runs on a side stream while the next range is reduced.  Ownership then interleaves: rank
r owns the r-th block of every range.  Every rank must use the same S.

TOUCHED-ROWS exchange (`exchange='sparse'`; 'auto' = `auto_item_exchange`; SURVEY.md section 8e "send only touched rows
(index-union + values) when 2B << I", north_star "only where users overlap items").  A step of B_local samples per rank
touches at most 2 B_local item rows per rank; when world x 2 B_local << I the dense exchange above moves I x 264 B for a
few thousand rows that changed.  Instead, with static shapes and no host synchronisation:

    ids_r        = the rank's touched item ids (ascending, padded with I)        | all_gather (world x cap_local x 4 B)
    union        = sorted distinct ids of all ranks - the SAME list on every rank (bit-identical replicas need that)
    buf[k]       = [gQ[union_k] | cnt[union_k]]   (zero rows for the padding)    | reduce_scatter (cap_union x (d+2) floats)
    owner apply  rank r: SGD on union rows [r*rows, (r+1)*rows) of Q            | all_gather (cap_union x d floats)
    Q[union_k]   = the gathered rows; the rank's touched rows of gQ / cnt are re-zeroed

SGD only: under torch's dense Adam every row of Q moves in every step (no row is untouched), so Adam keeps the dense exchange.

PHASE protocol (item modes 'chunked' / 'sorted', or a backend without the staged phases):
forward -> all_reduce(stats) -> item_grad -> all_reduce(gQ) overlapped with user_sgd -> dense
item_sgd_apply.  Kept for the modes the staged step does not cover.

DENSE-OPTIMISER protocol (`dense_opt=`: torch's Adagrad / RMSprop, and Adam for FM - AbstractRecommender.py:48-67): the
phase protocol with gradients instead of updates - forward -> all_reduce(stats) -> item_grad -> all_reduce(gQ)
overlapped with user_grad -> the dense optimiser on the rank's rows of P (its own state) and on the replicated Q (every
rank the same gradient, the same state, the same step).  FM: u_bias with its users, the item-bias gradient and the
coefficient sum all-reduced.  The wire volume is a dense all-reduce of gQ per step: the slow path, for the optimisers
the staged step does not apply.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _native as N


def user_range(user_num: int, world_size: int, rank: int):
    """Contiguous, near-equal split of user ids: rank r owns [lo, hi)."""
    base, rem = divmod(int(user_num), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def auto_exchange_slices(item_num: int, d: int, world_size: int, batch_per_rank: int, backend: str = "nccl",
                         bus_gbs: float | None = None) -> int:
    """How many item ranges the item pass of a user-sharded step is cut into (`slices='auto'`).  A pure function of
    numbers every rank holds identically (the GLOBAL batch / world size, not a rank's own share), so every rank arrives
    at the same cuts - the replicas stay bit-identical only then.

    The exchange of a step is fixed by the item table: reduce-scatter + all-gather of I x (d + 2) floats, 2 (N-1)/N of
    them on the wire per rank, at the bus rate RCCL reaches on the node (assumed 300 GB/s on an 8-GPU xGMI node,
    DESIGN.md section 5; `DAISY_XGMI_BUS_GBS` overrides).  The item pass it can hide under moves 2 B_local staged rows at
    about 5 TB/s.  With S slices all but the last range's exchange run under the next range's reduction; a cut costs
    about 0.15 % of the pass (4.77 -> 4.82 ms at S = 8).  One slice when the exchange is under a
    tenth of the pass or the backend has no asynchronous collectives worth overlapping (gloo: the CPU tests); otherwise
    the smallest power of two that leaves at most a tenth of the pass exposed - at most 16, no slice under ~100 us of item
    pass (a slice is a launch and three collectives of its own), blocks of at least 1024 rows per owner and slice."""
    import os
    if world_size <= 1 or backend != "nccl":
        return 1
    if bus_gbs is None:
        bus_gbs = float(os.environ.get("DAISY_XGMI_BUS_GBS", "300"))
    t_exchange = 2.0 * (world_size - 1) / world_size * item_num * (d + 2) * 4.0 / (bus_gbs * 1e9)
    t_item = 2.0 * batch_per_rank * 4.0 * d / 5.0e12
    if t_exchange < 0.1 * t_item:
        return 1
    want = t_exchange / max(0.1 * t_item, 1e-9)
    s_ = 2
    while s_ < want and s_ < 16:
        s_ *= 2
    # a slice is a kernel launch and three collectives of its own: none shorter than ~100 us of item pass (a pass that
    # short cannot hide much of the exchange anyway), no block under 1024 rows per owner and slice
    while s_ > 1 and (t_item / s_ < 100e-6 or item_num // (world_size * s_) < 1024):
        s_ //= 2
    return s_


def sparse_exchange_caps(item_num: int, world_size: int, max_local_batch: int, global_batch: int | None = None):
    """(cap_local, cap_union, rows_per_owner) of the touched-rows exchange: a rank whose share of a step is at most
    `max_local_batch` samples touches at most twice that many distinct items (positives and negatives); the union over
    the ranks at most 2 x `global_batch` (default: world x max_local_batch), never more than I; the union is cut into
    `world` equal owner blocks (padded)."""
    W = int(world_size)
    cap_local = max(1, min(int(item_num), 2 * int(max_local_batch)))
    gb = int(global_batch) if global_batch else W * int(max_local_batch)
    cap_union = max(1, min(int(item_num), 2 * gb, W * cap_local))
    rows = (cap_union + W - 1) // W
    return cap_local, rows * W, rows


def item_exchange_bytes(item_num: int, d: int, world_size: int, max_local_batch: int, global_batch: int | None = None,
                        slices: int = 1):
    """bytes a rank puts on the wire per step for the item exchange: {'dense': ..., 'sparse': ...} (ring collectives:
    (N-1)/N of the payload per rank for an all-gather or a reduce-scatter)"""
    W = int(world_size)
    if W <= 1:
        return {"dense": 0, "sparse": 0}
    f = (W - 1) / W
    rows = (int(item_num) + W * slices - 1) // (W * slices) * W * slices
    dense = f * rows * (d + 2) * 4 + f * rows * d * 4               # reduce-scatter of [gQ | cnt], all-gather of the Q rows
    cap_local, cap_union, _ = sparse_exchange_caps(item_num, W, max_local_batch, global_batch)
    sparse = f * W * cap_local * 4 + f * cap_union * (d + 2) * 4 + f * cap_union * d * 4
    return {"dense": int(dense), "sparse": int(sparse)}


def auto_item_exchange(item_num: int, d: int, world_size: int, max_local_batch: int, global_batch: int | None = None) -> str:
    """'sparse' (touched rows only) or 'dense' (the whole item table) for a user-sharded SGD step - like
    `auto_exchange_slices` a pure function of numbers every rank holds identically.  Sparse when it puts at most half of
    the dense exchange's bytes on the wire (its pack / unpack passes and the O(I) mask are not free): at d = 64, 8 ranks,
    I = 1 M that is B_local <= ~30 000; configs[2] at B_local >= 2 M touches nearly every item and stays dense."""
    if world_size <= 1:
        return "dense"
    b = item_exchange_bytes(item_num, d, world_size, max_local_batch, global_batch)
    return "sparse" if 2 * b["sparse"] <= b["dense"] else "dense"


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
                 gamma=1e-10, item_mode=N.ITEM_FUSED, group=None, overlap=True, always_collective=False, slices=1,
                 adam_steps=0, dense_opt=None, auto_batch=None, exchange="dense", global_batch=None):
        """adam_steps > 0: torch.optim.Adam instead of SGD (staged protocol only; the value sizes the table of per-step
        constants, it grows on demand): lazy on the rank's rows of P, dense on its own block(s) of Q (ops.ShardedAdam)
        exchange: 'dense' (the whole item table per step), 'sparse' (the union of the ranks' touched rows: module docstring)
        or 'auto' (`auto_item_exchange`); its buffers are sized by the LARGEST share of a step a rank can hold (the
        context's batch) and by `global_batch` (samples of a step over all ranks; default world x the context's batch);
        sparse needs the staged protocol with SGD and runs unsliced"""
        self.ctx, self.P, self.Q = ctx, P_local, Q
        self.user_lo = int(user_lo)
        self.lr, self.reg_1, self.reg_2 = float(lr), float(reg_1), float(reg_2)
        self.loss_type, self.gamma, self.item_mode = int(loss_type), float(gamma), int(item_mode)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.overlap = bool(overlap)
        # issue the collectives even in a group of one rank (they are identities there): lets a single-GPU box
        # drive the real RCCL entry points
        self.collective = self.world > 1 or (bool(always_collective) and dist.is_initialized())
        # dense_opt: an `ops.DenseOptimizer` (next_step() / step(W, g): consumes and clears g) - the dense-optimiser protocol
        self.dense = dense_opt
        if dense_opt is not None:
            if self.item_mode == N.ITEM_FUSED:
                self.item_mode = N.ITEM_CHUNKED
            self.gP = torch.zeros_like(P_local)
        self.staged = dense_opt is None and self.item_mode == N.ITEM_FUSED and hasattr(ctx, "staged_user")
        # FM (FMRecommender.py:61-95) on the staged protocol, SGD: u_bias rows live with their users (the user pass
        # updates the rank's slice in place), i_bias is replicated like Q - its gradient (one float per item, written
        # by the item pass next to gQ) is all-reduced and every rank applies the same update -, bias_ follows from the
        # all-reduced sum of the coefficients.  Without these exchanges the replicas would drift apart silently.
        self.fm = getattr(ctx, "_bias", None)
        if self.fm is not None and self.dense is None and (not self.staged or adam_steps):
            raise NotImplementedError("UserShardedBprTrainer: FM biases need the staged protocol (item_mode 'fused') "
                                      "with SGD, or a dense optimiser (dense_opt)")
        if self.item_mode == N.ITEM_FUSED and not self.staged:
            self.item_mode = N.ITEM_CHUNKED
        # collectives the backend lacks are emulated with the ones it has (gloo: no reduce_scatter);
        # "nccl" (= RCCL on ROCm) runs the real ones
        self._native_rs = self.collective and dist.get_backend(group) == "nccl"
        # auto_batch: interactions per rank and step, the same number on every rank (default: the context's batch)
        per_rank = int(auto_batch if auto_batch is not None else getattr(ctx, "max_batch", 0))
        max_local = int(getattr(ctx, "max_batch", 0) or per_rank)
        gbatch = int(global_batch) if global_batch else None
        can_sparse = self.staged and not adam_steps and self.world > 1 and max_local > 0
        if exchange == "auto":
            exchange = auto_item_exchange(Q.shape[0], Q.shape[1], self.world, max_local, gbatch) if can_sparse else "dense"
        if exchange not in ("dense", "sparse"):
            raise ValueError(f"UserShardedBprTrainer: exchange={exchange!r} (expected 'dense', 'sparse' or 'auto')")
        if exchange == "sparse" and not (self.staged and not adam_steps):
            raise NotImplementedError("UserShardedBprTrainer: the touched-rows exchange needs the staged protocol with SGD "
                                      "(under dense Adam every item row moves in every step)")
        self.sparse = exchange == "sparse" and self.collective and max_local > 0
        if slices in ("auto", 0, None):      # decided from numbers all ranks share: see auto_exchange_slices
            backend = dist.get_backend(group) if self.collective else "none"
            slices = auto_exchange_slices(Q.shape[0], Q.shape[1], self.world, per_rank, backend)
        self.slices = min(16, max(1, int(slices))) if self.staged and hasattr(ctx, "staged_item_slice") else 1
        if self.sparse:
            self.slices = 1
        if self.collective and self.world > 1:
            # every rank must cut and own the same blocks: a rank whose environment (DAISY_XGMI_BUS_GBS) or arguments
            # resolved differently would diverge silently - compare instead of trusting
            kb = (max_local, gbatch or 0)
            mine = torch.tensor([self.slices, -self.slices, int(self.sparse), -int(self.sparse), kb[0], -kb[0], kb[1], -kb[1]],
                                dtype=torch.int64, device=Q.device)
            dist.all_reduce(mine, op=dist.ReduceOp.MAX, group=group)
            lo_hi = [int(x) for x in mine.cpu()]
            if lo_hi[0] != -lo_hi[1] or lo_hi[2] != -lo_hi[3] or (self.sparse and (lo_hi[4] != -lo_hi[5] or lo_hi[6] != -lo_hi[7])):
                raise RuntimeError(f"UserShardedBprTrainer: the ranks disagree on the exchange (slices {-lo_hi[1]}..{lo_hi[0]}, "
                                   f"sparse {-lo_hi[3]}..{lo_hi[2]}, largest local batch {-lo_hi[5]}..{lo_hi[4]}, global batch "
                                   f"{-lo_hi[7]}..{lo_hi[6]}): pass the same slices / exchange / batch sizes on every rank")
        self.side = None
        self.timeline = None         # enable_timing(): per-step (start, compute queued, end) events
        if self.staged:
            I, d = Q.shape
            S = self.slices
            # item rows per owner AND slice; slice s = rows [s*world*rows, (s+1)*world*rows), its r-th block is rank r's
            self.rows = (I + self.world * S - 1) // (self.world * S)
            Ipad = self.rows * self.world * S
            self.bounds = [min(s_ * self.world * self.rows, I) for s_ in range(S)] + [Ipad]
            self.side = torch.cuda.Stream(device=Q.device) if (S > 1 and Q.is_cuda) else None
            self.gQ = torch.zeros(Ipad, d, dtype=torch.float32, device=Q.device)
            self.cnt = torch.zeros(Ipad, 2, dtype=torch.float32, device=Q.device)
            self.g_own = torch.zeros(self.rows, d, dtype=torch.float32, device=Q.device)
            self.c_own = torch.zeros(self.rows, 2, dtype=torch.float32, device=Q.device)
            self.Q_gather = Q if Ipad == I else torch.zeros(Ipad, d, dtype=torch.float32, device=Q.device)
            self.own_lo = self.rank * self.rows                     # (of slice 0; slice s: + s*world*rows)
            self.own_hi = min(self.own_lo + self.rows, I)
            if self.sparse:
                # gQ / cnt get one row more: row I stays zero and is what the padding of the id lists points at
                self.cap_local, self.cap_union, self.rows_s = sparse_exchange_caps(I, self.world, max_local, gbatch)
                dev = Q.device
                self.gQ = torch.zeros(I + 1, d, dtype=torch.float32, device=dev)
                self.cnt = torch.zeros(I + 1, 2, dtype=torch.float32, device=dev)
                self.ids_all = torch.empty(self.world * self.cap_local, dtype=torch.int32, device=dev)
                self.mask = torch.zeros(I + 1, dtype=torch.bool, device=dev)
                self.xbuf = torch.zeros(self.cap_union, d + 2, dtype=torch.float32, device=dev)
                self.x_own = torch.zeros(self.rows_s, d + 2, dtype=torch.float32, device=dev)
                self.q_all = torch.zeros(self.cap_union, d, dtype=torch.float32, device=dev)
                self.k_arange = torch.arange(self.cap_union, device=dev)
        # bytes this rank puts on the wire per step for the item exchange (ring collectives; bench.py reports both forms)
        self.wire_bytes = item_exchange_bytes(Q.shape[0], Q.shape[1], self.world, max(max_local, 1), gbatch, self.slices)
        self.wire_bytes["used"] = "sparse" if getattr(self, "sparse", False) else "dense"
        self.adam = None
        if adam_steps:
            if not self.staged:
                raise NotImplementedError("UserShardedBprTrainer: Adam needs the staged protocol (item_mode 'fused')")
            from .ops import ShardedAdam
            self.adam = ShardedAdam(P_local, self.rows * self.slices, Q.shape[1], lr, adam_steps)

    # -- instrumentation -------------------------------------------------------------------------
    def enable_timing(self, on=True):
        """Record three events per step on the step's stream: start / the step's own kernels all queued, the item
        exchange about to be waited for / end.  `step_split()` then reports, per step, the time up to the second
        event ("compute": kernels plus the two small all-reduces they wait for) and the time after it ("exposed
        exchange": reduce-scatter, owner update, all-gather as far as they did not hide under the item pass)."""
        self.timeline = [] if (on and self.Q.is_cuda) else None

    def _mark(self, which):
        if self.timeline is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.Q.device))
        if which == 0:
            self.timeline.append([ev, None, None])
        else:
            self.timeline[-1][which] = ev

    def step_split(self, last=None):
        """(compute_ms, exposed_exchange_ms) averaged over the last `last` timed steps (synchronises)"""
        if not self.timeline:
            return None
        torch.cuda.synchronize(self.Q.device)
        rows = [t for t in self.timeline if t[1] is not None and t[2] is not None]
        rows = rows[-last:] if last else rows
        if not rows:
            return None
        comp = sum(t[0].elapsed_time(t[1]) for t in rows) / len(rows)
        exch = sum(t[1].elapsed_time(t[2]) for t in rows) / len(rows)
        return comp, exch

    # -- collectives -----------------------------------------------------------------------------
    def _all_reduce(self, t, async_op=False):
        if not self.collective:
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _reduce_scatter(self, out, full):
        """out[rows] = sum over ranks of full[rank*rows:(rank+1)*rows]  (`full`: world*rows rows - the whole padded
        table, or one slice of it)"""
        if not self.collective:
            out.copy_(full[self.rank * self.rows:(self.rank + 1) * self.rows])
        elif self._native_rs:
            dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(full[self.rank * self.rows:(self.rank + 1) * self.rows])

    def _all_gather_rows(self, full, own):
        """full[r*rows:(r+1)*rows] = rank r's `own`; `own` may be that very slice of `full`"""
        if not self.collective:
            return
        if self._native_rs:
            dist.all_gather_into_tensor(full, own, group=self.group)
        else:
            parts = [torch.empty_like(own) for _ in range(self.world)]
            dist.all_gather(parts, own.clone(), group=self.group)
            for r, p in enumerate(parts):
                full[r * self.rows:(r + 1) * self.rows].copy_(p)

    # -- entry points ----------------------------------------------------------------------------
    def step_from_triples(self, triples, idx=None, start=0, B=None, validate=True):
        """One global step; `triples` are this rank's rows (global user ids)."""
        c = self.ctx
        c.set_batch_from_triples(triples, idx=idx, start=start, B=B, user_base=self.user_lo, validate=validate)
        return self._step()

    def step_from_plan(self, plan, k):
        """One global step on batch k of this rank's epoch plan.  A plan that holds a rank's share of a global
        epoch (`EpochPlan.build_positions`) may have no row of batch k (or `plan` is None: the rank owns no
        interaction at all): the rank then only takes part in the exchanges."""
        rows = getattr(plan, "batch_rows", None)
        if plan is None or (rows is not None and rows(k) == 0):
            return self._step_empty()
        self.ctx.set_batch_from_plan(plan, k)
        return self._step()

    def step(self, u_local, i, j, validate=True):
        self.ctx.set_batch(u_local, i, j, validate=validate)
        return self._step()

    def _step(self):
        if self.dense is not None:
            return self._step_dense(False)
        return self._step_staged() if self.staged else self._step_phases()

    def _step_dense(self, empty):
        """dense-optimiser protocol; `empty`: this rank holds no sample of the step (it contributes zero sums and zero
        gradients, and its rows still take the optimiser's step: Adam's and RMSprop's state moves without a gradient)"""
        c = self.ctx
        self.dense.next_step()
        if empty:
            c.stats.zero_()
        else:
            c.forward(self.P, self.Q, self.loss_type, self.gamma)
        self._all_reduce(c.stats[:7])
        c.finalize(self.reg_1, self.reg_2)                     # every rank: the GLOBAL loss and norms
        if not empty:
            c.item_grad(self.P, self.Q, self.reg_1, self.reg_2, self.item_mode)     # -> c.gQ (FM: g_i_bias)
        work = self._all_reduce(c.gQ, async_op=self.overlap)
        if not empty:
            c.user_grad(self.P, self.Q, self.reg_1, self.reg_2, self.gP)            # overlaps the all-reduce (FM: g_u_bias, g_bias)
        if work is not None:
            work.wait()
        if self.P.numel():
            self.dense.step(self.P, self.gP)
        self.dense.step(self.Q, c.gQ)
        if self.fm is not None:
            u_bias, i_bias, bias, g_u, g_i, g_b = self.fm
            if empty and g_b is not None:
                g_b.zero_()
            self._all_reduce(g_i)
            self._all_reduce(g_b)
            if u_bias.numel():
                self.dense.step(u_bias, g_u)
            self.dense.step(i_bias, g_i)
            self.dense.step(bias, g_b)
        return c.stats

    def _step_staged(self):
        c, I = self.ctx, self.Q.shape[0]
        self._mark(0)
        if self.adam is not None:
            self.adam.next_step()
            c.staged_adam_catchup_users(self.P, self.adam)       # the batch's rows of P -> step t-1, before anything reads them
        c.staged_prenorm(self.P)
        self._all_reduce(c.stats[N.ST_SQ_U_PRE:N.ST_SQ_U_PRE + 1])
        if self.adam is not None:
            c.staged_user_adam(self.P, self.Q, self.adam, self.reg_1, self.reg_2, self.loss_type, self.gamma)
        else:
            c.staged_user(self.P, self.Q, self.lr, self.reg_1, self.reg_2, self.loss_type, self.gamma)
        w0 = self._all_reduce(c.stats[:7], async_op=self.overlap)
        if self.fm is not None:                          # FM: dL/d bias_ = the sum of the coefficients over the GLOBAL batch
            self._all_reduce(c.stats[N.ST_SUM_COEF:N.ST_SUM_COEF + 1])
        if self.slices == 1:
            c.staged_item(self.lr, self.reg_1, self.reg_2, gQ=self.gQ[:I], cnt=self.cnt[:I],
                          loss_type=self.loss_type)      # overlaps the 56-byte all-reduce
            if w0 is not None:
                w0.wait()
            c.finalize(self.reg_1, self.reg_2)           # every rank: the GLOBAL loss and norms
            self._mark(1)
            self._fm_bias_step()
            out = self._exchange_items()
            self._mark(2)
            return out
        # item pass range by range; the exchange of range s runs on the side stream under the pass over range s+1
        c.staged_item_slices(self.bounds)
        fin = None
        for s_ in range(self.slices):
            c.staged_item_slice(s_, self.lr, self.reg_1, self.reg_2, self.gQ[:I], self.cnt[:I], loss_type=self.loss_type)
            if s_ == 0:
                if w0 is not None:
                    w0.wait()
                c.finalize(self.reg_1, self.reg_2)
            self._exchange_slice(s_)
        self._mark(1)
        self._fm_bias_step()
        self._join_side()
        self._mark(2)
        return c.stats

    def _fm_bias_step(self):
        """FM: i_bias -= lr * (all-reduced item-bias gradient), bias_ -= lr * (all-reduced coefficient sum) - the same
        arithmetic on every rank, so the replicas stay identical (FMRecommender.py:61-68; SGD)"""
        if self.fm is None:
            return
        _, i_bias, bias, _, g_i_bias, _ = self.fm
        self._all_reduce(g_i_bias)
        i_bias.view(-1).sub_(g_i_bias, alpha=self.lr)
        g_i_bias.zero_()
        bias.view(-1).sub_((self.lr * self.ctx.stats[N.ST_SUM_COEF]).to(bias.dtype))

    def _on_side(self):
        """context: the side stream, ordered behind everything queued on the current stream so far"""
        import contextlib
        if self.side is None:
            return contextlib.nullcontext()
        self.side.wait_stream(torch.cuda.current_stream(self.Q.device))
        return torch.cuda.stream(self.side)

    def _join_side(self):
        if self.side is not None:
            torch.cuda.current_stream(self.Q.device).wait_stream(self.side)

    def _exchange_slice(self, s_):
        """reduce-scatter (gQ, cnt) of item slice s_ -> the owner's SGD on its block -> all-gather of the slice's rows"""
        c, I = self.ctx, self.Q.shape[0]
        w = self.world * self.rows
        a = s_ * w                                        # first row of the slice (padded numbering)
        with self._on_side():
            self._reduce_scatter(self.g_own, self.gQ[a:a + w])
            self._reduce_scatter(self.c_own, self.cnt[a:a + w])
            self.gQ[a:a + w].zero_()
            self.cnt[a:a + w].zero_()
            lo = a + self.rank * self.rows
            hi = min(lo + self.rows, I)
            n_own = max(hi - lo, 0)
            if n_own > 0 and self.adam is not None:     # dense Adam over the owned block (moments of slice s_'s block)
                mo = s_ * self.rows
                c.item_apply_counts_adam(self.Q[lo:hi], self.g_own[:n_own], self.c_own[:n_own], self.adam.mQ[mo:mo + n_own],
                                         self.adam.vQ[mo:mo + n_own], self.adam, self.reg_1, self.reg_2)
            elif n_own > 0:
                c.item_apply_counts(self.Q[lo:hi], self.g_own[:n_own], self.c_own[:n_own], self.lr, self.reg_1, self.reg_2)
            if self.collective:
                if self.Q_gather is self.Q:
                    self._all_gather_rows(self.Q[a:a + w], self.Q[lo:lo + self.rows])
                else:                                    # I not a multiple of world*slices: padded staging
                    own = self.Q_gather[lo:lo + self.rows]
                    if n_own > 0:
                        own[:n_own].copy_(self.Q[lo:hi])
                    self._all_gather_rows(self.Q_gather[a:a + w], own)
                    top = min(a + w, I)
                    if top > a:
                        self.Q[a:top].copy_(self.Q_gather[a:top])

    def _exchange_items_sparse(self):
        """the touched-rows exchange (module docstring): static shapes, no host synchronisation, the same union - hence
        the same owners and the same arithmetic - on every rank"""
        c, Q = self.ctx, self.Q
        I, d, W = Q.shape[0], Q.shape[1], self.world
        touched = (self.cnt[:I, 0] + self.cnt[:I, 1]) > 0
        ids = torch.nonzero_static(touched, size=self.cap_local, fill_value=I).view(-1)       # ascending, padded with I
        mine32 = ids.to(torch.int32)
        if self._native_rs:
            dist.all_gather_into_tensor(self.ids_all, mine32, group=self.group)
        else:
            parts = [torch.empty_like(mine32) for _ in range(W)]
            dist.all_gather(parts, mine32, group=self.group)
            torch.cat(parts, out=self.ids_all)
        self.mask.zero_()
        self.mask[self.ids_all.long()] = True
        uid = torch.nonzero_static(self.mask[:I], size=self.cap_union, fill_value=I).view(-1)   # the union, ascending
        self.xbuf[:, :d] = self.gQ.index_select(0, uid)                # (row I of gQ / cnt is the zero row)
        self.xbuf[:, d:] = self.cnt.index_select(0, uid)
        r0 = self.rank * self.rows_s
        if self._native_rs:
            dist.reduce_scatter_tensor(self.x_own, self.xbuf, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(self.xbuf, op=dist.ReduceOp.SUM, group=self.group)
            self.x_own.copy_(self.xbuf[r0:r0 + self.rows_s])
        self.gQ.index_fill_(0, ids, 0.0)                                # the rank's touched rows (and row I) back to zero
        self.cnt.index_fill_(0, ids, 0.0)
        my = uid[r0:r0 + self.rows_s]
        q_rows = Q.index_select(0, my.clamp(max=I - 1))                 # (padding: some row, left as it is, never written back)
        g_own, c_own = self.x_own[:, :d].contiguous(), self.x_own[:, d:].contiguous()
        c.item_apply_counts(q_rows, g_own, c_own, self.lr, self.reg_1, self.reg_2)
        if self._native_rs:
            dist.all_gather_into_tensor(self.q_all, q_rows, group=self.group)
        else:
            parts = [torch.empty_like(q_rows) for _ in range(W)]
            dist.all_gather(parts, q_rows, group=self.group)
            torch.cat(parts, out=self.q_all)
        # unpack without a data-dependent shape: the padding aliases entry 0 (same target, same row: duplicate writes of
        # one value); a step nobody contributed a sample to (uid[0] == I) rewrites row I-1 with itself
        valid = uid < I
        src = torch.where(valid, self.k_arange, torch.zeros_like(self.k_arange))
        tgt = uid.index_select(0, src)
        rows = self.q_all.index_select(0, src)
        none = tgt >= I
        tgt = tgt.clamp(max=I - 1)
        rows = torch.where(none[:, None], Q.index_select(0, tgt), rows)
        Q.index_copy_(0, tgt, rows)
        return c.stats

    def _exchange_items(self):
        if self.sparse:
            return self._exchange_items_sparse()
        for s_ in range(self.slices):
            self._exchange_slice(s_)
        self._join_side()
        return self.ctx.stats

    def _step_empty(self):
        """This rank's part of a global step to which it contributes no sample (staged / dense-optimiser protocol)."""
        if self.dense is not None:
            return self._step_dense(True)
        if not self.staged:
            raise NotImplementedError("an empty local batch is only supported by the staged protocol")
        c = self.ctx
        self._mark(0)
        if self.adam is not None:
            self.adam.next_step()
        c.stats.zero_()
        self._all_reduce(c.stats[N.ST_SQ_U_PRE:N.ST_SQ_U_PRE + 1])
        self._all_reduce(c.stats[:7])
        if self.fm is not None:
            self._all_reduce(c.stats[N.ST_SUM_COEF:N.ST_SUM_COEF + 1])
        c.finalize(self.reg_1, self.reg_2)
        self._mark(1)
        self._fm_bias_step()
        out = self._exchange_items()
        self._mark(2)
        return out

    def _step_phases(self):
        c = self.ctx
        c.forward(self.P, self.Q, self.loss_type, self.gamma)
        split = self.item_mode == N.ITEM_CHUNKED and hasattr(c, "item_grad_data")
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
        c.item_sgd_apply(self.Q, self.lr, dense=self.collective)
        return c.stats


class RowShardedPropagation:
    """LightGCN's propagation over the GPUs of one node (BASELINE configs[4]; the reference is single device,
    LightGCNRecommender.py:117-129).  A_hat is row-sharded by node range: rank r owns rows [r*R, (r+1)*R) of every
    product Y = A_hat X (R = ceil(N / world)), computes them from the full X it holds, and the ranks all-gather
    their row blocks (RCCL over xGMI) - one all-gather of N*d floats per layer, forward and backward.  Everything
    else of the step (batch loss, gradient wrt the propagated rows, regulariser, optimiser) is replicated: it
    touches 3B rows, the products touch every edge.

    `graph` is a daisyrec_amd.ops.LgcnGraph holding the WHOLE adjacency (4.7 M entries x 20 B for Amazon-Book: the
    matrix is small, the work is not), or a stand-in with the same `spmm_rows`."""

    def __init__(self, graph, N, d, device, group=None, pieces=None):
        """pieces: each layer's row block is produced in that many sub-blocks, and sub-block k is all-gathered on a side
        stream while sub-block k+1 is reduced (default: 4 on RCCL, 1 elsewhere) - at Amazon-Book size a layer's
        exchange (37 MB) outweighs its compute on 8 GPUs, so the two must at least not be serialised as well."""
        self.g, self.N, self.d, self.group = graph, int(N), int(d), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._native = self.world > 1 and dist.get_backend(group) == "nccl"
        if pieces is None:
            pieces = 4 if self._native else 1
        self.pieces = max(1, int(pieces)) if self.world > 1 else 1
        rows = (self.N + self.world - 1) // self.world
        self.prow = (rows + self.pieces - 1) // self.pieces           # rows per sub-block
        self.rows = self.prow * self.pieces                            # rows per rank (padded to a multiple of the pieces)
        self.lo = min(self.rank * self.rows, self.N)
        self.hi = min(self.lo + self.rows, self.N)
        is_cuda = torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if (self.pieces > 1 and is_cuda) else None
        # one buffer per sub-block: its rows + a spare row on either side (the segmented reduction may spill a partial
        # neighbour row there); pieces in flight must not share storage
        self.blocks = [torch.zeros(self.prow + 2, d, dtype=torch.float32, device=device) for _ in range(self.pieces)]
        self.block = self.blocks[0]
        # gathered sub-blocks: full[p][r] = rank r's sub-block p, i.e. node rows [r*rows + p*prow, +prow)
        self.full = torch.zeros(self.pieces, self.world, self.prow, d, dtype=torch.float32, device=device)
        self.work = [torch.empty(self.N, d, dtype=torch.float32, device=device) for _ in range(2)]

    def spmm(self, X, Y):
        """Y = A_hat X on every rank"""
        if self.world == 1:
            own = self.g.spmm_rows(X, self.blocks[0], self.lo, self.hi)
            Y.copy_(own)
            return Y
        cur = torch.cuda.current_stream(X.device) if self.side is not None else None
        for p in range(self.pieces):
            a = min(self.lo + p * self.prow, self.N)
            b = min(a + self.prow, self.hi)
            blk = self.blocks[p]
            if b > a:
                self.g.spmm_rows(X, blk, a, b)
            mine = blk[1:1 + self.prow]                   # (rows past the rank's / the table's end: don't care)
            if self.side is not None:                     # the exchange of this piece runs under the next piece's reduction
                self.side.wait_stream(cur)
                with torch.cuda.stream(self.side):
                    self._gather(self.full[p], mine)
            else:
                self._gather(self.full[p], mine)
        if self.side is not None:
            cur.wait_stream(self.side)
        # node order: rank r, piece p, row i  <-  full[p][r][i]
        Y.copy_(self.full.permute(1, 0, 2, 3).reshape(self.world * self.rows, self.d)[:self.N])
        return Y

    def _gather(self, out, mine):
        """out[r] = rank r's `mine`"""
        if self._native:
            dist.all_gather_into_tensor(out.view(self.world * self.prow, self.d), mine, group=self.group)
        else:
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine.clone(), group=self.group)
            for r, part in enumerate(parts):
                out[r].copy_(part)

    def propagate(self, E0, num_layers, out):
        """out = mean_k A_hat^k E0  (LightGCNRecommender.py:117-129)"""
        from . import ops
        out.copy_(E0)
        x = E0
        for k in range(num_layers):
            y = self.spmm(x, self.work[k & 1])
            ops.axpby(y, 1.0, 1.0, out)                   # out += E_{k+1}
            x = y
        out.mul_(1.0 / (num_layers + 1))
        return out

    def backprop(self, G, num_layers, dE0):
        """dE0 += 1/(L+1) sum_k A_hat^k G  (Horner; A_hat is symmetric)"""
        from . import ops
        t = G
        for k in range(num_layers):
            y = self.spmm(t, self.work[k & 1])
            ops.axpby(G, 1.0, 1.0, y)                     # T <- G + A_hat T
            t = y
        ops.axpby(t, 1.0 / (num_layers + 1), 1.0, dE0)
        return dE0
