"""Host-side mirror of the reference's recommender base classes for the MF hot
path (reference: daisy/model/AbstractRecommender.py:10-137).

Same class names, constructor keys, attributes and method set, so callers such
as run_examples/test.py:90-95,120 keep working; but ``GeneralRecommender.fit``
drives the HIP kernels of ``libdaisyrec_hip.so`` instead of autograd +
``torch.optim``.  There is no CPU path: ``fit``/``rank`` raise when no HIP
device is present.
"""
from __future__ import annotations

import logging
import os

import torch
import torch.nn as nn

from .. import ops

try:  # the reference shows a tqdm bar (AbstractRecommender.py:116); optional here
    from tqdm import tqdm as _tqdm
except Exception:  # pragma: no cover
    _tqdm = None

NATIVE_OPTIMIZERS = ("sgd", "adam", "adagrad", "rmsprop")
KNOWN_OPTIMIZERS = ("adam", "sgd", "adagrad", "rmsprop", "sparse_adam")


# config['lazy_adam'] (Adam only).  With the default item_mode 'fused' the staged step's row owners ALWAYS apply the lazy
# form (bit-equal to the dense optimiser after every flush) unless lazy_adam is False, which routes the fit to the phase
# kernels + torch's dense pass.  For the explicit item modes 'chunked' / 'sorted' / 'atomic', True / 'auto' select
# ops.LazyAdam's row updates behind the phase kernels; there
# 'auto' picks the lazy Adam only when the tables are large enough for the dense pass to matter: below this many
# table bytes (ml-100k scale: 2.6 k rows) the dense optimiser is one trivial launch and the lazy form only adds two
# launches per step plus serial replays at every catch-up / flush
LAZY_ADAM_MIN_TABLE_BYTES = 64 << 20


def _parse_lazy_adam(value):
    """config['lazy_adam']: True / False / 'auto' (yaml and CLI strings included); anything else is an error -
    bool('false') would silently be True"""
    if isinstance(value, bool):
        return value
    if isinstance(value, (int, float)) and value in (0, 1):
        return bool(value)
    key = str(value).strip().lower()
    if key == "auto":
        return "auto"
    if key in ("true", "1", "yes", "on"):
        return True
    if key in ("false", "0", "no", "off"):
        return False
    raise ValueError(f"config['lazy_adam'] must be True, False or 'auto', got {value!r}")


class AbstractRecommender(nn.Module):
    """Reference: AbstractRecommender.py:10-93."""

    def __init__(self):
        super().__init__()
        self.optimizer = None
        self.initializer = None
        self.loss_type = None
        self.lr = 0.01
        self.logger = None
        # AbstractRecommender.py:19-31
        self.initializer_param_config = {
            "normal": {"mean": 0.0, "std": 0.01},
            "uniform": {"a": 0.0, "b": 1.0},
            "xavier_normal": {"gain": 1.0},
            "xavier_uniform": {"gain": 1.0},
        }
        self.initializer_config = {
            "normal": nn.init.normal_,
            "uniform": nn.init.uniform_,
            "xavier_normal": nn.init.xavier_normal_,
            "xavier_uniform": nn.init.xavier_uniform_,
        }

    # abstract surface, AbstractRecommender.py:33-46
    def calc_loss(self, batch):
        raise NotImplementedError

    def fit(self, train_loader):
        raise NotImplementedError

    def rank(self, test_loader):
        raise NotImplementedError

    def full_rank(self, u):
        raise NotImplementedError

    def predict(self, u, i):
        raise NotImplementedError

    def _resolve_optimizer(self, name=None):
        """AbstractRecommender.py:48-67: unknown names fall back to Adam with a log line.  'sparse_adam'
        builds optim.SparseAdam in the reference, which refuses the dense gradients of its (non-sparse)
        nn.Embedding tables at the first step: the same RuntimeError is raised here (pinned by
        tests/golden/kat_optimizers.npz)."""
        name = str(self.optimizer if name is None else name).lower()
        if name not in KNOWN_OPTIMIZERS:
            if self.logger is not None:
                self.logger.info("Received unrecognized optimizer, set default Adam optimizer")
            name = "adam"
        if name == "sparse_adam":
            raise RuntimeError("SparseAdam does not support dense gradients, please consider Adam instead")
        assert name in NATIVE_OPTIMIZERS
        return name

    def _init_weight(self, m):
        """AbstractRecommender.py:69-77 (applied in module order from the global torch RNG)."""
        if isinstance(m, nn.Linear):
            self.initializer_config[self.initializer](m.weight, **self.initializer_param_config[self.initializer])
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif isinstance(m, nn.Embedding):
            self.initializer_config[self.initializer](m.weight, **self.initializer_param_config[self.initializer])

    def _build_criterion(self, loss_type):
        """AbstractRecommender.py:79-93.  Pairwise losses are an epilogue of the
        forward kernel; the returned value is the native loss id."""
        key = str(loss_type).upper()
        if key not in ops.LOSS_IDS:
            raise NotImplementedError(f"Invalid loss type: {self.loss_type}...")
        return ops.LOSS_IDS[key]


class GeneralRecommender(AbstractRecommender):
    """Reference: AbstractRecommender.py:95-137 (device selection + the training loop)."""

    def __init__(self, config):
        super().__init__()
        # AbstractRecommender.py:99.  Under a one-process-per-GPU launcher (torchrun sets LOCAL_RANK, or the caller
        # has initialised torch.distributed) the launcher has given this process its device: hiding all GPUs but
        # config['gpu'] here would put every rank on the same one.
        import torch.distributed as _dist
        if "LOCAL_RANK" not in os.environ and not (_dist.is_available() and _dist.is_initialized()):
            os.environ["CUDA_VISIBLE_DEVICES"] = config["gpu"]
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        # torchrun exports LOCAL_RANK but neither narrows CUDA_VISIBLE_DEVICES nor selects a device: without this every
        # rank would allocate on GPU 0 and RCCL would refuse the duplicate device.  'cuda' then means the current
        # device, i.e. this rank's own (a caller that already chose another device keeps it).
        if self.device == "cuda" and "LOCAL_RANK" in os.environ and torch.cuda.device_count() > 1:
            local_rank = int(os.environ["LOCAL_RANK"])
            if torch.cuda.current_device() == 0 and 0 < local_rank < torch.cuda.device_count():
                torch.cuda.set_device(local_rank)
        self.logger = config.get("logger") or logging.getLogger("daisyrec_amd")
        # knobs of the native path (absent from the reference config: defaults keep its behaviour)
        # 'fused' (default): the staged step over the partitioned epoch plan (forward fused into the user
        # pass, item rows committed by their segment owner) for SGD + pairwise loss without biases, the
        # chunked phase kernels otherwise; 'chunked': the phase kernels; 'sorted': every item row summed
        # serially in plan order.  All three are bitwise reproducible run to run (fixed summation order, no
        # atomics).  'atomic': fp32 atomics (kept for A/B measurements, not reproducible)
        self.item_mode = str(config.get("item_mode", "fused")).lower()
        self.show_progress = bool(config.get("progress", True))
        # 'loader' (default): replay the DataLoader's torch RNG order (what the reference run does);
        # 'device': shuffle=True as a keyed permutation computed on the GPU (no host permutation,
        # no upload: the scalable choice, same distribution, different stream)
        self.shuffle_mode = str(config.get("shuffle_mode", "loader")).lower()
        self.seed = int(config.get("seed", 2022))
        # under torch.distributed (one process per GPU): split the users over the ranks (default) or let every
        # rank train the whole model
        self.shard_users = bool(config.get("shard_users", True))
        # 'auto' (default): sharding.auto_exchange_slices - the item exchange is pipelined under the item pass when it
        # is worth cutting (RCCL, exchange bytes / assumed bus rate against the pass); an integer forces the count
        xs = config.get("exchange_slices", "auto")
        self.exchange_slices = "auto" if str(xs).lower() == "auto" else int(xs)
        # config['item_exchange'] (sharded SGD fits): 'auto' (default; sharding.auto_item_exchange), 'dense' (the whole item
        # table per step) or 'sparse' (the union of the ranks' touched rows - what pays at the reference's batch sizes)
        self.item_exchange = str(config.get("item_exchange", "auto")).lower()
        if self.item_exchange not in ("auto", "dense", "sparse"):
            raise ValueError(f"config['item_exchange']={self.item_exchange!r}: expected 'auto', 'dense' or 'sparse'")
        # MF + Adam: the exact lazy row updates of ops.LazyAdam.  'auto': when a step references fewer rows than the
        # tables have (3B < U + I; measured: 1.55x at 10M x 1M with B = 2M, but 0.9x at 1M x 100K with B = 1M, where
        # every step touches most rows anyway and the dense streaming pass is cheaper than row-wise claims)
        self.lazy_adam = _parse_lazy_adam(config.get("lazy_adam", "auto"))
        self.epoch_losses = []

    # -- helpers ---------------------------------------------------------------
    def _require_device(self):
        if self.device != "cuda":
            raise RuntimeError("daisyrec_amd: no HIP device visible; the MF hot path has no CPU fallback")

    @staticmethod
    def _epoch_order(train_loader, n):
        """The index order one `for batch in DataLoader` pass would use, consuming the
        global torch RNG exactly like the reference run (dataset.py:5-7 ->
        torch DataLoader): `_BaseDataLoaderIter.__init__` draws `_base_seed`, then
        RandomSampler draws its own seed and calls torch.randperm(n, generator)."""
        from torch.utils.data import RandomSampler, SequentialSampler

        torch.empty((), dtype=torch.int64).random_(generator=train_loader.generator)  # _base_seed
        sampler = train_loader.sampler
        if isinstance(sampler, SequentialSampler):
            return None
        if isinstance(sampler, RandomSampler) and not sampler.replacement and sampler.num_samples == n:
            if sampler.generator is None:
                seed = int(torch.empty((), dtype=torch.int64).random_().item())
                gen = torch.Generator()
                gen.manual_seed(seed)
            else:
                gen = sampler.generator
            return torch.randperm(n, generator=gen)
        return torch.as_tensor(list(iter(sampler)), dtype=torch.int64)

    def _sharded_world(self):
        """ranks that share this fit: torch.distributed initialised with more than one rank and
        config['shard_users'] not switched off (the reference is single device, AbstractRecommender.py:99)"""
        import torch.distributed as dist
        if not self.shard_users or not dist.is_available() or not dist.is_initialized():
            return 1
        world = dist.get_world_size()
        return world if world > 1 and self.embed_user.weight.shape[0] >= world else 1

    def _epoch_positions(self, train_loader, n, epoch, row_ids):
        """epoch position of this rank's rows: the order one pass over the DataLoader would use (replayed from the
        torch RNG exactly like the single-device fit, so every rank draws the same permutation), the keyed device
        shuffle, or the identity"""
        from torch.utils.data import SequentialSampler
        if self.shuffle_mode == "device" and not isinstance(train_loader.sampler, SequentialSampler):
            return ops.feistel_positions_at(row_ids, n, self.seed, epoch)
        perm = self._epoch_order(train_loader, len(train_loader.dataset))
        if perm is None:
            return row_ids.clone()
        perm = perm.to(row_ids.device)
        if perm.numel() != n:
            raise NotImplementedError("drop_last with a shuffled loader is not supported on the HIP path "
                                      "(the dropped rows change per epoch)")
        inv = torch.empty(n, dtype=torch.int64, device=row_ids.device)
        inv[perm] = torch.arange(n, dtype=torch.int64, device=row_ids.device)
        return inv[row_ids].contiguous()

    def _fit_sharded(self, train_loader, triples, n, B, loss_id, opt="sgd", biases=None, dense_adam=False):
        """`fit` over the ranks of a torch.distributed job (one process per GPU).  Every rank calls it
        with the same loader and the same seeds.  Users - with their interactions and their rows of P - are split
        into contiguous ranges; Q is replicated.  Batch k of rank r = its rows among positions [k*B, (k+1)*B) of
        the epoch order, so the union over the ranks IS batch k of the single-device fit and the result equals it
        up to summation order (UserShardedBprTrainer: two small all-reduces, reduce-scatter of the item
        gradient, all-gather of the updated item rows per step).  At the end every rank holds the whole P."""
        import torch.distributed as dist
        from torch.utils.data import SequentialSampler
        from ..sharding import UserShardedBprTrainer, user_range
        world, rank = dist.get_world_size(), dist.get_rank()
        P, Q = self._tables()
        U, I, d = P.shape[0], Q.shape[0], P.shape[1]
        dist.broadcast(P, 0)               # replicas start from rank 0's tables (same seeds make this a no-op)
        dist.broadcast(Q, 0)
        lo, hi = user_range(U, world, rank)
        u = triples[:n, 0]
        row_ids = torch.nonzero((u >= lo) & (u < hi)).flatten()        # ascending: CSR order is kept
        mine = triples[row_ids].contiguous()
        n_loc = int(mine.shape[0])
        held = torch.tensor([n_loc], dtype=torch.int64, device=P.device)
        dist.all_reduce(held)
        if int(held.item()) != n:          # a user id outside [0, user_num) belongs to no rank: what the index would report
            raise ValueError(f"index out of range in the training triples: need 0 <= user < {U} "
                             "(the reference raises IndexError in nn.Embedding, MFRecommender.py:64-65)")
        P_loc = P[lo:hi]
        ctx = ops.BprContext(B, d, hi - lo, I, device=P.device)        # stage slots are positions inside a GLOBAL batch
        # Adagrad / RMSprop, and Adam with FM's biases: the dense-optimiser protocol (phase kernels + torch's dense
        # optimisers: sharding.py); SGD (MF, FM) and Adam (MF): the staged protocol
        # (config['lazy_adam']=False with Adam: torch's dense pass, i.e. the dense-optimiser protocol as well)
        dense = opt in ("adagrad", "rmsprop") or (opt == "adam" and (biases is not None or dense_adam))
        if biases is not None:             # FM: the rank's slice of u_bias, the replicated i_bias / bias_ (sharding.py)
            for b in biases:
                dist.broadcast(b, 0)
            zeros = lambda m: torch.zeros(m, dtype=torch.float32, device=P.device)      # noqa: E731
            ctx.set_bias(biases[0].view(-1)[lo:hi], biases[1], biases[2], g_u_bias=zeros(hi - lo) if dense else None,
                         g_i_bias=zeros(I), g_bias=zeros(1) if dense else None)
        if pointwise_rows := (loss_id in ops.POINTWISE_LOSSES):
            ctx.set_pointwise(True)
        index = plan = None
        trainer = UserShardedBprTrainer(ctx, P_loc, Q, lo, self.lr, self.reg_1, self.reg_2, loss_type=loss_id,
                                        item_mode=ops.ITEM_MODES["fused"], slices=self.exchange_slices,
                                        auto_batch=max(1, B // world),     # (the context's batch is the GLOBAL one here)
                                        # SGD on the staged protocol: only the item rows a step touched travel when
                                        # 2 B << I (sharding.auto_item_exchange: a pure function of I, d, world, B)
                                        exchange=self.item_exchange if (opt == "sgd" and not dense) else "dense", global_batch=B,
                                        adam_steps=(self.epochs * ((n + B - 1) // B)) if (opt == "adam" and not dense) else 0,
                                        dense_opt=ops.DenseOptimizer(opt, self.lr) if dense else None)
        self.logger.info("sharded HIP fit over %d ranks: item exchange %s (%d bytes per step and rank on the wire; the other "
                         "form would move %d), %d slice(s)", world, trainer.wire_bytes["used"],
                         trainer.wire_bytes[trainer.wire_bytes["used"]],
                         trainer.wire_bytes["dense" if trainer.wire_bytes["used"] == "sparse" else "sparse"], trainer.slices)
        self.last_exchange = dict(trainer.wire_bytes, slices=trainer.slices)
        acc = torch.zeros(2, dtype=torch.float64, device=P.device)
        nb = (n + B - 1) // B
        last_loss = 0.0
        try:
            if n_loc and not dense:
                index = ops.TrainIndex(mine, hi - lo, I, user_base=lo, pointwise=pointwise_rows)
                plan = ops.EpochPlan(n_loc, hi - lo, I, device=P.device)
            for epoch in range(1, self.epochs + 1):
                self.train()
                if n_loc and dense:
                    # the phase kernels read one sorted batch at a time: the rank's rows in epoch order, cut where the
                    # global batches end
                    pos_loc = self._epoch_positions(train_loader, n, epoch, row_ids)
                    order = torch.argsort(pos_loc)
                    cuts = torch.searchsorted(pos_loc[order].contiguous(),
                                              torch.arange(nb + 1, device=P.device, dtype=torch.int64) * B).cpu().tolist()
                elif n_loc:
                    plan.build_positions(index, self._epoch_positions(train_loader, n, epoch, row_ids), n, B)
                elif not (self.shuffle_mode == "device" and not isinstance(train_loader.sampler, SequentialSampler)):
                    # a rank without rows: the replayed DataLoader pass draws from the torch RNG, which has to stay in
                    # step with the other ranks; the device shuffle draws nothing (and has no rows to place here)
                    self._epoch_order(train_loader, len(train_loader.dataset))
                acc.zero_()
                for k in range(nb):
                    if dense and n_loc and cuts[k + 1] > cuts[k]:
                        stats = trainer.step_from_triples(mine, idx=order[cuts[k]:cuts[k + 1]].contiguous(), validate=(epoch == 1))
                    else:
                        stats = trainer.step_from_plan(plan, k)        # (plan None / no row of batch k: the rank only joins the exchanges)
                    loss = stats[ops.N.ST_LOSS]
                    acc[0] += loss
                    acc[1] += (~torch.isfinite(loss)).to(torch.float64)
                host = acc.cpu()
                current_loss = float(host[0])
                if float(host[1]) > 0 or current_loss != current_loss:
                    raise ValueError("Loss=Nan or Infinity: current settings does not fit the recommender")
                self.epoch_losses.append(current_loss)
                self.eval()
                delta_loss = float(current_loss - last_loss)
                if (abs(delta_loss) < 1e-5) and self.early_stop:       # AbstractRecommender.py:132-137
                    self.logger.info("Satisfy early stop mechanism")
                    break
                last_loss = current_loss
            if trainer.adam is not None:                               # the rows no batch referenced lately -> the last step
                trainer.adam.flush(ctx)
            for r in range(world):                                     # every rank ends with the whole user table
                a, b = user_range(U, world, r)
                if b > a:
                    dist.broadcast(P[a:b], r)
                    if biases is not None:                             # (and with every user's bias)
                        dist.broadcast(biases[0].view(-1)[a:b], r)
        finally:
            torch.cuda.synchronize()
            ctx.close()
            if plan is not None:
                plan.close()
            if index is not None:
                index.close()

    def fit(self, train_loader):
        """AbstractRecommender.py:103-137, natively: one enqueue per epoch, one host
        sync per epoch (for the loss the early-stop rule needs)."""
        pitch = getattr(self, "row_pitch", None)
        try:
            return self._fit(train_loader)
        finally:
            if pitch is not None:
                self.row_pitch = pitch          # (a sharded fit trains without the automatic pitch: not a lasting change)

    def _fit(self, train_loader):
        self._require_device()
        self.to(self.device)
        opt = self._resolve_optimizer()
        loss_id = self._build_criterion(self.loss_type)
        item_mode = ops.ITEM_MODES[self.item_mode]
        data = getattr(train_loader.dataset, "data", None)
        if data is None:
            raise TypeError("fit expects a DataLoader over BasicDataset (dataset.data = int32 [N,3] triples)")
        triples = torch.as_tensor(data).to(torch.int32).contiguous().to(self.device)
        n = triples.shape[0]
        B = int(train_loader.batch_size)
        if train_loader.drop_last:
            n = (n // B) * B
        self.epoch_losses = []
        if n == 0:
            # an empty loader: the reference's batch loop does not run, every epoch's loss is 0.0
            # (AbstractRecommender.py:117-137)
            for _ in range(self.epochs):
                self.epoch_losses.append(0.0)
                if self.early_stop:
                    self.logger.info("Satisfy early stop mechanism")
                    break
            return
        B = min(B, n)            # fewer rows than one batch: a single partial batch, like the DataLoader
        if self._sharded_world() > 1 and getattr(self, "row_pitch", None) == "auto":
            # a sharded fit moves whole table rows over the wire (broadcasts, the item exchange): the automatic row pitch
            # would ship its zero columns too (+28 % at d = 50), so it stays off there; an explicit pitch is honoured
            self.row_pitch = 0
        P, Q = self._tables(batch=B)    # (the padded buffers behind embed_*.weight where the model trains on a row pitch)
        ctx = ops.BprContext(B, P.shape[1], P.shape[0], Q.shape[0], device=P.device)
        plan = ops.EpochPlan(n, P.shape[0], Q.shape[0], device=P.device)
        biases = self._biases() if hasattr(self, "_biases") else None     # FM: (u_bias, i_bias, bias_)
        pointwise = loss_id in ops.POINTWISE_LOSSES      # rows are (user, item, label), sampler.py:93-98
        # The staged step over the partitioned plan (forward fused into the user update, item rows committed by their
        # segment owner): SGD and Adam, every loss of loss.py, FM's biases.  Adagrad / RMSprop run the phase kernels with
        # the dense optimisers.  Batches of a few hundred samples go through the sorted plan instead: fit_epoch_sgd then
        # runs the whole epoch inside one persistent workgroup (SGD, pairwise, no biases; csrc/bpr_small.hip).
        if opt == "adam" and self.lazy_adam is False and item_mode == ops.ITEM_MODES["fused"]:
            # config['lazy_adam']=False: torch's dense Adam pass over both tables behind the phase kernels (the staged
            # step's row owners apply the exact LAZY form - same bits after every flush, but the knob means what it says)
            item_mode = ops.ITEM_MODES["chunked"]
        staged = item_mode == ops.ITEM_MODES["fused"] and opt in ("sgd", "adam") and B > ops.SMALL_BATCH_MAX
        self.logger.info("HIP fit: %s, optimizer %s, B=%d: %s", type(self).__name__, opt, B,
                         "staged step over the partitioned plan" + (" (row owners apply lazy Adam)" if opt == "adam" else "")
                         if staged else ("small-batch epoch kernel" if (opt == "sgd" and B <= ops.SMALL_BATCH_MAX
                                                                         and item_mode in (ops.ITEM_MODES["fused"], ops.ITEM_MODES["chunked"]))
                                         else "phase kernels + dense optimiser"))
        adam = (_AdamState(P, Q, self.lr, biases, kind=opt, max_steps=self.epochs * ((n + B - 1) // B),
                           lazy=((3 * B < P.shape[0] + Q.shape[0]
                                  and (P.numel() + Q.numel()) * 4 >= LAZY_ADAM_MIN_TABLE_BYTES)
                                 if self.lazy_adam == "auto" else self.lazy_adam))
                if opt != "sgd" else None)                                                # any dense optimiser but SGD
        if biases is not None:
            g_i_bias = adam.g[1] if adam is not None else torch.zeros(Q.shape[0], device=P.device)
            ctx.set_bias(*biases, g_u_bias=adam.g[0] if adam is not None else None, g_i_bias=g_i_bias,
                         g_bias=adam.g[2] if adam is not None else None)
        user_sorted = ops.triples_user_sorted(triples[:n])
        if self._sharded_world() > 1:
            dense_adam = opt == "adam" and self.lazy_adam is False and self.item_mode == "fused"
            if item_mode == ops.ITEM_MODES["fused"] or opt in ("adagrad", "rmsprop") or dense_adam:
                ctx.close()
                plan.close()
                if (opt in ("adagrad", "rmsprop") or dense_adam or (opt == "adam" and biases is not None)) \
                        and Q.numel() * 4 > 64 * B * Q.shape[1]:
                    self.logger.warning("sharded fit with a dense optimiser: every step all-reduces the dense item gradient "
                                        "(%d MB) for a batch of %d rows - at this ratio the unsharded fit "
                                        "(config['shard_users']=False) is likely faster", Q.numel() * 4 >> 20, B)
                return self._fit_sharded(train_loader, triples, n, B, loss_id, opt, biases, dense_adam=dense_adam)
            self.logger.info("torch.distributed is initialised, but SGD / Adam with an explicit item_mode other than "
                             "'fused' do not shard the users over the ranks: every rank trains the whole model")
        if item_mode == ops.ITEM_MODES["fused"] and not staged and B > ops.SMALL_BATCH_MAX:
            item_mode = ops.ITEM_MODES["chunked"]
        index = None
        last_loss = 0.0
        try:
            if staged:      # indexed once per fit (also validates the id ranges)
                index = ops.TrainIndex(triples[:n], P.shape[0], Q.shape[0], user_sorted=user_sorted, pointwise=pointwise)
            epochs = range(1, self.epochs + 1)
            bar = _tqdm(epochs) if (_tqdm is not None and self.show_progress) else None
            for epoch in (bar if bar is not None else epochs):
                self.train()
                from torch.utils.data import SequentialSampler
                if self.shuffle_mode == "device" and not isinstance(train_loader.sampler, SequentialSampler):
                    order, perm = "feistel", None
                else:
                    perm = self._epoch_order(train_loader, triples.shape[0])
                    if perm is not None:
                        if n < triples.shape[0]:            # drop_last: the first n positions of the order
                            perm = perm[:n]
                            if bool((perm >= n).any()):
                                raise NotImplementedError("drop_last with a shuffled loader is not supported on "
                                                          "the HIP path (the dropped rows change per epoch)")
                        perm = perm.contiguous().to(self.device)
                    order = "identity" if perm is None else "perm"
                # the epoch laid out batch by batch, in the DataLoader's order
                if staged:
                    plan.build_indexed(index, B, order=order, perm=perm, seed=self.seed, epoch=epoch)
                else:
                    plan.build(triples, B, order=order, perm=perm, seed=self.seed, epoch=epoch, n_triples=n,
                               user_sorted=user_sorted, pointwise=pointwise)
                ctx.epoch_acc.zero_()
                if adam is None:
                    ctx.fit_epoch_sgd(plan, P, Q, self.lr, self.reg_1, self.reg_2, loss_type=loss_id,
                                      item_mode=item_mode)
                elif adam.kind == "adam" and biases is None and (
                        staged or (self.lazy_adam is not False and item_mode in (ops.ITEM_MODES["fused"], ops.ITEM_MODES["chunked"])
                                   and ops.LazyAdam.small_epoch_pays(ctx, plan, loss_id))):
                    # the epoch as ONE enqueue (daisy_bpr_fit_epoch_adam), like the SGD loop: at the reference's batch
                    # sizes a host round trip per batch would bound the fit.  B <= 256 (sorted plan): every step of the
                    # epoch inside one persistent workgroup (csrc/bpr_small.hip, the Adam form)
                    adam.staged_epoch(ctx, plan, self.reg_1, self.reg_2, loss_id)
                else:
                    for k in range(plan.num_batches):
                        ctx.set_batch_from_plan(plan, k)
                        adam.step(ctx, P, Q, self.reg_1, self.reg_2, loss_id, item_mode)
                    adam.flush()
                acc = ctx.epoch_acc.cpu()
                current_loss = float(acc[0])
                if float(acc[1]) > 0 or current_loss != current_loss:
                    # AbstractRecommender.py:122-123 (checked once per epoch instead of per batch)
                    raise ValueError("Loss=Nan or Infinity: current settings does not fit the recommender")
                self.epoch_losses.append(current_loss)
                log_path = os.environ.get("DAISY_AMD_EPOCH_LOG")    # evidence trail of driver runs (tests/test_gpu_driver.py)
                if log_path:
                    with open(log_path, "a") as fh:
                        fh.write(f"{type(self).__name__} epoch {epoch} loss {current_loss!r}\n")
                if bar is not None:
                    bar.set_description(f"[Epoch {epoch:03d}]")
                    bar.set_postfix(loss=current_loss)
                self.eval()
                delta_loss = float(current_loss - last_loss)
                if (abs(delta_loss) < 1e-5) and self.early_stop:       # AbstractRecommender.py:132-137
                    self.logger.info("Satisfy early stop mechanism")
                    break
                last_loss = current_loss
        finally:
            torch.cuda.synchronize()
            ctx.close()
            plan.close()
            if index is not None:
                index.close()


class _AdamState:
    """Dense optimiser state (torch.optim.Adam by default; Adagrad / RMSprop through `kind`) for the two tables
    and, for FM, the three bias tensors (AbstractRecommender.py:54-61)."""

    def __init__(self, P, Q, lr, biases=None, kind="adam", max_steps=0, lazy=False):
        self.kind = kind
        self.opt = ops.DenseOptimizer(kind, lr)
        self._gP = None
        self._P = P
        # FM: gradient buffers of (u_bias, i_bias, bias_)
        self.w = [] if biases is None else [b.view(-1) for b in biases]
        self.g = [torch.zeros_like(b) for b in self.w]
        # lazy=True, Adam: the exact lazy form - rows without a gradient are replayed when they are next needed instead
        # of being rewritten in every step (ops.LazyAdam: same bits as the dense optimiser); between flush() calls only
        # the rows of the batches seen so far are current.  The staged step (item_mode 'fused') always runs it.
        self._lazy_args = (P, Q, lr, max_steps)
        self.lazy = ops.LazyAdam(P, Q, lr, max_steps) if (lazy and kind == "adam" and biases is None) else None
        self.lazy_staged = None

    @property
    def gP(self):
        if self._gP is None:             # (the staged step forms no dense user gradient)
            self._gP = torch.zeros_like(self._P)
        return self._gP

    def step(self, ctx, P, Q, reg_1, reg_2, loss_id, item_mode):
        if self.kind == "adam" and item_mode == ops.ITEM_MODES["fused"]:
            # the staged step: forward fused into the user update, item rows committed by their segment owner, every
            # owner applies torch's Adam to its row (moments read and written once per touched row)
            if self.lazy_staged is None:
                self.lazy_staged = self.lazy if self.lazy is not None else ops.LazyAdam(*self._lazy_args)
                self.lazy = None
            self.opt.next_step()
            self.lazy_staged.staged_step(ctx, reg_1, reg_2, loss_id)
            if self.w:                   # FM: the biases through the dense optimiser (three small vectors)
                self.g[2].copy_(ctx.stats[ops.N.ST_SUM_COEF:ops.N.ST_SUM_COEF + 1])
                for w, g in zip(self.w, self.g):
                    self.opt.step(w, g)
            return
        if item_mode == ops.ITEM_MODES["fused"]:
            item_mode = ops.ITEM_MODES["chunked"]
        if self.lazy is not None:
            self.lazy.catchup(ctx)            # the rows this batch reads are brought to the previous step first
            ctx.forward(P, Q, loss_id)
            ctx.finalize(reg_1, reg_2)
            ctx.item_grad(P, Q, reg_1, reg_2, item_mode)
            ctx.user_grad(P, Q, reg_1, reg_2, self.gP)
            self.lazy.step(ctx, self.gP, ctx.gQ)
            return
        self.opt.next_step()
        ctx.forward(P, Q, loss_id)
        ctx.finalize(reg_1, reg_2)
        ctx.item_grad(P, Q, reg_1, reg_2, item_mode)
        ctx.user_grad(P, Q, reg_1, reg_2, self.gP)
        self.opt.step(P, self.gP)
        self.opt.step(Q, ctx.gQ)          # also zeroes gQ
        for w, g in zip(self.w, self.g):
            self.opt.step(w, g)

    def staged_epoch(self, ctx, plan, reg_1, reg_2, loss_id):
        """one epoch of the staged Adam step in one native call, the lazy rows flushed at its end (MF; see `step`)"""
        if self.lazy_staged is None:
            self.lazy_staged = self.lazy if self.lazy is not None else ops.LazyAdam(*self._lazy_args)
            self.lazy = None
        self.opt.t += plan.num_batches          # (the dense optimiser's shared step count: FM-free fits never read it)
        self.lazy_staged.fit_epoch(ctx, plan, reg_1, reg_2, loss_id)

    def flush(self):
        """every row up to the current step (end of an epoch: before the tables are read by anything but a step)"""
        for lz in (self.lazy, self.lazy_staged):
            if lz is not None:
                lz.flush()
