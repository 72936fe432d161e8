"""Uniform negative sampler with the reference's interface, run on the GPU.

Mirror of daisy/utils/sampler.py:3-103 (``AbstractSampler`` / ``BasicNegtiveSampler``
[sic]): same constructor (df, config), same ``sampling()`` result — an int32
``(N*num_ng, 3)`` array of ``(user, pos_item, neg_item)`` in train-set row order where
every interaction of user ``u`` carries the same ``num_ng`` negatives, drawn uniformly
with replacement from the items ``u`` has not interacted with (sampler.py:84-89,91,100-101).

The point-wise CL / SL layouts (sampler.py:93-98) are produced too.  Like the reference, sampling
raises ValueError when some user id in range(user_num) has interacted with every item
(np.random.choice on an empty complement, sampler.py:84-89).

Not reproduced: numpy's MT19937 stream (a device generator cannot follow it; the draws
are Philox4x32-10 keyed by config['seed']) and the popularity-mixed branches
(sampler.py:64-80) — outside the uniform hot path.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


class AbstractSampler(object):
    def __init__(self, config):
        self.uid_name = config["UID_NAME"]
        self.iid_name = config["IID_NAME"]
        self.item_num = config["item_num"]
        self.ur = config.get("train_ur")

    def sampling(self):
        raise NotImplementedError


class BasicNegtiveSampler(AbstractSampler):
    def __init__(self, df, config):
        super().__init__(config)
        self.user_num = config["user_num"]
        self.num_ng = config["num_ng"]
        self.inter_name = config.get("INTER_NAME", "rating")
        self.sample_method = config.get("sample_method", "uniform")
        self.sample_ratio = config.get("sample_ratio", 0)
        self.loss_type = str(config["loss_type"]).upper()
        self.seed = int(config.get("seed", 2022))
        self.epoch = int(config.get("sampler_epoch", 0))
        self.device = config.get("device", "cuda")
        assert self.sample_method in ["uniform", "low-pop", "high-pop"], \
            f"Invalid sampling method: {self.sample_method}"
        assert 0 <= self.sample_ratio <= 1, "Invalid sample ratio value"
        self.df = df
        self.pop_prob = None
        if self.sample_method in ("high-pop", "low-pop"):                     # sampler.py:44-54
            pop = df.groupby(self.iid_name).size()
            pop = pop / pop.sum()
            if self.sample_method == "high-pop":
                norm_pop = np.zeros(self.item_num)
                norm_pop[pop.index] = pop.values
            else:
                norm_pop = np.ones(self.item_num)
                norm_pop[pop.index] = 1 - pop.values
            self.pop_prob = norm_pop / norm_pop.sum()

    def _train_pairs(self):
        """(users, items) that define each user's positives: config['train_ur'] when the
        caller provides it (test.py:70-71, tune.py use the total train set), else df."""
        if self.ur is not None:
            us = np.fromiter((u for u, s in self.ur.items() for _ in s), dtype=np.int32)
            it = np.fromiter((i for _, s in self.ur.items() for i in s), dtype=np.int32)
            return us, it
        # no train_ur given (the reference would fail on self.ur[u]): the positives are the distinct
        # (user, item) pairs of df - duplicates would inflate the CSR rows the sampler searches
        pairs = np.unique(np.stack([self.df[self.uid_name].to_numpy().astype(np.int32),
                                    self.df[self.iid_name].to_numpy().astype(np.int32)], 1), axis=0)
        return np.ascontiguousarray(pairs[:, 0]), np.ascontiguousarray(pairs[:, 1])

    def sampling_device(self):
        """Triples as an int32 [N*num_ng, 3] DEVICE tensor."""
        if self.num_ng == 0:
            if self.loss_type in ("CL", "SL"):                                          # sampler.py:58-59
                cols = [self.uid_name, self.iid_name, self.inter_name]
                return torch.from_numpy(self.df[cols].to_numpy().astype(np.int32)).to(self.device)
            raise NotImplementedError("loss function (BPR, TL, HL) need num_ng > 0")   # sampler.py:61
        if self.loss_type not in ("BPR", "HL", "TL", "CL", "SL"):
            raise NotImplementedError(f"Invalid loss type: {self.loss_type}")
        if not torch.cuda.is_available():
            raise RuntimeError("daisyrec_amd sampler needs a HIP device (no CPU fallback)")
        pu, pi = self._train_pairs()
        indptr, csr = ops.build_user_csr(torch.from_numpy(pu).to(self.device),
                                         torch.from_numpy(pi).to(self.device), self.user_num)
        # sampler.py:65-81: with 'high-pop' / 'low-pop', int(sample_ratio * num_ng) of a user's negatives are drawn
        # from the popularity distribution over ALL items (positives not excluded, as in the reference), the rest
        # uniformly from the complement of the user's row; columns in that order
        other_num = int(self.sample_ratio * self.num_ng) if self.pop_prob is not None else 0
        uniform_num = self.num_ng - other_num
        js = torch.empty(self.user_num, self.num_ng, dtype=torch.int32, device=self.device)
        if uniform_num > 0:
            js[:, :uniform_num] = ops.sample_neg_per_user(indptr, csr, self.item_num, uniform_num, self.seed, self.epoch)
        if other_num > 0:
            cdf = torch.from_numpy(np.cumsum(self.pop_prob.astype(np.float64))).to(self.device)
            ops.sample_categorical(cdf, self.user_num, other_num, self.seed, ops.POP_STREAM | self.epoch, out=js,
                                   col0=uniform_num)
        if bool((js < 0).any().item()):
            # a user who has interacted with every item: np.random.choice(np.setdiff1d(...)) on an empty
            # array in the reference (sampler.py:84-89)
            raise ValueError("'a' cannot be empty unless no samples are taken")
        users = torch.from_numpy(self.df[self.uid_name].to_numpy().astype(np.int32)).to(self.device)
        items = torch.from_numpy(self.df[self.iid_name].to_numpy().astype(np.int32)).to(self.device)
        triples = ops.expand_triples(users, items, js)
        if self.loss_type in ("CL", "SL"):
            # sampler.py:93-98: positives (u, i, rating) stacked on negatives (u, neg, 0)
            rating = torch.from_numpy(self.df[self.inter_name].to_numpy().astype(np.int32)).to(self.device)
            pos = torch.stack([users, items, rating], 1)
            neg = torch.stack([triples[:, 0], triples[:, 2], torch.zeros_like(triples[:, 2])], 1)
            return torch.cat([pos, neg], 0).contiguous()
        return triples

    def sampling(self):
        """np.int32 (N*num_ng, 3) like sampler.py:100-101."""
        return self.sampling_device().cpu().numpy()


class SkipGramNegativeSampler(AbstractSampler):
    """sampler.py:105-160 (Item2Vec): (target, context, label) rows from windows over the user sequences, with as many
    uniform negatives per target as it has context items.  Same constructor, same `sampling()` return value layout
    (np.int array [rows, 3], user by user in `groupby` order, element by element: positives in window order, then
    the negatives); the positives are the reference's exactly, the negatives come from the device generator
    (Philox keyed by config['seed']) instead of MT19937 - same distribution, never one of the user's train items."""

    STREAM = 1 << 61

    def __init__(self, df, config, discard=False):
        super().__init__(config)
        self.context_window = config["context_window"]
        self.user_num = config["user_num"]
        self.seed = int(config.get("seed", 2022))
        self.epoch = int(config.get("sampler_epoch", 0))
        self.device = config.get("device", "cuda")
        if discard:                                                           # sampler.py:123-130 (host, vectorised there too)
            word_frequecy = df[self.iid_name].value_counts()
            prob_discard = 1 - np.sqrt(config["rho"] / word_frequecy)
            rnd_p = np.random.uniform(low=0.0, high=1.0, size=len(df))
            df = df[rnd_p >= df[self.iid_name].map(prob_discard).values]
        self.df = df

    def sampling_device(self):
        if not torch.cuda.is_available():
            raise RuntimeError("daisyrec_amd sampler needs a HIP device (no CPU fallback)")
        users = self.df[self.uid_name].to_numpy().astype(np.int64)
        items = self.df[self.iid_name].to_numpy().astype(np.int32)
        order = np.argsort(users, kind="stable")                              # groupby(user)[item].agg(list): train-set order kept
        seq_user = torch.from_numpy(users[order].astype(np.int32)).to(self.device)
        seq_items = torch.from_numpy(items[order]).to(self.device)
        seq_ptr = torch.zeros(self.user_num + 1, dtype=torch.int64, device=self.device)
        seq_ptr[1:] = torch.cumsum(torch.bincount(seq_user.long(), minlength=self.user_num), 0)
        us = np.fromiter((u for u, s in self.ur.items() for _ in s), dtype=np.int32)
        it = np.fromiter((i for _, s in self.ur.items() for i in s), dtype=np.int32)
        ur_ptr, ur_items = ops.build_user_csr(torch.from_numpy(us).to(self.device), torch.from_numpy(it).to(self.device),
                                              self.user_num)
        return ops.skipgram_samples(seq_items, seq_user, seq_ptr, self.context_window, ur_ptr, ur_items, self.item_num,
                                    self.seed, self.STREAM | self.epoch)

    def sampling(self):
        return self.sampling_device().cpu().numpy().astype(np.int64)           # np.array(list of python ints) is int64

