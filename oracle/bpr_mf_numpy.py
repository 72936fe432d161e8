"""CPU oracle (numpy) for the MF + BPR training hot path of AmazingDD/daisyRec.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product path (``daisyrec_amd``) never does and fails loudly when the HIP
library is missing.

Every function restates, in closed form, what the reference computes through
``torch.nn.Embedding`` + autograd + ``torch.optim`` (file:line relative to
/root/reference):

* forward            daisy/model/MFRecommender.py:63-68
* loss               daisy/model/MFRecommender.py:70-97, daisy/utils/loss.py:5-33
* backward + step    daisy/model/AbstractRecommender.py:48-67,119,125-126
* rank / full_rank   daisy/model/MFRecommender.py:106-133
* negative sampler   daisy/utils/sampler.py:55-103 (uniform branch :82-89)

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference (CPU PyTorch) in the build container and writes known-answer vectors
(per-step and ml-100k end-to-end) that ``tests/test_oracle_golden.py`` checks
this file against.
"""
from __future__ import annotations

import numpy as np

LOSS_BPR, LOSS_HL, LOSS_TL, LOSS_CL, LOSS_SL = 0, 1, 2, 3, 4
LOSS_IDS = {"BPR": LOSS_BPR, "HL": LOSS_HL, "TL": LOSS_TL, "CL": LOSS_CL, "SL": LOSS_SL}


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def pair_loss_coef(pos, neg, loss_type=LOSS_BPR, gamma=1e-10):
    """Per-sample loss terms and d(loss)/d(pos_score), d(loss)/d(neg_score).

    BPR  loss.py:10-13   -(gamma + sigmoid(pos-neg)).log().sum()
    HL   loss.py:20-23   clamp(1-(pos-neg), min=0).sum()   (grad passes at ==0)
    TL   loss.py:30-33   sigmoid(neg-pos).sum() + sigmoid(neg**2).sum()
    """
    x = pos - neg
    if loss_type == LOSS_BPR:
        s = _sigmoid(x)
        terms = -np.log(gamma + s)
        c = -(s * (1.0 - s)) / (gamma + s)
        return terms, c, -c
    if loss_type == LOSS_HL:
        terms = np.maximum(1.0 - x, 0.0)
        c = np.where((1.0 - x) >= 0.0, -1.0, 0.0).astype(x.dtype)
        return terms, c, -c
    if loss_type == LOSS_TL:
        s1 = _sigmoid(neg - pos)
        s2 = _sigmoid(neg * neg)
        terms = s1 + s2
        d1 = s1 * (1.0 - s1)
        cp = -d1
        cn = d1 + 2.0 * neg * s2 * (1.0 - s2)
        return terms, cp, cn
    raise NotImplementedError(f"Invalid loss type: {loss_type}")


def mf_forward(P, Q, u, i):
    """MFRecommender.py:63-68  pred[b] = sum_k P[u_b,k] * Q[i_b,k]."""
    return np.einsum("bk,bk->b", P[u], Q[i])


def mf_pair_grad(P, Q, u, i, j, reg_1, reg_2, loss_type=LOSS_BPR, gamma=1e-10,
                 dtype=np.float64):
    """Loss (scalar) and the DENSE gradients autograd produces for one batch.

    MFRecommender.py:70-97: the regularisers are non-squared norms over the
    whole gathered (B x d) matrices, duplicates counted:
        reg_1*(|Q[i]|_1 + |Q[j]|_1 + |P[u]|_1) + reg_2*(|Q[i]|_F + |Q[j]|_F + |P[u]|_F)
    d|X|_F/dX = X/|X|_F (0 when the norm is 0: torch's subgradient),
    d|X|_1/dX = sign(X).
    """
    P = np.asarray(P, dtype=dtype)
    Q = np.asarray(Q, dtype=dtype)
    u = np.asarray(u, dtype=np.int64)
    i = np.asarray(i, dtype=np.int64)
    j = np.asarray(j, dtype=np.int64)
    pu, qi, qj = P[u], Q[i], Q[j]
    pos = np.einsum("bk,bk->b", pu, qi)
    neg = np.einsum("bk,bk->b", pu, qj)
    terms, cp, cn = pair_loss_coef(pos, neg, loss_type, dtype(gamma))
    nU = np.sqrt((pu * pu).sum(dtype=dtype))
    nI = np.sqrt((qi * qi).sum(dtype=dtype))
    nJ = np.sqrt((qj * qj).sum(dtype=dtype))
    loss = terms.sum(dtype=dtype)
    loss += reg_1 * (np.abs(qi).sum(dtype=dtype) + np.abs(qj).sum(dtype=dtype))
    loss += reg_2 * (nI + nJ)
    loss += reg_1 * np.abs(pu).sum(dtype=dtype)
    loss += reg_2 * nU

    def _fro(x, n):
        return x / n if n > 0 else np.zeros_like(x)

    gP = np.zeros_like(P)
    gQ = np.zeros_like(Q)
    np.add.at(gP, u, cp[:, None] * qi + cn[:, None] * qj
              + reg_1 * np.sign(pu) + reg_2 * _fro(pu, nU))
    np.add.at(gQ, i, cp[:, None] * pu + reg_1 * np.sign(qi) + reg_2 * _fro(qi, nI))
    np.add.at(gQ, j, cn[:, None] * pu + reg_1 * np.sign(qj) + reg_2 * _fro(qj, nJ))
    return loss, gP, gQ


def mf_point_grad(P, Q, u, i, label, reg_1, reg_2, loss_type=LOSS_CL, dtype=np.float64):
    """Point-wise branch of MF.calc_loss (MFRecommender.py:75-81,93-95): rows (user, item, label),
    criterion = BCEWithLogitsLoss(sum) [CL] or MSELoss(sum) [SL] (AbstractRecommender.py:80-83),
    regularisers on Q[item] and P[user] only."""
    P = np.asarray(P, dtype=dtype)
    Q = np.asarray(Q, dtype=dtype)
    u = np.asarray(u, dtype=np.int64)
    i = np.asarray(i, dtype=np.int64)
    y = np.asarray(label, dtype=dtype)
    pu, qi = P[u], Q[i]
    x = np.einsum("bk,bk->b", pu, qi)
    if loss_type == LOSS_CL:
        terms = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
        c = _sigmoid(x) - y
    elif loss_type == LOSS_SL:
        terms = (x - y) ** 2
        c = 2.0 * (x - y)
    else:
        raise NotImplementedError(f"Invalid point-wise loss type: {loss_type}")
    nU = np.sqrt((pu * pu).sum(dtype=dtype))
    nI = np.sqrt((qi * qi).sum(dtype=dtype))
    loss = terms.sum(dtype=dtype) + reg_1 * np.abs(qi).sum(dtype=dtype) + reg_2 * nI \
        + reg_1 * np.abs(pu).sum(dtype=dtype) + reg_2 * nU

    def _fro(v, n):
        return v / n if n > 0 else np.zeros_like(v)

    gP = np.zeros_like(P)
    gQ = np.zeros_like(Q)
    np.add.at(gP, u, c[:, None] * qi + reg_1 * np.sign(pu) + reg_2 * _fro(pu, nU))
    np.add.at(gQ, i, c[:, None] * pu + reg_1 * np.sign(qi) + reg_2 * _fro(qi, nI))
    return loss, gP, gQ


def mf_sgd_step(P, Q, u, i, j, lr, reg_1, reg_2, loss_type=LOSS_BPR, gamma=1e-10,
                dtype=np.float64):
    """One `zero_grad / calc_loss / backward / SGD.step` (AbstractRecommender.py:119-126).

    Returns (loss, P_new, Q_new) with the tables cast back to float32 like the
    reference's parameters.  With dtype=float64 this is the "exact" batch
    synchronous result rounded once.
    """
    if loss_type in (LOSS_CL, LOSS_SL):       # j holds the labels
        loss, gP, gQ = mf_point_grad(P, Q, u, i, j, reg_1, reg_2, loss_type, dtype)
    else:
        loss, gP, gQ = mf_pair_grad(P, Q, u, i, j, reg_1, reg_2, loss_type, gamma, dtype)
    Pn = (np.asarray(P, dtype=dtype) - dtype(lr) * gP).astype(np.float32)
    Qn = (np.asarray(Q, dtype=dtype) - dtype(lr) * gQ).astype(np.float32)
    return float(loss), Pn, Qn


class DenseAdam:
    """torch.optim.Adam defaults (AbstractRecommender.py:54): betas (0.9, 0.999),
    eps 1e-8, no weight decay, no amsgrad, DENSE (every row moves every step)."""

    def __init__(self, shapes, lr, b1=0.9, b2=0.999, eps=1e-8, dtype=np.float64):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m = [np.zeros(s, dtype=dtype) for s in shapes]
        self.v = [np.zeros(s, dtype=dtype) for s in shapes]
        self.dtype = dtype

    def step(self, params, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        out = []
        for k, (w, g) in enumerate(zip(params, grads)):
            w = np.asarray(w, dtype=self.dtype)
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            denom = np.sqrt(self.v[k]) / np.sqrt(bc2) + self.eps
            out.append((w - (self.lr / bc1) * self.m[k] / denom).astype(np.float32))
        return out


class DenseAdagrad:
    """torch.optim.Adagrad defaults (AbstractRecommender.py:58): lr_decay 0, eps 1e-10, accumulator 0, dense:
    state_sum += g*g;  w -= lr * g / (sqrt(state_sum) + eps)   (rows with a zero gradient do not move)."""

    def __init__(self, shapes, lr, eps=1e-10, dtype=np.float64):
        self.lr, self.eps, self.dtype = lr, eps, dtype
        self.ss = [np.zeros(s, dtype=dtype) for s in shapes]

    def step(self, params, grads):
        out = []
        for k, (w, g) in enumerate(zip(params, grads)):
            self.ss[k] = self.ss[k] + g * g
            out.append((np.asarray(w, dtype=self.dtype) - self.lr * g / (np.sqrt(self.ss[k]) + self.eps)).astype(np.float32))
        return out


class DenseRMSprop:
    """torch.optim.RMSprop defaults (AbstractRecommender.py:60): alpha 0.99, eps 1e-8, no momentum, not centered:
    sq = alpha*sq + (1-alpha)*g*g;  w -= lr * g / (sqrt(sq) + eps)   (sq of EVERY row decays every step)."""

    def __init__(self, shapes, lr, alpha=0.99, eps=1e-8, dtype=np.float64):
        self.lr, self.alpha, self.eps, self.dtype = lr, alpha, eps, dtype
        self.sq = [np.zeros(s, dtype=dtype) for s in shapes]

    def step(self, params, grads):
        out = []
        for k, (w, g) in enumerate(zip(params, grads)):
            self.sq[k] = self.alpha * self.sq[k] + (1 - self.alpha) * g * g
            out.append((np.asarray(w, dtype=self.dtype) - self.lr * g / (np.sqrt(self.sq[k]) + self.eps)).astype(np.float32))
        return out


def mf_rank(P, Q, us, cands, topk):
    """MFRecommender.py:106-123: scores = bmm; argsort(descending); gather ids;
    first topk.  Ties broken by candidate position (stable), ids returned as
    float32 like the reference's `torch.tensor([])` concatenation."""
    P = np.asarray(P, dtype=np.float32)
    Q = np.asarray(Q, dtype=np.float32)
    us = np.asarray(us, dtype=np.int64)
    cands = np.asarray(cands, dtype=np.int64)
    scores = np.einsum("bk,bck->bc", P[us], Q[cands])
    order = np.argsort(-scores, axis=1, kind="stable")
    return np.take_along_axis(cands, order, axis=1)[:, :topk].astype(np.float32), scores


def mf_full_rank(P, Q, u, topk):
    """MFRecommender.py:126-133."""
    scores = np.asarray(Q, np.float32) @ np.asarray(P, np.float32)[u]
    return np.argsort(-scores, kind="stable")[:topk].astype(np.int64)


# --------------------------------------------------------------------------
# Uniform negative sampler (sampler.py:82-89): one `num_ng` vector per user,
# drawn with replacement, uniformly over {0..I-1} minus the user's train items.
# numpy's MT19937 stream cannot be matched by a counter-based device generator,
# so the oracle restates the DISTRIBUTION with the same counter-based generator
# the HIP kernel uses (Philox4x32-10), which makes the comparison bit-exact.
# --------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85
_M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11).  ctr: 4 uint32, key: 2 uint32."""
    c0, c1, c2, c3 = (int(x) & _M32 for x in ctr)
    k0, k1 = (int(x) & _M32 for x in key)
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        hi0, lo0 = (p0 >> 32) & _M32, p0 & _M32
        hi1, lo1 = (p1 >> 32) & _M32, p1 & _M32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _M32, lo1, (hi0 ^ c3 ^ k1) & _M32, lo0
        k0 = (k0 + _PHILOX_W0) & _M32
        k1 = (k1 + _PHILOX_W1) & _M32
    return c0, c1, c2, c3


def _draw_u64(seed, stream, index):
    """64 random bits for (seed, stream, index): counter = (index_lo, index_hi,
    stream_lo, stream_hi), key = (seed_lo, seed_hi); take words 0 and 1."""
    r = philox4x32_10((index & _M32, (index >> 32) & _M32, stream & _M32, (stream >> 32) & _M32),
                      (seed & _M32, (seed >> 32) & _M32))
    return (r[1] << 32) | r[0]


def kth_in_complement(row_sorted, r):
    """r-th (0-based) element of {0,1,...} minus the sorted, duplicate-free
    `row_sorted`: smallest t with row[t]-t > r, answer r+t."""
    lo, hi = 0, len(row_sorted)
    while lo < hi:
        mid = (lo + hi) // 2
        if int(row_sorted[mid]) - mid > r:
            hi = mid
        else:
            lo = mid + 1
    return r + lo


def sample_uniform_neg_per_user(indptr, items, item_num, num_ng, seed, epoch=0):
    """js[u, k] for every user id (sampler.py:63,84-89 loops over ALL user ids).
    stream = epoch, index = u*num_ng + k.  A user whose train row covers every
    item gets -1 (numpy would raise on an empty population)."""
    U = len(indptr) - 1
    out = np.empty((U, num_ng), dtype=np.int32)
    for u in range(U):
        row = items[indptr[u]:indptr[u + 1]]
        free = item_num - len(row)
        for k in range(num_ng):
            if free <= 0:
                out[u, k] = -1
                continue
            x = _draw_u64(seed, epoch, u * num_ng + k)
            r = (x * free) >> 64          # Lemire multiply-shift on 64 bits
            out[u, k] = kth_in_complement(row, r)
    return out


def sample_categorical(cdf, rows, k, seed, stream):
    """k draws per row from the categorical distribution with inclusive cumulative sums `cdf` (float64): inverse CDF
    on 53 Philox bits, index = row*k + c (np.random.choice(np.arange(I), size=k, p=prob) of sampler.py:76-80;
    numpy's own choice is the same inverse-CDF search on its MT19937 uniforms)."""
    cdf = np.asarray(cdf, dtype=np.float64)
    out = np.empty((rows, k), dtype=np.int32)
    total = cdf[-1]
    for r in range(rows):
        for c in range(k):
            x = _draw_u64(seed, stream, r * k + c)
            t = float(x >> 11) * (1.0 / 9007199254740992.0) * total
            out[r, c] = min(int(np.searchsorted(cdf, t, side="right")), len(cdf) - 1)
    return out


def skipgram_samples(seqs, ur_rows, item_num, context_window, seed, stream):
    """SkipGramNegativeSampler.sampling (sampler.py:133-155) restated with the Philox draws of the device path:
    `seqs` = list of (user, [items in train order]) in groupby order, `ur_rows[u]` = sorted distinct train items of u.
    Element by element: (target, context, 1) for the window, then as many (target, negative, 0); the k-th negative of
    an element whose rows start at `first` uses index first/2 + k."""
    out = []
    for u, seq in seqs:
        row = ur_rows[u]
        free = item_num - len(row)
        for i in range(len(seq)):
            first = len(out)
            j = i - context_window
            while j <= i + context_window and j < len(seq):
                if j >= 0 and j != i:
                    out.append([seq[i], seq[j], 1])
                j += 1
            c = len(out) - first
            for k in range(c):
                x = _draw_u64(seed, stream, first // 2 + k)
                out.append([seq[i], kth_in_complement(row, (x * free) >> 64), 0])
    return np.array(out, dtype=np.int64).reshape(-1, 3)


def sample_uniform_neg_per_interaction(indptr, items, users, item_num, num_ng, seed, epoch=0):
    """Per-interaction variant (one draw per (interaction, k)); stream = epoch | 1<<63."""
    n = len(users)
    out = np.empty((n, num_ng), dtype=np.int32)
    stream = epoch | (1 << 63)
    for e in range(n):
        u = int(users[e])
        row = items[indptr[u]:indptr[u + 1]]
        free = item_num - len(row)
        for k in range(num_ng):
            if free <= 0:
                out[e, k] = -1
                continue
            x = _draw_u64(seed, stream, e * num_ng + k)
            out[e, k] = kth_in_complement(row, (x * free) >> 64)
    return out


def expand_triples(users, pos_items, js):
    """sampler.py:91,100-101: df['neg_set']=js[user]; explode -> int32 (N*num_ng, 3)
    in train_set row order, each row repeated num_ng times consecutively."""
    users = np.asarray(users, dtype=np.int64)
    num_ng = js.shape[1]
    out = np.empty((len(users) * num_ng, 3), dtype=np.int32)
    out[:, 0] = np.repeat(users, num_ng)
    out[:, 1] = np.repeat(np.asarray(pos_items), num_ng)
    out[:, 2] = js[users].reshape(-1)
    return out


# --------------------------------------------------------------------------
# Device shuffle (DAISY_ORDER_FEISTEL): a keyed bijection of [0, n) that stands in
# for RandomSampler's torch.randperm (dataset.py:5-7, shuffle=True) on the
# throughput path.  4-round Feistel network on ceil(log2 n) bits (halves of a and b = a or a+1
# bits that swap places every round: balanced for an even bit count), a two-step
# multiply-xorshift round function on 24-bit multiplies (full-rate on CDNA),
# Philox round keys, cycle walking.
# --------------------------------------------------------------------------
def _mix32(x):
    x = np.asarray(x, dtype=np.uint64) & np.uint64(_M32)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(_M32)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(_M32)
    x ^= x >> np.uint64(16)
    return x


FEISTEL_ROUNDS = 4


def _mul24(a, b):
    """low 32 bits of (a mod 2^24) * (b mod 2^24): v_mul_u32_u24"""
    return ((a & np.uint64(0xFFFFFF)) * np.uint64(b & 0xFFFFFF)) & np.uint64(_M32)


def _feistel_round(r, key):
    x = (r ^ key) & np.uint64(_M32)
    x = _mul24(x, 0xCC9E2D)
    x ^= x >> np.uint64(15)
    x = _mul24(x, 0x85EBCB)
    x ^= x >> np.uint64(13)
    return x


def feistel_positions(n, seed, epoch=0):
    """pos[t] for t in 0..n-1 (a permutation of 0..n-1)."""
    assert n <= (1 << 30)
    bits = 2
    while bits < 30 and (1 << bits) < n:
        bits += 1                                   # the smallest power-of-two domain >= n
    a, b = bits // 2, bits - bits // 2              # halves of a and b bits (b = a or a + 1), swapping places every round
    mask_a, mask_b = np.uint64((1 << a) - 1), np.uint64((1 << b) - 1)
    keys = [np.uint64(_draw_u64(seed, epoch | (1 << 61), r) & _M32) for r in range(FEISTEL_ROUNDS)]
    x = np.arange(n, dtype=np.uint64)
    todo = np.ones(n, dtype=bool)
    while todo.any():
        v = x[todo]
        L, R = v >> np.uint64(b), v & mask_b
        for r in range(0, FEISTEL_ROUNDS, 2):
            L, R = R, L ^ (_feistel_round(R, keys[r]) & mask_a)
            L, R = R, L ^ (_feistel_round(R, keys[r + 1]) & mask_b)
        v = (L << np.uint64(b)) | R
        x[todo] = v
        todo[todo] = v >= np.uint64(n)
    return x.astype(np.int64)


def partitioned_plan(triples, pos, batch_size, user_base=0, pointwise=False):
    """The partitioned epoch plan (daisy_epoch_plan_build_indexed): what one pass over
    DataLoader(BasicDataset(triples), batch_size, shuffle) serves (dataset.py:5-27), laid out batch by
    batch.  `pos[t]` = position of triple t in the epoch order (identity, inverse of the sampler's
    permutation, or feistel_positions).  Static index: triples in CSR order (stable sort by user), their
    2n entries (item << 1 | slot) stably sorted; the plan is the stable partition of both by batch id.
    pointwise: rows are (user, item, label) (CL / SL, sampler.py:93-98) - ONE entry per row, `item << 1`.
    Returns (samples int64 [n,3] with user - user_base, sample_pos [n], entry_key [2n or n], entry_pos [2n or n])."""
    triples = np.asarray(triples, dtype=np.int64)
    pos = np.asarray(pos, dtype=np.int64)
    order = np.argsort(triples[:, 0], kind="stable")          # CSR order
    tri, p = triples[order], pos[order]
    n = len(tri)
    batch = p // batch_size
    so = np.argsort(batch, kind="stable")
    samples = tri[so].copy()
    samples[:, 0] -= user_base
    if pointwise:
        key = tri[:, 1] << 1
        t = np.arange(n)
    else:
        key = np.empty(2 * n, dtype=np.int64)
        key[0::2] = tri[:, 1] << 1
        key[1::2] = (tri[:, 2] << 1) | 1
        t = np.repeat(np.arange(n), 2)
    eo = np.argsort(key, kind="stable")                       # static item index
    ekey, et = key[eo], t[eo]
    bo = np.argsort(batch[et], kind="stable")
    return samples, p[so], ekey[bo], p[et[bo]]


def build_candidates(indptr_te, items_te, indptr_tr, items_tr, users, item_num, cand_num, seed):
    """utils.py:53-85 with the device generator: negatives (with replacement, uniform over the items
    in neither the test nor the train row) first, then the test items ascending; more than cand_num
    truths -> cand_num draws from the truths.  stream = 1<<60, index = row*cand_num + k."""
    out = np.empty((len(users), cand_num), dtype=np.int64)
    for row, u in enumerate(users):
        te = items_te[indptr_te[u]:indptr_te[u + 1]]
        tr = items_tr[indptr_tr[u]:indptr_tr[u + 1]]
        dte = len(te)
        for k in range(cand_num):
            x = _draw_u64(seed, 1 << 60, row * cand_num + k)
            if dte > cand_num:
                out[row, k] = te[(x * dte) >> 64]
                continue
            n_neg = cand_num - dte
            if k >= n_neg:
                out[row, k] = te[k - n_neg]
                continue
            taken = np.union1d(te, tr)
            free = item_num - len(taken)
            if free <= 0:
                out[row, k] = -1
                continue
            out[row, k] = kth_in_complement(taken, (x * free) >> 64)
    return out
