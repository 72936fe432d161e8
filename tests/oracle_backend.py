"""CPU stand-in for `daisyrec_amd.ops.BprContext` built on the numpy oracle, used ONLY by the
gloo tests to exercise the multi-rank protocol of `daisyrec_amd.sharding` without a GPU."""
import numpy as np
import torch

from oracle import bpr_mf_numpy as O


class OracleContext:
    def __init__(self, max_batch, d, user_num, item_num):
        self.d, self.user_num, self.item_num = d, user_num, item_num
        self.max_batch = int(max_batch)
        self.stats = torch.zeros(16, dtype=torch.float64)
        self.epoch_acc = torch.zeros(2, dtype=torch.float64)
        self.gQ = torch.zeros(item_num, d, dtype=torch.float32)
        self._bias = None

    def set_bias(self, u_bias=None, i_bias=None, bias=None, g_u_bias=None, g_i_bias=None, g_bias=None):
        """FM (FMRecommender.py:61-68): score += u_bias[u] + i_bias[item] + bias_"""
        self._bias = None if u_bias is None else (u_bias, i_bias, bias, g_u_bias, g_i_bias, g_bias)

    def set_batch_from_triples(self, triples, idx=None, start=0, B=None, user_base=0, validate=True):
        t = triples.numpy() if isinstance(triples, torch.Tensor) else triples
        rows = t[idx.numpy()] if idx is not None else t[start:start + (B if B is not None else len(t) - start)]
        self.u = rows[:, 0].astype(np.int64) - user_base
        self.i, self.j = rows[:, 1].astype(np.int64), rows[:, 2].astype(np.int64)

    def set_batch(self, u, i, j, validate=True):
        self.u, self.i, self.j = (np.asarray(x).astype(np.int64) for x in (u, i, j))

    def forward(self, P, Q, loss_type=0, gamma=1e-10):
        P64, Q64 = P.numpy().astype(np.float64), Q.numpy().astype(np.float64)
        pu, qi, qj = P64[self.u], Q64[self.i], Q64[self.j]
        sp, sn = (pu * qi).sum(1), (pu * qj).sum(1)
        if self._bias is not None:
            bu, bi, b0 = (x.numpy().astype(np.float64).reshape(-1) for x in self._bias[:3])
            sp, sn = sp + bu[self.u] + bi[self.i] + b0[0], sn + bu[self.u] + bi[self.j] + b0[0]
        terms, cp, cn = O.pair_loss_coef(sp, sn, loss_type, gamma)
        self.cp, self.cn = cp, cn
        s = self.stats
        s[0] = terms.sum()
        s[12] = (cp + cn).sum()                # dL/d bias_ over the LOCAL samples (FM)
        s[1], s[2], s[3] = np.abs(pu).sum(), np.abs(qi).sum(), np.abs(qj).sum()
        s[4], s[5], s[6] = (pu * pu).sum(), (qi * qi).sum(), (qj * qj).sum()

    def finalize(self, reg_1, reg_2, step_loss=None, accumulate=True):
        s = self.stats
        s[8], s[9], s[10] = s[4].sqrt(), s[5].sqrt(), s[6].sqrt()
        s[7] = s[0] + reg_1 * (s[1] + s[2] + s[3]) + reg_2 * (s[8] + s[9] + s[10])
        if accumulate:
            self.epoch_acc[0] += s[7]

    @staticmethod
    def _fro(x, n):
        return x / n if n > 0 else np.zeros_like(x)

    def item_grad(self, P, Q, reg_1, reg_2, item_mode=0, gQ=None):
        P64, Q64 = P.numpy().astype(np.float64), Q.numpy().astype(np.float64)
        pu, qi, qj = P64[self.u], Q64[self.i], Q64[self.j]
        nI, nJ = float(self.stats[9]), float(self.stats[10])
        g = np.zeros((self.item_num, self.d))
        np.add.at(g, self.i, self.cp[:, None] * pu + reg_1 * np.sign(qi) + reg_2 * self._fro(qi, nI))
        np.add.at(g, self.j, self.cn[:, None] * pu + reg_1 * np.sign(qj) + reg_2 * self._fro(qj, nJ))
        self.gQ += torch.from_numpy(g.astype(np.float32))

    def item_grad_data(self, P, Q, item_mode=2, gQ=None):
        P64 = P.numpy().astype(np.float64)
        pu = P64[self.u]
        g = np.zeros((self.item_num, self.d))
        np.add.at(g, self.i, self.cp[:, None] * pu)
        np.add.at(g, self.j, self.cn[:, None] * pu)
        self.gQ += torch.from_numpy(g.astype(np.float32))

    def item_grad_reg(self, Q, reg_1, reg_2, gQ=None):
        Q64 = Q.numpy().astype(np.float64)
        qi, qj = Q64[self.i], Q64[self.j]
        nI, nJ = float(self.stats[9]), float(self.stats[10])
        g = np.zeros((self.item_num, self.d))
        np.add.at(g, self.i, reg_1 * np.sign(qi) + reg_2 * self._fro(qi, nI))
        np.add.at(g, self.j, reg_1 * np.sign(qj) + reg_2 * self._fro(qj, nJ))
        self.gQ += torch.from_numpy(g.astype(np.float32))

    def user_sgd(self, P, Q, lr, reg_1, reg_2):
        P64, Q64 = P.numpy().astype(np.float64), Q.numpy().astype(np.float64)
        pu, qi, qj = P64[self.u], Q64[self.i], Q64[self.j]
        nU = float(self.stats[8])
        g = np.zeros_like(P64)
        np.add.at(g, self.u, self.cp[:, None] * qi + self.cn[:, None] * qj + reg_1 * np.sign(pu)
                  + reg_2 * self._fro(pu, nU))
        P.copy_(torch.from_numpy((P64 - lr * g).astype(np.float32)))

    def user_grad(self, P, Q, reg_1, reg_2, gP):
        """dL/dP of the local samples into gP (rows of this rank's users), regulariser from the GLOBAL norm"""
        P64, Q64 = P.numpy().astype(np.float64), Q.numpy().astype(np.float64)
        pu, qi, qj = P64[self.u], Q64[self.i], Q64[self.j]
        nU = float(self.stats[8])
        g = np.zeros_like(P64)
        np.add.at(g, self.u, self.cp[:, None] * qi + self.cn[:, None] * qj + reg_1 * np.sign(pu)
                  + reg_2 * self._fro(pu, nU))
        gP += torch.from_numpy(g.astype(np.float32))

    def item_sgd_apply(self, Q, lr, dense=False, gQ=None):
        Q.sub_(lr * self.gQ)
        self.gQ.zero_()

    # ---- the staged phases (csrc/bpr_staged.hip), restated: prenorm -> user -> finalize -> item -> apply
    def staged_prenorm(self, P):
        pu = P.numpy().astype(np.float64)[self.u]
        self.stats[13] = (pu * pu).sum()

    def staged_user(self, P, Q, lr, reg_1, reg_2, loss_type=0, gamma=1e-10):
        self.forward(P, Q, loss_type, gamma)                       # local batch sums + coefficients
        P64, Q64 = P.numpy().astype(np.float64), Q.numpy().astype(np.float64)
        self.pu_pre = P64[self.u].copy()                           # what the stage holds: pre-step rows
        qi, qj = Q64[self.i], Q64[self.j]
        nU = float(self.stats[13]) ** 0.5                          # GLOBAL |P[u]|_F (all-reduced before this call)
        g = np.zeros_like(P64)
        np.add.at(g, self.u, self.cp[:, None] * qi + self.cn[:, None] * qj + reg_1 * np.sign(self.pu_pre)
                  + reg_2 * self._fro(self.pu_pre, nU))
        P.copy_(torch.from_numpy((P64 - lr * g).astype(np.float32)))
        if self._bias is not None:             # the rank's slice of u_bias follows its users (SGD, in place)
            gb = np.zeros(P64.shape[0])
            np.add.at(gb, self.u, self.cp + self.cn)
            self._bias[0].view(-1).sub_(torch.from_numpy((lr * gb).astype(np.float32)))

    def staged_item(self, lr, reg_1, reg_2, Q=None, gQ=None, cnt=None, loss_type=0):
        assert Q is None, "the oracle backend only restates the gradient-output form"
        g = np.zeros((self.item_num, self.d))
        c = np.zeros((self.item_num, 2))
        np.add.at(g, self.i, self.cp[:, None] * self.pu_pre)
        np.add.at(g, self.j, self.cn[:, None] * self.pu_pre)
        np.add.at(c[:, 0], self.i, 1.0)
        np.add.at(c[:, 1], self.j, 1.0)
        if self._bias is not None:             # dL/d i_bias of the LOCAL samples, all-reduced by the trainer
            gb = np.zeros(self.item_num)
            np.add.at(gb, self.i, self.cp)
            np.add.at(gb, self.j, self.cn)
            self._bias[4].copy_(torch.from_numpy(gb.astype(np.float32)))
        gQ += torch.from_numpy(g.astype(np.float32))
        cnt += torch.from_numpy(c.astype(np.float32))

    def staged_item_slices(self, item_bounds):
        """ops.BprContext.staged_item_slices: the item ranges a sliced step reduces one after the other"""
        self._bounds = [int(b) for b in item_bounds]

    def staged_item_slice(self, s, lr, reg_1, reg_2, gQ, cnt, loss_type=0):
        lo, hi = self._bounds[s], self._bounds[s + 1]
        g = np.zeros((self.item_num, self.d))
        c = np.zeros((self.item_num, 2))
        mi, mj = (self.i >= lo) & (self.i < hi), (self.j >= lo) & (self.j < hi)
        np.add.at(g, self.i[mi], self.cp[mi, None] * self.pu_pre[mi])
        np.add.at(g, self.j[mj], self.cn[mj, None] * self.pu_pre[mj])
        np.add.at(c[:, 0], self.i[mi], 1.0)
        np.add.at(c[:, 1], self.j[mj], 1.0)
        gQ += torch.from_numpy(g.astype(np.float32))
        cnt += torch.from_numpy(c.astype(np.float32))

    def item_apply_counts(self, Q_rows, g_rows, cnt_rows, lr, reg_1, reg_2):
        nI, nJ = float(self.stats[9]), float(self.stats[10])
        q = Q_rows.numpy().astype(np.float64)
        npos, nneg = cnt_rows[:, 0:1].numpy().astype(np.float64), cnt_rows[:, 1:2].numpy().astype(np.float64)
        g = g_rows.numpy().astype(np.float64) + reg_1 * (npos + nneg) * np.sign(q) \
            + reg_2 * (npos * self._fro(q, nI) + nneg * self._fro(q, nJ))
        Q_rows.copy_(torch.from_numpy((q - lr * g).astype(np.float32)))
        g_rows.zero_()
        cnt_rows.zero_()

    # ---- torch.optim.Adam through the sharded protocol (sharding.UserShardedBprTrainer(adam_steps=...)): the HIP path
    # keeps the rank's rows of P in the lazy form and steps the owner's block of Q densely; restated here with the dense
    # oracle optimiser - every row of P_local in every step (zero gradient for rows without a sample; steps this rank
    # sat out are caught up first), every row of an owned block per owner call
    def staged_adam_catchup_users(self, P, adam):
        if not hasattr(self, "_p_adam"):
            self._p_adam = O.DenseAdam([tuple(P.shape)], adam.lr)
        self._catch_up_p(P, adam.t - 1)

    def _catch_up_p(self, P, upto):
        while self._p_adam.t < upto:                                   # zero-gradient steps
            (Pn,) = self._p_adam.step([P.numpy()], [np.zeros(tuple(P.shape))])
            P.copy_(torch.from_numpy(Pn))

    def oracle_flush_p(self, P, adam):
        if not hasattr(self, "_p_adam"):
            self._p_adam = O.DenseAdam([tuple(P.shape)], adam.lr)
        self._catch_up_p(P, adam.t)

    def staged_user_adam(self, P, Q, adam, reg_1, reg_2, loss_type=0, gamma=1e-10):
        self.forward(P, Q, loss_type, gamma)
        P64, Q64 = P.numpy().astype(np.float64), Q.numpy().astype(np.float64)
        self.pu_pre = P64[self.u].copy()
        qi, qj = Q64[self.i], Q64[self.j]
        nU = float(self.stats[13]) ** 0.5
        g = np.zeros_like(P64)
        np.add.at(g, self.u, self.cp[:, None] * qi + self.cn[:, None] * qj + reg_1 * np.sign(self.pu_pre)
                  + reg_2 * self._fro(self.pu_pre, nU))
        assert self._p_adam.t == adam.t - 1
        (Pn,) = self._p_adam.step([P.numpy()], [g])
        P.copy_(torch.from_numpy(Pn))

    def item_apply_counts_adam(self, Q_rows, g_rows, cnt_rows, m_rows, v_rows, adam, reg_1, reg_2):
        key = m_rows.data_ptr()                                        # one dense optimiser per owned block
        if not hasattr(self, "_q_adam"):
            self._q_adam = {}
        opt = self._q_adam.setdefault(key, O.DenseAdam([tuple(Q_rows.shape)], adam.lr))
        assert opt.t == adam.t - 1, "every owned block steps exactly once per global step"
        nI, nJ = float(self.stats[9]), float(self.stats[10])
        q = Q_rows.numpy().astype(np.float64)
        npos, nneg = cnt_rows[:, 0:1].numpy().astype(np.float64), cnt_rows[:, 1:2].numpy().astype(np.float64)
        g = g_rows.numpy().astype(np.float64) + reg_1 * (npos + nneg) * np.sign(q) \
            + reg_2 * (npos * self._fro(q, nI) + nneg * self._fro(q, nJ))
        (Qn,) = opt.step([Q_rows.numpy()], [g])
        Q_rows.copy_(torch.from_numpy(Qn))
        g_rows.zero_()
        cnt_rows.zero_()


class OracleGraph:
    """CPU stand-in for `daisyrec_amd.ops.LgcnGraph` (only what sharding.RowShardedPropagation.spmm calls): row ranges
    of A_hat X from the numpy oracle's CSR."""

    def __init__(self, users, items, user_num, item_num):
        from oracle import lightgcn_numpy as LG
        self.LG = LG
        self.csr = LG.norm_adj_csr(users, items, user_num, item_num)

    def spmm_rows(self, X, Yrows, row_lo, row_hi):
        assert Yrows.shape[0] >= row_hi - row_lo + 2 and Yrows.is_contiguous()
        full = self.LG.spmm(self.csr, X.numpy().astype(np.float64))
        Yrows[1:1 + row_hi - row_lo] = torch.from_numpy(full[row_lo:row_hi].astype(np.float32))
        return Yrows[1:1 + row_hi - row_lo]


class OracleDense:
    """CPU stand-in for `daisyrec_amd.ops.DenseOptimizer` (next_step() / step(W, g): consumes and clears g) on the oracle's
    dense optimisers, one state per parameter tensor"""

    def __init__(self, kind, lr):
        self.kind, self.lr, self.t, self._state = kind, lr, 0, {}

    def next_step(self):
        self.t += 1

    def step(self, W, g):
        st = self._state.get(W.data_ptr())
        if st is None:
            cls = {"adam": O.DenseAdam, "adagrad": O.DenseAdagrad, "rmsprop": O.DenseRMSprop}[self.kind]
            st = self._state[W.data_ptr()] = cls([tuple(W.shape)], self.lr)
        (Wn,) = st.step([W.numpy()], [g.numpy().astype(np.float64)])
        W.copy_(torch.from_numpy(Wn))
        g.zero_()

