"""CPU/PyTorch restatement of the reference's MF + BPR training step, used ONLY as the
`cpu_baseline` leg of bench.py and in tests (TEST INFRASTRUCTURE — the product path never
imports it).

It composes the same stock PyTorch pieces in the same order as the reference does, so its
host-side cost profile is the reference's (dense `[U,d]`/`[I,d]` gradients through
`embedding_dense_backward`, dense `optim.SGD.step`):
    tables       nn.Embedding x2, normal(0, 0.01)     MFRecommender.py:53-54,61
    forward      (E_u[u] * E_i[i]).sum(-1)            MFRecommender.py:63-68
    loss         -(gamma + sigmoid(pos-neg)).log().sum() + reg_1*L1 + reg_2*Frobenius
                                                      loss.py:10-13, MFRecommender.py:88-95
    step         zero_grad / backward / SGD.step      AbstractRecommender.py:119-126
Parity: pinned by tests/test_oracle_golden.py::test_torch_port_matches_golden.
"""
import torch
import torch.nn as nn


class TorchMFBPR(nn.Module):
    def __init__(self, user_num, item_num, d, lr=0.01, reg_1=0.001, reg_2=0.001, gamma=1e-10):
        super().__init__()
        self.embed_user = nn.Embedding(user_num, d)
        self.embed_item = nn.Embedding(item_num, d)
        nn.init.normal_(self.embed_user.weight, mean=0.0, std=0.01)
        nn.init.normal_(self.embed_item.weight, mean=0.0, std=0.01)
        self.reg_1, self.reg_2, self.gamma = reg_1, reg_2, gamma
        self.opt = torch.optim.SGD(self.parameters(), lr=lr)

    def score(self, u, i):
        return (self.embed_user(u) * self.embed_item(i)).sum(dim=-1)

    def loss(self, u, i, j):
        pos, neg = self.score(u, i), self.score(u, j)
        out = -(self.gamma + torch.sigmoid(pos - neg)).log().sum()
        out = out + self.reg_1 * (self.embed_item(i).norm(p=1) + self.embed_item(j).norm(p=1))
        out = out + self.reg_2 * (self.embed_item(i).norm() + self.embed_item(j).norm())
        out = out + self.reg_1 * self.embed_user(u).norm(p=1)
        out = out + self.reg_2 * self.embed_user(u).norm()
        return out

    def step(self, u, i, j):
        self.zero_grad()
        out = self.loss(u, i, j)
        out.backward()
        self.opt.step()
        return float(out.item())


class TorchNeuMF(nn.Module):
    """The reference's NeuMF training step out of stock PyTorch pieces (NeuMFRecommender.py:52-73,
    118-169; dense Adam, AbstractRecommender.py:54) - the `cpu_baseline` leg of tools/bench_neumf.py."""

    def __init__(self, user_num, item_num, d, num_layers, lr=0.001, reg_1=0.001, reg_2=0.001, dropout=0.0,
                 gamma=1e-10):
        super().__init__()
        dm = d * 2 ** (num_layers - 1)
        self.uG, self.iG = nn.Embedding(user_num, d), nn.Embedding(item_num, d)
        self.uM, self.iM = nn.Embedding(user_num, dm), nn.Embedding(item_num, dm)
        mods, n = [], 2 * dm
        for _ in range(num_layers):
            mods += [nn.Dropout(p=dropout), nn.Linear(n, n // 2), nn.ReLU()]
            n //= 2
        self.mlp = nn.Sequential(*mods)
        self.predict = nn.Linear(2 * d, 1)
        self.reg_1, self.reg_2, self.gamma = reg_1, reg_2, gamma
        self.opt = torch.optim.Adam(self.parameters(), lr=lr)

    def score(self, u, i):
        g = self.uG(u) * self.iG(i)
        x = self.mlp(torch.cat((self.uM(u), self.iM(i)), dim=-1))
        return self.predict(torch.cat((g, x), -1)).view(-1)

    def loss(self, u, i, j):
        pos, neg = self.score(u, i), self.score(u, j)
        out = -(self.gamma + torch.sigmoid(pos - neg)).log().sum()
        out = out + self.reg_1 * (self.iG(i).norm(p=1) + self.iG(j).norm(p=1))
        out = out + self.reg_1 * (self.iM(i).norm(p=1) + self.iG(j).norm(p=1))
        out = out + self.reg_2 * (self.iG(i).norm() + self.iG(j).norm())
        out = out + self.reg_2 * (self.iM(i).norm() + self.iG(j).norm())
        out = out + self.reg_1 * (self.uG(u).norm(p=1) + self.uM(u).norm(p=1))
        out = out + self.reg_2 * (self.uG(u).norm() + self.uM(u).norm())
        return out

    def step(self, u, i, j):
        self.zero_grad()
        out = self.loss(u, i, j)
        out.backward()
        self.opt.step()
        return float(out.item())


class TorchLightGCN(nn.Module):
    """The reference's LightGCN training step out of stock PyTorch pieces (LightGCNRecommender.py:109-169:
    torch.sparse.mm propagation recomputed per batch, dense Adam) - the `cpu_baseline` leg of
    tools/bench_lightgcn.py.  `adj` is a torch sparse COO tensor of the normalised adjacency."""

    def __init__(self, user_num, item_num, d, num_layers, adj, lr=0.01, gamma=1e-10):
        super().__init__()
        self.U, self.I, self.L, self.gamma = user_num, item_num, num_layers, gamma
        self.embed_user, self.embed_item = nn.Embedding(user_num, d), nn.Embedding(item_num, d)
        nn.init.xavier_uniform_(self.embed_user.weight)
        nn.init.xavier_uniform_(self.embed_item.weight)
        self.adj = adj
        self.opt = torch.optim.Adam(self.parameters(), lr=lr)

    def forward(self):
        e = torch.cat([self.embed_user.weight, self.embed_item.weight], 0)
        acc = [e]
        for _ in range(self.L):
            e = torch.sparse.mm(self.adj, e)
            acc.append(e)
        out = torch.stack(acc, 1).mean(1)
        return torch.split(out, [self.U, self.I])

    def step(self, u, i, j):
        self.zero_grad()
        eu, ei = self.forward()
        pos = (eu[u] * ei[i]).sum(1)
        neg = (eu[u] * ei[j]).sum(1)
        out = -(self.gamma + torch.sigmoid(pos - neg)).log().sum()
        out.backward()
        self.opt.step()
        return float(out.item())
