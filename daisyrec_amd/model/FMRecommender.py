"""FM recommender with the reference's interface, trained by the same HIP kernels as MF.

Mirror of daisy/model/FMRecommender.py:18-133 (class ``FM``): MF plus ``u_bias`` / ``i_bias``
(``nn.Embedding(n, 1)``, zero-initialised) and the scalar ``bias_`` added to every score
(FMRecommender.py:61-68).  The loss, its regularisers (embedding rows only) and the rank paths
are those of MF with the three terms added, so the class reuses ``MF``'s methods and only
declares the extra parameters; the kernels pick them up through ``daisy_bpr_ctx_set_bias`` /
``daisy_fm_*`` (include/daisyrec_amd.h).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .MFRecommender import MF
from .AbstractRecommender import GeneralRecommender


class FM(MF):
    def __init__(self, config):
        """Config keys as in FMRecommender.py:36-57: user_num, item_num, factors, epochs, lr, reg_1,
        reg_2, topk, loss_type, optimizer, init_method, early_stop, gpu, logger."""
        GeneralRecommender.__init__(self, config)
        self.epochs = config["epochs"]
        self.lr = config["lr"]
        self.reg_1 = config["reg_1"]
        self.reg_2 = config["reg_2"]

        # module order = the order self.apply(_init_weight) consumes the global torch RNG in
        # (FMRecommender.py:43-59): all four embeddings are drawn, then the biases are zeroed
        self.embed_user = nn.Embedding(config["user_num"], config["factors"])
        self.embed_item = nn.Embedding(config["item_num"], config["factors"])
        self.u_bias = nn.Embedding(config["user_num"], 1)
        self.i_bias = nn.Embedding(config["item_num"], 1)
        self.bias_ = nn.Parameter(torch.tensor([0.0]))

        self.loss_type = config["loss_type"]
        self.optimizer = config["optimizer"] if config["optimizer"] != "default" else "sgd"
        self.initializer = config["init_method"] if config["init_method"] != "default" else "normal"
        self.early_stop = config["early_stop"]
        self.topk = config["topk"]
        self.row_pitch = config.get("row_pitch", "auto")
        self._padded = {}

        self.apply(self._init_weight)
        nn.init.constant_(self.u_bias.weight, 0.0)
        nn.init.constant_(self.i_bias.weight, 0.0)

    def _biases(self):
        self._tables()                       # moves the module to the device if needed
        return self.u_bias.weight.data, self.i_bias.weight.data, self.bias_.data
