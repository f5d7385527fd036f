"""Fused mini-batch training step for LightGCN.

What RecBole's ``Trainer._train_epoch`` does per batch around the reference model [recbole==1.1.1]:
``optimizer.zero_grad(); loss = model.calculate_loss(interaction); loss.backward(); optimizer.step()``
with ``calculate_loss`` = lightgcn.py:83-110 and ``optimizer`` = ``torch.optim.Adam`` (RecBole's default learner).
Here the same arithmetic runs as five C-ABI calls on one stream — propagation, BPR gradient scatter, backward chain,
regulariser gradient, Adam — instead of ~30 small torch launches.
"""
from __future__ import annotations

import torch

from . import _lib, ops
from ._lib import c_vp, check, lib
from .models import LightGCN


class FusedBPRAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not isinstance(model, LightGCN):
            raise TypeError("FusedBPRAdam drives a LightGCN model")
        if not model.require_pow:
            raise NotImplementedError("the fused regulariser implements EmbLoss(require_pow=True) (LightGCN.yaml default)")
        self.model, self.lr, self.betas, self.eps = model, float(lr), betas, float(eps)
        self.step_count = 0
        dev = model.device
        n, d = model.n_users + model.n_items, model.latent_dim
        f = dict(dtype=torch.float32, device=dev)
        self.out_mean = torch.empty((n, d), **f)
        self.layers = torch.empty((max(model.n_layers, 1), n, d), **f)
        self.grad_mean = torch.empty((n, d), **f)
        self.grad_e0 = torch.empty((n, d), **f)
        self.work = torch.empty((n, d), **f)
        self.exp_avg = torch.zeros((n, d), **f)
        self.exp_avg_sq = torch.zeros((n, d), **f)
        self.loss = torch.zeros((), **f)

    @torch.no_grad()
    def step(self, interaction):
        """One optimisation step on a batch of (user, pos item, neg item) triples; returns the loss (device scalar)."""
        m = self.model
        if m.restore_user_e is not None or m.restore_item_e is not None:  # lightgcn.py:85-86
            m.restore_user_e, m.restore_item_e = None, None
        dev = m.device
        user = interaction[m.USER_ID].to(device=dev, dtype=torch.int64).contiguous()
        pos = interaction[m.ITEM_ID].to(device=dev, dtype=torch.int64).contiguous()
        neg = interaction[m.NEG_ITEM_ID].to(device=dev, dtype=torch.int64).contiguous()
        uw, iw = m.user_embedding.weight.data, m.item_embedding.weight.data
        d, k_layers, b = m.latent_dim, m.n_layers, user.shape[0]
        st = c_vp(torch.cuda.current_stream(dev).cuda_stream)
        g = m.graph
        with torch.cuda.device(dev):
            ops.lightgcn_forward_raw(g, uw, iw, k_layers, out=self.out_mean, layers=self.layers)
            check(lib.rbg_bpr_grad_f32(c_vp(self.out_mean.data_ptr()), m.n_users, m.n_items, c_vp(user.data_ptr()),
                                       c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, d, c_vp(self.grad_mean.data_ptr()),
                                       c_vp(self.loss.data_ptr()), st))
            arr = (c_vp * 1)(g.transpose().ptr)
            check(lib.rbg_lightgcn_backward_f32(arr, 1, c_vp(self.grad_mean.data_ptr()), c_vp(self.grad_e0.data_ptr()),
                                                c_vp(self.work.data_ptr()), d, k_layers, st))
            check(lib.rbg_emb_reg_grad_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users, c_vp(user.data_ptr()),
                                           c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, d, float(m.reg_weight),
                                           c_vp(self.grad_e0.data_ptr()), c_vp(self.loss.data_ptr()), st))
            self.step_count += 1
            check(lib.rbg_adam_step_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users, m.n_items, d,
                                        c_vp(self.grad_e0.data_ptr()), c_vp(self.exp_avg.data_ptr()),
                                        c_vp(self.exp_avg_sq.data_ptr()), self.step_count, self.lr, self.betas[0],
                                        self.betas[1], self.eps, st))
        return self.loss
