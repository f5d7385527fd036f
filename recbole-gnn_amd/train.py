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
from .graph import get_option as _get_option
from .models import LightGCN

_CONCURRENT_HALVES = __import__("os").environ.get("RBG_CONCURRENT_HALVES", "1") != "0"  # (A/B switch of _FusedStep._two_halves)


def fused_step_applies(model):
    """True when the model's training objective is exactly lightgcn.py:83-110 (what the fused step implements)."""
    return (isinstance(model, LightGCN) and type(model).calculate_loss is LightGCN.calculate_loss
            and type(model).forward is LightGCN.forward)


class FusedBPRAdam:
    """``step()`` enqueues the five calls; every one of them — incl. Adam, whose step count lives on the device — can be
    captured into a HIP graph by the caller (``torch.cuda.graph``) and replayed."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # exactly LightGCN: subclasses (SimGCL, XSimGCL) override forward / calculate_loss — contrastive terms, perturbed
        # passes, a layer mean without E0 — none of which this hard-wired BPR + reg step computes
        if not fused_step_applies(model):
            raise TypeError("FusedBPRAdam drives a plain LightGCN model (its forward and calculate_loss, not a subclass's)")
        self.model, self.lr, self.betas, self.eps = model, float(lr), betas, float(eps)
        self.step_count = 0
        dev = model.device
        n, d = model.n_users + model.n_items, model.latent_dim
        f = dict(dtype=torch.float32, device=dev)
        self.out_mean = torch.empty((n, d), **f)
        self.layers = torch.empty((max(model.n_layers, 1), n, d), **f)
        self.grad_mean = torch.zeros((n, d), **f)
        # r06, the lean step (rbg_lightgcn_step_head_f32 / _tail_f32: two launches instead of seven around the backward
        # propagation): grad_mean stays all-zero between steps (the tail zeroes the rows the head wrote), the batch's node
        # occurrences alternate between two tables by the step's parity, the loss is stored by the last workgroup to arrive
        self.lean = True
        self.row_count = torch.zeros((2, n), dtype=torch.int32, device=dev)
        self.scratch = torch.zeros(8, **f)
        self.loss_total = torch.zeros((), **f)  # the running sum of the steps' losses (the driver reads it once per epoch)
        self._lean_state_clean = True
        self.grad_e0 = torch.empty((n, d), **f)
        self.work = torch.empty((n, d), **f)
        self.exp_avg = torch.zeros((n, d), **f)
        self.exp_avg_sq = torch.zeros((n, d), **f)
        self.step_dev = torch.zeros((), dtype=torch.int64, device=dev)  # Adam's step count (device: replayable)
        self.adam_factors = torch.zeros(2, **f)
        self.loss = torch.zeros((), **f)
        self.reg_ws = torch.zeros(3, **f)  # the three block norms of EmbLoss(require_pow=False)

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
        lean = self.lean and m.require_pow and d % 4 == 0 and b > 0 and not _get_option("deterministic")
        if lean:
            with torch.cuda.device(dev):
                if not self._lean_state_clean:  # (a step of the other form left its gradient rows behind)
                    self.grad_mean.zero_()
                    self.row_count.zero_()
                    self._lean_state_clean = True
                ops.lightgcn_forward_raw(g, uw, iw, k_layers, out=self.out_mean, layers=self.layers)
                check(lib.rbg_lightgcn_step_head_f32(c_vp(self.out_mean.data_ptr()), c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users,
                                                     m.n_items, c_vp(user.data_ptr()), c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, d,
                                                     float(m.reg_weight), c_vp(self.grad_mean.data_ptr()), c_vp(self.row_count.data_ptr()),
                                                     c_vp(self.step_dev.data_ptr()), c_vp(self.scratch.data_ptr()), c_vp(self.loss.data_ptr()),
                                                     c_vp(self.loss_total.data_ptr()), self.lr, self.betas[0], self.betas[1], st))
                arr = (c_vp * 1)(g.transpose().ptr)
                check(lib.rbg_lightgcn_backward_f32(arr, 1, c_vp(self.grad_mean.data_ptr()), c_vp(self.grad_e0.data_ptr()),
                                                    c_vp(self.work.data_ptr()), d, k_layers, st))
                self.step_count += 1
                check(lib.rbg_lightgcn_step_tail_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users, m.n_items, d,
                                                     c_vp(self.grad_e0.data_ptr()), c_vp(self.grad_mean.data_ptr()),
                                                     c_vp(self.row_count.data_ptr()), float(m.reg_weight), b, c_vp(self.exp_avg.data_ptr()),
                                                     c_vp(self.exp_avg_sq.data_ptr()), c_vp(self.step_dev.data_ptr()),
                                                     c_vp(self.scratch.data_ptr()), self.lr, self.betas[0], self.betas[1], self.eps, st))
            return self.loss
        self._lean_state_clean = False
        with torch.cuda.device(dev):
            ops.lightgcn_forward_raw(g, uw, iw, k_layers, out=self.out_mean, layers=self.layers)
            check(lib.rbg_bpr_grad_f32(c_vp(self.out_mean.data_ptr()), m.n_users, m.n_items, c_vp(user.data_ptr()),
                                       c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, d, c_vp(self.grad_mean.data_ptr()),
                                       c_vp(self.loss.data_ptr()), st))
            arr = (c_vp * 1)(g.transpose().ptr)
            check(lib.rbg_lightgcn_backward_f32(arr, 1, c_vp(self.grad_mean.data_ptr()), c_vp(self.grad_e0.data_ptr()),
                                                c_vp(self.work.data_ptr()), d, k_layers, st))
            if m.require_pow:  # LightGCN.yaml: squared form
                check(lib.rbg_emb_reg_grad_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users, c_vp(user.data_ptr()),
                                               c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, d, float(m.reg_weight),
                                               c_vp(self.grad_e0.data_ptr()), c_vp(self.loss.data_ptr()), st))
            else:              # EmbLoss's default: the 2-norm of each gathered block
                check(lib.rbg_emb_reg_grad_nopow_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users, c_vp(user.data_ptr()),
                                                     c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, d, float(m.reg_weight),
                                                     c_vp(self.grad_e0.data_ptr()), c_vp(self.loss.data_ptr()),
                                                     c_vp(self.reg_ws.data_ptr()), st))
            self.step_count += 1
            if d % 4 == 0:
                check(lib.rbg_adam_step_dev_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users, m.n_items, d,
                                                c_vp(self.grad_e0.data_ptr()), c_vp(self.exp_avg.data_ptr()),
                                                c_vp(self.exp_avg_sq.data_ptr()), c_vp(self.step_dev.data_ptr()),
                                                c_vp(self.adam_factors.data_ptr()), self.lr, self.betas[0], self.betas[1], self.eps, st))
            else:  # (host-side step count: not replayable)
                check(lib.rbg_adam_step_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users, m.n_items, d,
                                            c_vp(self.grad_e0.data_ptr()), c_vp(self.exp_avg.data_ptr()),
                                            c_vp(self.exp_avg_sq.data_ptr()), self.step_count, self.lr, self.betas[0],
                                            self.betas[1], self.eps, st))
        self.loss_total += self.loss
        return self.loss


class _TableAdam:
    """torch.optim.Adam (no weight decay, no amsgrad) on the two embedding tables, one launch over both
    (``rbg_adam_step_dev_f32``: the step count lives on the device, so the call can be replayed from a HIP graph)."""

    def __init__(self, model, lr, betas, eps):
        self.model, self.lr, self.betas, self.eps = model, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        uw = model.user_embedding.weight
        n, d = model.n_users + model.n_items, uw.shape[1]
        f = dict(dtype=torch.float32, device=uw.device)
        self.exp_avg, self.exp_avg_sq = torch.zeros((n, d), **f), torch.zeros((n, d), **f)
        self.step_dev = torch.zeros((), dtype=torch.int64, device=uw.device)
        self.factors = torch.zeros(2, **f)
        self.loss_total = torch.zeros((), **f)  # r06: the running sum of the steps' losses (``step(grad, loss)``); a driver reads it per epoch

    def step(self, grad, loss=None):
        """``loss``: the step's finished loss (device scalar) — it joins ``loss_total`` in the launch that counts the step
        (rbg_adam_step_dev_total_f32): no add launch per step in the driver."""
        m = self.model
        uw, iw = m.user_embedding.weight.data, m.item_embedding.weight.data
        st = c_vp(torch.cuda.current_stream(uw.device).cuda_stream)
        if loss is not None and uw.shape[1] % 4 == 0:
            check(lib.rbg_adam_step_dev_total_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users, m.n_items, uw.shape[1],
                                                  c_vp(grad.data_ptr()), c_vp(self.exp_avg.data_ptr()), c_vp(self.exp_avg_sq.data_ptr()),
                                                  c_vp(self.step_dev.data_ptr()), c_vp(self.factors.data_ptr()), self.lr, self.betas[0],
                                                  self.betas[1], self.eps, c_vp(loss.data_ptr()), c_vp(self.loss_total.data_ptr()), st))
            return
        check(lib.rbg_adam_step_dev_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), m.n_users, m.n_items, uw.shape[1], c_vp(grad.data_ptr()),
                                        c_vp(self.exp_avg.data_ptr()), c_vp(self.exp_avg_sq.data_ptr()), c_vp(self.step_dev.data_ptr()),
                                        c_vp(self.factors.data_ptr()), self.lr, self.betas[0], self.betas[1], self.eps, st))
        if loss is not None:
            self.loss_total += loss


class _FusedStep:
    """What the autograd-free steps share: the batch's index tensors, and — ``graphed=True`` — the capture of one step's
    launches into a HIP graph after two eager warm-up steps (lazy allocations must exist before capture), replayed per batch
    of the captured size (a batch of another size — an epoch's last one — is enqueued eagerly on its own scratch) and
    re-captured when the graph handles the step reads change (SGL samples new views per epoch)."""

    def _init_graphed(self, graphed):
        self.graphed = bool(graphed)
        self._graph, self._static, self._calls, self._key = None, None, 0, None
        self._full = 0  # the batch size the graph is captured for: the largest seen so far (an epoch's last batch is shorter)

    def _handles(self):
        return (self.model.graph,)

    def _variant(self):
        """What else a captured step has baked in (NCL: whether the prototype term is part of the loss)."""
        return ()

    def _two_halves(self, dev, first, second):
        """The two InfoNCE halves of a step (users / items: disjoint row ranges of every table they touch, their own workspaces and
        loss words) on two streams (r06): each half is eight dependent launches of which six are latency-bound row kernels — issued
        side by side, one half's small kernels run in the shadow of the other's matrix-core launches.  ``first`` runs on the
        current stream, ``second`` on a side stream forked from it; joined before returning.  Sequential in "deterministic" mode
        (its fixed-point slots are per stream) and when ``self.concurrent_halves`` is False."""
        if not getattr(self, "concurrent_halves", _CONCURRENT_HALVES) or _get_option("deterministic"):
            first()
            second()
            return
        ops._fork_join([first, second], dev)

    def _side_stream(self, dev):
        """ONE side stream per stepper for the eager steps beside a captured graph (warm-up, an epoch's short last batch) — a fresh
        stream per such batch was a stream creation per epoch and model."""
        st = getattr(self, "_eager_side", None)
        if st is None:  # (a stepper lives on one device)
            st = self._eager_side = torch.cuda.Stream(device=dev)
        return st

    @torch.no_grad()
    def step(self, interaction):
        """One optimisation step on a batch of (user, pos item, neg item) triples; returns the loss (device scalar)."""
        m = self.model
        if m.restore_user_e is not None or m.restore_item_e is not None:  # (ngcf.py:108-109, sgl.py:212-213)
            m.restore_user_e, m.restore_item_e = None, None
        dev = m.device
        user, pos, neg = (interaction[k].to(device=dev, dtype=torch.int64).contiguous() for k in (m.USER_ID, m.ITEM_ID, m.NEG_ITEM_ID))
        if not self.graphed:
            self._enqueue(user, pos, neg)
            return self.loss
        # (the options "deterministic" and "lse_f16" pick kernels (and fixed-point slots) on the host at launch time: a captured graph has
        # the choice baked in, so a toggle re-warms and re-captures like new views do — ADVICE r05)
        key = tuple(id(h) for h in self._handles()) + tuple(self._variant()) + (int(_get_option("deterministic")), int(_get_option("lse_f16")))
        if self._graph is not None and key != self._key:
            self._graph, self._calls = None, 1  # new views: their plans exist, one eager step re-warms
        if self._graph is not None and user.shape[0] > self._full:  # (a larger batch than the captured one: capture again for it)
            self._graph, self._calls = None, 1
        self._full = max(self._full, int(user.shape[0]))
        if self._graph is None:
            self._calls += 1
            # warm-up steps — and any batch that is not of the full size (with three or fewer batches per epoch the
            # third call could be the epoch's short last batch; the graph would then have been captured for the odd size and every
            # full batch would have run eagerly for ever) — run eagerly on a side stream; the capture waits for a full batch
            if self._calls <= 2 or user.shape[0] < self._full:
                side = self._side_stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self._enqueue(user, pos, neg)
                torch.cuda.current_stream(dev).wait_stream(side)
                return self.loss
            self._static3 = torch.stack([user, pos, neg])  # ONE [3, B] block: a sampler's "_triples" view is copied in with one launch
            self._static = tuple(self._static3[i] for i in range(3))
            self._keep = self._handles()  # (a captured graph has their device pointers baked in)
            self._key = key
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._enqueue(*self._static)
        if user.shape != self._static[0].shape:  # (an epoch's last, shorter batch: the same launches, not replayed —
            side = self._side_stream(dev)  # on a side stream like the warm-up steps: its scratch is not the graph's)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._enqueue(user, pos, neg)
            torch.cuda.current_stream(dev).wait_stream(side)
            return self.loss
        trip = interaction.get("_triples") if isinstance(interaction, dict) else None
        if (trip is not None and trip.dtype == torch.int64 and trip.device == self._static3.device and trip.shape == self._static3.shape
                and trip[0].data_ptr() == user.data_ptr()):  # (driver.BPRSampler: the batch's three columns as one strided view)
            self._static3.copy_(trip)
        else:
            for dst, src in zip(self._static, (user, pos, neg)):
                dst.copy_(src)
        self._graph.replay()
        return self.loss


class FusedNGCFAdam(_FusedStep):
    """NGCF's training step (ngcf.py:106-126 + ``loss.backward()`` + Adam) as library calls on preallocated buffers, no autograd:
    per layer ONE forward call (``rbg_bignn_layer_f32``: product, both transforms, LeakyReLU, dropout mask, normalize) and ONE
    backward call (``rbg_bignn_backward_f32``); the loss on the rows of the concatenation ``cat(E_0..E_K)`` (ngcf.py:100,
    113-117) is ``rbg_concat_bpr_begin_f32`` + one ``rbg_concat_bpr_scatter_f32`` per layer, which adds a layer's sparse row
    gradients IN PLACE onto the dense gradient the layer above has just written (autograd: zeros + index_add + add per layer,
    and ~40 small launches for the loss).  The update: one library launch for the two embedding tables (``_TableAdam``: Adam's
    step count on the device), torch's fused Adam for the layers' weights — both on gradients that live in this object's buffers.
    ``graphed=True`` captures the whole step into one HIP graph after the first call (fixed batch size).

    The model keeps its parameters, so ``full_sort_predict`` etc. see the trained weights.  ``message_dropout`` draws its masks
    with torch's generator per step, ``node_dropout`` goes through the model's re-weighted views — both as in ``NGCF``."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graphed=False):
        from .models import NGCF
        if not (isinstance(model, NGCF) and type(model).calculate_loss is NGCF.calculate_loss and model.fused
                and isinstance(model.graph, ops.GraphHandle) and max(model.hidden_size_list) <= 128):
            raise TypeError("FusedNGCFAdam drives a plain NGCF model on a device graph handle with layer widths <= 128")
        self.model = model
        self._init_graphed(graphed)
        self.widths = list(model.hidden_size_list)
        if len(self.widths) > 8:
            raise ValueError("at most 7 layers (RBG_MAX_CONCAT tables)")
        dev, n = model.device, model.n_users + model.n_items
        f = dict(dtype=torch.float32, device=dev)
        self.e = [torch.empty((n, w), **f) for w in self.widths]         # E_0 (the ego table) .. E_K
        self.g = [torch.empty((n, w), **f) for w in self.widths]         # dLoss/dE_t
        self.p = [torch.empty((n, w), **f) for w in self.widths[:-1]]    # A E_{t} saved by layer t + 1's forward
        self.inv = [torch.empty(n, **f) for _ in self.widths[1:]]
        nbytes = _lib.c_i64()
        need = 8
        for d_in, d_out in zip(self.widths[:-1], self.widths[1:]):
            check(lib.rbg_bignn_backward_workspace(n, d_in, d_out, _lib.ctypes.byref(nbytes)))
            need = max(need, nbytes.value)
        self.work = torch.empty(need, dtype=torch.uint8, device=dev)
        self.coef, self._coef, self.sums, self.loss = None, {}, torch.zeros(3, **f), torch.zeros((), **f)
        self._tabs = (c_vp * len(self.e))(*[t.data_ptr() for t in self.e])
        self._widths = (_lib.c_int * len(self.e))(*self.widths)
        # gradients of the parameters live here; the embedding tables' are the two row ranges of g[0]
        nu = model.n_users
        model.user_embedding.weight.grad = self.g[0][:nu]  # (for inspection; the tables are updated from g[0] directly)
        model.item_embedding.weight.grad = self.g[0][nu:]
        self.gb = []
        for gnn in model.GNNlayers:
            gb = torch.zeros_like(gnn.lin1.bias)
            gnn.lin1.weight.grad, gnn.lin2.weight.grad = torch.zeros_like(gnn.lin1.weight), torch.zeros_like(gnn.lin2.weight)
            gnn.lin1.bias.grad, gnn.lin2.bias.grad = gb, gb  # (the two biases add into the same output: one gradient)
            self.gb.append(gb)
        # the two tables (all but 25 k of the 4.6 M parameters at the Gowalla shape): one library launch on the ego gradient;
        # the layers' weights and biases: torch's fused Adam, one multi-tensor launch
        self.table_opt = _TableAdam(model, lr, betas, eps)
        self.loss_total = self.table_opt.loss_total  # (driver.fit reads it once per epoch)
        self.opt = torch.optim.Adam([p for gnn in model.GNNlayers for p in gnn.parameters()], lr=lr, betas=betas, eps=eps,
                                    capturable=True, fused=True)

    # ---- one step's launches on the current stream --------------------------------------------------------------------------
    def _enqueue(self, user, pos, neg):
        m = self.model
        dev, nu, b = m.device, m.n_users, user.shape[0]
        graph = m._dropout_graph() if (m.node_dropout != 0 and m.training) else m.graph
        graph_t = graph.transpose()
        if b not in self._coef:  # (per batch size: a captured graph keeps reading the buffer of ITS size)
            self._coef[b] = torch.empty(b, dtype=torch.float32, device=dev)
        self.coef = self._coef[b]
        st = c_vp(torch.cuda.current_stream(dev).cuda_stream)
        k_layers = len(self.widths) - 1
        masks = []
        with torch.cuda.device(dev):
            torch.cat([m.user_embedding.weight.data, m.item_embedding.weight.data], dim=0, out=self.e[0])
            for t, gnn in enumerate(m.GNNlayers):
                d_in, d_out = self.widths[t], self.widths[t + 1]
                mask = None
                if m.message_dropout > 0:  # fresh nn.Dropout(p)(x) of ngcf.py:97: drawn on every forward
                    mask = ops.dropout_mask(self.e[0].shape[0], d_out, m.message_dropout, dev)
                masks.append(mask)
                check(lib.rbg_bignn_layer_f32(graph.ptr, c_vp(self.e[t].data_ptr()), d_in, c_vp(gnn.lin1.weight.data_ptr()),
                                              c_vp(gnn.lin1.bias.data_ptr()), c_vp(gnn.lin2.weight.data_ptr()), c_vp(gnn.lin2.bias.data_ptr()),
                                              c_vp(self.e[t + 1].data_ptr()), d_out, c_vp(self.p[t].data_ptr()), c_vp(self.inv[t].data_ptr()),
                                              c_vp(mask.data_ptr()) if mask is not None else None, d_in, d_out, 0.2, st))
            check(lib.rbg_concat_bpr_begin_f32(self._tabs, self._widths, len(self.e), nu, m.n_items, c_vp(user.data_ptr()),
                                               c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, 0, c_vp(self.coef.data_ptr()),
                                               c_vp(self.sums.data_ptr()), c_vp(self.loss.data_ptr()), st))
            self.g[k_layers].zero_()
            for t in range(k_layers, -1, -1):
                # the rows of E_t that the batch reads: their gradient on top of what layer t + 1's backward wrote
                check(lib.rbg_concat_bpr_scatter_f32(c_vp(self.e[t].data_ptr()), self.widths[t], nu, c_vp(user.data_ptr()),
                                                     c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, float(m.reg_weight), 0,
                                                     c_vp(self.coef.data_ptr()), c_vp(self.sums.data_ptr()), c_vp(self.g[t].data_ptr()),
                                                     c_vp(self.loss.data_ptr()) if t == k_layers else None, st))
                if t == 0:
                    break
                gnn, d_in, d_out = m.GNNlayers[t - 1], self.widths[t - 1], self.widths[t]
                mask = masks[t - 1]
                check(lib.rbg_bignn_backward_f32(graph_t.ptr, c_vp(self.g[t].data_ptr()), d_out, c_vp(self.e[t].data_ptr()), d_out,
                                                 c_vp(self.inv[t - 1].data_ptr()), c_vp(mask.data_ptr()) if mask is not None else None,
                                                 c_vp(self.e[t - 1].data_ptr()), d_in, c_vp(self.p[t - 1].data_ptr()),
                                                 c_vp(gnn.lin1.weight.data_ptr()), c_vp(gnn.lin2.weight.data_ptr()), d_in, d_out, 0.2,
                                                 c_vp(self.g[t - 1].data_ptr()), c_vp(gnn.lin1.weight.grad.data_ptr()),
                                                 c_vp(gnn.lin2.weight.grad.data_ptr()), c_vp(self.gb[t - 1].data_ptr()),
                                                 c_vp(self.work.data_ptr()), st))
            self.table_opt.step(self.g[0], self.loss)
            self.opt.step()


class FusedSGLAdam(_FusedStep):
    """SGL's training step (sgl.py:211-233 + ``loss.backward()`` + Adam) as library calls on preallocated buffers, no autograd:
    three propagations (the full graph and the two views, sgl.py:219-221), the sum-reduced BPR term on the propagated mean
    (``rbg_concat_bpr_*`` form 1), both InfoNCE halves with their table gradients (``rbg_infonce_f32``), three backward
    chains, EmbLoss on the ego rows, Adam on the two tables in one library launch (``_TableAdam``).  ``graphed=True``
    replays the step from a HIP graph, re-captured when the model samples new views (``SGL.train()``, sgl.py:94-98)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graphed=False):
        from .models import SGL
        if not (isinstance(model, SGL) and type(model).calculate_loss is SGL.calculate_loss and type(model).forward is SGL.forward
                and isinstance(model.graph, ops.GraphHandle) and model.user_embedding.weight.shape[1] <= 128):
            raise TypeError("FusedSGLAdam drives a plain SGL model on device graph handles with embedding width <= 128")
        self.model = model
        self._init_graphed(graphed)
        dev, nu = model.device, model.n_users
        n, d = nu + model.n_items, model.user_embedding.weight.shape[1]
        f = dict(dtype=torch.float32, device=dev)
        self.mean = [torch.empty((n, d), **f) for _ in range(3)]   # the full graph, view 1, view 2
        self._gm3 = torch.empty((3, n, d), **f)                    # (one allocation: ONE fill launch per step zeroes the three)
        self.gm = [self._gm3[0], self._gm3[1], self._gm3[2]]       # dLoss/d(mean)
        self.ge = [torch.empty((n, d), **f) for _ in range(3)]     # dLoss/dE0 through each propagation; ge[0] ends as the total
        self.layers = torch.empty((max(model.n_layers, 1), n, d), **f)
        self.work = torch.empty((n, d), **f)
        self.coef, self.nce_work, self._scratch = None, None, {}
        self.sums, self.reg_ws, self.loss = torch.zeros(3, **f), torch.zeros(3, **f), torch.zeros((), **f)
        self._tab = (c_vp * 1)(self.mean[0].data_ptr())
        self._wid = (_lib.c_int * 1)(d)
        if len(list(model.parameters())) != 2:
            raise TypeError("FusedSGLAdam updates the two embedding tables; this model has other parameters")
        self.table_opt = _TableAdam(model, lr, betas, eps)
        self.loss_total = self.table_opt.loss_total  # (driver.fit reads it once per epoch)
        model.user_embedding.weight.grad = self.ge[0][:nu]  # (for inspection; the tables are updated from ge[0] directly)
        model.item_embedding.weight.grad = self.ge[0][nu:]

    def _views(self):
        m = self.model
        if m.sub_graph1 is None:
            m.graph_construction()
        views = [[m.graph], [g for g, _ in m.sub_graph1], [g for g, _ in m.sub_graph2]]
        return [v[:1] if all(g is v[0] for g in v) else v for v in views]  # (RW: one graph per layer)

    def _handles(self):
        return tuple(g for v in self._views() for g in v)

    def _enqueue(self, user, pos, neg):
        m = self.model
        dev, nu, ni, b, k_layers = m.device, m.n_users, m.n_items, user.shape[0], m.n_layers
        uw, iw = m.user_embedding.weight.data, m.item_embedding.weight.data
        d = uw.shape[1]
        views = self._views()
        if b not in self._scratch:  # (per batch size: a captured graph keeps reading the buffers of ITS size)
            nbytes, need = _lib.c_i64(), 8
            for rows in (nu, ni):
                check(lib.rbg_infonce_workspace(b, rows, d, _lib.ctypes.byref(nbytes)))
                need = max(need, nbytes.value)
            self._scratch[b] = (torch.empty(b, dtype=torch.float32, device=dev), torch.empty(need, dtype=torch.uint8, device=dev),
                                torch.empty(need, dtype=torch.uint8, device=dev), torch.zeros((), dtype=torch.float32, device=dev))
        self.coef, self.nce_work, nce_work2, loss2 = self._scratch[b]
        st = c_vp(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t, row=0: c_vp(t.data_ptr() + 4 * row * d)  # noqa: E731  (rows [row, ...) of a contiguous [*, d] table)
        with torch.cuda.device(dev):
            for v in range(3):
                ops.lightgcn_forward_raw(views[v], uw, iw, k_layers, out=self.mean[v], layers=self.layers)
            # sgl.py:147-162 on the full graph's mean: value, then the rows' gradients onto zeros
            check(lib.rbg_concat_bpr_begin_f32(self._tab, self._wid, 1, nu, ni, ptr(user), ptr(pos), ptr(neg), b, 1, ptr(self.coef),
                                               ptr(self.sums), ptr(self.loss), st))
            self._gm3.zero_()
            check(lib.rbg_concat_bpr_scatter_f32(ptr(self.mean[0]), d, nu, ptr(user), ptr(pos), ptr(neg), b, 0.0, 0, ptr(self.coef),
                                                 ptr(self.sums), ptr(self.gm[0]), None, st))
            # sgl.py:176-209: users, then items, between the two views
            def half(row0, rows, idx, loss, work):
                def run():  # (the stream is looked up when it runs: the second half is issued on a side stream)
                    check(lib.rbg_infonce_f32(ptr(self.mean[1], row0), ptr(self.mean[2], row0), rows, d, ptr(idx), b, float(m.ssl_tau),
                                              float(m.ssl_weight), ptr(loss), ptr(self.gm[1], row0), ptr(self.gm[2], row0), ptr(work),
                                              c_vp(torch.cuda.current_stream(dev).cuda_stream)))
                return run

            def users_half():
                loss2.zero_()
                half(0, nu, user, loss2, nce_work2)()

            self._two_halves(dev, half(nu, ni, pos, self.loss, self.nce_work), users_half)
            self.loss.add_(loss2)
            for v in range(3):
                ts = [g.transpose() for g in views[v]]
                arr = (c_vp * len(ts))(*[g.ptr for g in ts])
                check(lib.rbg_lightgcn_backward_f32(arr, len(ts), ptr(self.gm[v]), ptr(self.ge[v]), ptr(self.work), d, k_layers, st))
            self.ge[0].add_(self.ge[1]).add_(self.ge[2])
            check(lib.rbg_emb_reg_grad_nopow_f32(ptr(uw), ptr(iw), nu, ptr(user), ptr(pos), ptr(neg), b, d, float(m.reg_weight),
                                                 ptr(self.ge[0]), ptr(self.loss), ptr(self.reg_ws), st))
            self.table_opt.step(self.ge[0], self.loss)


class _FusedContrastStep(_FusedStep):
    """What the SimGCL / XSimGCL steps share: buffers over the [N, d] table, the perturbed pass (simgcl.py:25-36: one
    ``rbg_spmm_noise_f32`` per layer on a fresh ``uniform_`` draw — the draws of ``torch.rand_like`` in the reference's order, so a
    model trained by the autograd path from the same seed sees the same noise), the contrast on the batch's gathered rows
    (``rbg_infonce_masked_f32`` with the one-occurrence mask: simgcl.py:38-57 over ``torch.unique`` without its data-dependent
    shape) and the backward chain.  sign() has no gradient, so a perturbed layer's backward is the plain product — and EVERY
    chain of a step is the same linear map of its incoming gradient: the gradients are summed first and ONE chain runs (autograd
    runs one per propagation)."""

    def _common_init(self, model, cls, lr, betas, eps, graphed):
        from .models import BPRLoss
        if not (type(model) is cls and isinstance(model.graph, ops.GraphHandle) and model.static_unique
                and model.user_embedding.weight.shape[1] <= 128 and model.user_embedding.weight.shape[1] % 4 == 0
                and type(model.mf_loss) is BPRLoss and model.mf_loss.gamma == 1e-10 and len(list(model.parameters())) == 2):
            raise TypeError(f"{type(self).__name__} drives a plain {cls.__name__} model (static_unique form) on a device graph handle, "
                            "embedding width <= 128 and a multiple of 4")
        self.model = model
        self._init_graphed(graphed)
        dev, nu = model.device, model.n_users
        n, d, k = nu + model.n_items, model.user_embedding.weight.shape[1], max(model.n_layers, 1)
        f = dict(dtype=torch.float32, device=dev)
        self.e0 = torch.empty((n, d), **f)
        self.noise = torch.empty((n, d), **f)
        self._noises = []
        self.gm = torch.empty((n, d), **f)     # dLoss / d(mean of the pass the BPR term reads); ends as the chain's input
        self.ge = torch.empty((n, d), **f)     # dLoss / dE0
        self.t0, self.t1, self.work = torch.empty((n, d), **f), torch.empty((n, d), **f), torch.empty((n, d), **f)
        self.loss, self.reg_ws = torch.zeros((), **f), torch.zeros(3, **f)
        self._scratch = {}
        self._once = {}
        self.table_opt = _TableAdam(model, lr, betas, eps)
        self.loss_total = self.table_opt.loss_total  # (driver.fit reads it once per epoch)
        model.user_embedding.weight.grad = self.ge[:nu]  # (for inspection; the tables are updated from ge directly)
        model.item_embedding.weight.grad = self.ge[nu:]
        return n, d, k, f

    @staticmethod
    def _ptr(t):
        return c_vp(t.data_ptr())

    def _mean_of(self, layers, k, out):
        srcs = (c_vp * k)(*[layers[i].data_ptr() for i in range(k)])
        check(lib.rbg_mean_f32(srcs, k, out.numel(), 1.0 / k, self._ptr(out), c_vp(torch.cuda.current_stream(out.device).cuda_stream)))

    def _noise_buffers(self, count):
        """``count`` [N, d] noise tables (r06): a step's draws are issued up front — in the reference's order, on a side stream — and
        run beside the propagations instead of between them (11 us of a launch that leaves most of the GPU idle, per layer and pass)."""
        while len(self._noises) < count:
            self._noises.append(torch.empty_like(self.noise))
        return self._noises[:count]

    def _draw_noise(self, dev, noises, beside):
        """uniform_ (= torch.rand_like: the same draws in the same order as SimGCL._layers) into every table of ``noises`` on a side
        stream while ``beside()`` issues launches on the current one; joined before returning."""
        def draw():
            for t in noises:
                t.uniform_()
        if not noises:
            beside()
        elif not getattr(self, "concurrent_halves", _CONCURRENT_HALVES):
            draw()
            beside()
        else:
            ops._fork_join([beside, draw], dev)

    def _perturbed_pass(self, layers, k, first_product=None, noises=None):
        """``first_product``: A E_0 where a plain pass has formed it already (r06: SimGCL's three passes start with the same
        product) — the first perturbed layer is then the noise epilogue alone (rbg_sign_noise_f32), one propagation less.
        ``noises``: the pass's k noise tables, drawn already (``_draw_noise``); None draws each layer's in place."""
        m, st = self.model, c_vp(torch.cuda.current_stream(self.model.device).cuda_stream)
        x = self.e0
        for i in range(k):
            if noises is None:
                self.noise.uniform_()  # (= torch.rand_like: the same draws in the same order as SimGCL._layers)
            nz = self.noise if noises is None else noises[i]
            if i == 0 and first_product is not None:
                check(lib.rbg_sign_noise_f32(self._ptr(first_product), self._ptr(nz), x.shape[0], x.shape[1], float(m.eps),
                                             self._ptr(layers[0]), st))
            else:
                check(lib.rbg_spmm_noise_f32(m.graph.ptr, self._ptr(x), self._ptr(layers[i]), self._ptr(nz), x.shape[1], float(m.eps), st))
            x = layers[i]

    def _contrast_scratch(self, b, d):
        """Two workspaces and a second loss word per batch size: the step's two contrasts run on two streams (_two_halves)."""
        if b not in self._scratch:
            m, nbytes = self.model, _lib.c_i64()
            check(lib.rbg_infonce_workspace(b, b, d, _lib.ctypes.byref(nbytes)))
            self._scratch[b] = (torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=m.device),
                                torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=m.device),
                                torch.zeros((), dtype=torch.float32, device=m.device))
        return self._scratch[b]

    def _contrasts(self, ta, tb, ga, gb, user, pos, b, mean_form):
        """The step's two contrasts — the users' rows and the positive items' rows (disjoint row ranges of every table) — side by
        side on two streams (r06): each is ten launches of which eight are latency-bound; their losses meet in self.loss."""
        m = self.model
        nu, ni = m.n_users, m.n_items
        work, work2, loss2 = self._contrast_scratch(b, ta.shape[1])

        def users():
            loss2.zero_()
            self._contrast(ta, tb, ga, gb, user, 0, nu, b, mean_form, loss2, work2)

        self._two_halves(m.device, lambda: self._contrast(ta, tb, ga, gb, pos, nu, ni, b, mean_form, self.loss, work), users)
        self.loss.add_(loss2)

    def _contrast(self, ta, tb, ga, gb, ids, row0, rows, b, mean_form, loss, work):
        """cl_rate x InfoNCE between rows `ids` of tables ta and tb (rows [row0, row0 + rows) of the [N, d] buffers): the value is
        added to `loss`, the gradients w.r.t. the two tables' rows are scattered onto ga / gb (either may be the same buffer)."""
        m = self.model
        d = ta.shape[1]
        st = c_vp(torch.cuda.current_stream(m.device).cuda_stream)
        # r06: the mask and the row weights in ONE launch (rbg_once_mask_f32) instead of scatter / gather / compare / cast (/ sum / divide)
        key = (b, int(rows), int(row0))  # (the users' and the items' contrast run side by side: their own buffers even when n_users = n_items)
        if key not in self._once:
            self._once[key] = (torch.empty(int(rows), dtype=torch.int64, device=m.device),
                               torch.empty(b, dtype=torch.float32, device=m.device), torch.empty(b, dtype=torch.float32, device=m.device))
        slot, once, row_w = self._once[key]  # (xsimgcl.py:54: mean_form = a mean over the distinct ids)
        check(lib.rbg_once_mask_f32(self._ptr(ids), b, int(rows), self._ptr(slot), int(bool(_get_option("deterministic"))), int(bool(mean_form)),
                                    self._ptr(once), self._ptr(row_w), st))
        # r06: the contrast among the batch's rows straight on the tables (rbg_infonce_batch_f32) — no gathered copies, no zero-filled
        # gradient blocks, no index_add_: the row kernels gather, the backward kernels scatter (rows of a repeated id carry zero
        # weight but one: the float atomics add zeros — the sums do not depend on the order)
        check(lib.rbg_infonce_batch_f32(self._ptr(ta[row0:row0 + rows]), self._ptr(tb[row0:row0 + rows]), d, self._ptr(ids), b,
                                        float(m.temperature), float(m.cl_rate), self._ptr(row_w), self._ptr(once), self._ptr(loss),
                                        self._ptr(ga[row0:row0 + rows]), self._ptr(gb[row0:row0 + rows]), self._ptr(work), st))

    def _reg_and_adam(self, user, pos, neg, b, d):
        m = self.model
        uw, iw = m.user_embedding.weight.data, m.item_embedding.weight.data
        st = c_vp(torch.cuda.current_stream(m.device).cuda_stream)
        p = self._ptr
        if m.require_pow:
            check(lib.rbg_emb_reg_grad_f32(p(uw), p(iw), m.n_users, p(user), p(pos), p(neg), b, d, float(m.reg_weight), p(self.ge), p(self.loss), st))
        else:
            check(lib.rbg_emb_reg_grad_nopow_f32(p(uw), p(iw), m.n_users, p(user), p(pos), p(neg), b, d, float(m.reg_weight), p(self.ge),
                                                 p(self.loss), p(self.reg_ws), st))
        self.table_opt.step(self.ge, self.loss)


class FusedSimGCLAdam(_FusedContrastStep):
    """SimGCL's training step (simgcl.py:45-61 + ``loss.backward()`` + Adam) as library calls, no autograd: the plain pass (K
    products, mean of layers 1..K), two perturbed passes, BPR + EmbLoss, the two contrasts, ONE backward chain for the three
    passes' summed gradients (K products), Adam on the two tables in one launch.  ``graphed=True`` replays it from a HIP graph."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graphed=False):
        from .models import SimGCL
        n, d, k, f = self._common_init(model, SimGCL, lr, betas, eps, graphed)
        self.lay = [torch.empty((k, n, d), **f) for _ in range(3)]   # the plain pass, the two perturbed ones
        self.mean = [torch.empty((n, d), **f) for _ in range(3)]
        self.g12 = torch.empty((n, d), **f)                           # d(contrast) / d(mean of pass 1) + / d(mean of pass 2)

    def _enqueue(self, user, pos, neg):
        m = self.model
        dev, nu, ni, b, k = m.device, m.n_users, m.n_items, user.shape[0], m.n_layers
        uw, iw = m.user_embedding.weight.data, m.item_embedding.weight.data
        d = uw.shape[1]
        st = c_vp(torch.cuda.current_stream(dev).cuda_stream)
        p = self._ptr
        with torch.cuda.device(dev):
            torch.cat([uw, iw], dim=0, out=self.e0)
            # simgcl.py:25-36 without noise: E_1 .. E_K row-major, their mean (no E_0)
            noises = self._noise_buffers(2 * k)  # (pass 1's k draws, then pass 2's: the reference's order)

            def plain():
                ops.lightgcn_forward_raw(m.graph, uw, iw, k, keep_layers=True, out=self.work, layers=self.lay[0])
                self._mean_of(self.lay[0], k, self.mean[0])

            self._draw_noise(dev, noises, plain)  # the 2 k draws run beside the plain pass
            for v in (1, 2):
                self._perturbed_pass(self.lay[v], k, first_product=self.lay[0][0], noises=noises[(v - 1) * k:v * k])  # (the three passes share A E_0)
                self._mean_of(self.lay[v], k, self.mean[v])
            # lightgcn.py:93-100 on the plain pass (zeroes gm and the loss), then the contrasts between the perturbed passes
            check(lib.rbg_bpr_grad_f32(p(self.mean[0]), nu, ni, p(user), p(pos), p(neg), b, d, p(self.gm), p(self.loss), st))
            self.g12.zero_()
            # (the gradients w.r.t. the two perturbed means go through the same linear chain: they are added into ONE buffer)
            self._contrasts(self.mean[1], self.mean[2], self.g12, self.g12, user, pos, b, False)
            self.gm.add_(self.g12)
            # out = (A + A^2 + .. + A^K) E0 / K for every pass  =>  dE0 = A (g + A g + .. + A^(K-1) g) / K
            if k > 1:
                arr = (c_vp * 1)(m.graph.transpose().ptr)
                check(lib.rbg_lightgcn_backward_f32(arr, 1, p(self.gm), p(self.t0), p(self.work), d, k - 1, st))
                src = self.t0
            else:
                src = self.gm
            check(lib.rbg_spmm_f32(m.graph.transpose().ptr, p(src), p(self.ge), d, 0, st))
            self._reg_and_adam(user, pos, neg, b, d)


class FusedXSimGCLAdam(_FusedContrastStep):
    """XSimGCL's training step (xsimgcl.py:56-90 + backward + Adam) as library calls, no autograd: ONE perturbed pass serves the
    BPR term (its layer mean) and the contrast between that mean and the embedding after layer ``layer_cl``; the backward is one
    Horner chain with the contrast's second gradient injected at that layer."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graphed=False):
        from .models import XSimGCL
        n, d, k, f = self._common_init(model, XSimGCL, lr, betas, eps, graphed)
        self.lay = torch.empty((k, n, d), **f)
        self.mean = torch.empty((n, d), **f)
        self.gcl = torch.empty((n, d), **f)  # d(contrast) / d(the layer_cl embedding)

    def _enqueue(self, user, pos, neg):
        m = self.model
        dev, nu, ni, b, k = m.device, m.n_users, m.n_items, user.shape[0], m.n_layers
        uw, iw = m.user_embedding.weight.data, m.item_embedding.weight.data
        d = uw.shape[1]
        st = c_vp(torch.cuda.current_stream(dev).cuda_stream)
        p = self._ptr
        lc = m.layer_cl if 1 <= m.layer_cl <= k else 0  # 0: the contrast's second view is E_0 itself (xsimgcl.py:29, 39-41)
        with torch.cuda.device(dev):
            torch.cat([uw, iw], dim=0, out=self.e0)
            noises = self._noise_buffers(k)
            noises[0].uniform_()  # (the first layer needs its draw at once; the others' run beside its propagation)

            def first_layer():
                check(lib.rbg_spmm_noise_f32(m.graph.ptr, p(self.e0), p(self.lay[0]), p(noises[0]), d, float(m.eps),
                                             c_vp(torch.cuda.current_stream(dev).cuda_stream)))

            self._draw_noise(dev, noises[1:], first_layer)
            x = self.lay[0]
            for i in range(1, k):
                check(lib.rbg_spmm_noise_f32(m.graph.ptr, p(x), p(self.lay[i]), p(noises[i]), d, float(m.eps), st))
                x = self.lay[i]
            self._mean_of(self.lay, k, self.mean)
            cl = self.lay[lc - 1] if lc else self.e0
            check(lib.rbg_bpr_grad_f32(p(self.mean), nu, ni, p(user), p(pos), p(neg), b, d, p(self.gm), p(self.loss), st))
            self.gcl.zero_()
            self._contrasts(self.mean, cl, self.gm, self.gcl, user, pos, b, True)
            # mean = (E_1 + .. + E_K) / K, E_j = A E_(j-1) (+ noise, no gradient): dE0 = A (g_1 + A (g_2 + .. A g_K)), g_j = gm / K
            # (+ gcl at j = layer_cl); Horner from the top layer down
            gt = m.graph.transpose().ptr
            torch.mul(self.gm, 1.0 / k, out=self.gm)
            # r06: every step is ONE launch  nxt = g_j + A cur  (rbg_spmm_add_f32) — no copy of the addend into the output first;
            # the layer that also receives the contrast's gradient takes gm + gcl (formed in place in gcl, once)
            if lc >= 1:
                self.gcl.add_(self.gm)
            cur = self.gcl if lc == k else self.gm
            bufs = (self.t0, self.t1)
            for n, j in enumerate(range(k - 1, 0, -1)):
                nxt = bufs[n & 1]
                check(lib.rbg_spmm_add_f32(gt, p(cur), p(self.gcl if lc == j else self.gm), p(nxt), d, st))
                cur = nxt
            if lc == 0:
                check(lib.rbg_spmm_add_f32(gt, p(cur), p(self.gcl), p(self.ge), d, st))
            else:
                check(lib.rbg_spmm_f32(gt, p(cur), p(self.ge), d, 0, st))
            self._reg_and_adam(user, pos, neg, b, d)


class FusedNCLAdam(_FusedStep):
    """NCL's training step (ncl.py:167-199 + NCLTrainer._train_epoch's sum, trainer.py:130-133 + backward + Adam) without the
    autograd graph of the propagation: the L = max(n_layers, 2 hyper_layers) layers in one chain call (every layer kept,
    ncl.py:93-104), BPR + EmbLoss by the library, the structure contrast (ncl.py:137-165: layer 2 h against E_0 over ALL rows) by
    ``rbg_infonce_f32`` with both table gradients, and ONE Horner chain of L products for the mean's gradient and the context
    layer's injected at layer 2 h.  The prototype contrast (ncl.py:106-135: the batch's rows of E_0 against the k centroids, the
    positive = the row's cluster) is one ``rbg_infonce_map_f32`` per side (r06; the model's formula under ``torch.autograd.grad``
    before: ~ 70 launches, 385 us per step); the users' and the items' contrasts run side by side on two streams.  ``with_proto``
    (False during the trainer's warm-up epochs) is part of the captured step: toggling it re-captures."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graphed=False):
        from .models import NCL
        d = model.user_embedding.weight.shape[1] if hasattr(model, "user_embedding") else 0
        if not (type(model) is NCL and isinstance(model.graph, ops.GraphHandle) and d <= 128 and d % 4 == 0
                and len(list(model.parameters())) == 2 and 1 <= 2 * model.hyper_layers):
            raise TypeError("FusedNCLAdam drives a plain NCL model on a device graph handle, embedding width <= 128 and a multiple of 4")
        self.model = model
        self._init_graphed(graphed)
        self.with_proto = True
        dev, nu = model.device, model.n_users
        n = nu + model.n_items
        self.L = max(model.n_layers, 2 * model.hyper_layers)
        if self.L > _lib.MAX_FUSED_LAYERS:
            raise TypeError("too many layers for one chain call")
        f = dict(dtype=torch.float32, device=dev)
        self.e0 = torch.empty((n, d), **f)
        self.lay = torch.empty((self.L, n, d), **f)
        self.mean, self.chain_mean = torch.empty((n, d), **f), torch.empty((n, d), **f)
        self._gz = torch.empty((2, n, d), **f)  # (gctx and g0 side by side: ONE fill launch per step zeroes both)
        self.gm, self.gctx, self.g0 = torch.empty((n, d), **f), self._gz[0], self._gz[1]
        self.ge, self.t0, self.t1 = torch.empty((n, d), **f), torch.empty((n, d), **f), torch.empty((n, d), **f)
        self.loss, self.reg_ws = torch.zeros((), **f), torch.zeros(3, **f)
        self._scratch = {}
        self.table_opt = _TableAdam(model, lr, betas, eps)
        self.loss_total = self.table_opt.loss_total  # (driver.fit reads it once per epoch)
        model.user_embedding.weight.grad = self.ge[:nu]
        model.item_embedding.weight.grad = self.ge[nu:]

    def _variant(self):
        return (bool(self.with_proto),)

    def _enqueue(self, user, pos, neg):
        m = self.model
        dev, nu, ni, b, k, L = m.device, m.n_users, m.n_items, user.shape[0], m.n_layers, self.L
        uw, iw = m.user_embedding.weight.data, m.item_embedding.weight.data
        d = uw.shape[1]
        st = c_vp(torch.cuda.current_stream(dev).cuda_stream)
        p = lambda t, row=0: c_vp(t.data_ptr() + 4 * row * d)  # noqa: E731
        if b not in self._scratch:
            nbytes, need = _lib.c_i64(), 8
            for rows in (nu, ni, int(m.k)):  # (the structure contrast over all rows of a side, the prototype contrast over the k centroids)
                check(lib.rbg_infonce_workspace(b, rows, d, _lib.ctypes.byref(nbytes)))
                need = max(need, nbytes.value)
            self._scratch[b] = (torch.empty(need, dtype=torch.uint8, device=dev), torch.empty(need, dtype=torch.uint8, device=dev),
                                torch.zeros((), dtype=torch.float32, device=dev))
        work, work2, loss2 = self._scratch[b]
        with torch.cuda.device(dev):
            torch.cat([uw, iw], dim=0, out=self.e0)
            # ncl.py:93-104: E_1 .. E_L row-major; the mean over E_0 .. E_K (the chain's own mean when K = L)
            ops.lightgcn_forward_raw(m.graph, uw, iw, L, keep_layers=True, out=self.chain_mean, layers=self.lay)
            if k == L:
                mean = self.chain_mean
            else:
                srcs = (c_vp * (k + 1))(self.e0.data_ptr(), *[self.lay[i].data_ptr() for i in range(k)])
                check(lib.rbg_mean_f32(srcs, k + 1, self.mean.numel(), 1.0 / (k + 1), p(self.mean), st))
                mean = self.mean
            check(lib.rbg_bpr_grad_f32(p(mean), nu, ni, p(user), p(pos), p(neg), b, d, p(self.gm), p(self.loss), st))
            # ncl.py:137-165: the context layer E_(2 h) against the center E_0, users then items (x alpha)
            ctx = self.lay[2 * m.hyper_layers - 1]
            self._gz.zero_()  # gctx, g0
            if self.with_proto and m.user_centroids is None:
                raise RuntimeError("NCL.e_step() has not run: no prototypes yet (NCLTrainer calls it before the first epoch)")

            def half(row0, rows, idx, wgt, loss, ws, centroids, node2cluster):
                def run():  # (the stream is looked up when it runs: the second half is issued on a side stream)
                    sth = c_vp(torch.cuda.current_stream(dev).cuda_stream)
                    check(lib.rbg_infonce_f32(p(ctx, row0), p(self.e0, row0), rows, d, p(idx), b, float(m.ssl_temp), float(wgt), p(loss),
                                              p(self.gctx, row0), p(self.g0, row0), p(ws), sth))
                    if self.with_proto:
                        # ncl.py:106-135, r06: the batch's rows of E_0 against the side's k centroids (constants), the positive = the row's
                        # cluster — one library call (value + the rows' gradient onto g0) instead of ~ 35 torch launches under autograd
                        check(lib.rbg_infonce_map_f32(p(self.e0, row0), c_vp(centroids.data_ptr()), centroids.shape[0], d, p(idx),
                                                      c_vp(node2cluster.data_ptr()), b, float(m.ssl_temp), float(m.proto_reg), p(loss),
                                                      p(self.g0, row0), None, p(ws), sth))
                return run

            def users_half():
                loss2.zero_()
                half(0, nu, user, m.ssl_reg, loss2, work2, m.user_centroids, m.user_2cluster)()

            self._two_halves(dev, half(nu, ni, pos, m.ssl_reg * m.alpha, self.loss, work, m.item_centroids, m.item_2cluster), users_half)
            self.loss.add_(loss2)
            # mean = (E_0 + .. + E_K) / (K + 1), E_j = A E_(j-1):  dE0 = g_0 + A (g_1 + A (g_2 + .. A g_L)),
            # g_j = gm / (K + 1) for j <= K  (+ gctx at j = 2 h),  g_0 = gm / (K + 1) + the direct gradients on E_0
            gt = m.graph.transpose().ptr
            torch.mul(self.gm, 1.0 / (k + 1), out=self.gm)
            h2 = 2 * m.hyper_layers
            # r06: every step is ONE launch  nxt = g_j + A cur  (rbg_spmm_add_f32) instead of "copy g_j, then accumulate";
            # g_j = gm (j <= K) + gctx (j = 2 h): the sums that occur are formed in place, once (gctx += gm, g0 += gm)
            if h2 <= k:
                self.gctx.add_(self.gm)
            self.g0.add_(self.gm)

            def g_of(j):
                if j == h2:
                    return self.gctx  # (gm + gctx when 2 h <= K, gctx alone above K)
                if j <= k:
                    return self.gm
                return None  # no incoming gradient at this layer

            cur = g_of(L)
            bufs = (self.t0, self.t1)
            for n, j in enumerate(range(L - 1, 0, -1)):
                nxt, add = bufs[n & 1], g_of(j)
                if add is None:
                    check(lib.rbg_spmm_f32(gt, p(cur), p(nxt), d, 0, st))
                else:
                    check(lib.rbg_spmm_add_f32(gt, p(cur), p(add), p(nxt), d, st))
                cur = nxt
            check(lib.rbg_spmm_add_f32(gt, p(cur), p(self.g0), p(self.ge), d, st))
            check(lib.rbg_emb_reg_grad_nopow_f32(p(uw), p(iw), nu, p(user), p(pos), p(neg), b, d, float(m.reg_weight), p(self.ge), p(self.loss),
                                                 p(self.reg_ws), st))
            self.table_opt.step(self.ge, self.loss)



def fused_stepper(model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graphed=True):
    """The autograd-free training step of `model` if this package has one — ``FusedBPRAdam`` (plain LightGCN),
    ``FusedNGCFAdam`` (plain NGCF, fused layers, widths <= 128), ``FusedSGLAdam`` (plain SGL, d <= 128), ``FusedSimGCLAdam`` /
    ``FusedXSimGCLAdam`` (their ``static_unique`` form), ``FusedNCLAdam``, all on device graph handles — else None (``GraphedStep`` / an eager loop
    serve every other model).  ``graphed`` applies to all but the first."""
    from .models import NCL, NGCF, SGL, SimGCL, XSimGCL
    if not next(model.parameters()).is_cuda:
        return None
    if fused_step_applies(model) and isinstance(model.graph, ops.GraphHandle):
        return FusedBPRAdam(model, lr=lr, betas=betas, eps=eps)
    for cls, step in ((NGCF, FusedNGCFAdam), (SGL, FusedSGLAdam), (SimGCL, FusedSimGCLAdam), (XSimGCL, FusedXSimGCLAdam),
                      (NCL, FusedNCLAdam)):
        if type(model) is cls:
            try:
                return step(model, lr=lr, betas=betas, eps=eps, graphed=graphed)
            except TypeError:
                return None
    return None


def _total(loss):
    """RecBole's trainer sums a tuple of loss terms (XSimGCL returns mf, reg, cl separately, xsimgcl.py:90)."""
    return sum(loss) if isinstance(loss, tuple) else loss


def total_without_last(loss):
    """NCLTrainer._train_epoch during the warm-up epochs (trainer.py:130-133): the prototype term is left out."""
    return sum(loss[:-1]) if isinstance(loss, tuple) else loss


class GraphedStep:
    """One whole training step — ``zero_grad; calculate_loss; backward; Adam.step`` (RecBole ``Trainer._train_epoch``
    [recbole==1.1.1]) — captured ONCE into a HIP graph and replayed per batch.

    The autograd path of NGCF / SGL issues ~300 kernels per step from Python; on MI355X those kernels add up to less than
    half of the step's wall time (NGCF, Gowalla shape: 1.5 ms of kernels in a 3.5 ms step, r01) — the rest is launch
    latency.  A graph replay submits the same kernels, in the same order on the same buffers, with one call.  Works for
    every model of this package whose loss has static shapes (all device work is enqueued on torch's current stream and
    nothing synchronises): LightGCN, NGCF, SGL, and SimGCL / XSimGCL with their ``static_unique`` form (the reference's
    ``torch.unique`` of the batch, simgcl.py:52-53, is a data-dependent shape: a device-to-host sync, which stream capture
    forbids, and a replay would bake in the first batch's unique count; the models' default writes the same loss with a
    one-occurrence mask).  Models that declare ``graph_capturable = False`` (``static_unique = False``, NCL) are refused here.

    ``step(batch)`` copies the batch's index tensors into the captured input buffers and replays; batches must have the
    size of ``example_batch`` (RecBole's last, shorter batch of an epoch goes through ``eager_step``)."""

    def __init__(self, model, example_batch, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, warmup=2, reduce=None):
        self.reduce = reduce or _total  # how a tuple of loss terms becomes the trained scalar (``set_reduce`` re-captures)
        if not next(model.parameters()).is_cuda:
            raise RuntimeError("GraphedStep needs the model on a GPU")
        if not getattr(model, "graph_capturable", True):
            raise RuntimeError(f"{type(model).__name__}.calculate_loss has data-dependent shapes (torch.unique) and cannot be "
                               "captured into a HIP graph; train it eagerly")
        self.model = model
        # fused: one multi-tensor kernel per step instead of the foreach implementation's seven passes over the parameters
        # (NGCF, Gowalla shape: 135 us of a 1 090 us step); same update rule (torch.optim.Adam, RecBole's default learner)
        self.opt = torch.optim.Adam(model.parameters(), lr=lr, betas=betas, eps=eps, capturable=True, fused=True)
        self.static = {k: v.detach().clone() for k, v in example_batch.items()}
        if not model.training:
            model.train()  # (not unconditionally: SGL.train() re-samples its augmented views, sgl.py:94-98)
        # warm-up on a side stream (allocator / lazy-initialisation effects must be out of the way before capture); the
        # parameters and the optimiser state are put back afterwards, so the warm-up leaves no trace in the training run
        saved = [p.detach().clone() for p in model.parameters()]
        side = self._side = torch.cuda.Stream(device=self.static[next(iter(self.static))].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for p, s in zip(model.parameters(), saved):
                p.copy_(s)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        self._capture()

    @staticmethod
    def _graph_objects(model):
        """The graph handles / edge tensors a step of `model` reads.  A captured HIP graph has their device pointers
        baked in, so they are (a) kept alive by this object and (b) compared by identity before every replay."""
        found = []
        for name in ("graph", "sub_graph1", "sub_graph2", "edge_index", "edge_weight"):
            v = getattr(model, name, None)
            found.extend(v if isinstance(v, (list, tuple)) else [v])
        return found

    def _capture(self):
        # the previous capture's loss holds that step's autograd graph, which holds the graph handles it ran on: let go of it
        # BEFORE the capture starts — dropped inside it, the last reference to an old epoch's views would destroy them there
        # (hipFree during a capture invalidates it: SGL's second epoch failed with hipErrorStreamCaptureInvalidated)
        self.loss = None
        self.graph = None
        self._captured_graphs = self._graph_objects(self.model)
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self.reduce(self.model.calculate_loss(self.static))
            self.loss.backward()
            self.opt.step()

    def set_reduce(self, reduce):
        """Another combination of the model's loss terms from now on (NCL after its warm-up epochs): the step is re-captured,
        parameters and optimizer state carry over."""
        reduce = reduce or _total
        if reduce is not self.reduce:
            self.reduce = reduce
            self._capture()

    def _eager(self):
        self.opt.zero_grad(set_to_none=True)
        loss = self.reduce(self.model.calculate_loss(self.static))
        loss.backward()
        self.opt.step()
        return loss

    def step(self, batch):
        for k, v in batch.items():
            if v.shape != self.static[k].shape:
                raise ValueError(f"batch field {k} has shape {tuple(v.shape)}, captured {tuple(self.static[k].shape)}: use eager_step")
            self.static[k].copy_(v, non_blocking=True)
        now = self._graph_objects(self.model)
        if len(now) != len(self._captured_graphs) or any(a is not b for a, b in zip(now, self._captured_graphs)):
            self._capture()  # e.g. SGL.train() sampled new augmented views for this epoch (sgl.py:94-98)
        self.graph.replay()
        return self.loss

    def eager_step(self, batch):
        """The same step without the graph (odd-sized batches).  Gradients left by the graph's buffers are dropped first.
        Runs on the warm-up's side stream, like everything this object does outside a replay: autograd binds the parameters'
        AccumulateGrad nodes to the stream of their first use, and nodes bound to the DEFAULT stream invalidate the next
        capture (SGL re-captures every epoch, right after an epoch's odd-sized last batch came through here)."""
        cur = torch.cuda.current_stream(self._side.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self.opt.zero_grad(set_to_none=True)
            loss = self.reduce(self.model.calculate_loss(batch))
            loss.backward()
            self.opt.step()
            loss = loss.detach()
        cur.wait_stream(self._side)
        return loss
