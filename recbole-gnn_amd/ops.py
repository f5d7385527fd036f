"""Operators: thin torch wrappers over the C ABI (tensors go in as ``data_ptr()``, the current HIP
stream as ``torch.cuda.current_stream().cuda_stream``).  Mirrors ``recbole_gnn/model/layers.py``
(LightGCNConv :8-23, BiGNNConv :41-67) so call sites read like the reference's.

Everything here needs a GPU-resident graph: there is no CPU implementation.
"""
from __future__ import annotations

import ctypes
import weakref

import torch
import torch.nn as nn

from . import _lib
from ._lib import c_vp, check, lib
from .graph import GraphHandle


def _stream(t):
    return c_vp(torch.cuda.current_stream(t.device).cuda_stream)


def _check_dense(t, name, graph=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (the HIP engine has no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if graph is not None and graph.device_index != t.device.index:
        raise RuntimeError(f"{name} is on {t.device} but the graph is on cuda:{graph.device_index}")


def _require_device_graph(graph):
    if not isinstance(graph, GraphHandle):
        raise TypeError(f"expected a GraphHandle, got {type(graph)}")
    if not graph.is_device:
        raise RuntimeError("operator called with a host graph; create it with device='cuda'")


def spmm_raw(graph, x, out=None, accumulate=False):
    """Y = Â·X (no autograd).  x: [n_cols, d] contiguous fp32 on the graph's GPU."""
    _require_device_graph(graph)
    _check_dense(x, "x", graph)
    if x.dim() != 2 or x.shape[0] != graph.n_cols:
        raise ValueError(f"x must be [{graph.n_cols}, d], got {tuple(x.shape)}")
    x = x.contiguous()
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an explicit out tensor")
        out = torch.empty((graph.n_rows, x.shape[1]), dtype=torch.float32, device=x.device)
    else:
        _check_dense(out, "out", graph)
        if tuple(out.shape) != (graph.n_rows, x.shape[1]) or not out.is_contiguous():
            raise ValueError("out must be a contiguous [n_rows, d] tensor")
    if x.shape[1] in (32, 64, 128):
        _auto_sell(graph, x.shape[1])  # (eligible handles only: built from interactions; rbg_spmm_f32 then runs over the plan)
    with torch.cuda.device(x.device):
        check(lib.rbg_spmm_f32(graph.ptr, c_vp(x.data_ptr()), c_vp(out.data_ptr()), x.shape[1], int(bool(accumulate)),
                               _stream(x)))
    return out


class _Spmm(torch.autograd.Function):
    """Autograd for Y = Â·X.  backward = Â^T·dY — the same kernel, because Â is symmetric
    (dataset.py:62-64 builds both directions; SGL views too, sgl.py:113-115)."""

    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        return spmm_raw(graph, x)

    @staticmethod
    def backward(ctx, grad_out):
        return spmm_raw(ctx.graph.transpose(), grad_out.contiguous()), None


def spmm(graph, x):
    return _Spmm.apply(x, graph)


def spmm_add_raw(graph, x, z, out=None):
    """Y = Z + Â·X in one launch (no autograd): a Horner step with its own addend (rbg_spmm_add_f32)."""
    _require_device_graph(graph)
    _check_dense(x, "x", graph)
    _check_dense(z, "z", graph)
    if x.dim() != 2 or x.shape[0] != graph.n_cols or tuple(z.shape) != (graph.n_rows, x.shape[1]):
        raise ValueError(f"x must be [{graph.n_cols}, d] and z [{graph.n_rows}, d]")
    x, z = x.contiguous(), z.contiguous()
    if out is None:
        out = torch.empty((graph.n_rows, x.shape[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.rbg_spmm_add_f32(graph.ptr, c_vp(x.data_ptr()), c_vp(z.data_ptr()), c_vp(out.data_ptr()), x.shape[1], _stream(x)))
    return out


def spmm_noise_raw(graph, x, noise, eps, out=None):
    """Y = Â·X;  Y += sign(Y) * normalize(noise, dim=-1) * eps  (simgcl.py:29-34), no autograd."""
    _require_device_graph(graph)
    _check_dense(x, "x", graph)
    _check_dense(noise, "noise", graph)
    if x.dim() != 2 or x.shape[0] != graph.n_cols or tuple(noise.shape) != (graph.n_rows, x.shape[1]):
        raise ValueError(f"x must be [{graph.n_cols}, d] and noise [{graph.n_rows}, d]")
    x, noise = x.contiguous(), noise.contiguous()
    if out is None:
        out = torch.empty((graph.n_rows, x.shape[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.rbg_spmm_noise_f32(graph.ptr, c_vp(x.data_ptr()), c_vp(out.data_ptr()), c_vp(noise.data_ptr()), x.shape[1],
                                     float(eps), _stream(x)))
    return out


class _SpmmNoise(torch.autograd.Function):
    """The perturbation sign(y) * const has zero gradient (as in torch autograd of simgcl.py:33): backward = Â^T·dY."""

    @staticmethod
    def forward(ctx, x, noise, graph, eps):
        ctx.graph = graph
        return spmm_noise_raw(graph, x, noise, eps)

    @staticmethod
    def backward(ctx, grad_out):
        return spmm_raw(ctx.graph.transpose(), grad_out.contiguous()), None, None, None


def spmm_noise(graph, x, noise, eps):
    return _SpmmNoise.apply(x, noise, graph, float(eps))


# Graph handles for models that hold the reference's dense pair (edge_index, edge_weight).  The cache is keyed by the
# identity of the two tensor OBJECTS (weak references + version counters), never by their device addresses: the
# caching allocator hands the same address to a different graph's tensors as soon as the old ones are freed.
_pair_cache = {}


def graph_from_pair(edge_index, edge_weight, num_nodes, device):
    key = (id(edge_index), id(edge_weight))
    hit = _pair_cache.get(key)
    if hit is not None:
        ref_i, ref_w, ver_i, ver_w, nodes, dev, graph = hit
        if (ref_i() is edge_index and ref_w() is edge_weight and ver_i == edge_index._version
                and ver_w == edge_weight._version and nodes == int(num_nodes) and dev == str(device)):
            return graph
    for k in [k for k, v in _pair_cache.items() if v[0]() is None or v[1]() is None]:
        del _pair_cache[k]  # the tensors died: drop their handles
    graph = GraphHandle.from_edge_index(edge_index, edge_weight, num_nodes, device=device, symmetric=False)
    _pair_cache[key] = (weakref.ref(edge_index), weakref.ref(edge_weight), edge_index._version, edge_weight._version,
                        int(num_nodes), str(device), graph)
    return graph


class LightGCNConv(nn.Module):
    """recbole_gnn/model/layers.py:8-23.  ``forward(x, edge_index, edge_weight)``: ``edge_index`` is
    either a GraphHandle (the ``enable_sparse`` branch, where the reference holds a SparseTensor and
    ``edge_weight`` is None) or the int64 ``[2, nnz]`` tensor with fp32 ``edge_weight`` (the default
    branch); both run the same CSR kernel."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x, edge_index, edge_weight=None):
        if isinstance(edge_index, GraphHandle):
            return spmm(edge_index, x)
        return spmm(graph_from_pair(edge_index, edge_weight, x.shape[0], x.device), x)

    def __repr__(self):
        return "{}({})".format(self.__class__.__name__, self.dim)


# ---- fused LightGCN propagation ------------------------------------------------------------

def lightgcn_forward_raw(graphs, user_w, item_w, n_layers, keep_layers=False, out=None, layers=None):
    """cat -> K x (Â·) -> mean, one C call (lightgcn.py:60-81).  Returns (mean [N,d], layers or None).
    ``out`` / ``layers`` let a caller reuse its own buffers ([N,d] and [max(K,1),N,d])."""
    graphs = list(graphs) if isinstance(graphs, (list, tuple)) else [graphs]
    for g in graphs:
        _require_device_graph(g)
    _check_dense(user_w, "user embedding", graphs[0])
    _check_dense(item_w, "item embedding", graphs[0])
    user_w, item_w = user_w.contiguous(), item_w.contiguous()
    n_users, d = user_w.shape
    n = n_users + item_w.shape[0]
    if item_w.shape[1] != d:
        raise ValueError("user and item embeddings differ in width")
    if n != graphs[0].n_rows:
        raise ValueError(f"graph has {graphs[0].n_rows} nodes but the tables hold {n} rows")
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=user_w.device)
    if layers is None:
        layers = torch.empty((max(n_layers, 1), n, d), dtype=torch.float32, device=user_w.device)
    if tuple(out.shape) != (n, d) or tuple(layers.shape) != (max(n_layers, 1), n, d) or \
            not (out.is_contiguous() and layers.is_contiguous()):
        raise ValueError("out must be contiguous [N, d] and layers contiguous [max(K,1), N, d]")
    _check_dense(out, "out", graphs[0])
    _check_dense(layers, "layers", graphs[0])
    if d in (32, 64, 128):
        for g in graphs:
            _auto_sell(g, d)
    arr = (c_vp * len(graphs))(*[g.ptr for g in graphs])
    # (no caller of this wrapper reads `layers` unless keep_layers: the library may use it as scratch in any layout)
    flags = _lib.FWD_KEEP_LAST_LAYER if keep_layers else _lib.FWD_LAYERS_SCRATCH
    with torch.cuda.device(user_w.device):
        check(lib.rbg_lightgcn_forward_f32(arr, len(graphs), n_users, c_vp(user_w.data_ptr()), c_vp(item_w.data_ptr()),
                                           c_vp(out.data_ptr()), c_vp(layers.data_ptr()), d, n_layers, flags,
                                           _stream(user_w)))
    return out, (layers if keep_layers else None)


def _auto_sell(graph, d):
    """A handle created while option "sell_auto" was off (or before a re-plan was wanted) gets its column-slab plan on its
    first propagation: one call into the library's planner (``rbg_graph_plan_sell``).  Handles created normally already carry
    one — ``rbg_graph_create*`` plans — and a failed attempt is remembered, not retried."""
    if graph.__dict__.get("_sell_tried") or torch.cuda.is_current_stream_capturing():  # (planning allocates and synchronises)
        return
    graph._sell_tried = True
    from . import graph as _g
    if not _g.get_option("sell") or not graph.sell_eligible(d) or graph.has_sell(d):
        return
    try:
        graph.plan_sell()
    except _lib.RbgError as ex:
        if ex.code != _lib.RBG_EUNSUPPORTED:  # (not applicable: sell_status() holds the reason, the binned kernel serves the call)
            import warnings
            warnings.warn(f"column-slab plan not built ({ex}); the propagation runs on the binned SpMM kernel")


class _LightGCNForward(torch.autograd.Function):
    """mean_k(Â^k E0) is linear in E0, so backward needs no saved activations:
    dE0 = c + Â_0(c + Â_1(... + Â_{K-1} c)), c = dOut/(K+1)  (Horner, K SpMMs; Â symmetric)."""

    @staticmethod
    def forward(ctx, user_w, item_w, n_layers, *graphs):
        ctx.graphs, ctx.n_layers, ctx.n_users = graphs, n_layers, user_w.shape[0]
        out, _ = lightgcn_forward_raw(graphs, user_w, item_w, n_layers)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        k_layers = ctx.n_layers
        g = grad_out.contiguous()
        _check_dense(g, "grad_out", ctx.graphs[0])
        grad_e0 = torch.empty_like(g)
        work = torch.empty_like(g) if k_layers >= 2 else None
        graphs = [gr.transpose() for gr in ctx.graphs]  # the handle itself when symmetric
        arr = (c_vp * len(graphs))(*[gr.ptr for gr in graphs])
        with torch.cuda.device(g.device):
            check(lib.rbg_lightgcn_backward_f32(arr, len(graphs), c_vp(g.data_ptr()), c_vp(grad_e0.data_ptr()),
                                                c_vp(work.data_ptr()) if work is not None else None, g.shape[1], k_layers,
                                                _stream(g)))
        return (grad_e0[:ctx.n_users], grad_e0[ctx.n_users:], None) + (None,) * len(ctx.graphs)


def lightgcn_forward(graphs, user_w, item_w, n_layers):
    """Differentiable fused propagation -> [N, d] mean embedding."""
    graphs = list(graphs) if isinstance(graphs, (list, tuple)) else [graphs]
    return _LightGCNForward.apply(user_w, item_w, n_layers, *graphs)


class _BprEmbLoss(torch.autograd.Function):
    """BPRLoss on the propagated rows + reg_weight * EmbLoss on the ego rows (lightgcn.py:93-110; ncl.py:186-196) with their
    gradients in two or three library launches (rbg_bpr_grad_f32 + rbg_emb_reg_grad[_nopow]_f32) instead of six index_selects,
    ~20 elementwise / reduction launches and — in backward — six zero-filled tables with an index_add each."""

    @staticmethod
    def forward(ctx, mean, uw, iw, user, pos, neg, reg_weight, require_pow):
        nu, d = uw.shape
        ni, b = iw.shape[0], user.shape[0]
        gm, ge = torch.empty_like(mean), torch.zeros_like(mean)
        loss, reg = torch.zeros((), dtype=torch.float32, device=mean.device), torch.zeros((), dtype=torch.float32, device=mean.device)
        st = _stream(mean)
        ptr = lambda t: c_vp(t.data_ptr())  # noqa: E731
        with torch.cuda.device(mean.device):
            check(lib.rbg_bpr_grad_f32(ptr(mean), nu, ni, ptr(user), ptr(pos), ptr(neg), b, d, ptr(gm), ptr(loss), st))
            if reg_weight != 0:
                if require_pow:
                    check(lib.rbg_emb_reg_grad_f32(ptr(uw), ptr(iw), nu, ptr(user), ptr(pos), ptr(neg), b, d, float(reg_weight), ptr(ge), ptr(reg), st))
                else:
                    ws = torch.empty(3, dtype=torch.float32, device=mean.device)
                    check(lib.rbg_emb_reg_grad_nopow_f32(ptr(uw), ptr(iw), nu, ptr(user), ptr(pos), ptr(neg), b, d, float(reg_weight), ptr(ge),
                                                         ptr(reg), ptr(ws), st))
        ctx.save_for_backward(gm, ge)
        ctx.nu = nu
        return loss, reg

    @staticmethod
    def backward(ctx, go_mf, go_reg):
        gm, ge = ctx.saved_tensors
        ge = ge * go_reg
        return gm * go_mf, ge[: ctx.nu], ge[ctx.nu:], None, None, None, None, None


def bpr_emb_loss(user_all, item_all, user_w, item_w, user, pos, neg, reg_weight, require_pow):
    """``(BPRLoss(<u, p>, <u, n>), reg_weight * EmbLoss(ego rows))`` of lightgcn.py:93-110 / xsimgcl.py:78-85 for ``user_all,
    item_all`` = the two halves of ONE contiguous [N, d] tensor (what ``torch.split`` of the propagated mean returns), or None
    when the inputs do not have that form (the caller then spells the loss in torch)."""
    base = user_all._base if user_all._base is not None and user_all._base is item_all._base else None
    nu, ni = user_w.shape[0], item_w.shape[0]
    if (base is None or not base.is_cuda or base.dtype != torch.float32 or base.dim() != 2 or not base.is_contiguous()
            or base.shape[0] != nu + ni or user_all.shape[0] != nu or item_all.shape[0] != ni
            or user_all.data_ptr() != base.data_ptr() or item_all.data_ptr() != base.data_ptr() + 4 * nu * base.shape[1]
            or user_w.shape[1] != base.shape[1] or not (user_w.is_contiguous() and item_w.is_contiguous())):
        return None
    dev = base.device
    user, pos, neg = (t.to(device=dev, dtype=torch.int64).contiguous() for t in (user, pos, neg))
    return _BprEmbLoss.apply(base, user_w, item_w, user, pos, neg, float(reg_weight), bool(require_pow))


class _LayerMean(torch.autograd.Function):
    """mean over a list of equally shaped layer outputs — ``torch.mean(torch.stack(list, dim=1), dim=1)`` of lightgcn.py:77-78 /
    simgcl.py:34-35 / ncl.py:99-100 — as ONE launch (rbg_mean_f32) instead of a stack copy and a reduction over it; the backward
    is one scaling whose result every input shares."""

    @staticmethod
    def forward(ctx, *layers):
        xs = [t.contiguous() for t in layers]
        out = torch.empty_like(xs[0])
        arr = (c_vp * len(xs))(*[t.data_ptr() for t in xs])
        with torch.cuda.device(out.device):
            check(lib.rbg_mean_f32(arr, len(xs), out.numel(), 1.0 / len(xs), c_vp(out.data_ptr()), _stream(out)))
        ctx.k = len(xs)
        return out

    @staticmethod
    def backward(ctx, grad):
        g = grad * (1.0 / ctx.k)
        return (g,) * ctx.k


def layer_mean(layers):
    """Mean of a list of [N, d] tensors (differentiable).  Falls back to the reference's stack + mean off the GPU, for other
    dtypes or more layers than one launch of the library takes."""
    layers = list(layers)
    if (len(layers) > _lib.MAX_FUSED_LAYERS + 1 or not layers[0].is_cuda or any(t.dtype != torch.float32 or t.shape != layers[0].shape for t in layers)):
        return torch.mean(torch.stack(layers, dim=1), dim=1)
    return _LayerMean.apply(*layers)


# ---- independent propagations of one E0 on concurrent HIP streams (SGL: the full graph + two views) ----------------------

_SIDE_STREAMS = {}


def _side_streams(device, n):
    key = torch.device(device).index or 0
    pool = _SIDE_STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _fork_join(fns, device):
    """Run fns[0] on the current stream and fns[1:] on side streams forked from it by an event, then join.  The callables
    only LAUNCH on memory the caller allocated on the current stream beforehand (nothing is allocated or freed on a side
    stream, so the caching allocator — and a HIP-graph capture of the caller — see one stream's worth of lifetimes)."""
    main = torch.cuda.current_stream(device)
    side = _side_streams(device, len(fns) - 1)
    fork = torch.cuda.Event()
    fork.record(main)
    for s, fn in zip(side, fns[1:]):
        s.wait_event(fork)
        with torch.cuda.stream(s):
            fn()
    fns[0]()
    for s in side:
        done = torch.cuda.Event()
        done.record(s)
        main.wait_event(done)


class _LightGCNForwardViews(torch.autograd.Function):
    """V propagations of the SAME E0 over V graphs (sgl.py:219-221: the full graph and the two augmented views), each a
    chain of K dependent launches that leaves a third of the GPU idle in its ramps and tails: the chains are
    issued on V HIP streams and run concurrently, forward and backward.  Values are those of V ``lightgcn_forward`` calls.
    MEASURED SLOWER than the sequential issue (328 vs 268 us for SGL's three forwards at the Gowalla shape): every launch
    already covers all XCDs and owns their L2s (one slab of one table each); concurrent launches evict each other.  Kept,
    tested, off by default (``SGL.concurrent_views``)."""

    @staticmethod
    def forward(ctx, user_w, item_w, n_layers, *graphs):
        ctx.graphs, ctx.n_layers, ctx.n_users = graphs, n_layers, user_w.shape[0]
        user_w, item_w = user_w.contiguous(), item_w.contiguous()
        n, d = user_w.shape[0] + item_w.shape[0], user_w.shape[1]
        f = dict(dtype=torch.float32, device=user_w.device)
        outs = [torch.empty((n, d), **f) for _ in graphs]
        scratch = [torch.empty((max(n_layers, 1), n, d), **f) for _ in graphs]
        for g in graphs:
            if d in (64, 128):
                _auto_sell(g, d)  # (plans are built on the caller's stream, before the fork)
        _fork_join([lambda g=g, o=o, l=l: lightgcn_forward_raw(g, user_w, item_w, n_layers, out=o, layers=l)
                    for g, o, l in zip(graphs, outs, scratch)], user_w.device)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        k_layers, graphs = ctx.n_layers, ctx.graphs
        live = [(gr.transpose(), g.contiguous()) for gr, g in zip(graphs, grads) if g is not None]
        if not live:
            return (None, None, None) + (None,) * len(graphs)
        dev = live[0][1].device
        parts = [torch.empty_like(g) for _, g in live]
        works = [torch.empty_like(g) if k_layers >= 2 else None for _, g in live]

        def chain(gr, g, out, work):
            arr = (c_vp * 1)(gr.ptr)
            with torch.cuda.device(dev):
                check(lib.rbg_lightgcn_backward_f32(arr, 1, c_vp(g.data_ptr()), c_vp(out.data_ptr()),
                                                    c_vp(work.data_ptr()) if work is not None else None, g.shape[1], k_layers,
                                                    _stream(g)))
        _fork_join([lambda gr=gr, g=g, o=o, w=w: chain(gr, g, o, w) for (gr, g), o, w in zip(live, parts, works)], dev)
        total = parts[0]
        for p in parts[1:]:
            total = total + p
        return (total[:ctx.n_users], total[ctx.n_users:], None) + (None,) * len(graphs)


def lightgcn_forward_views(graphs, user_w, item_w, n_layers):
    """[mean_v] for V single-graph propagations of one E0, differentiable, issued on concurrent streams."""
    return list(_LightGCNForwardViews.apply(user_w, item_w, n_layers, *graphs))


# ---- NGCF ----------------------------------------------------------------------------------

def bignn_conv_raw(graph, x, w1, b1, w2, b2, out=None, leaky_norm=False, slope=0.2):
    """BiGNNConv.forward (layers.py:54-58) [+ LeakyReLU + L2-normalize, ngcf.py:96,98].
    x may be a column slice of a wider row-major buffer (row stride = x.stride(0)); same for out.
    Returns (out, P) with P = Â·x."""
    _require_device_graph(graph)
    _check_dense(x, "x", graph)
    for t, nm in ((w1, "W1"), (b1, "b1"), (w2, "W2"), (b2, "b2")):
        _check_dense(t, nm, graph)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be 2-D with unit column stride")
    n, d_in = x.shape
    d_out = w1.shape[0]
    if tuple(w1.shape) != (d_out, d_in) or tuple(w2.shape) != (d_out, d_in):
        raise ValueError("W1/W2 must be [d_out, d_in]")
    if out is None:
        out = torch.empty((n, d_out), dtype=torch.float32, device=x.device)
    elif out.stride(1) != 1 or tuple(out.shape) != (n, d_out):
        raise ValueError("out must be [N, d_out] with unit column stride")
    p = torch.empty((n, d_in), dtype=torch.float32, device=x.device)
    flags = _lib.BIGNN_LEAKY_NORM if leaky_norm else _lib.BIGNN_CONV_ONLY
    with torch.cuda.device(x.device):
        check(lib.rbg_bignn_conv_f32(graph.ptr, c_vp(x.data_ptr()), x.stride(0), c_vp(w1.contiguous().data_ptr()),
                                     c_vp(b1.contiguous().data_ptr()), c_vp(w2.contiguous().data_ptr()),
                                     c_vp(b2.contiguous().data_ptr()), c_vp(out.data_ptr()), out.stride(0),
                                     c_vp(p.data_ptr()), d_in, d_out, flags, float(slope), _stream(x)))
    return out, p


def bignn_dense_raw(p, x, w1, b1, w2, b2, out=None, leaky_norm=False, slope=0.2):
    """The dense half of BiGNNConv.forward from a product P = Â·x the caller holds (layers.py:56-58) [+ LeakyReLU +
    L2-normalize]: rbg_bignn_dense_f32.  x / out may be column slices of wider row-major buffers."""
    for t, nm in ((p, "P"), (x, "x"), (w1, "W1"), (b1, "b1"), (w2, "W2"), (b2, "b2")):
        _check_dense(t, nm)
    if x.dim() != 2 or x.stride(1) != 1 or not p.is_contiguous() or tuple(p.shape) != tuple(x.shape):
        raise ValueError("x must be 2-D with unit column stride and P a contiguous tensor of the same shape")
    n, d_in = x.shape
    d_out = w1.shape[0]
    if tuple(w1.shape) != (d_out, d_in) or tuple(w2.shape) != (d_out, d_in):
        raise ValueError("W1/W2 must be [d_out, d_in]")
    if out is None:
        out = torch.empty((n, d_out), dtype=torch.float32, device=x.device)
    elif out.stride(1) != 1 or tuple(out.shape) != (n, d_out):
        raise ValueError("out must be [N, d_out] with unit column stride")
    flags = _lib.BIGNN_LEAKY_NORM if leaky_norm else _lib.BIGNN_CONV_ONLY
    with torch.cuda.device(x.device):
        check(lib.rbg_bignn_dense_f32(c_vp(p.data_ptr()), c_vp(x.data_ptr()), x.stride(0) if n > 1 else d_in,
                                      c_vp(w1.contiguous().data_ptr()), c_vp(b1.contiguous().data_ptr()),
                                      c_vp(w2.contiguous().data_ptr()), c_vp(b2.contiguous().data_ptr()), c_vp(out.data_ptr()),
                                      out.stride(0) if n > 1 else d_out, n, d_in, d_out, flags, float(slope), _stream(x)))
    return out


def bignn_wgrad_raw(g, p, x):
    """(G^T (P + X), G^T (P * X), sum_rows G): the weight / bias gradients of BiGNNConv (layers.py:54-58)."""
    for t, name in ((g, "g"), (p, "p"), (x, "x")):
        _check_dense(t, name)
    g = g if g.stride(1) == 1 else g.contiguous()
    x = x if x.stride(1) == 1 else x.contiguous()
    p = p.contiguous()
    n, d_out = g.shape
    d_in = x.shape[1]
    gw1 = torch.empty((d_out, d_in), dtype=torch.float32, device=g.device)
    gw2 = torch.empty_like(gw1)
    gb = torch.empty(d_out, dtype=torch.float32, device=g.device)
    nbytes = _lib.c_i64()
    check(lib.rbg_bignn_wgrad_workspace(n, d_in, d_out, ctypes.byref(nbytes)))
    work = torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device):
        check(lib.rbg_bignn_wgrad_f32(c_vp(g.data_ptr()), g.stride(0) if n > 1 else d_out, c_vp(p.data_ptr()), c_vp(x.data_ptr()),
                                      x.stride(0) if n > 1 else d_in, n, d_in, d_out, c_vp(gw1.data_ptr()), c_vp(gw2.data_ptr()),
                                      c_vp(gb.data_ptr()), c_vp(work.data_ptr()), _stream(g)))
    return gw1, gw2, gb


class _BiGNNConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, graph):
        out, p = bignn_conv_raw(graph, x, w1, b1, w2, b2)
        ctx.graph = graph
        ctx.save_for_backward(x, p, w1, w2)
        return out

    @staticmethod
    def backward(ctx, g):
        x, p, w1, w2 = ctx.saved_tensors
        g = g.contiguous()
        gt = g @ w1            # d/d(P+X)
        gi = g @ w2            # d/d(P*X)
        gp = gt + gi * x
        gx = gt + gi * p + spmm_raw(ctx.graph.transpose(), gp.contiguous())
        if x.shape[1] <= 128 and g.shape[1] <= 128:
            gw1, gw2, gb = bignn_wgrad_raw(g, p, x)  # one pass over G, P, X instead of two 64 x 64 x N rocBLAS GEMMs
        else:
            gw1 = g.t() @ (p + x)
            gw2 = g.t() @ (p * x)
            gb = g.sum(dim=0)
        return gx, gw1, gb, gw2, gb, None


class _BiGNNLayer(torch.autograd.Function):
    """One NGCF layer with its tail — BiGNNConv -> LeakyReLU(slope) -> F.normalize (layers.py:54-58, ngcf.py:96,98) — as
    one forward call (rbg_bignn_layer_f32) and one backward call (rbg_bignn_backward_f32: tail backward, the two
    G·W products, the weight / bias gradients and the propagated input gradient)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, graph, slope, mask):
        _require_device_graph(graph)
        _check_dense(x, "x", graph)
        x = x if x.stride(1) == 1 else x.contiguous()
        w1, b1, w2, b2 = w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()
        n, d_in = x.shape
        d_out = w1.shape[0]
        y = torch.empty((n, d_out), dtype=torch.float32, device=x.device)
        p = torch.empty((n, d_in), dtype=torch.float32, device=x.device)
        inv = torch.empty(n, dtype=torch.float32, device=x.device)
        if mask is not None:
            _check_dense(mask, "mask", graph)
            if tuple(mask.shape) != (n, d_out):
                raise ValueError(f"mask must be [{n}, {d_out}]")
            mask = mask.contiguous()
        with torch.cuda.device(x.device):
            check(lib.rbg_bignn_layer_f32(graph.ptr, c_vp(x.data_ptr()), x.stride(0) if n > 1 else d_in, c_vp(w1.data_ptr()),
                                          c_vp(b1.data_ptr()), c_vp(w2.data_ptr()), c_vp(b2.data_ptr()), c_vp(y.data_ptr()), d_out,
                                          c_vp(p.data_ptr()), c_vp(inv.data_ptr()), c_vp(mask.data_ptr()) if mask is not None else None,
                                          d_in, d_out, float(slope), _stream(x)))
        ctx.graph, ctx.slope, ctx.has_mask = graph, float(slope), mask is not None
        ctx.save_for_backward(x, p, y, inv, w1, w2, *([mask] if mask is not None else []))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, p, y, inv, w1, w2 = ctx.saved_tensors[:6]
        mask = ctx.saved_tensors[6] if ctx.has_mask else None
        gx, gw1, gw2, gb = bignn_backward_raw(ctx.graph.transpose(), gy, y, inv, mask, x, p, w1, w2, ctx.slope)
        return gx, gw1, gb, gw2, gb, None, None, None


def bignn_backward_raw(graph_t, gy, y, inv, mask, x, p, w1, w2, slope=0.2):
    """rbg_bignn_backward_f32: the backward of one NGCF layer (autograd of layers.py:54-58 [+ ngcf.py:96-98 when ``inv`` —
    the rows' 1 / norm saved by the forward — is given; ``mask`` is the forward's scaled dropout mask]) from the upstream
    gradient ``gy``, the saved output ``y``, input ``x`` and product ``p`` = Â·x.  ``graph_t`` is the TRANSPOSED graph.
    Returns (dX, dW1, dW2, db)."""
    gy = gy if gy.stride(1) == 1 else gy.contiguous()
    n, d_in = x.shape
    d_out = w1.shape[0]
    gx = torch.empty((n, d_in), dtype=torch.float32, device=x.device)
    gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
    gb = torch.empty(d_out, dtype=torch.float32, device=x.device)
    nbytes = _lib.c_i64()
    check(lib.rbg_bignn_backward_workspace(n, d_in, d_out, ctypes.byref(nbytes)))
    work = torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.rbg_bignn_backward_f32(graph_t.ptr, c_vp(gy.data_ptr()), gy.stride(0) if n > 1 else d_out,
                                         c_vp(y.data_ptr()) if y is not None else None, d_out,
                                         c_vp(inv.data_ptr()) if inv is not None else None,
                                         c_vp(mask.data_ptr()) if mask is not None else None, c_vp(x.data_ptr()),
                                         x.stride(0) if n > 1 else d_in, c_vp(p.data_ptr()), c_vp(w1.data_ptr()),
                                         c_vp(w2.data_ptr()), d_in, d_out, float(slope), c_vp(gx.data_ptr()), c_vp(gw1.data_ptr()),
                                         c_vp(gw2.data_ptr()), c_vp(gb.data_ptr()), c_vp(work.data_ptr()), _stream(x)))
    return gx, gw1, gw2, gb


_ones_cache = {}
DROPOUT_ONE_LAUNCH = __import__("os").environ.get("RBG_DROPOUT_ONE_LAUNCH", "1") != "0"


def dropout_mask(n, d, p_drop, device):
    """The scaled keep mask of ``nn.Dropout(p)`` (0 or 1 / (1 - p)) as an fp32 [n, d] tensor on torch's generator.  r06: ONE
    launch — torch's own dropout kernel on a cached table of ones (draw, keep, scale: what ``nn.Dropout`` itself runs; NGCF epoch
    0.302 -> 0.293 s); ``RBG_DROPOUT_ONE_LAUNCH=0``: Bernoulli draw + scale, two launches (r04; the compare / cast / divide
    spelling before that cost four passes over [N, d] per layer, 140 us of a 660 us NGCF step at the Gowalla shape)."""
    if p_drop >= 1.0:  # nn.Dropout(p = 1): everything dropped
        return torch.zeros((n, d), dtype=torch.float32, device=device)
    if DROPOUT_ONE_LAUNCH:  # r06: torch's own dropout kernel on a table of ones — ONE launch (draw, keep, scale)
        key = (int(n), int(d), str(device))
        ones = _ones_cache.get(key)
        if ones is None:
            if len(_ones_cache) >= 8:
                _ones_cache.clear()
            ones = _ones_cache[key] = torch.ones((n, d), dtype=torch.float32, device=device)
        return torch.nn.functional.dropout(ones, p_drop, training=True)
    return torch.empty((n, d), dtype=torch.float32, device=device).bernoulli_(1.0 - p_drop).mul_(1.0 / (1.0 - p_drop))


def bignn_layer(x, w1, b1, w2, b2, graph, slope=0.2, p_drop=0.0, mask=None):
    """normalize(dropout(LeakyReLU(BiGNNConv(x)))) — one NGCF layer (ngcf.py:94-98) with fused forward and backward;
    d_in, d_out <= 128.  ``p_drop`` > 0 draws the scaled keep mask (``dropout_mask``) with torch's RNG; ``mask`` supplies one."""
    if mask is None and p_drop > 0:
        mask = dropout_mask(x.shape[0], w1.shape[0], p_drop, x.device)
    return _BiGNNLayer.apply(x, w1, b1, w2, b2, graph, float(slope), mask)


class BiGNNConv(nn.Module):
    """recbole_gnn/model/layers.py:41-67: lin1(ÂX + X) + lin2(ÂX ⊙ X)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin1 = nn.Linear(in_features=in_channels, out_features=out_channels)
        self.lin2 = nn.Linear(in_features=in_channels, out_features=out_channels)

    def forward(self, x, edge_index, edge_weight=None):
        graph = edge_index if isinstance(edge_index, GraphHandle) else graph_from_pair(edge_index, edge_weight,
                                                                                       x.shape[0], x.device)
        return _BiGNNConv.apply(x, self.lin1.weight, self.lin1.bias, self.lin2.weight, self.lin2.bias, graph)

    def __repr__(self):
        return "{}({},{})".format(self.__class__.__name__, self.in_channels, self.out_channels)


# ---- scoring -------------------------------------------------------------------------------

def score(u, items):
    """u [B, d] · items [n, d]^T -> [B, n]  (lightgcn.py:131), fp32 MFMA."""
    _check_dense(u, "u")
    _check_dense(items, "items")
    if u.dim() != 2 or items.dim() != 2 or u.shape[1] != items.shape[1]:
        raise ValueError("u must be [B, d] and items [n, d]")
    if u.stride(1) != 1:
        u = u.contiguous()
    if items.stride(1) != 1:
        items = items.contiguous()
    out = torch.empty((u.shape[0], items.shape[0]), dtype=torch.float32, device=u.device)
    with torch.cuda.device(u.device):
        check(lib.rbg_score_f32(c_vp(u.data_ptr()), u.stride(0) if u.shape[0] > 1 else u.shape[1],
                                c_vp(items.data_ptr()), items.stride(0) if items.shape[0] > 1 else items.shape[1],
                                c_vp(out.data_ptr()), u.shape[0], items.shape[0], u.shape[1], _stream(u)))
    return out


def gather_rows(src, idx):
    """src[idx] for a 2-D fp32 src with unit column stride (lightgcn.py:128)."""
    _check_dense(src, "src")
    idx = idx.to(device=src.device, dtype=torch.int64).contiguous()
    if src.stride(1) != 1:
        src = src.contiguous()
    out = torch.empty((idx.shape[0], src.shape[1]), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        check(lib.rbg_gather_rows_f32(c_vp(src.data_ptr()), src.stride(0) if src.shape[0] > 1 else src.shape[1],
                                      c_vp(idx.data_ptr()), c_vp(out.data_ptr()), idx.shape[0], src.shape[1],
                                      _stream(src)))
    return out


def _history_csr(graph):
    """(rowptr, col) of a graph handle as device tensors (cached on the handle): user u's history = its row."""
    hit = getattr(graph, "_hist_csr", None)
    if hit is None:
        rowptr, col, _ = graph.export_csr()
        hit = (torch.from_numpy(rowptr).to(graph.device), torch.from_numpy(col.astype("int64")).to(graph.device))
        graph._hist_csr = hit
    return hit


def full_sort_topk(history, user_all, item_all, users, k):
    """Top-k items per user of ``user_all[users] @ item_all.T`` with the PAD item and each user's training history
    masked (full_sort_predict + RecBole's ``_full_sort_batch_eval`` masking + ``torch.topk``), without ever writing the
    [B, n_items] score matrix.  ``history``: the training GraphHandle (or None).  Returns (values [B,k], item ids [B,k])."""
    _check_dense(user_all, "user_all")
    _check_dense(item_all, "item_all")
    if history is not None:
        _require_device_graph(history)
    user_all, item_all = user_all.contiguous(), item_all.contiguous()
    users = users.to(device=user_all.device, dtype=torch.int64).contiguous()
    b, (n_users, d), n_items = users.shape[0], user_all.shape, item_all.shape[0]
    if k > 32 or d > 256:
        # beyond the fused kernel's list capacity (e.g. RecBole configs with topk: [50]): the reference's own sequence —
        # score matrix, PAD and history to -inf, torch.topk (Trainer._full_sort_batch_eval [recbole==1.1.1])
        scores = score(gather_rows(user_all, users), item_all)
        scores[:, 0] = float("-inf")
        if history is not None:
            rowptr, col = _history_csr(history)
            cnt = rowptr[users + 1] - rowptr[users]
            rows = torch.repeat_interleave(torch.arange(b, device=users.device), cnt)
            start = torch.repeat_interleave(rowptr[users] - (torch.cumsum(cnt, 0) - cnt), cnt)
            items = col[start + torch.arange(rows.shape[0], device=users.device)] - n_users
            scores[rows, items] = float("-inf")
        kk = min(k, n_items)
        vals, idx = torch.topk(scores, kk, dim=1)
        idx = torch.where(torch.isinf(vals) & (vals < 0), torch.full_like(idx, -1), idx)
        if kk < k:
            vals = torch.cat([vals, vals.new_full((b, k - kk), float("-inf"))], 1)
            idx = torch.cat([idx, idx.new_full((b, k - kk), -1)], 1)
        return vals, idx
    nbytes = _lib.c_i64()
    check(lib.rbg_full_sort_topk_workspace(b, n_items, k, ctypes.byref(nbytes)))
    work = torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=user_all.device)
    vals = torch.empty((b, k), dtype=torch.float32, device=user_all.device)
    idx = torch.empty((b, k), dtype=torch.int64, device=user_all.device)
    with torch.cuda.device(user_all.device):
        check(lib.rbg_full_sort_topk_f32(history.ptr if history is not None else None, c_vp(user_all.data_ptr()),
                                         c_vp(item_all.data_ptr()), c_vp(users.data_ptr()), b, n_users, n_items, d, k,
                                         c_vp(vals.data_ptr()), c_vp(idx.data_ptr()), c_vp(work.data_ptr()),
                                         _stream(user_all)))
    return vals, idx


# ---------------------------------------------------------------------------------------------------------------------
# InfoNCE denominator (sgl.py:195-198, :204-207) without the [B, n] matrix
# ---------------------------------------------------------------------------------------------------------------------
def _lse_workspace(b, n, d, device):
    nbytes = _lib.c_i64()
    check(lib.rbg_lse_rows_workspace(b, n, d, ctypes.byref(nbytes)))
    return torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=device)


def lse_rows_raw(q, c, scale, shift=0.0):
    """lse[b] = log sum_j exp(scale * <q[b], c[j]>) (no autograd)."""
    _check_dense(q, "q")
    _check_dense(c, "c")
    if q.shape[1] != c.shape[1]:
        raise ValueError(f"q is [*, {q.shape[1]}] but c is [*, {c.shape[1]}]")
    q, c = q.contiguous(), c.contiguous()
    b, d = q.shape
    out = torch.empty(b, dtype=torch.float32, device=q.device)
    work = _lse_workspace(b, c.shape[0], d, q.device)
    with torch.cuda.device(q.device):
        check(lib.rbg_lse_rows_f32(c_vp(q.data_ptr()), d, b, c_vp(c.data_ptr()), d, c.shape[0], d, float(scale),
                                   float(shift), c_vp(out.data_ptr()), c_vp(work.data_ptr()), _stream(q)))
    return out


class _LseRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, c, scale, shift):
        q, c = q.contiguous(), c.contiguous()
        lse = lse_rows_raw(q, c, scale, shift)
        ctx.save_for_backward(q, c, lse)
        ctx.scale, ctx.shift = float(scale), float(shift)
        return lse

    @staticmethod
    def backward(ctx, grad_lse):
        q, c, lse = ctx.saved_tensors
        b, d = q.shape
        n = c.shape[0]
        grad_lse = grad_lse.contiguous().to(torch.float32)
        gq = torch.empty_like(q) if ctx.needs_input_grad[0] else None
        gc = torch.empty_like(c) if ctx.needs_input_grad[1] else None
        work = _lse_workspace(b, n, d, q.device)
        with torch.cuda.device(q.device):
            check(lib.rbg_lse_rows_backward_f32(c_vp(q.data_ptr()), d, b, c_vp(c.data_ptr()), d, n, d, ctx.scale, ctx.shift,
                                                c_vp(lse.data_ptr()), c_vp(grad_lse.data_ptr()),
                                                c_vp(gq.data_ptr()) if gq is not None else None,
                                                c_vp(gc.data_ptr()) if gc is not None else None,
                                                c_vp(work.data_ptr()), _stream(q)))
        return gq, gc, None, None


def lse_rows(q, c, scale, shift=0.0):
    """Differentiable ``torch.logsumexp(scale * q @ c.T, dim=1)`` that never writes the [B, n] matrix."""
    return _LseRows.apply(q, c, float(scale), float(shift))


class _InfoNCE(torch.autograd.Function):
    """Loss and both table gradients come out of ONE library call; backward only scales them."""

    @staticmethod
    def forward(ctx, t1, t2, idx, tau, row_w, col_w):
        t1, t2 = t1.contiguous(), t2.contiguous()
        idx = idx.to(device=t1.device, dtype=torch.int64).contiguous()
        n, d = t2.shape
        b = idx.shape[0]
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g1 = torch.zeros_like(t1) if need1 else None
        g2 = torch.zeros_like(t2) if need2 else None
        loss = torch.zeros(1, dtype=torch.float32, device=t1.device)
        nbytes = _lib.c_i64()
        check(lib.rbg_infonce_workspace(b, n, d, ctypes.byref(nbytes)))
        work = torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=t1.device)
        ptr = lambda t: c_vp(t.data_ptr()) if t is not None else None  # noqa: E731
        with torch.cuda.device(t1.device):
            if row_w is None and col_w is None:
                check(lib.rbg_infonce_f32(ptr(t1), ptr(t2), n, d, ptr(idx), b, float(tau), 1.0, ptr(loss), ptr(g1), ptr(g2), ptr(work), _stream(t1)))
            else:
                row_w = row_w.to(device=t1.device, dtype=torch.float32).contiguous() if row_w is not None else None
                col_w = col_w.to(device=t1.device, dtype=torch.float32).contiguous() if col_w is not None else None
                check(lib.rbg_infonce_masked_f32(ptr(t1), ptr(t2), n, d, ptr(idx), b, float(tau), 1.0, ptr(row_w), ptr(col_w), ptr(loss),
                                                 ptr(g1), ptr(g2), ptr(work), _stream(t1)))
        ctx.save_for_backward(*(g for g in (g1, g2) if g is not None))
        ctx.have = (need1, need2)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        saved = list(ctx.saved_tensors)
        g1 = saved.pop(0) * grad_out if ctx.have[0] else None
        g2 = saved.pop(0) * grad_out if ctx.have[1] else None
        return g1, g2, None, None, None, None


def info_nce(t1, t2, idx, tau, row_w=None, col_w=None):
    """-sum_b log( exp(<a_b,p_b>/tau) / sum_j exp(<a_b,c_j>/tau) ) with a = normalize(t1[idx]), p = normalize(t2[idx]),
    c = normalize(t2): one half of SGL.calc_ssl_loss (sgl.py:191-208) — rbg_infonce_f32.  ``row_w`` [B] weights the batch rows of
    the sum, ``col_w`` [n] the candidates inside every denominator (rbg_infonce_masked_f32; constants: no gradient w.r.t. them):
    with the batch's gathered rows as both tables and a one-occurrence mask for both this is SimGCL's contrast over
    ``torch.unique`` of the batch (simgcl.py:38-57) without its data-dependent shape."""
    _check_dense(t1, "t1")
    _check_dense(t2, "t2")
    if t1.shape != t2.shape:
        raise ValueError(f"the two views differ in shape: {tuple(t1.shape)} vs {tuple(t2.shape)}")
    if row_w is not None and tuple(row_w.shape) != (idx.shape[0],):
        raise ValueError("row_w must have one weight per batch row")
    if col_w is not None and tuple(col_w.shape) != (t2.shape[0],):
        raise ValueError("col_w must have one weight per row of t2")
    return _InfoNCE.apply(t1, t2, idx, float(tau), row_w, col_w)


# ---------------------------------------------------------------------------------------------------------------------
# k-means on the device (NCL's prototypes, ncl.py:60-81: faiss.Kmeans(d, k, gpu=True).train(x) + index.search(x, 1))
# ---------------------------------------------------------------------------------------------------------------------
def nearest_centroid(x, centroids):
    """argmin_j ||x_i - c_j||^2 for every row of x, without the [n, k] distance matrix: it is argmax_j of
    <x_i, c_j> - ||c_j||^2 / 2, i.e. the fused scoring + top-1 kernel (rbg_full_sort_topk_f32, exact-fp32 MFMA) on operands
    widened by one column; slot 0 of the item side is that kernel's masked [PAD] item, so the centroids sit at 1..k."""
    _check_dense(x, "x")
    _check_dense(centroids, "centroids")
    n, d = x.shape
    k = centroids.shape[0]
    dp = (d + 4) // 4 * 4
    xa = torch.zeros((n, dp), dtype=torch.float32, device=x.device)
    xa[:, :d] = x
    xa[:, d] = 1.0
    ca = torch.zeros((k + 1, dp), dtype=torch.float32, device=x.device)
    ca[1:, :d] = centroids
    ca[1:, d] = -0.5 * (centroids * centroids).sum(dim=1)
    out = torch.empty(n, dtype=torch.int64, device=x.device)
    step = 1 << 20
    for s0 in range(0, n, step):
        users = torch.arange(s0, min(n, s0 + step), device=x.device)
        _, idx = full_sort_topk(None, xa, ca, users, 1)
        out[s0:s0 + users.shape[0]] = idx[:, 0] - 1
    return out


def kmeans(x, k, niter=25, seed=1234, max_points_per_centroid=256, init=None):
    """Lloyd's k-means as faiss.Kmeans runs it [faiss: third-party, un-pinned, not under the reference tree]: centroids
    start as k distinct random training points (seed 1234), ``niter`` = 25 rounds of {assign every point to its nearest
    centroid (L2); centroid = mean of its points}, an empty cluster is re-seeded by splitting a populated one (its
    centroid copied with a +-1/1024 relative perturbation, faiss ``split_clusters``), and at most
    ``max_points_per_centroid * k`` randomly chosen points train.  Returns (centroids [k, d], assignment [n] of ALL points).
    The assignment step runs on the fused MFMA scoring + top-1 kernel; sums / counts are index_add_ / bincount."""
    _check_dense(x, "x")
    n, d = x.shape
    if n < k:
        raise ValueError(f"k-means needs at least as many points ({n}) as clusters ({k})")
    gen = torch.Generator(device=x.device).manual_seed(int(seed))
    train = x
    if n > max_points_per_centroid * k:
        train = x.index_select(0, torch.randperm(n, generator=gen, device=x.device)[: max_points_per_centroid * k])
    m = train.shape[0]
    if init is not None:  # caller-supplied starting centroids (tests: the same start as the CPU restatement)
        cent = init.to(device=x.device, dtype=torch.float32).clone()
    else:
        cent = train.index_select(0, torch.randperm(m, generator=gen, device=x.device)[:k]).clone()
    gen_cpu = torch.Generator().manual_seed(int(seed))
    eps = 1.0 / 1024.0
    sign = torch.where(torch.arange(d, device=x.device) % 2 == 0, 1.0 + eps, 1.0 - eps)
    for _ in range(niter):
        assign = nearest_centroid(train, cent)
        cnt = torch.bincount(assign, minlength=k).to(torch.float32)
        sums = torch.zeros((k, d), dtype=torch.float32, device=x.device).index_add_(0, assign, train)
        cent = torch.where(cnt[:, None] > 0, sums / cnt.clamp(min=1.0)[:, None], cent)
        empty = torch.nonzero(cnt == 0).flatten()
        if empty.numel():  # rare: handled on the host like faiss does (data-dependent control flow)
            cnt_h = cnt.cpu().clone()
            for ci in empty.tolist():
                p = (cnt_h - 1.0).clamp(min=0.0)
                cj = int(torch.multinomial(p / p.sum(), 1, generator=gen_cpu)) if float(p.sum()) > 0 else int(cnt_h.argmax())
                cent[ci] = cent[cj] * sign
                cent[cj] = cent[cj] * (2.0 - sign)
                cnt_h[ci] = cnt_h[cj] / 2
                cnt_h[cj] -= cnt_h[ci]
    return cent, nearest_centroid(x, cent)
