"""Graph handles: the normalized user-item adjacency resident in HBM.

Host-side mirror of ``GeneralGraphDataset.get_norm_adj_mat`` / ``edge_index_to_adj_t``
(reference ``recbole_gnn/data/dataset.py:41-79``) above the C ABI of ``include/rbgnn.h``.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import c_i64, c_int, c_vp, check, lib


def _np_i64(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr(a):
    return None if a is None else a.ctypes.data


def _device_index(device):
    """None / 'cpu' -> -1 (host graph); 'cuda' / 'cuda:i' / int / torch.device -> GPU index."""
    if device is None:
        return -1
    if isinstance(device, int):
        return device
    dev = torch.device(device)
    if dev.type == "cpu":
        return -1
    if dev.type != "cuda":
        raise ValueError(f"unsupported device {device!r}")
    return torch.cuda.current_device() if dev.index is None else dev.index


def find_communities(uid, iid, n_users, n_items, n_parts=8, iters=24, seed=0):
    """Balanced label propagation on the bipartite interaction graph (host, numpy): every node repeatedly adopts the
    label most common among its neighbours, votes scaled by (target load / label load)^2, half of the nodes moving per
    round (a bipartite graph otherwise oscillates).  Returns (labels int32 [n_users + n_items], cut, imbalance) where
    ``cut`` is the fraction of interactions whose two ends carry different labels and ``imbalance`` is the largest
    label's share of the nnz over the mean.  Feeds ``rbg_graph_create_partitioned`` when the caller has no partition."""
    uid, iid = _np_i64(uid), _np_i64(iid)
    n = int(n_users) + int(n_items)
    rng = np.random.default_rng(seed)
    rows = np.concatenate([uid, iid + n_users])
    cols = np.concatenate([iid + n_users, uid])
    deg = np.bincount(rows, minlength=n).astype(np.float64)
    label = rng.integers(0, n_parts, n)
    target = max(deg.sum() / n_parts, 1.0)
    for _ in range(iters):
        votes = np.bincount(rows * n_parts + label[cols], minlength=n * n_parts).reshape(n, n_parts).astype(np.float64)
        load = np.bincount(label, weights=deg, minlength=n_parts)
        score = votes * (target / np.maximum(load, 1.0)) ** 2 + rng.random(votes.shape) * 1e-3
        new = score.argmax(1)
        new[deg == 0] = label[deg == 0]
        label = np.where(rng.random(n) < 0.5, new, label)
    load = np.bincount(label, weights=deg, minlength=n_parts)
    cut = float(np.mean(label[uid] != label[n_users + iid])) if len(uid) else 0.0
    return label.astype(np.int32), cut, float(load.max() / max(load.mean(), 1.0))


_PARKED = []  # native handles whose Python owner died during a stream capture (GraphHandle.destroy)


def flush_parked():
    """Destroy the handles parked during a capture; a no-op inside one."""
    if _PARKED and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        while _PARKED:
            lib.rbg_graph_destroy(_PARKED.pop())


class GraphHandle:
    """Owns one ``rbg_graph*``.  Opaque to models, exactly like the reference's ``self.edge_index``
    (a torch_sparse SparseTensor when ``enable_sparse``; abstract_recommender.py:15-18)."""

    def __init__(self, ptr, symmetric, n_users=None):
        flush_parked()
        self._ptr = c_vp(ptr)
        self.symmetric = bool(symmetric)
        n_rows, n_cols, nnz, dev = c_i64(), c_i64(), c_i64(), c_int()
        check(lib.rbg_graph_info(self._ptr, ctypes.byref(n_rows), ctypes.byref(n_cols), ctypes.byref(nnz),
                                 ctypes.byref(dev)))
        self.n_rows, self.n_cols, self.nnz, self.device_index = n_rows.value, n_cols.value, nnz.value, dev.value
        self.n_users = n_users
        self._transpose = None

    # ---- constructors -----------------------------------------------------------------------
    @classmethod
    def from_interactions(cls, uid, iid, n_users, n_items, device=None, keep=None, flags=0, xcd_part=None):
        """dataset.py:60-75 (keep=None) or one SGL view (sgl.py:107-126) when ``keep`` is a mask.
        ``xcd_part``: optional int array [n_users + n_items] with a community id per node (1, 2, 4 or 8 communities):
        each community's rows are pinned to its own XCD(s) for L2 locality; results are unchanged."""
        if isinstance(uid, torch.Tensor) and uid.is_cuda:
            return cls._from_device_interactions(uid, iid, n_users, n_items, device, keep, flags, xcd_part)
        uid, iid = _np_i64(uid), _np_i64(iid)
        if uid.shape != iid.shape or uid.ndim != 1:
            raise ValueError("uid and iid must be 1-D arrays of equal length")
        out = c_vp()
        if isinstance(xcd_part, str):
            if xcd_part != "auto":
                raise ValueError("xcd_part must be an array or 'auto'")
            # use discovered communities only if they are real (most interactions stay inside) and balanced;
            # an unstructured graph keeps the default user-row / item-row XCD split
            kept = np.flatnonzero(keep) if keep is not None else slice(None)
            part, cut, imbalance = find_communities(uid[kept], iid[kept], n_users, n_items)
            xcd_part = part if (cut < 0.3 and imbalance < 1.1) else None  # measured to pay up to a ~20 % cut, DESIGN.md §6.4
        if xcd_part is not None:
            part = np.ascontiguousarray(xcd_part, dtype=np.int32)
            if part.shape != (n_users + n_items,):
                raise ValueError("xcd_part must have one entry per node")
            n_parts = int(part.max()) + 1 if part.size else 1
            if keep is not None:
                keep = np.ascontiguousarray(keep.detach().cpu().numpy() if isinstance(keep, torch.Tensor) else keep, dtype=np.uint8)
                if keep.shape != uid.shape:  # the C side reads keep[e] for every interaction
                    raise ValueError("keep mask must have one entry per interaction")
            check(lib.rbg_graph_create_partitioned(ctypes.byref(out), n_users, n_items, uid.shape[0], _ptr(uid), _ptr(iid),
                                                   _ptr(keep), _ptr(part), n_parts, _device_index(device), flags))
            return cls(out.value, symmetric=True, n_users=int(n_users))
        if keep is None:
            check(lib.rbg_graph_create(ctypes.byref(out), n_users, n_items, uid.shape[0], _ptr(uid), _ptr(iid),
                                       _device_index(device), flags))
        else:
            if isinstance(keep, torch.Tensor):
                keep = keep.detach().cpu().numpy()
            keep = np.ascontiguousarray(keep, dtype=np.uint8)
            if keep.shape != uid.shape:
                raise ValueError("keep mask must have one entry per interaction")
            check(lib.rbg_graph_create_masked(ctypes.byref(out), n_users, n_items, uid.shape[0], _ptr(uid), _ptr(iid),
                                              _ptr(keep), _device_index(device), flags))
        return cls(out.value, symmetric=True, n_users=int(n_users))

    @classmethod
    def _from_device_interactions(cls, uid, iid, n_users, n_items, device, keep, flags, xcd_part):
        """uid / iid (int64) and keep (uint8 / bool) already live on the graph's GPU: the device builder reads them in
        place (RBG_GRAPH_INPUTS_ON_DEVICE) — the path of SGL views sampled on the device."""
        if xcd_part is not None:
            raise ValueError("xcd_part needs host interactions")
        idx = _device_index(device if device is not None else uid.device)
        if not (iid.is_cuda and uid.device == iid.device and uid.device.index == idx):
            raise ValueError("uid, iid and the graph must be on the same GPU")
        uid, iid = uid.to(torch.int64).contiguous(), iid.to(torch.int64).contiguous()
        if uid.shape != iid.shape or uid.dim() != 1:
            raise ValueError("uid and iid must be 1-D tensors of equal length")
        kp = None
        if keep is not None:
            keep = torch.as_tensor(keep, device=uid.device).to(torch.uint8).contiguous()
            if keep.shape != uid.shape:
                raise ValueError("keep mask must have one entry per interaction")
            kp = c_vp(keep.data_ptr())
        torch.cuda.current_stream(uid.device).synchronize()  # the builder runs on the null stream
        out = c_vp()
        with torch.cuda.device(uid.device):
            check(lib.rbg_graph_create_masked(ctypes.byref(out), n_users, n_items, uid.shape[0], c_vp(uid.data_ptr()),
                                              c_vp(iid.data_ptr()), kp, idx, flags | _lib.GRAPH_INPUTS_ON_DEVICE))
        return cls(out.value, symmetric=True, n_users=int(n_users))

    @classmethod
    def from_csr(cls, rowptr, col, val, n_cols, device=None, symmetric=False, flags=0, n_class0_rows=None):
        """``n_class0_rows``: rows [0, n_class0_rows) and the rest reference disjoint column sets (a shard's user / item
        rows): the two classes are pinned to different XCDs like a graph built from interactions."""
        rowptr = _np_i64(rowptr)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        out = c_vp()
        check(lib.rbg_graph_create_csr_classes(ctypes.byref(out), rowptr.shape[0] - 1, n_cols, _ptr(rowptr), _ptr(col),
                                               _ptr(val), -1 if n_class0_rows is None else int(n_class0_rows),
                                               _device_index(device), flags))
        return cls(out.value, symmetric=symmetric)

    @classmethod
    def from_edge_index(cls, edge_index, edge_weight, num_nodes, device=None, symmetric=True, flags=0):
        """The reference's dense-branch pair (dataset.py:77-79): target rows gather from sources."""
        ei = _np_i64(edge_index)
        if ei.ndim != 2 or ei.shape[0] != 2:
            raise ValueError("edge_index must be [2, nnz]")
        w = edge_weight.detach().cpu().numpy() if isinstance(edge_weight, torch.Tensor) else edge_weight
        w = np.ascontiguousarray(w, dtype=np.float32)
        if w.shape != (ei.shape[1],):
            raise ValueError("edge_weight must be [nnz]")
        out = c_vp()
        check(lib.rbg_graph_create_coo(ctypes.byref(out), num_nodes, ei.shape[1], _ptr(ei), _ptr(w),
                                       _device_index(device), flags))
        return cls(out.value, symmetric=symmetric)

    # ---- queries ----------------------------------------------------------------------------
    @property
    def ptr(self):
        if not self._ptr:
            raise RuntimeError("graph handle was destroyed")
        return self._ptr

    @property
    def is_device(self):
        return self.device_index >= 0

    @property
    def device(self):
        return torch.device("cuda", self.device_index) if self.is_device else torch.device("cpu")

    def export_csr(self):
        rowptr = np.empty(self.n_rows + 1, dtype=np.int64)
        col = np.empty(self.nnz, dtype=np.int32)
        val = np.empty(self.nnz, dtype=np.float32)
        check(lib.rbg_graph_export_csr(self.ptr, _ptr(rowptr), _ptr(col), _ptr(val)))
        return rowptr, col, val

    def bins(self, d):
        vals = [c_i64() for _ in range(5)]
        check(lib.rbg_graph_bins(self.ptr, d, *[ctypes.byref(v) for v in vals]))
        keys = ("n_short", "n_wave", "n_block_tasks", "n_split_rows", "grid_blocks")
        return dict(zip(keys, (v.value for v in vals)))

    def spmm_kernel_name(self, d):
        """The kernel an SpMM of width d launches on this handle under the current options (as rocprofv3 prints it)."""
        buf = ctypes.create_string_buffer(128)
        check(lib.rbg_spmm_kernel_name(self.ptr, d, buf, 128))
        return buf.value.decode()

    def propagation_kernel_name(self, d, scratch_layers=True):
        """The kernel ``rbg_lightgcn_forward_f32`` launches per layer on this handle (the slab kernel when a SELL plan is
        attached and the caller does not read the layers, else the SpMM kernel)."""
        buf = ctypes.create_string_buffer(128)
        flags = 2 if scratch_layers else 0
        check(lib.rbg_lightgcn_forward_kernel_name(self.ptr, d, flags, buf, 128))
        return buf.value.decode()

    # ---- column-slab propagation (csrc/sell_plan.hip + csrc/sell.hip) -----------------------------------------------------
    def sell_eligible(self, d):
        """A SELL plan can serve width d on this handle: built from interactions (a user / item boundary), on a device."""
        return (self.is_device and self.n_users is not None and 0 < self.n_users < self.n_rows and self.n_rows == self.n_cols
                and d in (32, 64, 128) and not getattr(self, "_is_view", False))

    def has_sell(self, d):
        return bool(lib.rbg_graph_has_sell(self.ptr, d))

    def sell_status(self):
        """"planned" / "attached" / "view of a planned graph", or the reason this handle runs the binned kernel."""
        buf = ctypes.create_string_buffer(256)
        check(lib.rbg_graph_sell_status(self.ptr, buf, 256))
        return buf.value.decode()

    def sell_info(self):
        """Shape of the installed plan (raises without one)."""
        W, chunk, n_ent, fac, rm = c_int(), c_int(), c_i64(), c_int(), c_int()
        nu = (ctypes.c_int32 * 2)()
        check(lib.rbg_graph_sell_info(self.ptr, ctypes.byref(W), ctypes.byref(chunk), ctypes.byref(n_ent), nu, ctypes.byref(fac),
                                      ctypes.byref(rm)))
        return {"W": W.value, "chunk": chunk.value, "n_ent": n_ent.value, "n_units": [nu[0], nu[1]], "factored": bool(fac.value),
                "rowmajor": bool(rm.value), "padding": n_ent.value / max(self.nnz, 1)}

    def sell_arrays(self):
        """The installed plan's device arrays as torch tensors that alias them (read-only by contract; they keep this handle
        alive): ent [n_ent, 2], head [n_units, 4], orig [N] (int32), factors [N] float32 or None, src [n_ent] int32 or None."""
        info = self.sell_info()
        ptrs = [c_vp() for _ in range(5)]
        check(lib.rbg_graph_sell_arrays(self.ptr, *[ctypes.byref(q) for q in ptrs]))
        n_units = sum(info["n_units"])
        shapes = [((info["n_ent"], 2), "<i4"), ((n_units, 4), "<i4"), ((self.n_rows,), "<i4"), ((self.n_rows,), "<f4"),
                  ((info["n_ent"],), "<i4")]
        out = []
        for q, (shape, ts) in zip(ptrs, shapes):
            n = int(np.prod(shape))
            if not q.value:
                out.append(None)
            elif n == 0:
                out.append(torch.empty(shape, dtype=torch.int32 if ts == "<i4" else torch.float32, device=self.device))
            else:
                out.append(torch.as_tensor(self._view(q.value, n, ts), device=self.device).view(*shape))
        return dict(zip(("ent", "head", "orig", "factors", "src"), out))

    def plan_sell(self, W=32, chunk=0):
        """Plan the column-slab propagation inside the library (``rbg_graph_plan_sell``: rocPRIM sorts and one-pass kernels on
        this handle's device CSR) and install it.  ``rbg_graph_create*`` already did this for every eligible handle (option
        "sell_auto"); call it to re-plan with another slab width / chunk.  A plan of slab width 32 serves d = 32, 64 and 128;
        W = 64 serves d = 128 with two slabs (slower).  Raises ``RbgError`` (code RBG_EUNSUPPORTED) when the graph is outside
        what the plan serves; ``sell_status()`` then says why."""
        with torch.cuda.device(self.device):
            check(lib.rbg_graph_plan_sell(self.ptr, int(W), int(chunk)))
        return self.sell_info()

    def attach_sell(self, d=64, chunk=None, W=32):
        """Kept name of r03 for ``plan_sell`` with a width check.  (The executable specification of the plan's layout — torch ops,
        attached through ``rbg_graph_attach_sell`` — lives with the tests: ``tests/sell_spec.py``; the GPU tests compare the
        library's planner with it bit for bit.)"""
        if not self.sell_eligible(d):
            raise ValueError("a SELL plan needs a device graph built from interactions and d in (32, 64, 128)")
        if W not in (32, 64) or (W == 64 and d != 128):
            raise ValueError(f"slab width {W} does not serve d = {d}")
        return self.plan_sell(W, 0 if chunk is None else chunk)

    def detach_sell(self):
        check(lib.rbg_graph_detach_sell(self.ptr))

    def refresh_values(self):
        """After rewriting the ``vals`` tensor of a re-weighted view in place: refresh the copy of the values the view's
        column-slab plan holds (``rbg_graph_refresh_values``, on the current stream).  A no-op on handles without a plan."""
        if getattr(self, "_is_view", False):
            with torch.cuda.device(self.device):
                check(lib.rbg_graph_refresh_values(self.ptr, c_vp(torch.cuda.current_stream(self.device).cuda_stream)))

    def _view(self, ptr, n, typestr):
        handle = self

        class _View:  # the CUDA array interface: torch wraps the memory without copying it
            def __init__(self):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr or 0, False), "version": 2}
                self._owner = handle

        return _View()

    def transpose(self):
        """Handle of Â^T (needed for the backward of a non-symmetric graph)."""
        if self.symmetric:
            return self
        if self._transpose is None:
            rowptr, col, val = self.export_csr()
            rows = np.repeat(np.arange(self.n_rows, dtype=np.int64), np.diff(rowptr))
            order = np.lexsort((rows, col))
            t_rowptr = np.zeros(self.n_cols + 1, dtype=np.int64)
            np.add.at(t_rowptr, col.astype(np.int64) + 1, 1)
            t_rowptr = np.cumsum(t_rowptr)
            self._transpose = GraphHandle.from_csr(t_rowptr, rows[order].astype(np.int32), val[order], self.n_rows,
                                                   device=self.device if self.is_device else None)
            self._transpose._transpose = self
        return self._transpose

    def device_csr(self):
        """(rowptr int32 [n_rows+1], col int32 [nnz], val fp32 [nnz]) as torch tensors that ALIAS the handle's HBM arrays
        (no copy; read-only BY CONTRACT — torch has no read-only tensors, so nothing stops a caller from breaking the graph by
        writing to them; they keep this handle alive)."""
        ptrs = [c_vp(), c_vp(), c_vp()]
        check(lib.rbg_graph_device_arrays(self.ptr, *[ctypes.byref(q) for q in ptrs]))
        handle = self

        class _View:  # the CUDA array interface: torch wraps the memory without copying it
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr or 0, False), "version": 2}
                self._owner = handle

        out = []
        for q, n, ts in zip(ptrs, (self.n_rows + 1, self.nnz, self.nnz), ("<i4", "<i4", "<f4")):
            out.append(torch.as_tensor(_View(q.value, n, ts), device=self.device) if n and q.value else
                       torch.empty(0, dtype=torch.int32 if ts == "<i4" else torch.float32, device=self.device))
        return tuple(out)

    def values(self):
        """The fp32 edge weights in CSR entry order as a tensor on the graph's device (a copy)."""
        _, _, val = self.export_csr()
        return torch.from_numpy(val).to(self.device)

    def transpose_map(self):
        """int32 device tensor [nnz]: position of every entry's transposed partner (structure must be symmetric)."""
        m = torch.empty(max(self.nnz, 1), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.rbg_graph_transpose_map(self.ptr, c_vp(m.data_ptr()), c_vp(torch.cuda.current_stream(self.device).cuda_stream)))
        return m[: self.nnz]

    def reweighted(self, vals, symmetric=False):
        """A view with this graph's structure and launch plan whose edge weights are ``vals`` (fp32 device tensor [nnz], CSR
        entry order): NGCF edge dropout (ngcf.py:74-90) without rebuilding a sparse matrix.  RULE: after EVERY in-place rewrite
        of ``vals`` call ``view.refresh_values()`` (or use ``view.update_values(new)``, which does both).  On a planned base
        graph the view runs the column-slab kernel on a COPY of the values taken by the last refresh; only a view that was never
        refreshed (or whose base has no plan) reads ``vals`` at launch time — a rewrite without a refresh trains on stale
        weights with no error.  While views exist the base handle's plan can be neither detached nor re-planned (RbgError,
        RBG_EUNSUPPORTED); destroy the views first.  The view keeps this handle and ``vals`` alive."""
        if not (vals.is_cuda and vals.dtype == torch.float32 and vals.is_contiguous() and vals.numel() == self.nnz):
            raise ValueError("vals must be a contiguous fp32 device tensor with one entry per edge")
        out = c_vp()
        check(lib.rbg_graph_create_reweighted(ctypes.byref(out), self.ptr, c_vp(vals.data_ptr())))
        view = GraphHandle(out.value, symmetric=symmetric, n_users=self.n_users)
        view._keep_alive = (self, vals)
        view._is_view = True  # (the values are the caller's: the view borrows the base's plan and owns a refreshed copy of them)
        return view

    def update_values(self, new_vals):
        """Re-weighted views: copy ``new_vals`` into the view's ``vals`` tensor and refresh the plan's copy, in one call."""
        if not getattr(self, "_is_view", False):
            raise ValueError("update_values: not a re-weighted view")
        self._keep_alive[1].copy_(new_vals)
        self.refresh_values()

    def to(self, device):
        """Mirror of ``SparseTensor.to(device)`` (abstract_recommender.py:18)."""
        idx = _device_index(device)
        if idx == self.device_index:
            return self
        rowptr, col, val = self.export_csr()
        out = c_vp()
        check(lib.rbg_graph_create_csr(ctypes.byref(out), self.n_rows, self.n_cols, _ptr(rowptr), _ptr(col), _ptr(val),
                                       idx, 0))
        return GraphHandle(out.value, symmetric=self.symmetric, n_users=self.n_users)

    def destroy(self):
        if getattr(self, "_ptr", None):
            ptr, self._ptr = self._ptr, c_vp()
            # rbg_graph_destroy frees HBM (hipFree): inside a stream capture that invalidates the capture — and a handle can die
            # there through no fault of the capturing code: Python's cyclic collector runs whenever it likes and handles do sit in
            # cycles (NGCF's two edge-dropout views point at each other).  Such a handle is parked and destroyed by the next
            # destroy / create outside a capture (seen as a 1-in-6 hipErrorStreamCaptureInvalidated in tests/test_driver.py).
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                _PARKED.append(ptr)
                return
            lib.rbg_graph_destroy(ptr)
            flush_parked()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def __repr__(self):
        return (f"GraphHandle({self.n_rows}x{self.n_cols}, nnz={self.nnz}, "
                f"device={'cuda:%d' % self.device_index if self.is_device else 'host'})")


def norm_edges(uid, iid, n_users, n_items):
    """``get_norm_adj_mat(enable_sparse=False)`` (dataset.py:60-66,77-79) -> CPU tensors
    (edge_index int64 [2, 2E], edge_weight fp32 [2E])."""
    uid, iid = _np_i64(uid), _np_i64(iid)
    e = uid.shape[0]
    edge_index = np.empty((2, 2 * e), dtype=np.int64)
    edge_weight = np.empty(2 * e, dtype=np.float32)
    check(lib.rbg_norm_edges(n_users, n_items, e, _ptr(uid), _ptr(iid), _ptr(edge_index), _ptr(edge_weight)))
    return torch.from_numpy(edge_index), torch.from_numpy(edge_weight)


def set_tuning(short_max=-1, wave_max=-1, seg_len=-1):
    check(lib.rbg_set_tuning(short_max, wave_max, seg_len))


def get_tuning():
    a, b, c = c_int(), c_int(), c_int()
    check(lib.rbg_get_tuning(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return {"short_max": a.value, "wave_max": b.value, "seg_len": c.value}


def set_option(key, value):
    check(lib.rbg_set_option(key.encode(), int(value)))


def get_option(key):
    v = c_i64()
    check(lib.rbg_get_option(key.encode(), ctypes.byref(v)))
    return v.value


def device_count():
    n = c_int()
    check(lib.rbg_device_count(ctypes.byref(n)))
    return n.value


class InteractionDataset:
    """The slice of ``GeneralGraphDataset`` the path touches (dataset.py:24-79): the training
    interactions plus ``user_num`` / ``item_num`` (both include the PAD id 0), ``is_sparse`` and
    ``get_norm_adj_mat``."""

    def __init__(self, uid, iid, n_users, n_items):
        self.uid = torch.as_tensor(_np_i64(uid))
        self.iid = torch.as_tensor(_np_i64(iid))
        self.user_num = int(n_users)
        self.item_num = int(n_items)
        self.is_sparse = True  # the HIP engine is the sparse backend; it is always available

    def num(self, field):
        return self.user_num if field == "user_id" else self.item_num

    def get_norm_adj_mat(self, enable_sparse=False, device=None, xcd_part=None):
        """dataset.py:49-79.  enable_sparse truthy -> (GraphHandle, None) [the SparseTensor branch];
        falsy (None / False, the reference default) -> (edge_index, edge_weight) CPU tensors.
        ``xcd_part`` (engine extension): community array or "auto", see GraphHandle.from_interactions."""
        if enable_sparse:
            return GraphHandle.from_interactions(self.uid, self.iid, self.user_num, self.item_num, device=device,
                                                 xcd_part=xcd_part), None
        return norm_edges(self.uid, self.iid, self.user_num, self.item_num)
