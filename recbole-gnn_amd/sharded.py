"""Node-range sharding of the propagation across the GPUs of one node.

The reference is single-device (no distributed code at all), so there is no reference interface to
mirror here; this module extends ``LightGCN.forward`` (lightgcn.py:70-81) to P ranks:

* rank p owns a set of nodes (users AND items, so loads balance by nnz) — rows of Â and of every
  layer's embedding matrix;
* Y[owned] = Â[owned, :]·X needs X for the columns its rows reference: local ones plus a HALO.
  The halo is trimmed: per peer q only the rows p actually references (``send lists``), packed with
  ``rbg_gather_rows_f32`` and exchanged with one all-to-all-v per layer (RCCL over xGMI);
* the local CSR is split into an interior part (columns owned by p) and a halo part (columns in the
  receive buffer), so the interior SpMM runs on the compute stream WHILE the exchange runs on a
  second stream; then Y += Â_halo·X_halo.

Host logic (partition plan, send lists) is numpy; the weights are computed from GLOBAL degrees with
the same fp32 operations as the single-GPU builder, so a sharded run reproduces the single-GPU
matrix bit for bit.  The compute backend is injectable: the default is the HIP engine (needs a GPU);
tests inject a CPU backend to check plan + exchange under gloo.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


# ---- partition plan (pure numpy) ---------------------------------------------------------------

def balanced_ranges(deg, world):
    """Contiguous ranges with (almost) equal total degree.  Returns owner[len(deg)]."""
    n = len(deg)
    owner = np.zeros(n, dtype=np.int32)
    if n == 0 or world == 1:
        return owner
    c = np.cumsum(deg.astype(np.float64) + 1e-3)  # +eps keeps isolated nodes spread too
    bounds = np.searchsorted(c, c[-1] * np.arange(1, world) / world, side="left")
    owner[:] = np.searchsorted(bounds, np.arange(n), side="right")
    return owner


def default_partition(uid, iid, n_users, n_items, world):
    """Users and items are each cut into `world` contiguous nnz-balanced ranges."""
    du = np.bincount(uid, minlength=n_users)
    di = np.bincount(iid, minlength=n_items)
    return np.concatenate([balanced_ranges(du, world), balanced_ranges(di, world)])


def striped_partition(n_users, n_items, world):
    """Node r of each side -> rank (r-1) % world (PAD id 0 -> rank 0): the layout of the synthetic
    generator's community blocks (synth.powerlaw_bipartite(n_blocks=world))."""
    pu = (np.arange(n_users) - 1) % world
    pi = (np.arange(n_items) - 1) % world
    pu[0] = 0
    pi[0] = 0
    return np.concatenate([pu, pi]).astype(np.int32)


def degree_striped_partition(uid, iid, n_users, n_items, world):
    """Each side's nodes in degree order (descending, ties by id) dealt to the ranks in snake order (0..P-1, P-1..0, ...):
    every rank gets the same number of rows (+-1), the same share of the hubs and (almost) the same nnz.  Contiguous
    nnz-balanced ranges of popularity-sorted ids (``default_partition``) instead give rank 0 a few very heavy rows whose
    neighbourhood is the whole other side: at N = 2 on the Amazon-Book shape 14 822 owned rows against a 129 266-row halo."""
    out = []
    for ids, n in ((uid, n_users), (iid, n_items)):
        deg = np.bincount(ids, minlength=n)
        order = np.argsort(-deg, kind="stable")
        k = np.arange(n)
        rnd, pos = k // world, k % world
        owner = np.empty(n, dtype=np.int32)
        owner[order] = np.where(rnd % 2 == 0, pos, world - 1 - pos)
        out.append(owner)
    return np.concatenate(out)


def partition_stats(uid, iid, n_users, n_items, owner, world):
    """Per-rank load of a partition from the degrees alone: owned rows and nnz of the owned rows (= interior + halo
    entries); the halo ROW count needs the plan.  Returns {"rows": [...], "nnz": [...]}."""
    deg = np.concatenate([np.bincount(uid, minlength=n_users), np.bincount(iid, minlength=n_items)])
    owner = np.asarray(owner)
    return {"rows": np.bincount(owner, minlength=world).astype(np.int64).tolist(),
            "nnz": np.bincount(owner, weights=deg, minlength=world).astype(np.int64).tolist()}


def choose_partition(uid, iid, n_users, n_items, world, mode="auto"):
    """(owner, name, stats of both candidates).  ``auto`` keeps the candidate with the smaller max over ranks of
    (nnz of the owned rows + owned rows): the per-layer work of the slowest rank (its products plus its output rows)."""
    cands = {"ranges": default_partition(uid, iid, n_users, n_items, world),
             "striped": degree_striped_partition(uid, iid, n_users, n_items, world)}
    stats = {k: partition_stats(uid, iid, n_users, n_items, v, world) for k, v in cands.items()}
    if mode == "auto":
        cost = {k: max(a + b for a, b in zip(st["nnz"], st["rows"])) for k, st in stats.items()}
        mode = min(cost, key=cost.get)
    return cands[mode], mode, stats


class ShardPlan:
    """Everything rank `rank` needs: its rows, the two local CSR blocks, send / receive lists."""

    def __init__(self, rank, world, owned, n_users_owned, int_csr, halo_csr, halo_ids, recv_counts, send_idx,
                 send_counts):
        self.rank, self.world = rank, world
        self.owned = owned                    # global node ids, ascending (users first)
        self.n_users_owned = n_users_owned
        self.int_csr = int_csr                # (rowptr, col(local idx), val), n_cols = len(owned)
        self.halo_csr = halo_csr              # (rowptr, col(halo slot), val), n_cols = len(halo_ids)
        self.halo_ids = halo_ids              # global ids of the halo slots, grouped by owner rank
        self.recv_counts = recv_counts        # [world] rows received from each peer
        self.send_idx = send_idx              # local row indices to pack, grouped by destination rank
        self.send_counts = send_counts        # [world]

    @property
    def n_owned(self):
        return len(self.owned)

    @property
    def n_halo(self):
        return len(self.halo_ids)

    def cat_csr(self):
        """[A_interior | A_halo] as ONE rectangular CSR over the table [owned rows | halo rows] (r06, the fused layer): row r's
        interior entries, then its halo entries with the column moved past the owned rows.  (rowptr, col, val); n_cols =
        n_owned + n_halo."""
        ip, ic, iv = (np.asarray(a) for a in self.int_csr)
        hp, hc, hv = (np.asarray(a) for a in self.halo_csr)
        n = self.n_owned
        ip, hp = ip.astype(np.int64), hp.astype(np.int64)
        rp = ip + hp
        col = np.empty(int(rp[-1]), dtype=np.int32)
        val = np.empty(int(rp[-1]), dtype=np.float32)
        rows_i = np.repeat(np.arange(n), np.diff(ip))
        rows_h = np.repeat(np.arange(n), np.diff(hp))
        pos_i = rp[rows_i] + (np.arange(len(ic)) - ip[rows_i])
        pos_h = rp[rows_h] + (ip[rows_h + 1] - ip[rows_h]) + (np.arange(len(hc)) - hp[rows_h])
        col[pos_i], val[pos_i] = ic, iv
        col[pos_h], val[pos_h] = np.asarray(hc, dtype=np.int64) + n, hv
        return rp, col, val

    def cat_windows(self, window_rows):
        """cat_csr() cut into COLUMN windows of at most ``window_rows`` table rows: [(row_lo, row_hi, (rowptr, col - row_lo, val))].
        A rectangular plan addresses its table with 32-bit byte offsets (4 M rows at d = 128); config #5's 8-rank shards gather
        a 15 M-row table, so their layer is one launch per window, the later ones accumulating into Y (r06)."""
        rp, col, val = self.cat_csr()
        n_cols = self.n_owned + self.n_halo
        if n_cols <= window_rows:
            return [(0, n_cols, (rp, col, val))]
        rows = np.repeat(np.arange(self.n_owned), np.diff(rp))
        out = []
        for lo in range(0, n_cols, window_rows):
            hi = min(lo + window_rows, n_cols)
            m = (col >= lo) & (col < hi)
            ptr = np.zeros(self.n_owned + 1, dtype=np.int64)
            ptr[1:] = np.cumsum(np.bincount(rows[m], minlength=self.n_owned))
            out.append((lo, hi, (ptr, (col[m] - lo).astype(np.int32), val[m])))
        return out


def _csr_from_sorted(rows_local, cols, vals, n_rows):
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, rows_local + 1, 1)
    return np.cumsum(rowptr), cols.astype(np.int32), vals.astype(np.float32)


def build_plans(uid, iid, n_users, n_items, world, owner=None, ranks=None, keep=None):
    """Plans for `ranks` (default: all).  Deterministic and communication-free: every rank holds the
    interaction list (as in the reference, where the dataset is replicated on the host).
    ``keep`` (one flag per interaction): an SGL edge-drop view (sgl.py:107-126) — the kept sub-graph, normalized on its
    OWN global degrees (sgl.py:119-124); the partition is still the full graph's unless ``owner`` says otherwise.
    (numpy reference implementation; ``plan_from_csr`` cuts the same plan out of a built CSR with device-side sorts.)"""
    uid = np.ascontiguousarray(uid, dtype=np.int64)
    iid = np.ascontiguousarray(iid, dtype=np.int64)
    n = n_users + n_items
    if owner is None:
        owner = default_partition(uid, iid, n_users, n_items, world)
    if keep is not None:
        keep = np.asarray(keep).astype(bool)
        if keep.shape != uid.shape:
            raise ValueError("keep mask must have one entry per interaction")
        uid, iid = uid[keep], iid[keep]
    owner = np.asarray(owner, dtype=np.int32)
    assert owner.shape == (n,) and owner.min(initial=0) >= 0 and owner.max(initial=0) < world
    # directed edges of the symmetric graph: target row <- source col  (dataset.py:60-64)
    rows = np.concatenate([uid, iid + n_users])
    cols = np.concatenate([iid + n_users, uid])
    # gcn_norm on GLOBAL degrees, fp32 exactly as the single-GPU builder (graph_build.cpp)
    deg = np.bincount(rows, minlength=n).astype(np.float32)
    with np.errstate(divide="ignore"):
        dis = (np.float32(1.0) / np.sqrt(deg)).astype(np.float32)
    dis[np.isinf(dis)] = 0.0
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    vals = (dis[rows] * np.float32(1.0)) * dis[cols]
    row_owner, col_owner = owner[rows], owner[cols]
    # who needs what: unique (needing rank, remote column) pairs
    remote = row_owner != col_owner
    need_key = np.unique(row_owner[remote].astype(np.int64) * n + cols[remote])
    need_rank, need_col = need_key // n, need_key % n
    need_owner = owner[need_col]
    local_index = np.full(n, -1, dtype=np.int64)
    plans = {}
    for p in (range(world) if ranks is None else ranks):
        owned = np.flatnonzero(owner == p)
        local_index[:] = -1
        local_index[owned] = np.arange(len(owned))
        mine = row_owner == p
        r_l = local_index[rows[mine]]
        c, v, interior = cols[mine], vals[mine], col_owner[mine] == p
        int_csr = _csr_from_sorted(r_l[interior], local_index[c[interior]], v[interior], len(owned))
        # halo slots: remote columns this rank needs, ordered by (owner, global id)
        sel = need_rank == p
        h_cols, h_owner = need_col[sel], need_owner[sel]
        ho = np.lexsort((h_cols, h_owner))
        halo_ids = h_cols[ho]
        recv_counts = np.bincount(h_owner, minlength=world).astype(np.int64)
        slot = np.full(n, -1, dtype=np.int64)
        slot[halo_ids] = np.arange(len(halo_ids))
        halo_csr = _csr_from_sorted(r_l[~interior], slot[c[~interior]], v[~interior], len(owned))
        # send lists: what every q needs from p, in q's halo order (ascending global id per owner)
        sel = need_owner == p
        s_rank, s_col = need_rank[sel], need_col[sel]
        so = np.lexsort((s_col, s_rank))
        send_idx = local_index[s_col[so]]
        send_counts = np.bincount(s_rank, minlength=world).astype(np.int64)
        plans[p] = ShardPlan(p, world, owned, int(np.count_nonzero(owned < n_users)), int_csr, halo_csr, halo_ids,
                             recv_counts, send_idx, send_counts)
    return plans


def plan_from_csr(rowptr, col, val, n_users, owner, rank, world):
    """Rank `rank`'s ShardPlan cut out of the GLOBAL normalized CSR (torch tensors on any device — e.g.
    ``GraphHandle.device_csr()`` of a graph the device builder made, so that nothing of size nnz is sorted on the host:
    at BASELINE config #5 that is 4e8 directed edges per rank for the numpy planner).  The matrix must be structurally
    symmetric (it is: dataset.py:62-64): what a peer q needs from this rank — the rows of mine that q's rows reference — is
    then exactly the set of my rows with a neighbour owned by q, so a rank plans from its own rows alone.  Equal to
    ``build_plans(...)[rank]`` (tested)."""
    dev = col.device
    rowptr = rowptr.to(torch.int64)
    n = rowptr.numel() - 1
    owner_t = torch.as_tensor(np.asarray(owner), device=dev).to(torch.int64)
    owned = torch.nonzero(owner_t == rank).flatten()
    n_owned = int(owned.numel())
    beg = rowptr[owned]
    cnt = rowptr[owned + 1] - beg
    total = int(cnt.sum())
    r_local = torch.repeat_interleave(torch.arange(n_owned, device=dev), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    idx = torch.repeat_interleave(beg - first, cnt) + torch.arange(total, device=dev)
    c = col[idx].to(torch.int64)
    v = val[idx]
    co = owner_t[c]
    interior = co == rank
    local_index = torch.full((n,), -1, dtype=torch.int64, device=dev)
    local_index[owned] = torch.arange(n_owned, device=dev)

    def block(mask, cols):
        ptr = torch.zeros(n_owned + 1, dtype=torch.int64, device=dev)
        ptr[1:] = torch.cumsum(torch.bincount(r_local[mask], minlength=n_owned), 0)
        return ptr.cpu().numpy(), cols.to(torch.int32).cpu().numpy(), v[mask].cpu().numpy()

    int_csr = block(interior, local_index[c[interior]])
    remote = ~interior
    key = co[remote] * n + c[remote]                      # (owner, global id): the halo order
    ukey = torch.unique(key)
    halo_ids = (ukey % n).cpu().numpy()
    recv_counts = torch.bincount(ukey // n, minlength=world).cpu().numpy().astype(np.int64)
    halo_csr = block(remote, torch.searchsorted(ukey, key))
    skey = torch.unique(co[remote] * max(n_owned, 1) + r_local[remote])   # (destination, my row): what each peer needs of mine
    send_idx = (skey % max(n_owned, 1)).cpu().numpy()
    send_counts = torch.bincount(skey // max(n_owned, 1), minlength=world).cpu().numpy().astype(np.int64)
    owned_np = owned.cpu().numpy()
    return ShardPlan(rank, world, owned_np, int(np.count_nonzero(owned_np < n_users)), int_csr, halo_csr, halo_ids, recv_counts,
                     send_idx, send_counts)


def self_exchange_plan(plan, every=3):
    """A world-size-1 plan that exchanges every `every`-th row with ITSELF (test / probe device: one GPU then runs the
    pack, the collective, the interior and the halo product of the multi-GPU path).  Same result as the plain plan."""
    if plan.world != 1:
        raise ValueError("self_exchange_plan takes a world-size-1 plan")
    rp, col, val = (np.asarray(a) for a in plan.int_csr)
    n = plan.n_owned
    halo_nodes = np.arange(0, n, every)
    slot = -np.ones(n, dtype=np.int64)
    slot[halo_nodes] = np.arange(len(halo_nodes))
    rows = np.repeat(np.arange(n), np.diff(rp))
    is_halo = slot[col] >= 0

    def csr(mask, cols):
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(ptr, rows[mask] + 1, 1)
        return np.cumsum(ptr), cols.astype(np.int32), val[mask].astype(np.float32)

    return ShardPlan(0, 1, plan.owned, plan.n_users_owned, csr(~is_halo, col[~is_halo]), csr(is_halo, slot[col[is_halo]]),
                     halo_nodes, np.array([len(halo_nodes)]), halo_nodes.copy(), np.array([len(halo_nodes)]))


# ---- compute backends ----------------------------------------------------------------------------

class HipBackend:
    """The product backend: librbgnn.so kernels on this rank's GPU.  The hot calls go straight to the C ABI (the
    shapes were validated when the plan was built), so a layer costs a handful of ctypes calls on the host."""

    def __init__(self, device):
        from . import _lib, ops
        from .graph import GraphHandle
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipBackend needs a cuda device (no CPU path)")
        self._ops, self._GraphHandle, self._lib = ops, GraphHandle, _lib

    def make_graph(self, csr, n_cols, n_user_rows=None):
        """``n_user_rows``: the shard's leading user rows — they gather item rows only, the rest user rows only, so the
        launch plan pins the two classes to different XCDs (as for a single-GPU graph)."""
        return self._GraphHandle.from_csr(csr[0], csr[1], csr[2], n_cols, device=self.device, n_class0_rows=n_user_rows)

    def layer_ctx(self):
        """Context (two events) for layer_begin / layer_end; freed with the backend object."""
        import ctypes
        ctx = self._lib.c_vp()
        self._lib.check(self._lib.lib.rbg_shard_ctx_create(ctypes.byref(ctx), self.device.index or 0))
        self._ctxs = getattr(self, "_ctxs", []) + [ctx]
        return ctx

    def __del__(self):
        if torch is None or torch.cuda is None:  # (interpreter shutdown: the modules are already gone)
            return
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return  # (collected inside somebody's stream capture: leaked rather than invalidating it — see GraphHandle.destroy)
        for ctx in getattr(self, "_ctxs", []):
            try:
                self._lib.lib.rbg_shard_ctx_destroy(ctx)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def layer_begin(self, ctx, g_int, x, y, send_idx, n_send, send_buf, main_h, comm_h):
        """comm stream: wait for x, pack the send rows; main stream: y = A_int x  (one call, rbg_shard_layer_begin)."""
        lib, vp = self._lib.lib, self._lib.c_vp
        self._lib.check(lib.rbg_shard_layer_begin(ctx, g_int.ptr, vp(x.data_ptr()), vp(y.data_ptr()), vp(send_idx.data_ptr()),
                                                  n_send, vp(send_buf.data_ptr()), x.shape[1], vp(main_h), vp(comm_h)))

    def layer_end(self, ctx, g_halo, halo, y, main_h, comm_h):
        """main stream: wait for the comm stream, y += A_halo halo  (rbg_shard_layer_end)."""
        lib, vp = self._lib.lib, self._lib.c_vp
        self._lib.check(lib.rbg_shard_layer_end(ctx, g_halo.ptr if g_halo is not None else None, vp(halo.data_ptr()),
                                                vp(y.data_ptr()), y.shape[1], vp(main_h), vp(comm_h)))

    # `stream`: raw HIP stream handle (int) to launch on; None = torch's current stream.  The sharded propagation passes
    # handles it looked up once per call: torch.cuda.current_stream() costs ~5 us, and a layer would make five of them.
    def _stream(self, stream):
        return self._lib.c_vp(torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream)

    def spmm(self, graph, x, out, accumulate, stream=None):
        lib, vp = self._lib.lib, self._lib.c_vp
        self._lib.check(lib.rbg_spmm_f32(graph.ptr, vp(x.data_ptr()), vp(out.data_ptr()), x.shape[1], int(bool(accumulate)),
                                         self._stream(stream)))
        return out

    def mean(self, srcs, out, stream=None):
        """out = (srcs[0] + srcs[1] + ...) / len(srcs), one launch (rbg_mean_f32)."""
        lib, vp = self._lib.lib, self._lib.c_vp
        arr = (vp * len(srcs))(*[t.data_ptr() for t in srcs])
        self._lib.check(lib.rbg_mean_f32(arr, len(srcs), out.numel(), 1.0 / len(srcs), vp(out.data_ptr()), self._stream(stream)))
        return out

    def spmm_mean(self, graph, x, partial, srcs, out, stream=None):
        """out = (srcs[0] + ... + (partial + A x)) / (len(srcs) + 1): the last layer with the layer mean in its epilogue
        (rbg_spmm_mean_f32); partial may be None."""
        lib, vp = self._lib.lib, self._lib.c_vp
        arr = (vp * len(srcs))(*[t.data_ptr() for t in srcs])
        self._lib.check(lib.rbg_spmm_mean_f32(graph.ptr, vp(x.data_ptr()), vp(partial.data_ptr()) if partial is not None else None,
                                              arr, len(srcs), vp(out.data_ptr()), x.shape[1], self._stream(stream)))
        return out

    def bignn_dense(self, p, x, w1, b1, w2, b2, out, leaky_norm=True, slope=0.2, stream=None):
        """lin1(P + X) + lin2(P * X) [+ LeakyReLU + L2-normalize] on this rank's rows from its product P (rbg_bignn_dense_f32);
        x / out may be column slices of wider row-major buffers."""
        lib, vp = self._lib.lib, self._lib.c_vp
        n, d_in = x.shape
        with torch.cuda.device(self.device):  # the entry point takes no graph handle: it launches on the CURRENT device
            self._bignn_dense_call(lib, vp, p, x, w1, b1, w2, b2, out, n, d_in, leaky_norm, slope, stream)
        return out

    def _bignn_dense_call(self, lib, vp, p, x, w1, b1, w2, b2, out, n, d_in, leaky_norm, slope, stream):
        self._lib.check(lib.rbg_bignn_dense_f32(vp(p.data_ptr()), vp(x.data_ptr()), x.stride(0) if n > 1 else d_in, vp(w1.data_ptr()),
                                                vp(b1.data_ptr()), vp(w2.data_ptr()), vp(b2.data_ptr()), vp(out.data_ptr()),
                                                out.stride(0) if n > 1 else out.shape[1], n, d_in, out.shape[1],
                                                self._lib.BIGNN_LEAKY_NORM if leaky_norm else self._lib.BIGNN_CONV_ONLY, float(slope),
                                                self._stream(stream)))
        return out

    def gather_rows(self, src, idx, out=None, stream=None):
        if out is None:
            return self._ops.gather_rows(src, idx)
        lib, vp = self._lib.lib, self._lib.c_vp
        self._lib.check(lib.rbg_gather_rows_f32(vp(src.data_ptr()), src.shape[1], vp(idx.data_ptr()), vp(out.data_ptr()),
                                                idx.shape[0], src.shape[1], self._stream(stream)))
        return out


# ---- halo push without a collective (r06) ------------------------------------------------------------

class PushExchange:
    """Peer-to-peer halo PUSH (csrc/ipc.hip): this rank's layer tables [owned rows | halo rows] live in memory it exports
    (hipIpcGetMemHandle); every peer maps them and its pack kernel stores the rows this rank needs straight into the table's
    tail — no send buffer, no collective, no receive-side copy.  Ordering by flag words in the same exported block:

    * ``ready[k][p]``  (mine, written by sender p after its pushes into my table k): my stream waits for all of them before the
      layer launch;
    * ``free[k][q]``   (mine, written by receiver q after ITS launch consumed its table k): my stream waits for it before it
      pushes into q's table k again.

    Values are the exchange's sequence number (monotone: no reset).  Waits are bounded spins on the GPU (``timeout_ms``; a
    time-out sets an error word that ``check()`` raises on).  Set-up is collective over ``group`` (handles and geometry travel
    by ``all_gather_object``); a layer afterwards is launches on the rank's own stream only.  Exercised by two processes on ONE
    GPU (tests); between different GPUs the mapping goes over xGMI — never run in this project (no multi-GPU box)."""

    def __init__(self, plan, device, d, n_tables, group=None, timeout_ms=2000):
        import ctypes
        from . import _lib
        self._lib, self.plan, self.device, self.d, self.n_tables = _lib, plan, torch.device(device), int(d), int(n_tables)
        self.world, self.rank, self.timeout_ms = plan.world, plan.rank, int(timeout_ms)
        lib, vp = _lib.lib, _lib.c_vp
        dev_i = self.device.index or 0
        self.rows = plan.n_owned + plan.n_halo
        self.table_bytes = -(-self.rows * self.d * 4 // 256) * 256
        nf = self.n_tables * self.world
        self.flag_bytes = -(-(2 * nf * 8 + 64) // 4096) * 4096     # ready[k][p], free[k][q] (uint64), one error word
        self._base = vp()
        _lib.check(lib.rbg_ipc_alloc(ctypes.byref(self._base), self.flag_bytes + self.n_tables * self.table_bytes, dev_i))
        h = ctypes.create_string_buffer(64)
        _lib.check(lib.rbg_ipc_export(self._base, h))
        meta = {"handle": h.raw, "n_owned": int(plan.n_owned), "recv_counts": [int(c) for c in plan.recv_counts],
                "table_bytes": self.table_bytes, "flag_bytes": self.flag_bytes, "d": self.d, "n_tables": self.n_tables}
        metas = [None] * self.world
        dist.all_gather_object(metas, meta, group=group)
        for m in metas:
            if m["d"] != self.d or m["n_tables"] != self.n_tables:
                raise ValueError("PushExchange: the ranks disagree on d / n_tables")
        self._metas = metas
        self._peer = {}
        for q, m in enumerate(metas):
            if q == self.rank:
                continue
            ptr = vp()
            _lib.check(lib.rbg_ipc_open(m["handle"], ctypes.byref(ptr), dev_i))
            self._peer[q] = ptr.value
        # what I push to q: my rows send_idx[so_q : so_q + sc_q]; where they land in q's tail: after the rows of the ranks before me
        self._send = []
        so = 0
        for q in range(self.world):
            sc = int(plan.send_counts[q])
            if sc and q != self.rank:
                row_off = metas[q]["n_owned"] + sum(metas[q]["recv_counts"][:self.rank])
                self._send.append((q, so, sc, row_off))
            so += sc
        self._senders = [p for p in range(self.world) if p != self.rank and int(plan.recv_counts[p]) > 0]
        self.send_idx = torch.as_tensor(plan.send_idx, dtype=torch.int64, device=self.device)
        self._seq = 0
        self._last_use = [0] * self.n_tables
        self._tables = [self._view(self._base.value + self.flag_bytes + k * self.table_bytes, (self.rows, self.d)) for k in range(self.n_tables)]
        # the words of ranks that never signal me (no rows exchanged in that direction, and my own) are born satisfied, so ONE
        # wait launch covers a table's row of words
        flags = self._view(self._base.value, (2, self.n_tables, self.world), typestr="<i8")
        never_ready = [p for p in range(self.world) if p not in self._senders]
        never_free = [q for q in range(self.world) if q not in [t[0] for t in self._send]]
        if never_ready:
            flags[0][:, never_ready] = -1   # (= 2^64 - 1 as the unsigned word the kernels compare)
        if never_free:
            flags[1][:, never_free] = -1
        torch.cuda.synchronize(self.device)
        # every rank has mapped everyone (and initialised its words) before anyone pushes
        dist.barrier(group=group)

    def _view(self, ptr, shape, typestr="<f4"):
        owner = self

        class _V:
            def __init__(self):
                self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}
                self._owner = owner
        return torch.as_tensor(_V(), device=self.device)

    def table(self, k):
        return self._tables[k]

    def _flag(self, base, kind, k, who):
        """address of ready (kind 0) / free (kind 1) word [k][who] in the block at `base`"""
        return base + ((kind * self.n_tables + k) * self.world + who) * 8

    @property
    def _err_ptr(self):
        return self._base.value + 2 * self.n_tables * self.world * 8

    def exchange(self, k, stream):
        """Push the rows my peers need of table k's head into THEIR table k, then wait until mine has everyone's rows."""
        lib, vp, chk = self._lib.lib, self._lib.c_vp, self._lib.check
        self._seq += 1
        seq, st = self._seq, vp(stream)
        x = self._tables[k]
        # the receivers have consumed what I pushed into their table k last time
        if self._last_use[k] and self._send:
            chk(lib.rbg_ipc_wait(vp(self._flag(self._base.value, 1, k, 0)), self.world, self._last_use[k], self.timeout_ms, vp(self._err_ptr), st))
        for q, so, sc, row_off in self._send:
            dst = self._peer[q] + self._metas[q]["flag_bytes"] + k * self._metas[q]["table_bytes"] + row_off * self.d * 4
            chk(lib.rbg_gather_rows_f32(vp(x.data_ptr()), self.d, vp(self.send_idx.data_ptr() + so * 8), vp(dst), sc, self.d, st))
            chk(lib.rbg_ipc_signal(vp(self._flag(self._peer[q], 0, k, self.rank)), seq, st))
        if self._senders:
            chk(lib.rbg_ipc_wait(vp(self._flag(self._base.value, 0, k, 0)), self.world, seq, self.timeout_ms, vp(self._err_ptr), st))
        self._last_use[k] = seq

    def consumed(self, k, stream):
        """After the launch that read table k: its senders may overwrite the tail."""
        lib, vp, chk = self._lib.lib, self._lib.c_vp, self._lib.check
        for p in self._senders:
            chk(lib.rbg_ipc_signal(vp(self._flag(self._peer[p], 1, k, self.rank)), self._last_use[k], vp(stream)))

    def check(self):
        """Synchronise and raise if a wait timed out (a peer that never pushed)."""
        torch.cuda.synchronize(self.device)
        e = int(self._view(self._err_ptr, (1,), typestr="<i4")[0])
        if e:
            raise RuntimeError(f"PushExchange: rank {self.rank} timed out waiting for flag word {e - 1} ({self.timeout_ms} ms)")

    def close(self):
        if getattr(self, "_base", None) is not None and self._base.value:
            torch.cuda.synchronize(self.device)
            for ptr in self._peer.values():
                self._lib.lib.rbg_ipc_close(self._lib.c_vp(ptr))
            self._peer = {}
            self._tables = []
            self._lib.lib.rbg_ipc_free(self._base)
            self._base = None

    def __del__(self):
        try:
            if not torch.cuda.is_current_stream_capturing():  # (close() synchronises and frees: never inside a capture)
                self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


# ---- the sharded propagation ---------------------------------------------------------------------

class ShardedPropagation:
    """K-layer LightGCN propagation over a ShardPlan.  ``transport``: "nccl" (device buffers straight
    into RCCL all_to_all_single) or "staged" (buffers staged through the host and exchanged with
    point-to-point send/recv; works with gloo — used by tests and when several ranks must share one GPU).

    ``overlap`` (nccl transport): False = one stream per layer (pack, exchange, interior product, halo product), True =
    pack + collective on a second, high-priority stream beside the interior product (rbg_shard_layer_begin/end).
    Which is faster depends on the machine: on one GPU with a world-size-1 RCCL group (r01, devtools/nccl1_probe.py) the
    second stream cost ~35 us of cross-stream event latency per layer and hid nothing (there was no link time to hide:
    325 vs 259 us per propagation); with real peers the single-stream layer exposes the whole collective.  ``autotune``
    measures both on the actual group and keeps the faster (``bench.py --gpus N`` does).  Default (``overlap=None``): the single-stream
    layer — the overlapped form has never run between different GPUs (no multi-GPU box in five rounds: ADVICE r04), so it is
    an explicit choice (``overlap=True`` / ``set_overlap``) or autotune's, not a default.

    r06 — ``fused`` (default whenever ``overlap`` is off and the rank has a halo): the rank's block [A_interior | A_halo] is ONE
    rectangular handle over the table [owned rows | halo rows] (``ShardPlan.cat_csr``), planned for the column-slab kernel like any
    other handle (rectangular form, csrc/sell_plan.hip).  A layer = pack -> exchange into the table's tail -> ONE launch that
    writes the next layer's table head: no accumulate pass over Y, every entry on ``sell_spmm_kernel`` (r05: the halo block —
    (P - 1) / P of the entries on an unstructured graph — ran the binned kernel and re-read Y).  ``overlap=True`` keeps the
    two-handle form (interior product beside the exchange); its halo block is planned too (r06) — only widths other than
    32 / 64 / 128 stay on the binned kernel.
    ``halo_bytes_per_layer`` says what a rank receives per layer — on an unstructured power-law graph almost the whole table
    (Amazon-Book shape, P = 4: 106 056 of 108 183 foreign rows): see ``colsharded.py`` for the sharding that exchanges nothing."""

    def halo_bytes_per_layer(self, d):
        """bytes this rank RECEIVES per layer at width d (fp32 rows of its halo), and the share of the foreign rows that is"""
        n_total = sum(int(c) for c in self.plan.recv_counts)
        return {"recv_bytes": n_total * d * 4, "halo_rows": int(self.plan.n_halo), "owned_rows": int(self.plan.n_owned)}

    def __init__(self, plan, backend, group=None, transport="nccl", overlap=None, fused=None, push_tables=4, push_timeout_ms=2000,
                 cat_window_rows=None):
        # overlap = None (default): off — unmeasured between real peers; ``autotune`` measures both forms on the actual group
        if overlap is None:
            overlap = False
        self.plan, self.backend, self.group, self.transport = plan, backend, group, transport
        dev = getattr(backend, "device", torch.device("cpu"))
        self.device = dev
        self._g_int = self._g_halo = self._g_cat = None   # handles are built on first use: a fused rank never builds the pair
        self._want_fused = (not overlap) if fused is None else bool(fused)
        # transport "push" (r06): no collective — the peers' pack kernels store into this rank's exported layer tables
        # (PushExchange); the fused layer only, forward / spmm / backward (halo_of and the staged helpers are not served)
        self.push, self._push_tables, self._push_timeout_ms = None, int(push_tables), int(push_timeout_ms)
        if transport == "push" and (overlap or fused is False):
            raise ValueError('transport "push" is the fused single-stream layer')
        self.send_idx = torch.as_tensor(plan.send_idx, dtype=torch.int64, device=dev)
        self.comm_stream, self._comm_h, self._ctx = None, None, None
        self.overlap = False
        self.set_overlap(overlap)
        # a rectangular plan addresses its table with 32-bit byte offsets (4.19 M rows at d = 128).  A longer table — config #5's
        # 8-rank shards gather 12 M rows — CAN be cut into column windows, one planned handle and one launch each
        # (``cat_window_rows``), but measured at that scale the windows lose to the two-handle form (4.93 vs 4.24 ms per layer,
        # profiles/r06_config5_shard.json: every extra launch re-reads and re-writes the rank's 0.96 GB of Y and the gathers are
        # fabric-bound on either kernel), so by default such a rank keeps two handles: interior on the column-slab kernel, halo binned
        self._cat_window_rows = int(cat_window_rows) if cat_window_rows else None
        if self._want_fused and transport != "push" and self._cat_window_rows is None and (plan.n_owned + plan.n_halo) * 256 >= 0x7ffffff0:
            self._want_fused = False
        self._g_cats = None
        if self.fused:
            # blocks the planner does not serve (no two row classes, ...) would put EVERY entry on the binned kernel; the two-handle
            # form keeps at least the interior block on the column-slab kernel (the r05 state)
            gs = self.g_cats
            if transport != "push" and any(hasattr(g, "sell_status") and g.sell_status() != "planned" for _, _, g in gs):
                self._want_fused, self._g_cat, self._g_cats = False, None, None
            del gs
        if not self.fused:
            _ = self.g_int, self.g_halo
        self._n_send = len(plan.send_idx)
        self._buf_d = None  # per-width buffers, allocated on first use: halo, send, ping-pong outputs
        self._recv_splits = [int(c) for c in plan.recv_counts]
        self._send_splits = [int(c) for c in plan.send_counts]
        self.tuned = None  # filled by autotune(): {"single_stream_us", "overlap_us", "chosen"}

    @property
    def fused(self):
        """One handle, one launch per layer (no interior / halo split): whenever asked for, the rank has a halo and the
        overlapped two-stream form is off."""
        if self.transport == "push":  # (every rank runs the table form, also one without a halo of its own: its peers push and wait)
            return self.plan.world > 1 and self.plan.n_owned > 0
        return self._want_fused and self.plan.n_halo > 0 and self.plan.n_owned > 0 and not self.overlap

    def set_fused(self, fused):
        self._want_fused = bool(fused)

    @property
    def g_int(self):
        if self._g_int is None:
            self._g_int = self._make_graph(self.plan.int_csr, self.plan.n_owned)
        return self._g_int

    @property
    def g_halo(self):
        if self._g_halo is None and self.plan.n_halo:
            self._g_halo = self._make_graph(self.plan.halo_csr, max(self.plan.n_halo, 1))
        return self._g_halo

    @property
    def g_cats(self):
        """[(row_lo, row_hi, handle)]: the [interior | halo] block per column window of the layer table (one window unless the
        table is beyond the rectangular plan's 32-bit offsets)."""
        if self._g_cats is None:
            if self._g_cat is not None:  # (a handle handed over by a sibling propagation of the same plan)
                self._g_cats = [(0, self.plan.n_owned + self.plan.n_halo, self._g_cat)]
            else:
                self._g_cats = [(lo, hi, self._make_graph(csr, hi - lo))
                                for lo, hi, csr in self.plan.cat_windows(self._cat_window_rows or (self.plan.n_owned + self.plan.n_halo))]
                self._g_cat = self._g_cats[0][2]
        return self._g_cats

    @property
    def g_cat(self):
        return self.g_cats[0][2]

    def kernel_status(self):
        """What the handles of the active form run: {"form", "cat" | "interior" / "halo": plan status} (HIP backend)."""
        st = lambda g: g.sell_status() if (g is not None and hasattr(g, "sell_status")) else None  # noqa: E731
        if self.fused:
            gs = self.g_cats
            return {"form": "fused", "cat": st(gs[0][2])} if len(gs) == 1 else {"form": "fused", "cat": [st(g) for _, _, g in gs], "windows": len(gs)}
        return {"form": "two handles", "interior": st(self.g_int), "halo": st(self.g_halo)}

    def set_overlap(self, overlap):
        self.overlap = bool(overlap) and self.transport == "nccl" and self.device.type == "cuda"
        if self.overlap and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.device, priority=-1)
            self._comm_h = self.comm_stream.cuda_stream
            self._ctx = self.backend.layer_ctx() if hasattr(self.backend, "layer_ctx") else None

    def _make_graph(self, csr, n_cols):
        return self.backend.make_graph(csr, n_cols, n_user_rows=self.plan.n_users_owned)

    def _buffers(self, x):
        d = x.shape[1]
        if self._buf_d != d:
            plan = self.plan
            f = dict(dtype=x.dtype, device=x.device)
            self._halo = torch.empty((max(plan.n_halo, 1), d), **f)
            self._send = torch.empty((max(len(plan.send_idx), 1), d), **f)
            # layer tables [owned rows | halo rows]: a fused layer gathers one and writes the head of the next; the two-handle
            # form uses their heads as its ping-pong outputs
            if self.transport == "push" and plan.world > 1:   # (collective set-up: every rank reaches its first layer together)
                if self.push is not None:
                    self.push.close()
                self.push = PushExchange(plan, x.device, d, self._push_tables, group=self.group, timeout_ms=self._push_timeout_ms)
                self._cat = [self.push.table(k) for k in range(self._push_tables)]
            else:
                self._cat = [torch.empty((plan.n_owned + plan.n_halo, d), **f) for _ in range(2)]
            self._y = [c[: plan.n_owned] for c in self._cat]
            self._mean = torch.empty((plan.n_owned, d), **f)
            self._halo_view = self._halo[: plan.n_halo]
            self._flip = 0
            self._buf_d = d
        return self._halo, self._send

    # -- halo exchange ---------------------------------------------------------------------------
    def _exchange_staged(self, x, halo):
        plan = self.plan
        send = self.backend.gather_rows(x, self.send_idx).cpu() if len(plan.send_idx) else x.new_zeros((0, x.shape[1])).cpu()
        recv = torch.empty((plan.n_halo, x.shape[1]), dtype=x.dtype)
        ops, so, ro = [], 0, 0
        for q in range(plan.world):
            sc, rc = int(plan.send_counts[q]), int(plan.recv_counts[q])
            if q != plan.rank:
                # (the plan counts ranks inside ITS group — a column group of hybrid.py is a subgroup; P2POp names peers globally)
                peer = q if self.group is None else dist.get_global_rank(self.group, q)
                if sc:
                    ops.append(dist.P2POp(dist.isend, send[so:so + sc].contiguous(), peer, group=self.group))
                if rc:
                    ops.append(dist.P2POp(dist.irecv, recv[ro:ro + rc], peer, group=self.group))
            so += sc
            ro += rc
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        halo.copy_(recv)

    def _exchange_nccl(self, x, halo, stream=None):
        """Pack + all_to_all_single on torch's current stream (the collective is enqueued there); `stream` = that
        stream's raw handle, to spare the pack launch a lookup."""
        _, send = self._buffers(x)
        n_send = len(self.plan.send_idx)
        if n_send:
            self.backend.gather_rows(x, self.send_idx, out=send[:n_send], stream=stream)
        dist.all_to_all_single(halo, send[:n_send], output_split_sizes=self._recv_splits,
                               input_split_sizes=self._send_splits, group=self.group)

    def halo_of(self, x, out=None):
        """The halo rows of x under this plan — ONE exchange (the collective of a layer without its products), on the
        caller's stream.  Propagations that start from the same x can share it (SGL's three forwards all start from E0,
        sgl.py:129: SURVEY §8(e) "views share layer-1's all-gather"): pass the result as ``first_halo`` / ``halo_rows``,
        re-indexed with ``remap_halo`` for a plan whose halo is a subset of this one's (an edge-drop view of this graph
        under the same partition).  Every rank of the group must call it, like any layer."""
        plan = self.plan
        d = x.shape[1]
        if out is None:
            out = torch.empty((max(plan.n_halo, 1), d), dtype=x.dtype, device=x.device)
        if plan.world == 1:
            return out
        if self.transport == "nccl":
            self._buffers(x)
            self._exchange_nccl(x, out[: plan.n_halo], stream=torch.cuda.current_stream(x.device).cuda_stream)
        else:
            self._exchange_staged(x, out[: plan.n_halo])
        return out

    def halo_map_from(self, other_plan):
        """For every halo slot of this plan its slot in `other_plan`'s halo (int64 tensor on this object's device).  Raises
        unless this plan's halo ids are a subset of the other's — true for an edge-drop view of the other plan's graph
        under the same partition (a view drops edges, so it can only need fewer remote rows)."""
        mine, theirs = np.asarray(self.plan.halo_ids), np.asarray(other_plan.halo_ids)
        order = np.argsort(theirs, kind="stable")
        pos = np.searchsorted(theirs[order], mine)
        if len(mine) and (pos.max(initial=0) >= len(theirs) or not np.array_equal(theirs[order][np.minimum(pos, len(theirs) - 1)], mine)):
            raise ValueError("this plan's halo is not a subset of the other plan's halo")
        return torch.as_tensor(order[pos] if len(mine) else np.zeros(0, dtype=np.int64), dtype=torch.int64, device=self.device)

    def remap_halo(self, halo_other, index, out=None):
        """halo_other[index] ([n_halo, d], padded to one row for an empty halo): the other plan's exchanged halo in this
        plan's slot order; ``index`` = halo_map_from(other_plan).  A local gather, no communication."""
        n = int(index.shape[0])
        if out is None:
            out = torch.empty((max(n, 1), halo_other.shape[1]), dtype=halo_other.dtype, device=halo_other.device)
        if n:
            if halo_other.device.type == "cuda":
                self.backend.gather_rows(halo_other, index, out=out[:n])
            else:
                out[:n] = self.backend.gather_rows(halo_other, index)
        return out

    def _halo_product(self, halo, y, finish, main_h=None):
        """The second half of a layer: Y += A_halo·halo — or, on the last layer (finish = (srcs, out)), the same product
        with the layer mean in its epilogue: out = (srcs... + (Y + A_halo·halo)) / (len(srcs) + 1)."""
        kw = {} if main_h is None else {"stream": main_h}
        if finish is None:
            if self.g_halo is not None:
                self.backend.spmm(self.g_halo, halo, y, True, **kw)
            return y
        srcs, out = finish
        if self.g_halo is not None:
            return self.backend.spmm_mean(self.g_halo, halo, y, srcs, out, **kw)
        return self.backend.mean(list(srcs) + [y], out, **kw)  # no halo on this rank: plain mean of the kept layers

    def _more_tables(self, n):
        if self.push is not None and n > len(self._cat):
            raise ValueError(f"transport \"push\" was set up with {len(self._cat)} layer tables; this call needs {n} (push_tables=)")
        while len(self._cat) < n:
            self._cat.append(torch.empty_like(self._cat[0]))
            self._y.append(self._cat[-1][: self.plan.n_owned])

    def _table_of(self, x, avoid=None):
        """The layer table whose head holds x: x itself when it already is one (a previous layer's output), else a table
        (not the one `avoid` lives in) it is copied into."""
        n = self.plan.n_owned
        for c in self._cat:
            if c.data_ptr() == x.data_ptr() and x.shape[0] == n and x.is_contiguous():
                return c
        for c in self._cat:
            if avoid is None or c.data_ptr() != avoid.data_ptr():
                c[:n].copy_(x)
                return c
        raise RuntimeError("no free layer table")

    def _spmm_fused(self, x, out, main, finish, halo_rows):
        """One fused layer: the halo of x lands behind x's rows in its layer table, then ONE launch over [A_int | A_halo]."""
        plan = self.plan
        n, d = plan.n_owned, x.shape[1]
        if x.device.type == "cuda":
            self._buffers(x)
            main_h = (main or torch.cuda.current_stream(x.device)).cuda_stream
            kw = {"stream": main_h}
            xcat = self._table_of(x, avoid=out)
        else:  # injected CPU backend (tests): fresh tensors
            main_h, kw = None, {}
            xcat = torch.empty((n + plan.n_halo, d), dtype=x.dtype)
            xcat[:n] = x
        tail = xcat[n:]
        if halo_rows is not None:
            if halo_rows.shape[0] < max(plan.n_halo, 1):
                raise ValueError("halo_rows has fewer rows than this plan's halo")
            if halo_rows.data_ptr() != tail.data_ptr():
                tail.copy_(halo_rows[: plan.n_halo])
        elif self.transport == "push":
            pk = next(k for k, c in enumerate(self._cat) if c.data_ptr() == xcat.data_ptr())
            self.push.exchange(pk, main_h)
        elif self.transport == "nccl":
            if self._n_send:
                self.backend.gather_rows(xcat[:n], self.send_idx, out=self._send[: self._n_send], stream=main_h)
            dist.all_to_all_single(tail, self._send[: self._n_send], output_split_sizes=self._recv_splits,
                                   input_split_sizes=self._send_splits, group=self.group)
        else:
            self._exchange_staged(xcat[:n], tail)
        pushed = self.transport == "push" and halo_rows is None
        wins = self.g_cats
        if out is not None:
            y = out
        elif x.device.type == "cuda":
            y = next(c for c in self._cat if c.data_ptr() != xcat.data_ptr())[:n]
        else:
            y = torch.empty((n, d), dtype=x.dtype)
        for w, (lo, hi, g) in enumerate(wins[:-1]):  # (column windows of a table beyond 32-bit offsets: the later ones accumulate)
            self.backend.spmm(g, xcat[lo:hi], y, w > 0, **kw)
        lo, hi, g = wins[-1]
        if finish is not None:
            srcs, mean_out = finish
            if hasattr(self.backend, "spmm_mean"):
                res = self.backend.spmm_mean(g, xcat[lo:hi], y if len(wins) > 1 else None, srcs, mean_out, **kw)
            else:  # injected CPU backend (tests)
                self.backend.spmm(g, xcat[lo:hi], y, len(wins) > 1)
                res = mean_out.copy_((sum(srcs) + y) / float(len(srcs) + 1))
        else:
            res = self.backend.spmm(g, xcat[lo:hi], y, len(wins) > 1, **kw)
        if pushed:
            self.push.consumed(pk, main_h)  # the table's senders may overwrite its tail from here on (stream order)
        return res

    def spmm(self, x, out=None, main=None, finish=None, halo_rows=None):
        """Y[owned] = Â[owned,:]·X with X given as this rank's owned rows.  `main`: the torch stream the caller runs on
        (looked up once per propagation by forward()).  `finish` = (srcs, out_mean): this is the last layer of a
        propagation — returns out_mean = (sum(srcs) + Y) / (len(srcs) + 1) instead of Y (HIP backend only).
        `halo_rows` ([n_halo, d], from halo_of / remap_halo): the halo of x is already here — no exchange, no collective."""
        plan = self.plan
        d = x.shape[1]
        if self.fused and plan.world > 1:
            return self._spmm_fused(x, out, main, finish, halo_rows)
        if not x.is_contiguous():
            x = x.contiguous()
        if halo_rows is not None and plan.world > 1:
            if halo_rows.shape[0] < max(plan.n_halo, 1):
                raise ValueError("halo_rows has fewer rows than this plan's halo")
            if x.device.type == "cuda":
                self._buffers(x)
                y = out if out is not None else self._y[0 if self._y[0].data_ptr() != x.data_ptr() else 1]
                kw = {"stream": (main or torch.cuda.current_stream(x.device)).cuda_stream}
            else:
                y = out if out is not None else torch.empty((plan.n_owned, d), dtype=x.dtype, device=x.device)
                kw = {}
            self.backend.spmm(self.g_int, x, y, False, **kw)
            return self._halo_product(halo_rows, y, finish, kw.get("stream"))
        if x.device.type == "cuda":
            halo, _ = self._buffers(x)
            if out is not None:
                y = out
            else:
                self._flip ^= 1
                y = self._y[self._flip]  # ping-pong: x may be the other buffer (the previous layer's output)
                if y.data_ptr() == x.data_ptr():
                    self._flip ^= 1
                    y = self._y[self._flip]
        else:
            y = torch.empty((plan.n_owned, d), dtype=x.dtype, device=x.device)
            halo = torch.empty((max(plan.n_halo, 1), d), dtype=x.dtype, device=x.device)
        if plan.world == 1:
            if finish is not None:
                return self.backend.spmm_mean(self.g_int, x, None, finish[0], finish[1])
            return self.backend.spmm(self.g_int, x, y, False)
        if self.transport == "nccl":
            # all_to_all_single is a collective: every rank takes part every layer, even one whose
            # own send and receive lists are empty.
            if main is None:
                main = torch.cuda.current_stream(x.device)
            main_h = main.cuda_stream
            if not self.overlap:  # one stream: pack, exchange, interior, halo — no cross-stream events
                if self._n_send:
                    self.backend.gather_rows(x, self.send_idx, out=self._send[: self._n_send], stream=main_h)
                dist.all_to_all_single(self._halo_view, self._send[: self._n_send], output_split_sizes=self._recv_splits,
                                       input_split_sizes=self._send_splits, group=self.group)
                self.backend.spmm(self.g_int, x, y, False, stream=main_h)
                return self._halo_product(halo, y, finish, main_h)
            if self._ctx is not None:
                # begin: comm waits for x and packs, main runs the interior SpMM; the collective goes on the comm stream;
                # end: main waits for the comm stream; the halo product follows on main.
                self.backend.layer_begin(self._ctx, self.g_int, x, y, self.send_idx, self._n_send, self._send, main_h, self._comm_h)
                with torch.cuda.stream(self.comm_stream):
                    dist.all_to_all_single(self._halo_view, self._send[: self._n_send], output_split_sizes=self._recv_splits,
                                           input_split_sizes=self._send_splits, group=self.group)
                self.backend.layer_end(self._ctx, None, halo, y, main_h, self._comm_h)  # the wait only
                return self._halo_product(halo, y, finish, main_h)
            self.comm_stream.wait_stream(main)            # x is ready
            with torch.cuda.stream(self.comm_stream):
                self._exchange_nccl(x, self._halo_view, stream=self._comm_h)
            self.backend.spmm(self.g_int, x, y, False, stream=main_h)    # overlaps with the exchange
            main.wait_stream(self.comm_stream)
            return self._halo_product(halo, y, finish, main_h)
        self.backend.spmm(self.g_int, x, y, False)
        self._exchange_staged(x, halo[: plan.n_halo])  # point-to-point: only non-empty pairs talk
        return self._halo_product(halo, y, finish)

    # (Capturing the whole propagation in a HIP graph was tried on a world-size-1 RCCL group, devtools/nccl1_probe.py: a
    # lone all_to_all_single captures and replays, but the capture of this method — the collective on a second stream
    # forked from the capturing one — segfaults inside torch.cuda.graph on torch 2.10 / ROCm 7.2; with the single-stream
    # layer the capture works (236 vs 260 us) but the process hung in process-group teardown.  The N > 1 path stays
    # eager: a crash or hang cannot be caught and voted on the way an exception is.)
    def forward(self, e0, n_layers, out=None, first_halo=None):
        """mean(E_0..E_K) for the owned rows (lightgcn.py:70-81); rows [0, n_users_owned) are users.
        The result lives in a buffer this object re-uses: it is valid until the next forward() / spmm() call on this
        object (propagating two views back to back: pass ``out=`` or clone the first result).
        ``first_halo``: the halo rows of e0 in this plan's slot order (halo_of / remap_halo) — the first layer then
        skips its exchange (K - 1 collectives instead of K)."""
        if hasattr(self.backend, "spmm_mean") and e0.device.type == "cuda" and 1 <= n_layers <= 8:
            # keep the K - 1 first layer outputs; the K-th product carries the layer mean in its epilogue
            self._buffers(e0)
            self._more_tables(n_layers + 1)
            main = torch.cuda.current_stream(e0.device) if e0.device.type == "cuda" else None
            # (fused: E0 is copied behind nothing — into table 0's head, its halo lands in that table's tail; layer k writes
            # table k + 1's head, so the mean's addends E_1 .. E_{K-1} stay where the layers left them)
            srcs, x = [e0], e0
            if self.fused and self.plan.world > 1:
                self._cat[0][: self.plan.n_owned].copy_(e0)
                x = self._cat[0][: self.plan.n_owned]
            for k in range(n_layers - 1):
                x = self.spmm(x, out=self._y[k + 1], main=main, halo_rows=first_halo if k == 0 else None)
                srcs.append(x)
            return self.spmm(x, out=self._y[n_layers], main=main, finish=(srcs, self._mean if out is None else out),
                             halo_rows=first_halo if n_layers == 1 else None)
        acc = e0.clone()
        x = e0
        for k in range(n_layers):
            x = self.spmm(x, halo_rows=first_halo if k == 0 else None)
            acc += x
        acc /= float(n_layers + 1)
        if out is not None:
            out.copy_(acc)
            return out
        return acc

    def ngcf_forward(self, e0, layer_params, slope=0.2):
        """NGCF.forward (ngcf.py:92-104, message_dropout = node_dropout = 0) for the owned rows: per layer the sharded
        product P = (Â X)[owned] (one halo exchange), then BiGNNConv's dense half + LeakyReLU + L2-normalize on the rank's
        own rows, written into its column block of the [n_owned, sum(d)] concat buffer.  layer_params: [(W1, b1, W2, b2)]."""
        widths = [e0.shape[1]] + [w1.shape[0] for w1, _, _, _ in layer_params]
        out = torch.empty((self.plan.n_owned, sum(widths)), dtype=e0.dtype, device=e0.device)
        out[:, : widths[0]] = e0
        off = 0
        for (w1, b1, w2, b2), d_in, d_out in zip(layer_params, widths[:-1], widths[1:]):
            x = out[:, off: off + d_in]
            p = self.spmm(x)   # (a column block: the two-handle form makes it contiguous, the fused one copies it into a layer table)
            y = out[:, off + d_in: off + d_in + d_out]
            if hasattr(self.backend, "bignn_dense"):
                self.backend.bignn_dense(p, x, w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous(), y, True, slope)
            else:  # injected CPU backend (tests)
                z = (p + x) @ w1.T + b1 + (p * x) @ w2.T + b2
                y.copy_(torch.nn.functional.normalize(torch.nn.functional.leaky_relu(z, slope), p=2, dim=1))
            off += d_in
        return out

    def backward(self, grad_out, n_layers):
        """dL/dE0 for the owned rows given dL/d(mean) for the owned rows.  The propagation is linear and the GLOBAL matrix
        is symmetric (dataset.py:62-64 builds both directions), so  (Â^T g)[owned] = (Â g)[owned]: the backward of the
        sharded product is the same sharded product — same plan, same halo exchange, applied to the gradient:
            dE0 = (g + Â(g + Â(... + Â g))) / (K + 1)        (Horner form, K exchanges)."""
        g = grad_out.contiguous()
        x = g
        for i in range(n_layers):
            x = self.spmm(x)
            if self.fused and x.device.type == "cuda" and self.plan.world > 1:
                x.add_(g)   # the layer table's head (ours): the next step gathers it where it lies
            else:
                x = x + g  # a fresh tensor: the ping-pong buffer is free again
        return x / float(n_layers + 1)

    def autotune(self, e0, n_layers, iters=10, try_push=False):
        """nccl transport, N > 1: time a few propagations with each stream structure on the real group — and (r06) with the halo
        PUSH that needs no collective when ``try_push`` is set — take the MAX over ranks, keep the fastest on every rank.  Returns
        the record it also stores in ``self.tuned``.  (``try_push`` is opt-in: the push has only ever run between processes that
        share ONE GPU; a flag word that never arrives is caught by a time-out, a peer mapping that faults is not.)"""
        if not (self.transport == "nccl" and self.plan.world > 1 and e0.device.type == "cuda"):
            return None
        res = {}
        for ov in (False, True):   # False: the fused one-launch layer (when asked for); True: two handles, two streams
            self.set_overlap(ov)
            for _ in range(3):
                self.forward(e0, n_layers)
            torch.cuda.synchronize(e0.device)
            dist.barrier(group=self.group)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                self.forward(e0, n_layers)
            b.record()
            torch.cuda.synchronize(e0.device)
            t = torch.tensor([a.elapsed_time(b) * 1e3 / iters], device=e0.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            res[ov] = float(t[0])
        choice = res[True] < res[False]
        self.set_overlap(choice)
        self.tuned = {"single_stream_us": res[False], "overlap_us": res[True], "chosen": "overlap" if choice else "single_stream"}
        # r06: the halo push without a collective (PushExchange) as a third candidate; every rank tries, a failure anywhere (IPC
        # not available between these devices) is voted on, so nobody waits on a peer that gave up
        if try_push and self._want_fused and n_layers + 1 <= 9:
            sib, err = None, None
            try:
                sib = ShardedPropagation(self.plan, self.backend, group=self.group, transport="push", push_tables=n_layers + 1, push_timeout_ms=500)
                sib._g_cat = self._g_cat
            except Exception as ex:  # noqa: BLE001
                err = str(ex)[:160]
            ok = torch.tensor([0.0 if err else 1.0], device=e0.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if float(ok[0]) == 1.0:
                # one propagation first, checked: flag words that never arrive (a mapping this machine does not keep coherent) time
                # out in a few seconds instead of being timed ten times over
                try:
                    sib.forward(e0, n_layers)
                    sib.push.check()
                except Exception as ex:  # noqa: BLE001
                    err = str(ex)[:160]
                ok = torch.tensor([0.0 if err else 1.0], device=e0.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if float(ok[0]) == 1.0:
                for _ in range(2):
                    sib.forward(e0, n_layers)
                torch.cuda.synchronize(e0.device)
                dist.barrier(group=self.group)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(iters):
                    sib.forward(e0, n_layers)
                b.record()
                torch.cuda.synchronize(e0.device)
                t = torch.tensor([a.elapsed_time(b) * 1e3 / iters], device=e0.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                self.tuned["push_us"] = float(t[0])
                if float(t[0]) < min(res.values()):  # adopt the sibling's transport and tables
                    self.set_overlap(False)
                    self.transport, self.push, self._push_tables = "push", sib.push, sib._push_tables
                    self._cat, self._y, self._mean, self._buf_d = sib._cat, sib._y, sib._mean, sib._buf_d
                    self._halo, self._send, self._halo_view, self._flip = sib._halo, sib._send, sib._halo_view, sib._flip
                    self.tuned["chosen"] = "push"
                else:
                    sib.push.close()
            else:
                self.tuned["push_error"] = err or "a peer could not set the exchange up"
        return self.tuned

    # -- full-sort scoring over a sharded table (lightgcn.py:123-133: scores = u @ item_all.T) -------------------------------
    def gather_item_table(self, mean_local, n_users, n_items):
        """All-gather the item rows of the propagated embeddings once per evaluation: every rank ends with the
        full [n_items, d] table in global item order; the user rows stay sharded."""
        plan = self.plan
        d = mean_local.shape[1]
        items_local = mean_local[plan.n_users_owned:]
        ids_local = torch.as_tensor(plan.owned[plan.n_users_owned:] - n_users, dtype=torch.int64)
        table = torch.empty((n_items, d), dtype=mean_local.dtype, device=mean_local.device)
        if plan.world == 1:
            table[ids_local.to(table.device)] = items_local
            return table
        # per-rank item counts on the SAME group as the gathers below (a propagation built on a sub-group must not enter a
        # collective of the default group: the ranks outside it would never join).  A tensor collective: works for RCCL too.
        mine = torch.tensor([int(items_local.shape[0])], dtype=torch.int64, device=mean_local.device if self.transport == "nccl" else "cpu")
        counts_t = [torch.zeros_like(mine) for _ in range(plan.world)]
        dist.all_gather(counts_t, mine, group=self.group)
        counts = [int(c) for c in counts_t]
        cap = max(counts)
        staged = self.transport != "nccl"
        pad = torch.zeros((cap, d), dtype=mean_local.dtype, device="cpu" if staged else mean_local.device)
        pad[: items_local.shape[0]] = items_local.cpu() if staged else items_local
        pid = torch.full((cap,), -1, dtype=torch.int64, device=pad.device)
        pid[: ids_local.shape[0]] = ids_local.to(pad.device)
        rows = [torch.empty_like(pad) for _ in range(plan.world)]
        ids = [torch.empty_like(pid) for _ in range(plan.world)]
        dist.all_gather(rows, pad, group=self.group)
        dist.all_gather(ids, pid, group=self.group)
        for r, i, c in zip(rows, ids, counts):
            table[i[:c].to(table.device)] = r[:c].to(table.device)
        return table

    def full_sort_scores(self, mean_local, local_users, n_users, n_items, item_table=None):
        """scores [B, n_items] of this rank's users (indices into its owned user rows) against ALL items."""
        if item_table is None:
            item_table = self.gather_item_table(mean_local, n_users, n_items)
        u = mean_local[: self.plan.n_users_owned].index_select(0, local_users.to(mean_local.device))
        if mean_local.device.type == "cuda":
            from . import ops
            return ops.score(u.contiguous(), item_table)
        return u @ item_table.T


class LayeredShardedPropagation:
    """A propagation whose layer k has its OWN matrix (SGL's "RW" augmentation, sgl.py:89-91: one sub-graph per layer; the
    reference passes a list of K (edge_index, edge_weight) pairs to forward, sgl.py:137-139): one ShardedPropagation per
    layer, all on the same partition, so a layer's output rows are the next layer's input rows.  Same calls as
    ShardedPropagation where ``sharded_sgl_forward`` / ``ShardedTrainer`` use them; the halo of E0 (first_halo) is the
    FIRST layer's."""

    def __init__(self, layer_plans, backend, group=None, transport="nccl", overlap=False, fused=None):
        if not layer_plans:
            raise ValueError("at least one layer plan")
        owned = layer_plans[0].owned
        for pl in layer_plans[1:]:
            if not np.array_equal(pl.owned, owned):
                raise ValueError("the layer plans must share one partition")
        self.layers = [ShardedPropagation(pl, backend, group=group, transport=transport, overlap=overlap, fused=fused) for pl in layer_plans]
        self.plan, self.backend = layer_plans[0], backend
        self.device = self.layers[0].device

    @property
    def overlap(self):
        return self.layers[0].overlap

    def halo_map_from(self, other_plan):
        return self.layers[0].halo_map_from(other_plan)

    def remap_halo(self, halo_other, index, out=None):
        return self.layers[0].remap_halo(halo_other, index, out=out)

    def forward(self, e0, n_layers, out=None, first_halo=None):
        if n_layers != len(self.layers):
            raise ValueError(f"{len(self.layers)} layer plans but n_layers = {n_layers}")
        acc, x = e0.clone(), e0
        for k, lay in enumerate(self.layers):
            x = lay.spmm(x, halo_rows=first_halo if k == 0 else None)
            acc += x
        acc /= float(n_layers + 1)
        if out is not None:
            out.copy_(acc)
            return out
        return acc

    def backward(self, grad_out, n_layers):
        """dE0 = (g + A_1 (g + A_2 (... (g + A_K g)))) / (K + 1): every A_k symmetric (sgl.py:113-115 rebuilds both
        directions), so step i is layer K - 1 - i's sharded product."""
        g = grad_out.contiguous()
        x = g
        for lay in reversed(self.layers):
            x = lay.spmm(x)
            x = x + g
        return x / float(n_layers + 1)


class _ShardedLightGCN(torch.autograd.Function):
    """Autograd over ShardedPropagation.forward (fused mean) / .backward (Horner chain of the same sharded product)."""

    @staticmethod
    def forward(ctx, e0, prop, n_layers):
        ctx.prop, ctx.n_layers = prop, n_layers
        return prop.forward(e0, n_layers).clone()

    @staticmethod
    def backward(ctx, grad_out):
        return ctx.prop.backward(grad_out, ctx.n_layers), None, None


def sharded_lightgcn_forward(prop, e0, n_layers):
    """Differentiable mean(E_0..E_K) of this rank's rows; e0 = the rank's rows of the embedding tables (users first)."""
    return _ShardedLightGCN.apply(e0, prop, n_layers)


class _ShardedSGLForward(torch.autograd.Function):
    """SGL's three propagations (sgl.py:128-145 on the full graph, :219-221 on the two views) of one E0 shard with ONE
    exchange of E0's halo instead of three: the views' halos are subsets of the full graph's (a view drops edges), so their
    first layers re-index the full plan's halo rows locally.  3 K - 2 collectives per forward instead of 3 K
    ("views share layer-1's all-gather of E0").  The backward has nothing to share: each propagation's
    transposed chain starts from its own gradient."""

    @staticmethod
    def forward(ctx, e0, main, views, maps, n_layers):
        ctx.props, ctx.n_layers = [main] + list(views), n_layers
        halo0 = main.halo_of(e0)
        outs = [main.forward(e0, n_layers, first_halo=halo0).clone()]
        for v, m in zip(views, maps):
            outs.append(v.forward(e0, n_layers, first_halo=v.remap_halo(halo0, m)).clone())
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        total = None
        for prop, g in zip(ctx.props, grads):
            if g is None:
                continue
            gi = prop.backward(g.contiguous(), ctx.n_layers)
            total = gi.clone() if total is None else total + gi
        return total, None, None, None, None


def sharded_sgl_forward(main, views, e0, n_layers, maps=None):
    """(mean_full, mean_view_1, ...) for this rank's rows, differentiable; `main` / `views`: ShardedPropagation objects over
    the full graph's plan and the edge-drop view plans built on the SAME partition (build_plans(owner=..., keep=mask));
    `maps`: [view.halo_map_from(main.plan)] if the caller keeps them across steps (the views change once per epoch)."""
    if maps is None:
        maps = [v.halo_map_from(main.plan) for v in views]
    return _ShardedSGLForward.apply(e0, main, tuple(views), tuple(maps), n_layers)


# ---- the same path behind the C ABI (no torch.distributed): rbg_comm_* / rbg_graph_create_sharded / rbg_*_sharded_f32 ------------

def comm_unique_id():
    """128 bytes rank 0 draws (ncclGetUniqueId inside the library) and passes to every rank by whatever channel the host
    program has; RCCL is bound at run time by librbgnn.so."""
    import ctypes
    from . import _lib
    buf = ctypes.create_string_buffer(128)
    _lib.check(_lib.lib.rbg_comm_unique_id(buf))
    return buf.raw


class RcclShard:
    """A rank's shard driven entirely through the C ABI: the library owns the RCCL communicator, the exchange buffers, the
    comm stream and the interior / halo graph handles; a layer is ONE host call (rbg_spmm_sharded_f32), a propagation ONE
    (rbg_lightgcn_forward_sharded_f32) — what a host program without torch.distributed binds."""

    def __init__(self, plan, comm_id, device, nranks=None, rank=None, d_max=128):
        import ctypes
        from . import _lib
        self._lib = _lib
        lib, vp = _lib.lib, _lib.c_vp
        self.plan = plan
        self.device = torch.device(device)
        nranks = plan.world if nranks is None else nranks
        rank = plan.rank if rank is None else rank
        self._comm, self._shard = vp(), vp()
        with torch.cuda.device(self.device):
            _lib.check(lib.rbg_comm_create(ctypes.byref(self._comm), nranks, rank, comm_id, self.device.index or 0))
            a = lambda x, dt: np.ascontiguousarray(x, dtype=dt)  # noqa: E731
            ir, ic, iv = a(plan.int_csr[0], np.int64), a(plan.int_csr[1], np.int32), a(plan.int_csr[2], np.float32)
            hr, hc, hv = a(plan.halo_csr[0], np.int64), a(plan.halo_csr[1], np.int32), a(plan.halo_csr[2], np.float32)
            si, sc, rc = a(plan.send_idx, np.int64), a(plan.send_counts, np.int64), a(plan.recv_counts, np.int64)
            p = lambda x: vp(x.ctypes.data)  # noqa: E731
            _lib.check(lib.rbg_graph_create_sharded(ctypes.byref(self._shard), self._comm, plan.n_owned, plan.n_users_owned, p(ir), p(ic),
                                                    p(iv), plan.n_halo, p(hr), p(hc), p(hv), p(si), p(sc), p(rc), int(d_max)))

    def _stream(self):
        return self._lib.c_vp(torch.cuda.current_stream(self.device).cuda_stream)

    def status(self):
        """"fused: <plan status>" (one handle over [owned | halo], one launch per layer) or "two handles: interior ..., halo ..."."""
        import ctypes
        buf = ctypes.create_string_buffer(512)
        self._lib.check(self._lib.lib.rbg_shard_status(self._shard, buf, 512))
        return buf.value.decode()

    def spmm(self, x, out=None):
        out = torch.empty_like(x) if out is None else out
        self._lib.check(self._lib.lib.rbg_spmm_sharded_f32(self._shard, self._lib.c_vp(x.data_ptr()), self._lib.c_vp(out.data_ptr()),
                                                           x.shape[1], self._stream()))
        return out

    def forward(self, e0, n_layers, out=None):
        out = torch.empty_like(e0) if out is None else out
        layers = torch.empty((n_layers,) + tuple(e0.shape), dtype=e0.dtype, device=e0.device)
        self._lib.check(self._lib.lib.rbg_lightgcn_forward_sharded_f32(self._shard, self._lib.c_vp(e0.data_ptr()), self._lib.c_vp(out.data_ptr()),
                                                                       self._lib.c_vp(layers.data_ptr()), e0.shape[1], n_layers,
                                                                       self._stream()))
        self._layers = layers  # stream-ordered use: keep the buffer alive until the next call
        return out

    def forward_into(self, e0, n_layers, out, layers):
        """The same propagation into caller-owned buffers (``layers``: [n_layers, n_owned, d]): no allocation, so the call
        can sit inside a HIP-graph capture — the library issues its grouped ncclSend / ncclRecv on its own comm stream,
        forked from and joined back to the capturing stream by events."""
        self._lib.check(self._lib.lib.rbg_lightgcn_forward_sharded_f32(self._shard, self._lib.c_vp(e0.data_ptr()), self._lib.c_vp(out.data_ptr()),
                                                                       self._lib.c_vp(layers.data_ptr()), e0.shape[1], n_layers,
                                                                       self._stream()))
        return out

    def close(self):
        if getattr(self, "_shard", None):
            torch.cuda.synchronize(self.device)
            self._lib.lib.rbg_shard_destroy(self._shard)
            self._lib.lib.rbg_comm_destroy(self._comm)
            self._shard = self._comm = None

    def __del__(self):
        try:
            if not torch.cuda.is_current_stream_capturing():  # (close() synchronises and frees: never inside a capture)
                self.close()
        except Exception:  # noqa: BLE001
            pass
