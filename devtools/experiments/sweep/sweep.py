"""Column-sweep launch plans for the SpMM (``rbg_graph_attach_sweep``).

Host-side planner: re-cuts a graph's CSR (every entry once, values untouched) into the per-lane-group piece
streams the sweep kernel executes.  Nothing here is on the hot path — a plan is built once per graph and width.

The reference has no counterpart (torch_sparse's kernel takes the CSR as is, layers.py:19-20); what the plan must
preserve is the operator's result: ``simulate`` executes a plan on the CPU exactly as the kernel does (piece by piece,
slot by slot) so the planner is testable without a GPU.

Plan layout (all int32 / uint32 numpy arrays, see include/rbgnn.h):
  lg_ptr [n_wg * lgs + 1], pieces [n_pieces, 2], ent [n_ent, 2], wg_row_ptr [n_wg + 1], rows [n_desc, 4],
  optional hot tile: wg_hot [n_wg, 2], hot_rows [n_hot], hot_base.
"""
from __future__ import annotations

import ctypes
import heapq

import numpy as np

from ._lib import c_vp, check, lib

PIECE_FIRST, PIECE_HOT = 1, 2


class SweepPlan:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def lgs(self):
        return self.threads // (self.d // 4)

    def stats(self):
        lg_ent = np.zeros(self.n_wg * self.lgs, dtype=np.int64)
        cnt = (self.pieces[:, 1] >> 16) & 0xFF
        np.add.at(lg_ent, np.repeat(np.arange(self.n_wg * self.lgs), np.diff(self.lg_ptr)), cnt)
        per_wg = lg_ent.reshape(self.n_wg, self.lgs)
        return {"n_wg": self.n_wg, "threads": self.threads, "lds_bytes": self.lds_floats * 4, "n_pieces": int(len(self.pieces)),
                "n_ent": int(len(self.ent)), "entries_per_piece": float(len(self.ent)) / max(len(self.pieces), 1),
                "lg_entries_max": int(lg_ent.max(initial=0)), "lg_entries_mean": float(lg_ent.mean()) if len(lg_ent) else 0.0,
                "wg_entries_max": int(per_wg.sum(1).max(initial=0)), "wg_entries_mean": float(per_wg.sum(1).mean()),
                "n_phases": self.n_phases, "n_hot": int(len(self.hot_rows)) if self.hot_rows is not None else 0}


def _lpt(costs, n_bins):
    """Longest-processing-time assignment: items (descending cost order assumed) -> least loaded bin."""
    heap = [(0.0, b) for b in range(n_bins)]
    out = np.empty(len(costs), dtype=np.int64)
    for i, c in enumerate(costs):
        load, b = heapq.heappop(heap)
        out[i] = b
        heapq.heappush(heap, (load + float(c), b))
    return out


def build_plan(rowptr, col, val, n_users, d, *, n_wg=768, threads=512, range_bytes=2 << 20, xcd_users=4,
               lds_bytes=48 * 1024, hot_rows_per_class=0, piece_cost=2.0, part_frac=0.5):
    """Plan for the square graph (rowptr, col, val) whose rows [0, n_users) gather columns [n_users, N) and vice versa
    (n_users <= 0 or >= N: one row class over all XCDs).
      n_wg, threads : persistent grid (workgroup b runs on XCD b % 8; XCDs [0, xcd_users) serve user rows)
      range_bytes   : size of one column range of the gathered table (what an XCD's L2 should hold at a time)
      lds_bytes     : LDS per workgroup (accumulator slots + hot tile); rows are spread so that every workgroup fits
      hot_rows_per_class : the h highest-degree columns of each gathered table are served from an LDS copy
    """
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.float32)
    n = len(rowptr) - 1
    nnz = int(rowptr[-1])
    lpr = d // 4
    lgs = threads // lpr
    pm = min(16, lpr)
    deg = np.diff(rowptr)
    two = 0 < n_users < n
    classes = [(0, n_users, n_users, n), (n_users, n, 0, n_users)] if two else [(0, n, 0, n)]
    wg_class = np.array([(0 if (b % 8) < xcd_users else 1) if two else 0 for b in range(n_wg)])
    range_rows = max(1, int(range_bytes // (4 * d)))
    hot_n = int(hot_rows_per_class)
    hot_floats = hot_n * d
    slot_cap = (lds_bytes // 4 - hot_floats) // d
    if slot_cap < 2:
        raise ValueError("no LDS left for accumulators")

    row_wg = np.full(n, -1, dtype=np.int64)
    row_slot0 = np.zeros(n, dtype=np.int64)
    row_k = np.ones(n, dtype=np.int64)           # lane-groups (= slots) a row is cut over
    part_ptr = np.zeros(n + 1, dtype=np.int64)   # row -> range in part_lg
    part_lg_list = [None] * n
    wg_slots = np.zeros(n_wg, dtype=np.int64)
    e_row = np.repeat(np.arange(n), deg)
    phase = np.zeros(nnz, dtype=np.int64)
    is_hot = np.zeros(nnz, dtype=bool)
    hot_index = np.zeros(nnz, dtype=np.int64)
    hot_rows_all, wg_hot = [], np.zeros((n_wg, 2), dtype=np.int32)
    n_phases = 0
    col_deg = np.bincount(col, minlength=n) if nnz else np.zeros(n, dtype=np.int64)
    row_chunks = np.zeros(n, dtype=np.int64)

    for cls, (r0, r1, c0, c1) in enumerate(classes):
        wgs = np.flatnonzero(wg_class == cls)
        if len(wgs) == 0:
            raise ValueError("a row class has no workgroup")
        sel = (e_row >= r0) & (e_row < r1)
        ph = (col[sel] - c0) // range_rows + 1      # phase 0 is the hot tile
        if hot_n > 0:
            cand = np.arange(c0, c1)
            top = cand[np.argsort(-col_deg[cand], kind="stable")[:hot_n]]
            top = top[col_deg[top] > 0]
            lut = np.full(n, -1, dtype=np.int64)
            lut[top] = np.arange(len(top))
            h = lut[col[sel]]
            ph = np.where(h >= 0, 0, ph)
            is_hot[sel] = h >= 0
            hot_index[sel] = np.maximum(h, 0)
            base = sum(len(x) for x in hot_rows_all)
            hot_rows_all.append(top.astype(np.int32))
            wg_hot[wgs, 0] = base
            wg_hot[wgs, 1] = len(top)
        phase[sel] = ph
        n_phases = max(n_phases, int(ph.max(initial=0)) + 1)
        # chunks (<= pm entries of one row inside one phase) per row: a row is cut over at most that many lane-groups
        gkey, gcnt = np.unique(e_row[sel] * (int(ph.max(initial=0)) + 2) + ph, return_counts=True)
        np.add.at(row_chunks, gkey // (int(ph.max(initial=0)) + 2), (gcnt + pm - 1) // pm)
        t_est = max(1, (c1 - c0 + range_rows - 1) // range_rows)
        rows = np.arange(r0, r1)
        order = rows[np.argsort(-deg[rows], kind="stable")]
        cost = deg[order] + piece_cost * np.minimum(deg[order], t_est) + 1.0
        which = _lpt(cost, len(wgs))
        row_wg[order] = wgs[which]
        by_wg = np.argsort(which, kind="stable")  # keeps the descending-degree order inside a workgroup
        cut = np.searchsorted(which[by_wg], np.arange(len(wgs) + 1))
        # inside a workgroup: rows in descending degree; a row larger than the lane-group target is cut over several
        for w_i, w in enumerate(wgs):
            pick = by_wg[cut[w_i]:cut[w_i + 1]]
            mine = order[pick]
            c_m = cost[pick]
            target = max(float(c_m.sum()) / lgs, 8.0)
            heap = [(0.0, l) for l in range(lgs)]
            slot = 0
            for r, c in zip(mine.tolist(), c_m.tolist()):
                k = 1 if deg[r] == 0 else int(min(lgs, row_chunks[r], max(1, np.ceil(c / (part_frac * target)))))
                got = [heapq.heappop(heap) for _ in range(k)]
                for load, l in got:
                    heapq.heappush(heap, (load + c / k, l))
                part_lg_list[r] = [l for _, l in got]
                row_k[r] = k
                row_slot0[r] = slot
                slot += k if deg[r] > 0 else 0
            if slot > slot_cap:
                raise ValueError(f"workgroup {w} needs {slot} accumulator slots but the LDS holds {slot_cap}: "
                                 f"raise n_wg or lds_bytes")
            wg_slots[w] = slot
    part_ptr[1:] = np.cumsum(row_k)
    part_lg = np.concatenate([np.asarray(x, dtype=np.int64) for x in part_lg_list]) if n else np.zeros(0, dtype=np.int64)

    # ---- entries -> (workgroup, lane-group, phase, slot) ---------------------------------------------------------------
    if nnz:
        if hot_n > 0:  # a row's hot entries (phase 0) are scattered over its column order: regroup by (row, phase, col)
            pre = np.lexsort((col, phase, e_row))
            col, val, phase, is_hot, hot_index, e_row = col[pre], val[pre], phase[pre], is_hot[pre], hot_index[pre], e_row[pre]
        idx = np.arange(nnz)
        grp_start = np.ones(nnz, dtype=bool)                         # start of a (row, phase) group
        grp_start[1:] = (e_row[1:] != e_row[:-1]) | (phase[1:] != phase[:-1])
        gid = np.cumsum(grp_start) - 1
        g_first = idx[grp_start]
        rank = idx - g_first[gid]
        chunk_in_grp = rank // pm
        g_len = np.diff(np.append(g_first, nnz))
        g_chunks = (g_len + pm - 1) // pm
        g_row = e_row[g_first]
        # chunks of earlier groups of the same row (exclusive scan restarted per row)
        csum = np.cumsum(g_chunks) - g_chunks
        row_first_grp = np.ones(len(g_first), dtype=bool)
        row_first_grp[1:] = g_row[1:] != g_row[:-1]
        base = csum[row_first_grp][np.cumsum(row_first_grp) - 1]
        chunk_row = (csum - base)[gid] + chunk_in_grp
        part = chunk_row % row_k[e_row]
        e_lg = part_lg[part_ptr[e_row] + part]
        e_slot = row_slot0[e_row] + part
        e_wg = row_wg[e_row]
        # hot entries of a (row, phase 0) group are column-sorted by id, not by hot index: fine, order only fixes the sum order
        order = np.lexsort((idx, e_slot, phase, e_lg, e_wg))
        s_wg, s_lg, s_ph, s_slot = e_wg[order], e_lg[order], phase[order], e_slot[order]
        run_start = np.ones(nnz, dtype=bool)
        run_start[1:] = (s_wg[1:] != s_wg[:-1]) | (s_lg[1:] != s_lg[:-1]) | (s_ph[1:] != s_ph[:-1]) | (s_slot[1:] != s_slot[:-1])
        rid = np.cumsum(run_start) - 1
        r_first = idx[run_start]
        in_run = idx - r_first[rid]
        piece_start = (in_run % pm) == 0
        p_first = idx[piece_start]
        p_cnt = np.diff(np.append(p_first, nnz))
        p_wg, p_lg, p_slot, p_ph = s_wg[p_first], s_lg[p_first], s_slot[p_first], s_ph[p_first]
        # FIRST: the first piece of every (workgroup, slot) in stream order
        key = p_wg * (int(wg_slots.max(initial=0)) + 1) + p_slot
        _, first_idx = np.unique(key, return_index=True)
        flags = np.zeros(len(p_first), dtype=np.uint32)
        flags[first_idx] |= PIECE_FIRST
        if hot_n > 0:
            flags[is_hot[order][p_first]] |= PIECE_HOT
        pieces = np.empty((len(p_first), 2), dtype=np.uint32)
        pieces[:, 0] = p_first.astype(np.uint32)
        pieces[:, 1] = p_slot.astype(np.uint32) | (p_cnt.astype(np.uint32) << 16) | (flags << 24)
        ent = np.empty((nnz, 2), dtype=np.int32)
        ent[:, 0] = np.where(is_hot[order], hot_index[order], col[order]).astype(np.int32)
        ent[:, 1] = val[order].view(np.int32)
        lg_global = p_wg * lgs + p_lg
        lg_ptr = np.zeros(n_wg * lgs + 1, dtype=np.int64)
        np.add.at(lg_ptr, lg_global + 1, 1)
        lg_ptr = np.cumsum(lg_ptr)
    else:
        pieces = np.zeros((0, 2), dtype=np.uint32)
        ent = np.zeros((0, 2), dtype=np.int32)
        lg_ptr = np.zeros(n_wg * lgs + 1, dtype=np.int64)

    # ---- row descriptors ------------------------------------------------------------------------------------------------
    order_r = np.lexsort((np.arange(n), row_wg))
    rows = np.zeros((n, 4), dtype=np.int32)
    rows[:, 0] = order_r
    rows[:, 1] = row_slot0[order_r]
    rows[:, 2] = np.where(deg[order_r] > 0, row_k[order_r], 0)
    wg_row_ptr = np.zeros(n_wg + 1, dtype=np.int64)
    np.add.at(wg_row_ptr, row_wg + 1, 1)
    wg_row_ptr = np.cumsum(wg_row_ptr)
    slots_max = int(wg_slots.max(initial=1))
    hot_base = slots_max * d
    lds_floats = hot_base + hot_floats
    return SweepPlan(d=d, threads=threads, n_wg=n_wg, lds_floats=max(lds_floats, d), lg_ptr=lg_ptr.astype(np.int32),
                     pieces=pieces, ent=ent, wg_row_ptr=wg_row_ptr.astype(np.int32), rows=rows,
                     wg_hot=wg_hot if hot_n > 0 else None,
                     hot_rows=np.concatenate(hot_rows_all).astype(np.int32) if hot_n > 0 else None, hot_base=hot_base,
                     n_phases=n_phases, n_rows=n)


def simulate(plan, x):
    """Execute the plan on the CPU the way the kernel does (float32 accumulation in the plan's order) -> Y [n_rows, d]."""
    x = np.asarray(x, dtype=np.float32)
    d, lgs = plan.d, plan.lgs
    y = np.zeros((plan.n_rows, d), dtype=np.float32)
    n_slots = plan.hot_base // d
    for w in range(plan.n_wg):
        acc = np.full((n_slots, d), np.nan, dtype=np.float32)
        hot = None
        if plan.wg_hot is not None:
            hb, hn = plan.wg_hot[w]
            hot = x[plan.hot_rows[hb:hb + hn]]
        for lg in range(lgs):
            for q in range(plan.lg_ptr[w * lgs + lg], plan.lg_ptr[w * lgs + lg + 1]):
                beg, meta = int(plan.pieces[q, 0]), int(plan.pieces[q, 1])
                slot, cnt, flags = meta & 0xFFFF, (meta >> 16) & 0xFF, meta >> 24
                s = np.zeros(d, dtype=np.float32)
                for j in range(cnt):
                    c, v = plan.ent[beg + j, 0], plan.ent[beg + j, 1:2].view(np.float32)[0]
                    src = hot[c] if (flags & PIECE_HOT) else x[c]
                    s = (v * src + s).astype(np.float32)
                acc[slot] = s if (flags & PIECE_FIRST) else (acc[slot] + s).astype(np.float32)
        for r in range(plan.wg_row_ptr[w], plan.wg_row_ptr[w + 1]):
            row, s0, ns = plan.rows[r, 0], plan.rows[r, 1], plan.rows[r, 2]
            s = np.zeros(d, dtype=np.float32)
            for i in range(ns):
                s = (s + acc[s0 + i]).astype(np.float32)
            y[row] = s
    return y


def _p(a):
    return None if a is None else c_vp(a.ctypes.data)


def attach(graph, plan):
    """Upload the plan; SpMM launches of width plan.d on this handle then use the sweep kernel (option "sweep")."""
    arrs = {k: (None if getattr(plan, k) is None else np.ascontiguousarray(getattr(plan, k)))
            for k in ("lg_ptr", "pieces", "ent", "wg_row_ptr", "rows", "wg_hot", "hot_rows")}
    check(lib.rbg_graph_attach_sweep(graph.ptr, plan.d, plan.threads, plan.n_wg, plan.lds_floats, _p(arrs["lg_ptr"]),
                                     _p(arrs["pieces"]), len(arrs["pieces"]), _p(arrs["ent"]), len(arrs["ent"]),
                                     _p(arrs["wg_row_ptr"]), _p(arrs["rows"]), len(arrs["rows"]), _p(arrs["wg_hot"]),
                                     _p(arrs["hot_rows"]), 0 if arrs["hot_rows"] is None else len(arrs["hot_rows"]),
                                     plan.hot_base))


def detach(graph, d=0):
    check(lib.rbg_graph_detach_sweep(graph.ptr, d))


def plan_for_graph(graph, d, **kw):
    """Build a plan from a graph handle's CSR (exported from the device) and attach it."""
    rowptr, col, val = graph.export_csr()
    plan = build_plan(rowptr, col, val, graph.n_users if graph.n_users is not None else 0, d, **kw)
    attach(graph, plan)
    return plan
