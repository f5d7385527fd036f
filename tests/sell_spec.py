"""Launch plan of the column-slab SpMM (csrc/sell.hip): SELL-C-sigma over lane-groups.

The propagation kernel of ``rbg_lightgcn_forward_f32`` at d = 64 keeps the layers as two column slabs ``[2][row][32]`` in an
INTERNAL row numbering and reads the graph in a sliced-ELL form cut for its wave layout.  This module builds that form from
the graph handle's device CSR with torch ops (on the handle's GPU: ~10 ms at the Gowalla shape) and attaches it
(``rbg_graph_attach_sell``).  Per row class (user rows / item rows):

* a row longer than ``chunk`` entries is cut into ``parts`` equal pieces (a power of two <= the LGW = 64 / (W/4) lane-groups
  of a wave); a row longer than ``chunk * LGW`` into U = ceil(degree / (chunk LGW)) UNITS of LGW pieces each ("wide" rows;
  r06: any number of units — their partial sums meet in a scratch slot per unit and the last unit to arrive adds them in unit
  order — where r03-r05 had exactly four, the four waves of a workgroup, and refused graphs with longer hub rows);
* rows are renumbered in PROCESSING order — (parts, degree) descending — so the rows of a unit (= the lane-groups of one
  wave) are consecutive and of similar length, heaviest first (the hardware dispatcher then balances the load);
* a unit's entries are stored unit-major and padded to its longest piece (rounded up to 2): batches of 8 slots,
  ``[batch][lane-group][slot]``, so one wave-wide 16-byte load fetches a batch and no lane masks anything.

Entry = {byte offset of the slab row (internal column * W * 4), bits of val}; padding = {K_PAST, 0.0} (a buffer load past
the table returns zeros).  ``factors`` = r with val_ij = r_i r_j when the values are the symmetric normalisation (else None).
Unit header = {first entry, first row, slots << 16 | j, log2(parts) | rows << 8 | wide << 16 | U << 17}: j = the unit's index
among the U units of its wide row (0, 0 otherwise).  The wide units are the FIRST units of their class.
The same planner runs on CPU tensors (tests)."""
from __future__ import annotations

import torch

K_PAST = 0x7FFFFFF0
CHUNK = 128  # (sweep, profiles/r03_sell_chunk_probe.jsonl: 24..192 within 4 %, best 128-192; from 256 on the longest piece is the critical path again)
MAX_UNITS = 32767  # units of one wide row (15 bits of the header)


class NotApplicable(ValueError):
    """The graph is outside what the slab path serves (a table beyond 32-bit slab offsets, a row of more than
    MAX_UNITS x chunk x LGW entries): the caller keeps the binned kernel."""


def _round2(x):
    return (x + 1) // 2 * 2


def build_plan(rowptr, col, val, n_users, n_items, W=32, chunk=CHUNK, wide=True):
    """rowptr / col / val: the normalized CSR (torch tensors, any device) in the reference's numbering (users [0, n_users),
    items after).  Returns a dict of tensors on the same device: ent [n_ent + 128, 2] int32, head [n_units, 4] int32,
    orig [N] int32 (original node id of (class, internal row)), and python ints unit_base / n_units / n_class."""
    dev = col.device
    i64 = dict(dtype=torch.int64, device=dev)
    rowptr = rowptr.to(torch.int64)
    col = col.to(torch.int64)
    n = [int(n_users), int(n_items)]
    base = [0, int(n_users)]
    deg = rowptr[1:] - rowptr[:-1]
    lgw = 64 // (W // 4)
    if max(n) * W * 4 >= K_PAST:
        raise NotApplicable("table too large for 32-bit slab offsets")
    max_deg = int(deg.max()) if deg.numel() else 0
    if max_deg > MAX_UNITS * chunk * lgw:
        raise NotApplicable(f"a row of {max_deg} entries is longer than {MAX_UNITS} units of {chunk * lgw}")
    order, inv, parts_of = [], [], []
    for c in (0, 1):
        d = deg[base[c]:base[c] + n[c]]
        p = torch.ones(n[c], **i64)
        big = d > chunk
        need = (d[big] + chunk - 1) // chunk
        p[big] = torch.clamp(2 ** torch.ceil(torch.log2(need.to(torch.float64))).to(torch.int64), max=lgw)
        if wide:
            w = d > chunk * lgw
            p[w] = (d[w] + chunk * lgw - 1) // (chunk * lgw) * lgw  # U units of lgw pieces
        o = torch.argsort(-d, stable=True)
        o = o[torch.argsort(-p[o], stable=True)]  # parts descending, then degree descending, then id
        i = torch.empty(n[c], **i64)
        i[o] = torch.arange(n[c], **i64)
        order.append(o)
        inv.append(i)
        parts_of.append(p[o])
    ents, heads, unit_base, n_units = [], [], [], []
    ent_off = 0
    lg = torch.arange(lgw, **i64)
    for c in (0, 1):
        rows = order[c] + base[c]
        rdeg = deg[rows]
        ptr = torch.zeros(n[c] + 1, **i64)
        ptr[1:] = torch.cumsum(rdeg, 0)
        tot = int(ptr[-1])
        rid = torch.repeat_interleave(torch.arange(n[c], **i64), rdeg)
        src = torch.repeat_interleave(rowptr[rows] - ptr[:-1], rdeg) + torch.arange(tot, **i64)
        ci = inv[1 - c][col[src] - base[1 - c]]
        v = val[src]
        o = torch.argsort(rid * n[1 - c] + ci)  # ascending internal column inside a row (keys are unique per edge)
        ci, v = ci[o], v[o]
        del rid, src, o
        p = parts_of[c]
        # ---- units: consecutive rows of equal `parts` ---------------------------------------------------------------------
        vals_p, counts = torch.unique_consecutive(p, return_counts=True)
        u_row0, u_nrows, u_lp, u_pbase, u_pp = [], [], [], [], []
        b = 0
        for pb, cnt in zip(vals_p.tolist(), counts.tolist()):
            e = b + cnt
            if pb > lgw:  # wide rows: one row = U = pb / lgw units; unit j holds parts [j lgw, (j + 1) lgw)
                rows_w = torch.repeat_interleave(torch.arange(b, e, **i64), pb // lgw)
                u_row0.append(rows_w)
                u_nrows.append(torch.ones(len(rows_w), **i64))
                u_lp.append(torch.full((len(rows_w),), lgw.bit_length() - 1, **i64))
                u_pbase.append((torch.arange(pb // lgw, **i64) * lgw).repeat(cnt))
                u_pp.append(torch.full((len(rows_w),), pb, **i64))
            else:
                per = lgw // pb
                starts = torch.arange(b, e, per, **i64)
                u_row0.append(starts)
                u_nrows.append(torch.clamp(e - starts, max=per))
                u_lp.append(torch.full((len(starts),), pb.bit_length() - 1, **i64))
                u_pbase.append(torch.zeros(len(starts), **i64))
                u_pp.append(torch.full((len(starts),), pb, **i64))
            b = e
        u_row0, u_nrows, u_lp = torch.cat(u_row0), torch.cat(u_nrows), torch.cat(u_lp)
        u_pbase, u_pp = torch.cat(u_pbase), torch.cat(u_pp)
        u_wide = (u_pp > lgw).to(torch.int64)
        nu_ = int(u_row0.shape[0])
        # ---- pieces: (unit, lane-group) -> entry range -----------------------------------------------------------------------
        sub = lg[None, :] >> u_lp[:, None]
        pvalid = sub < u_nrows[:, None]
        prow = torch.where(pvalid, u_row0[:, None] + sub, torch.zeros((), **i64))
        ppart = u_pbase[:, None] + (lg[None, :] & ((1 << u_lp[:, None]) - 1))
        pp = u_pp[:, None]
        r0, dg = ptr[prow], rdeg[prow]
        ca = torch.where(pvalid, r0 + dg * ppart // pp, torch.zeros((), **i64))
        cb = torch.where(pvalid, r0 + dg * (ppart + 1) // pp, torch.zeros((), **i64))
        pc = cb - ca
        u_nc = _round2(pc.max(dim=1).values) if nu_ else torch.zeros(0, **i64)
        u_off = torch.zeros(nu_ + 1, **i64)
        u_off[1:] = torch.cumsum(lgw * u_nc, 0)
        n_ent = int(u_off[-1])
        e = torch.zeros((n_ent, 2), dtype=torch.int32, device=dev)
        e[:, 0] = K_PAST
        # ---- scatter the entries to their slots --------------------------------------------------------------------------------
        fv = pvalid.reshape(-1)
        fa, ln = ca.reshape(-1)[fv], pc.reshape(-1)[fv]
        fu = torch.repeat_interleave(torch.arange(nu_, **i64), lgw)[fv]
        flg = lg.repeat(nu_)[fv]
        piece = torch.repeat_interleave(torch.arange(len(fa), **i64), ln)
        first = torch.cumsum(ln, 0) - ln
        eidx = torch.repeat_interleave(fa - first, ln) + torch.arange(int(ln.sum()), **i64)
        if int(eidx.shape[0]) != tot:
            raise RuntimeError("SELL plan does not cover every entry")
        i_sec = eidx - fa[piece]
        un = fu[piece]
        k, j = i_sec // 8, i_sec % 8
        sb = torch.clamp(u_nc[un] - 8 * k, max=8)
        pos = u_off[un] + lgw * 8 * k + flg[piece] * sb + j
        e[pos, 0] = (ci[eidx] * (W * 4)).to(torch.int32)
        e[pos, 1] = v[eidx].contiguous().view(torch.int32)
        head = torch.stack([u_off[:-1] + ent_off, u_row0, (u_nc << 16) | (u_wide * (u_pbase // lgw)),
                            u_lp | (u_nrows << 8) | (u_wide << 16) | ((u_wide * (u_pp // lgw)) << 17)], dim=1)
        unit_base.append(sum(n_units))
        n_units.append(nu_)
        heads.append(head)
        ents.append(e)
        ent_off += n_ent
        del ci, v, piece, eidx, pos
    if ent_off >= 2 ** 31 - 256:
        raise NotApplicable("SELL plan too large for 32-bit entry offsets")
    ent = torch.cat(ents + [torch.zeros((128, 2), dtype=torch.int32, device=dev)])
    head = torch.cat(heads).to(torch.int32)
    orig = torch.cat([order[0] + base[0], order[1] + base[1]]).to(torch.int32)
    return dict(ent=ent.contiguous(), head=head.contiguous(), orig=orig.contiguous(), unit_base=unit_base, n_units=n_units, n_class=n,
                W=W, n_ent=ent_off, factors=_row_factors(rowptr, col, val, deg, orig))


def _row_factors(rowptr, col, val, deg, orig):
    """r [N] float32 in the PLAN's numbering with val_ij = r_i r_j when the graph is the symmetric normalisation
    D^-1/2 A D^-1/2 with D = the row counts (dataset.py:41-79; SGL's edge-drop views alike), else None.  With factors the
    slab chains keep r (.) E_k between the layers and read 4 bytes per entry (csrc/sell.hip, rbg_graph_sell_set_factors)."""
    if val.numel() == 0:
        return None
    r = torch.where(deg > 0, deg.to(torch.float64).clamp(min=1).pow(-0.5), torch.zeros((), dtype=torch.float64, device=deg.device))
    rows = torch.repeat_interleave(torch.arange(deg.numel(), dtype=torch.int64, device=deg.device), deg)
    f = r[rows] * r[col]
    ok = bool(((val.to(torch.float64) - f).abs() <= 4e-7 * f).all())
    return r[orig.to(torch.int64)].to(torch.float32).contiguous() if ok else None


def emulate(plan, x):
    """float64 Y = A X computed from the plan alone (a check of the layout the kernel reads); x, Y in the reference's
    numbering.  Small graphs only (python loop over the units)."""
    import numpy as np
    W, lgw = plan["W"], 64 // (plan["W"] // 4)
    n, orig = plan["n_class"], plan["orig"].cpu().numpy()
    ent, head = plan["ent"].cpu().numpy(), plan["head"].cpu().numpy().astype(np.int64)
    y = np.zeros((n[0] + n[1], x.shape[1]))
    for c in (0, 1):
        obase, base = (n[0], 0) if c == 0 else (0, n[0])
        yc = np.zeros((n[c], x.shape[1]))
        for off, row0, hc, lr in head[plan["unit_base"][c]:plan["unit_base"][c] + plan["n_units"][c]]:
            nc, lp, nrows = (hc >> 16) & 0xFFFF, lr & 0xFF, (lr >> 8) & 0xFF
            for k in range(0, nc, 8):
                sb = min(8, nc - k)
                blk = ent[off + lgw * k: off + lgw * k + lgw * sb].reshape(lgw, sb, 2)
                for g in range(lgw):
                    r = g >> lp
                    cols = blk[g, :, 0].astype(np.int64)
                    vals = blk[g, :, 1].copy().view(np.float32).astype(np.float64)
                    ok = cols != K_PAST
                    if r >= nrows:
                        assert not ok.any()
                        continue
                    cc = cols[ok] // (W * 4)
                    yc[row0 + r] += (vals[ok, None] * x[orig[obase + cc]]).sum(axis=0)
        y[orig[base:base + n[c]]] = yc
    return y


def attach(handle, W=32, chunk=None):
    """Build the plan of ``handle`` (a device GraphHandle) with this specification and attach it through the C ABI
    (``rbg_graph_attach_sell`` + ``rbg_graph_sell_set_factors``): what ``GraphHandle.attach_sell(planner="spec")`` did in r03/r04."""
    import ctypes
    import recbole_gnn_amd as rbg
    lib, check, vp = rbg._lib.lib, rbg._lib.check, ctypes.c_void_p
    rowptr, col, val = handle.device_csr()
    with torch.cuda.device(handle.device):
        plan = build_plan(rowptr, col, val, handle.n_users, handle.n_rows - handle.n_users, W=W, chunk=CHUNK if chunk is None else chunk)
        ub = (ctypes.c_int32 * 2)(*plan["unit_base"])
        nu = (ctypes.c_int32 * 2)(*plan["n_units"])
        check(lib.rbg_graph_attach_sell(handle.ptr, W, vp(plan["ent"].data_ptr()), plan["n_ent"], vp(plan["head"].data_ptr()), ub, nu,
                                        vp(plan["orig"].data_ptr())))
        if plan["factors"] is not None:
            check(lib.rbg_graph_sell_set_factors(handle.ptr, vp(plan["factors"].data_ptr())))
    return handle.sell_info()
