"""numpy planner of the column-slab SpMM prototype (devtools/slab/slab.hip): SELL-C-sigma over lane-groups.

Input: the normalized CSR of the bipartite graph in the reference's numbering (users [0, n_users), items after).
Per class (user rows / item rows):
  * hot set = the `hot_rows` highest-degree rows (they get internal ids [0, hot_rows): the LDS tile of the gathered slab);
  * a row is cut into `parts` equal pieces when it is longer than `chunk` (parts a power of two <= lane-groups per wave);
  * rows are renumbered in PROCESSING order: hot-set rows first, each segment by (parts, cold entries, hot entries)
    descending, so the rows of a unit (= the lane-groups of one wave) are consecutive and of similar length;
  * a unit's entries are stored unit-major and padded to the unit's longest piece (hot and cold sections separately,
    lengths rounded up to 2): section = batches of 8 slots, [batch][lane-group][slot], so that one wave-wide 16-byte load
    fetches a whole batch (2 entries per lane, 4 lanes per lane-group) and no lane ever masks anything.
Entry = {byte offset of the slab row (col * W * 4), bits of val}; padding = {0 | out-of-range, 0.0}."""
import numpy as np

K_PAST = 0x7FFFFFF0


def _round2(x):
    return (x + 1) // 2 * 2


def build(rowptr, col, val, n_users, n_items, W, hot_rows, chunk=64, wide=False):
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    n = [int(n_users), int(n_items)]
    base = [0, int(n_users)]
    deg = np.diff(rowptr)
    lgw = 64 // (W // 4)
    # ---- hot sets (by degree), then costs, then the processing order = internal numbering --------------------------------
    is_hot = np.zeros(n[0] + n[1], dtype=bool)
    hot = [min(hot_rows, n[0]), min(hot_rows, n[1])]
    for c in (0, 1):
        o = np.argsort(-deg[base[c]:base[c] + n[c]], kind="stable")
        is_hot[base[c] + o[:hot[c]]] = True
    rid_all = np.repeat(np.arange(n[0] + n[1]), deg)
    nhot_row = np.bincount(rid_all[is_hot[col]], minlength=n[0] + n[1]).astype(np.int64)
    order, inv, parts_of = [], [], []
    for c in (0, 1):
        d = deg[base[c]:base[c] + n[c]]
        nh = nhot_row[base[c]:base[c] + n[c]]
        p = np.ones(n[c], dtype=np.int64)
        big = d > chunk
        need = -(-d[big] // chunk)
        p[big] = np.minimum(lgw, 1 << np.ceil(np.log2(need)).astype(np.int64))
        if wide:  # rows too long for the lane-groups of ONE wave: 4 lgw parts over the 4 waves of a workgroup (LDS reduce)
            p[d > chunk * lgw] = 4 * lgw
        h = is_hot[base[c]:base[c] + n[c]]
        o = np.lexsort((-nh, -(d - nh), -p, ~h))  # hot-set rows first; then parts, cold, hot descending
        i = np.empty(n[c], dtype=np.int64)
        i[o] = np.arange(n[c])
        order.append(o)
        inv.append(i)
        parts_of.append(p[o])
    # hot-set rows must own the internal ids [0, hot): true by construction (they sort first)
    ents, heads, unit_base, n_units, stats = [], [], [], [], {}
    ent_off = 0
    for c in (0, 1):
        rows = order[c] + base[c]
        rdeg = deg[rows]
        ptr = np.concatenate([[0], np.cumsum(rdeg)])
        tot = int(ptr[-1])
        rid = np.repeat(np.arange(n[c]), rdeg)
        src = np.repeat(rowptr[rows], rdeg) + (np.arange(tot) - np.repeat(ptr[:-1], rdeg))
        ci = inv[1 - c][col[src] - base[1 - c]]
        v = val[src]
        o = np.lexsort((ci, rid))  # ascending internal column inside a row: the hot ones (ids < hot) come first
        ci, v = ci[o], v[o]
        nh_row = np.bincount(rid[ci < hot[1 - c]], minlength=n[c]).astype(np.int64)
        p = parts_of[c]
        # ---- units: consecutive rows of equal `parts`, lgw / parts rows per unit ---------------------------------------
        # run-length over p (it is non-increasing inside each of the two segments)
        change = np.nonzero(np.diff(p))[0] + 1
        seg_b = np.concatenate([[0], change])
        seg_e = np.concatenate([change, [n[c]]])
        u_row0, u_nrows, u_lp, u_pbase, u_pp = [], [], [], [], []
        for b, e in zip(seg_b, seg_e):
            pb = int(p[b])
            if pb > lgw:  # wide rows: one row = 4 units (a whole workgroup), unit j holds parts [j lgw, (j + 1) lgw)
                assert pb == 4 * lgw and b % 1 == 0
                rows_w = np.repeat(np.arange(b, e), 4)
                u_row0.append(rows_w)
                u_nrows.append(np.ones(len(rows_w), dtype=np.int64))
                u_lp.append(np.full(len(rows_w), lgw.bit_length() - 1))
                u_pbase.append(np.tile(np.arange(4) * lgw, e - b))
                u_pp.append(np.full(len(rows_w), pb))
                continue
            per = lgw // pb
            starts = np.arange(b, e, per)
            u_row0.append(starts)
            u_nrows.append(np.minimum(per, e - starts))
            u_lp.append(np.full(len(starts), pb.bit_length() - 1))
            u_pbase.append(np.zeros(len(starts), dtype=np.int64))
            u_pp.append(np.full(len(starts), pb))
        u_row0, u_nrows, u_lp = np.concatenate(u_row0), np.concatenate(u_nrows), np.concatenate(u_lp)
        u_pbase, u_pp = np.concatenate(u_pbase), np.concatenate(u_pp)
        u_wide = (u_pp > lgw).astype(np.int64)
        if wide:
            assert u_wide.sum() % 4 == 0 and (u_wide[: u_wide.sum()] == 1).all()  # wide units first: workgroup-aligned
        nu_ = len(u_row0)
        # ---- pieces: (unit, lane-group) -> (row, a, b) -------------------------------------------------------------------
        lg = np.arange(lgw)
        prow = u_row0[:, None] + (lg[None, :] >> u_lp[:, None])                      # [nu, lgw] row of the lane-group
        pvalid = (lg[None, :] >> u_lp[:, None]) < u_nrows[:, None]
        prow_c = np.where(pvalid, prow, 0)
        ppart = u_pbase[:, None] + (lg[None, :] & ((1 << u_lp[:, None]) - 1))
        pp = u_pp[:, None]
        # a piece takes the k-th slice of the row's hot entries AND the k-th slice of its cold entries (both balanced)
        r0 = ptr[prow_c]
        nhr = nh_row[prow_c]
        ncr = rdeg[prow_c] - nhr
        ha = r0 + nhr * ppart // pp
        hb = r0 + nhr * (ppart + 1) // pp
        ca = r0 + nhr + ncr * ppart // pp
        cb = r0 + nhr + ncr * (ppart + 1) // pp
        ph = np.where(pvalid, hb - ha, 0)
        pc = np.where(pvalid, cb - ca, 0)
        u_nh = _round2(ph.max(axis=1))
        u_nc = _round2(pc.max(axis=1))
        u_slots = lgw * (u_nh + u_nc)
        u_off = np.concatenate([[0], np.cumsum(u_slots)])                             # in entries, class-local
        n_ent = int(u_off[-1])
        e = np.zeros((n_ent, 2), dtype=np.int32)
        # padding: hot sections {0, 0.0}; cold sections {K_PAST, 0.0}
        sec_cold_start = u_off[:-1] + lgw * u_nh
        cold_mark = np.zeros(n_ent + 1, dtype=np.int64)
        np.add.at(cold_mark, sec_cold_start, 1)
        np.add.at(cold_mark, u_off[1:], -1)
        e[np.cumsum(cold_mark[:-1]) > 0, 0] = K_PAST
        # ---- scatter the real entries (hot and cold sections alike) -----------------------------------------------------
        flat_valid = pvalid.reshape(-1)
        fu = np.repeat(np.arange(nu_), lgw)[flat_valid]
        flg = np.tile(lg, nu_)[flat_valid]
        covered = 0
        for sec, (xa, xb, u_len) in enumerate(((ha, hb, u_nh), (ca, cb, u_nc))):
            fa, fb = xa.reshape(-1)[flat_valid], xb.reshape(-1)[flat_valid]
            ln = fb - fa
            piece = np.repeat(np.arange(len(fa)), ln)
            eidx = np.repeat(fa, ln) + (np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln))
            covered += len(eidx)
            i_sec = eidx - fa[piece]
            sec_len = u_len[fu[piece]]
            k, j = i_sec // 8, i_sec % 8
            sb = np.minimum(8, sec_len - 8 * k)
            pos = u_off[fu[piece]] + (lgw * u_nh[fu[piece]] if sec else 0) + lgw * 8 * k + flg[piece] * sb + j
            e[pos, 0] = (ci[eidx] * (W * 4)).astype(np.int32)
            e[pos, 1] = v[eidx].view(np.int32)
        assert covered == tot
        head = np.stack([u_off[:-1] + ent_off, u_row0, u_nh | (u_nc << 16), u_lp | (u_nrows << 8) | (u_wide << 16)], axis=1)
        unit_base.append(sum(n_units))
        n_units.append(nu_)
        heads.append(head)
        ents.append(e)
        ent_off += n_ent
        stats[c] = dict(rows=n[c], nnz=tot, slots=n_ent, pad=round(n_ent / max(tot, 1), 3), hot_share=float(nh_row.sum() / max(tot, 1)),
                        hot_slots=int((lgw * u_nh).sum()), cold_slots=int((lgw * u_nc).sum()), units=nu_)
    ent = np.concatenate(ents + [np.zeros((128, 2), dtype=np.int32)])
    head = np.concatenate(heads).astype(np.int32)
    orig = np.concatenate([order[0] + base[0], order[1] + base[1]]).astype(np.int32)
    slab_off = np.zeros((2, 4), dtype=np.int64)
    for c in (0, 1):
        for q in range(64 // W):
            slab_off[c, q] = base[c] * 64 + q * n[c] * W
    return dict(ent=ent, head=head, unit_base=unit_base, n_units=n_units, orig=orig, slab_off=slab_off, n_class=n, W=W,
                hot_rows=hot_rows, stats=stats)


def emulate(pl, X):
    """float64 Y = A X from the plan alone (checks the layout the kernel reads); X, Y in the reference's numbering."""
    W, lgw = pl["W"], 64 // (pl["W"] // 4)
    n, orig, ent, head = pl["n_class"], pl["orig"], pl["ent"], pl["head"]
    Y = np.zeros((n[0] + n[1], X.shape[1]))
    for c in (0, 1):
        hb = head[pl["unit_base"][c]:pl["unit_base"][c] + pl["n_units"][c]].astype(np.int64)
        obase = 0 if c == 1 else n[0]
        base = 0 if c == 0 else n[0]
        yc = np.zeros((n[c], X.shape[1]))
        for off, row0, hc, lr in hb:
            nh, nc, lp, nrows = hc & 0xFFFF, hc >> 16, lr & 0xFF, (lr >> 8) & 0xFF
            for sec, (s0, ln) in enumerate(((off, nh), (off + lgw * nh, nc))):
                for k in range(0, ln, 8):
                    sb = min(8, ln - k)
                    blk = ent[s0 + lgw * k: s0 + lgw * k + lgw * sb].reshape(lgw, sb, 2)
                    for g in range(lgw):
                        r = g >> lp
                        if r >= nrows:
                            assert (blk[g, :, 1] == 0).all()
                            continue
                        cols = blk[g, :, 0].astype(np.int64)
                        vals = blk[g, :, 1].view(np.float32).astype(np.float64)
                        ok = cols != K_PAST
                        if sec == 0:
                            assert (cols[vals != 0] // (W * 4) < pl["hot_rows"]).all()
                        else:
                            assert (cols[ok] // (W * 4) >= min(pl["hot_rows"], n[1 - c])).all()
                        cc = cols[ok] // (W * 4)
                        yc[row0 + r] += (vals[ok, None] * X[orig[obase + cc]]).sum(axis=0)
        Y[orig[base:base + n[c]]] = yc
    return Y
