"""Deterministic synthetic power-law bipartite interaction graphs.

p_user(r) ∝ (r+10)^-0.75, p_item(r) ∝ (r+10)^-0.75; pairs sampled by inverse CDF, de-duplicated,
topped up to exactly ``n_inter`` unique pairs; ids shifted by +1 so that id 0 of each side stays the
empty [PAD] row RecBole reserves.  ``n_users`` / ``n_items`` INCLUDE that PAD id.
An optional community structure (``n_blocks``, ``p_in``) makes a fraction ``p_in`` of every user's
interactions fall inside the user's own block of items — the locality knob the multi-GPU numbers
depend on; it is always reported next to them.
"""
from __future__ import annotations

import numpy as np

SHAPES = {
    # name: (n_users, n_items, n_inter) with PAD rows included
    "ml-100k": (944, 1683, 100_000),
    "gowalla": (29_859, 40_982, 1_027_370),
    "yelp2018": (31_669, 38_049, 1_561_406),
    "amazon-book": (52_644, 91_600, 2_984_108),
    "g-1.3m": (550_000, 750_000, 18_850_000),
    "toy": (347, 1_125, 5_999),
}


def _powerlaw_cdf(n, alpha=0.75, shift=10.0):
    w = (np.arange(n, dtype=np.float64) + shift) ** (-alpha)
    c = np.cumsum(w)
    return c / c[-1]


def contiguous_blocks(n, n_blocks):
    """Relabelling striped -> contiguous communities: 0-based rank r (block r % P, position r // P) moves to
    block * ceil(n / P) + position.  Returns (new_rank[r], block_of_new_rank[...])."""
    r = np.arange(n)
    per = (n + n_blocks - 1) // n_blocks
    new = (r % n_blocks) * per + r // n_blocks
    # compact (the last block may be shorter): rank of each new position among the used ones
    order = np.argsort(new, kind="stable")
    compact = np.empty(n, dtype=np.int64)
    compact[order] = np.arange(n)
    block = np.empty(n, dtype=np.int32)
    block[compact] = (r % n_blocks).astype(np.int32)
    return compact, block


def powerlaw_bipartite(n_users, n_items, n_inter, seed=2020, alpha=0.75, n_blocks=1, p_in=1.0, layout="striped"):
    """Returns (uid, iid) int64 arrays of exactly ``n_inter`` unique pairs, ids in [1, n).
    ``layout="contiguous"`` relabels the nodes so that every community occupies one contiguous id range (what a
    partitioner-driven relabelling produces); ``partition_of`` then gives the community of every node."""
    nu, ni = n_users - 1, n_items - 1  # real (non-PAD) ids
    if n_inter > nu * ni:
        raise ValueError("more interactions than user-item pairs")
    rng = np.random.default_rng(seed)
    cu, ci = _powerlaw_cdf(nu, alpha), _powerlaw_cdf(ni, alpha)
    keys = np.empty(0, dtype=np.int64)
    need = n_inter
    while need > 0:
        m = int(need * 1.25) + 1024
        u = np.searchsorted(cu, rng.random(m), side="right").astype(np.int64)
        i = np.searchsorted(ci, rng.random(m), side="right").astype(np.int64)
        np.minimum(u, nu - 1, out=u)
        np.minimum(i, ni - 1, out=i)
        if n_blocks > 1:
            # users/items are striped over blocks (rank r -> block r % n_blocks) so every block keeps
            # the same power-law degree profile; an "inside" interaction re-maps the item to the
            # user's block at (almost) the same popularity rank.
            inside = rng.random(m) < p_in
            ub = u % n_blocks
            i_in = (i // n_blocks) * n_blocks + ub
            i_in = np.where(i_in >= ni, i_in - n_blocks, i_in)
            i = np.where(inside, i_in, i)
        keys = np.unique(np.concatenate([keys, u * ni + i]))
        need = n_inter - keys.shape[0]
    keys = rng.permutation(keys)[:n_inter]
    u, i = keys // ni, keys % ni
    if n_blocks > 1 and layout == "contiguous":
        u = contiguous_blocks(nu, n_blocks)[0][u]
        i = contiguous_blocks(ni, n_blocks)[0][i]
    return u + 1, i + 1


def powerlaw_bipartite_device(n_users, n_items, n_inter, device, seed=2020, alpha=0.75):
    """The same generator (inverse-CDF samples of the two power laws, de-duplicated, topped up to exactly ``n_inter`` unique
    pairs, shuffled) run with torch on a GPU: seconds instead of minutes at BASELINE config #5's size (200 M interactions:
    181 s with numpy on the test box).  Same distribution, NOT the same stream as the numpy generator — a graph made here is
    labelled as such by its users (bench.py's config5 workload); tests and every other workload use the numpy one."""
    import torch
    nu, ni = n_users - 1, n_items - 1
    if n_inter > nu * ni:
        raise ValueError("more interactions than user-item pairs")
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)

    def cdf(n):
        w = (torch.arange(n, dtype=torch.float64, device=dev) + 10.0) ** (-alpha)
        c = torch.cumsum(w, 0)
        return c / c[-1]

    cu, ci = cdf(nu), cdf(ni)
    keys = torch.empty(0, dtype=torch.int64, device=dev)
    need = n_inter
    while need > 0:
        m = int(need * 1.25) + 1024
        u = torch.searchsorted(cu, torch.rand(m, generator=g, device=dev, dtype=torch.float64), right=True).clamp_(max=nu - 1)
        i = torch.searchsorted(ci, torch.rand(m, generator=g, device=dev, dtype=torch.float64), right=True).clamp_(max=ni - 1)
        keys = torch.unique(torch.cat([keys, u * ni + i]))
        del u, i
        need = n_inter - int(keys.numel())
    keys = keys[torch.randperm(int(keys.numel()), generator=g, device=dev)[:n_inter]]
    uid, iid = (keys // ni + 1).cpu().numpy(), (keys % ni + 1).cpu().numpy()
    del keys
    torch.cuda.empty_cache()
    return uid, iid


def partition_of(n_users, n_items, n_blocks, layout="striped"):
    """Community of every node (users then items; the PAD ids go to community 0) for the generator's layouts."""
    out = []
    for n in (n_users - 1, n_items - 1):
        if layout == "contiguous":
            blk = contiguous_blocks(n, n_blocks)[1]
        else:
            blk = (np.arange(n) % n_blocks).astype(np.int32)
        out.append(np.concatenate([[0], blk]).astype(np.int32))
    return np.concatenate(out)


def shape(name):
    return SHAPES[name.lower()]


def make(name, seed=2020, **kw):
    n_users, n_items, n_inter = shape(name)
    uid, iid = powerlaw_bipartite(n_users, n_items, n_inter, seed=seed, **kw)
    return uid, iid, n_users, n_items


def algorithmic_bytes(n_nodes, nnz, d, n_layers):
    """B_layer = 4(N+1) + 8 nnz + 8 N d ; B_prop = K B_layer + 4 N d (K+2)."""
    b_layer = 4 * (n_nodes + 1) + 8 * nnz + 8 * n_nodes * d
    b_prop = n_layers * b_layer + 4 * n_nodes * d * (n_layers + 2)
    return b_layer, b_prop
