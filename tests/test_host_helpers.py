"""Host-side helpers of the model mirrors that need no GPU: the one-occurrence mask behind SimGCL's / XSimGCL's static-shape
contrast, and the fall-backs of the fused autograd ops when their inputs are not what the library call takes."""
import numpy as np
import torch


def test_once_mask_keeps_one_occurrence_per_id(rbg):
    from recbole_gnn_amd.models import _once_mask
    gen = torch.Generator().manual_seed(0)
    for n, hi in ((1, 5), (64, 7), (500, 40), (300, 100000)):
        ids = torch.randint(0, hi, (n,), generator=gen)
        m = _once_mask(ids)
        assert m.dtype == torch.bool and m.shape == ids.shape
        kept = ids[m]
        assert kept.numel() == torch.unique(ids).numel() and torch.equal(torch.sort(kept).values, torch.unique(ids))
    assert torch.equal(_once_mask(torch.tensor([3, 3, 3])).sum(), torch.tensor(1))


def test_masked_contrast_equals_the_unique_contrast_in_float64(rbg):
    """simgcl.py:38-57 over torch.unique of the batch == the same loss with one-occurrence row / column weights over the whole
    batch (what rbg_infonce_masked_f32 computes on the GPU), in plain torch."""
    from recbole_gnn_amd.models import _once_mask
    gen = torch.Generator().manual_seed(1)
    table1, table2 = torch.randn(30, 16, generator=gen, dtype=torch.float64), torch.randn(30, 16, generator=gen, dtype=torch.float64)
    ids = torch.randint(0, 30, (200,), generator=gen)
    tau = 0.2

    def contrast(x1, x2, w=None):
        x1, x2 = torch.nn.functional.normalize(x1, dim=-1), torch.nn.functional.normalize(x2, dim=-1)
        pos = torch.exp((x1 * x2).sum(-1) / tau)
        logits = torch.exp(x1 @ x2.T / tau)
        if w is None:
            return -torch.log(pos / logits.sum(1)).sum()
        return -(torch.log(pos / (logits * w[None, :]).sum(1)) * w).sum()

    u = torch.unique(ids)
    w = _once_mask(ids).double()
    a, b = contrast(table1[u], table2[u]), contrast(table1[ids], table2[ids], w)
    assert abs(float(a) - float(b)) <= 1e-10 * abs(float(a))


def test_fused_autograd_ops_fall_back_off_the_gpu(rbg):
    layers = [torch.randn(10, 4) for _ in range(3)]
    assert torch.allclose(rbg.ops.layer_mean(layers), torch.mean(torch.stack(layers, dim=1), dim=1))
    mean = torch.randn(10, 4)
    ua, ia = torch.split(mean, [4, 6])
    idx = torch.tensor([1, 2])
    assert rbg.ops.bpr_emb_loss(ua, ia, torch.randn(4, 4), torch.randn(6, 4), idx, idx, idx, 1e-4, False) is None  # (CPU tensors)
    assert rbg.ops.bpr_emb_loss(torch.randn(4, 4), torch.randn(6, 4), torch.randn(4, 4), torch.randn(6, 4), idx, idx, idx, 1e-4, False) is None
    m = rbg.ops.dropout_mask(5, 3, 1.0, torch.device("cpu"))
    assert float(m.abs().sum()) == 0.0
    m = rbg.ops.dropout_mask(2000, 8, 0.25, torch.device("cpu"))
    vals = np.unique(m.numpy())
    assert len(vals) == 2 and vals[0] == 0.0 and abs(float(vals[1]) - 1 / 0.75) < 1e-6
    assert abs(float((m == 0).float().mean()) - 0.25) < 0.03
