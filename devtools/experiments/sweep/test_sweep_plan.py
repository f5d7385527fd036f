"""The column-sweep planner (recbole-gnn_amd/sweep.py) on the CPU: a plan must re-cut the CSR without changing the
operator.  ``simulate`` executes a plan exactly as the kernel does (piece by piece into LDS-slot accumulators, rows
finished from their slots), so the planner is checked against the oracle's product without a GPU; the kernel itself is
checked against the same oracle in tests/test_gpu_parity.py::test_sweep_kernel."""
import numpy as np
import pytest

from oracle import coracle as C


@pytest.mark.parametrize("cfg", [dict(d=64, threads=256, n_wg=16), dict(d=64, threads=512, n_wg=16, range_bytes=16 * 1024),
                                 dict(d=32, threads=256, n_wg=16), dict(d=128, threads=256, n_wg=24),
                                 dict(d=64, threads=256, n_wg=16, hot_rows_per_class=40),
                                 dict(d=64, threads=1024, n_wg=24, hot_rows_per_class=64, range_bytes=32 * 1024)])
def test_plan_reproduces_the_product(rbg, ref_inter, cfg):
    from recbole_gnn_amd import sweep
    uid, iid, nu, ni = ref_inter
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    n = nu + ni
    cfg = dict(cfg)
    d = cfg.pop("d")
    cfg.setdefault("range_bytes", 64 * 1024)
    plan = sweep.build_plan(rowptr, col, val, nu, d, lds_bytes=64 * 1024, **cfg)
    # every CSR entry exactly once, values untouched
    cnt = (plan.pieces[:, 1] >> 16) & 0xFF
    assert int(cnt.sum()) == len(col) == len(plan.ent) and cnt.max() <= min(16, d // 4)
    assert np.array_equal(np.sort(plan.ent[:, 1].view(np.float32)), np.sort(val))
    assert np.array_equal(np.sort(plan.rows[:, 0]), np.arange(n))
    assert plan.lds_floats * 4 <= 64 * 1024
    x = np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)
    y = sweep.simulate(plan, x)
    ref = C.spmm(rowptr, col, val, x)
    assert np.abs(y - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    assert not np.any(y[[0, nu]])  # the PAD rows stay empty


def test_plan_single_class_and_capacity_error(rbg, ref_inter):
    from recbole_gnn_amd import sweep
    uid, iid, nu, ni = ref_inter
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    plan = sweep.build_plan(rowptr, col, val, 0, 64, n_wg=8, threads=256, lds_bytes=160 * 1024)  # one row class on all XCDs
    x = np.random.default_rng(1).standard_normal((nu + ni, 64)).astype(np.float32)
    assert np.abs(sweep.simulate(plan, x) - C.spmm(rowptr, col, val, x)).max() <= 1e-5 * 4
    with pytest.raises(ValueError):
        sweep.build_plan(rowptr, col, val, nu, 64, n_wg=8, threads=256, lds_bytes=8 * 1024)
