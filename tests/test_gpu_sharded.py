"""The sharded path with the PRODUCT backend (HIP kernels) on the single GPU of the test box: two gloo
processes share cuda:0 and exchange halos through the host ("staged" transport).  This covers every kernel
the multi-GPU path launches (interior / halo CSR blocks, accumulate epilogue, send-list gather); only the
RCCL all_to_all call itself is left to the multi-GPU driver run."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from oracle import coracle as C

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, uid, iid, nu, ni, k_layers, d, out_q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recbole_gnn_amd as rbg
        sh = rbg.sharded
        dev = torch.device("cuda:0")
        plan = sh.build_plans(uid, iid, nu, ni, world, ranks=[rank])[rank]
        e0 = np.random.default_rng(1).standard_normal((nu + ni, d)).astype(np.float32)
        prop = sh.ShardedPropagation(plan, sh.HipBackend(dev), transport="staged")
        mean_local = prop.forward(torch.from_numpy(e0[plan.owned]).to(dev), k_layers)
        torch.cuda.synchronize()
        rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
        ref = C.lightgcn_forward(rowptr, col, val, e0[:nu], e0[nu:], k_layers)
        err = float(np.abs(mean_local.cpu().numpy() - ref[plan.owned]).max())
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, err, plan.n_halo))
        if rank == 0:
            out_q.put(gathered)
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu(ref_inter):
    uid, iid, nu, ni = ref_inter
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, uid, iid, nu, ni, 3, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, n_halo in res:
        assert err <= 1e-5 and n_halo > 0, (rank, err, n_halo)


def test_world_size_one_is_the_plain_path(rbg, cuda, ref_inter):
    uid, iid, nu, ni = ref_inter
    sh = rbg.sharded
    plan = sh.build_plans(uid, iid, nu, ni, 1)[0]
    assert plan.n_halo == 0 and plan.n_owned == nu + ni
    e0 = torch.randn(nu + ni, 64, generator=torch.Generator().manual_seed(2))
    prop = sh.ShardedPropagation(plan, sh.HipBackend(cuda), transport="staged")
    got = prop.forward(e0.to(cuda), 3)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    ref = C.lightgcn_forward(rowptr, col, val, e0[:nu].numpy(), e0[nu:].numpy(), 3)
    assert np.abs(got.cpu().numpy() - ref).max() <= 1e-5
