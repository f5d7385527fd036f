"""The NCCL (RCCL) transport of the sharded propagation on ONE GPU: a world-size-1 process group, where all_to_all_single
is a self-copy.  A 1-rank plan is given an artificial halo (its own first rows sent to itself) so that every call of the
real N>1 path happens: pack kernel, all_to_all on the second stream, interior SpMM, halo SpMM.  Times the eager
propagation and its phases (host overhead of the N>1 path) next to the single-GPU fused propagation."""
import faulthandler, json, os, sys
faulthandler.enable()
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from recbole_gnn_amd import sharded as sh
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
uid, iid, nu, ni = rbg.synth.make("gowalla")
plan = sh.build_plans(uid, iid, nu, ni, 1)[0]
m = 5000  # artificial halo: the rank's first m rows, exchanged with itself
rng = np.random.default_rng(0)
rows = np.sort(rng.integers(0, plan.n_owned, 100_000))
rowptr = np.zeros(plan.n_owned + 1, dtype=np.int64); np.add.at(rowptr, rows + 1, 1); rowptr = np.cumsum(rowptr)
plan.halo_csr = (rowptr, rng.integers(0, m, 100_000).astype(np.int32), np.full(100_000, 1e-3, dtype=np.float32))
plan.halo_ids = np.arange(m)
plan.send_idx = np.arange(m); plan.send_counts = np.array([m]); plan.recv_counts = np.array([m])
plan.world = 2  # take the exchange branch of spmm(); the split lists keep the length of the 1-rank group
prop = sh.ShardedPropagation(plan, sh.HipBackend(dev), transport="nccl")
e0 = torch.randn(plan.n_owned, 64, device=dev) * 0.1


def timed(fn, iters=100):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


ref = prop.forward(e0, 3).clone()
out = {"kind": "nccl_world1_sharded_propagation", "eager_us": round(timed(lambda: prop.forward(e0, 3)), 1)}
phase = {}
halo_buf = torch.empty((m, 64), device=dev); y_buf = torch.empty((plan.n_owned, 64), device=dev)
phase["exchange(pack + all_to_all)"] = round(timed(lambda: prop._exchange_nccl(e0, halo_buf)), 1)
phase["interior_spmm"] = round(timed(lambda: prop.backend.spmm(prop.g_int, e0, y_buf, False)), 1)
phase["halo_spmm"] = round(timed(lambda: prop.backend.spmm(prop.g_halo, halo_buf, y_buf, True)), 1)
out["phase_us"] = phase
single = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
uw, iw = e0[:nu].contiguous(), e0[nu:].contiguous()
out["single_gpu_fused_propagation_us"] = round(timed(lambda: rbg.ops.lightgcn_forward_raw(single, uw, iw, 3)), 1)
# the two-stream variant (overlap=True: high-priority comm stream, begin / end calls with their own events)
prop2 = sh.ShardedPropagation(plan, sh.HipBackend(dev), transport="nccl", overlap=True)
got2 = prop2.forward(e0, 3)
out["max_abs_diff_overlap_vs_single_stream"] = float((got2 - ref).abs().max())
out["two_streams_high_priority_comm_us"] = round(timed(lambda: prop2.forward(e0, 3)), 1)
print(json.dumps(out))
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats, io
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(200):
        prop.forward(e0, 3)
    torch.cuda.synchronize()
    pr.disable()
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14)
    print(st.getvalue()[:3500])
dist.destroy_process_group()
