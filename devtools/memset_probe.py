"""Does a hipMemsetAsync node of a captured HIP graph keep its value over many replays?  (rbg_concat_bpr_begin_f32 with B = 0 is two
memsets and nothing else.)"""
import ctypes, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from recbole_gnn_amd._lib import c_vp, check, lib
dev = torch.device("cuda:0")
tab = torch.zeros(4, 64, device=dev)
sums, loss, coef = torch.ones(3, device=dev), torch.ones((), device=dev), torch.ones(1, device=dev)
ptrs = (c_vp * 1)(tab.data_ptr()); wid = (ctypes.c_int * 1)(64)
idx = torch.zeros(1, dtype=torch.int64, device=dev)


def call():
    check(lib.rbg_concat_bpr_begin_f32(ptrs, wid, 1, 2, 2, c_vp(idx.data_ptr()), c_vp(idx.data_ptr()), c_vp(idx.data_ptr()), 0, 0, c_vp(coef.data_ptr()),
                                       c_vp(sums.data_ptr()), c_vp(loss.data_ptr()), c_vp(torch.cuda.current_stream(dev).cuda_stream)))


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    call()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    call()
bad = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    sums.fill_(1.0); loss.fill_(1.0)
    g.replay()
    torch.cuda.synchronize()
    if float(sums.abs().sum()) != 0.0 or float(loss) != 0.0:
        bad.append((it, sums.tolist(), float(loss)))
print(json.dumps({"replays": it + 1, "first_bad": bad[:3], "n_bad": len(bad)}))
