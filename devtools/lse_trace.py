import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
F = torch.nn.functional
n, d, B = 40982, 64, 2048
q = F.normalize(torch.randn(B, d, device=dev), dim=1)
c = F.normalize(torch.randn(n, d, device=dev), dim=1)
for _ in range(3):
    rbg.ops.lse_rows_raw(q, c, 5.0, 5.0)
torch.cuda.synchronize()
trace = torch.zeros(4096, 4, dtype=torch.int64, device=dev)
os.environ["RBG_LSE_TRACE"] = str(trace.data_ptr())
rbg.ops.lse_rows_raw(q, c, 5.0, 5.0)
torch.cuda.synchronize()
t = trace.cpu().numpy()
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # us (100 MHz)
hw, xcc = t[:, 2], t[:, 3]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = xcc * 64 + se * 16 + sh * 8 + 0 * cu + cu
print("blocks", len(t), "kernel span us", en.max())
print("start pct [0,50,90,100]", np.percentile(st, [0, 50, 90, 100]).round(1))
print("end   pct [0,10,50,90,100]", np.percentile(en, [0, 10, 50, 90, 100]).round(1))
print("dur   pct [0,10,50,90,100]", np.percentile(en - st, [0, 10, 50, 90, 100]).round(1))
u, cnt = np.unique(np.stack([xcc, se, sh, cu], 1), axis=0, return_counts=True)
print("distinct CUs used", len(u), "blocks per CU histogram", np.bincount(cnt))
print("per XCD blocks", np.bincount(xcc, minlength=8))
