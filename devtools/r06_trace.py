#!/usr/bin/env python
"""r06 (from r05_trace.py): where a layer's time goes, wave by wave, after the single-wave workgroups and the multi-unit rows.  Runs the K = 3 propagation at a shape with the
RBG_SELL_TRACE build of the library (devtools/microbench/build_sell_trace.sh; loaded through RBGNN_LIB) and reads the per-wave
clocks of its three launches (valued first layer, compact middle layer, compact last layer), for the one-wave-per-unit launch
and the resident-round launch.  JSON lines -> gpurun_out/r05_trace.jsonl
usage: r05_trace.py [shape] [d]"""
import ctypes, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ["RBGNN_LIB"] = os.path.join(HERE, "microbench", "librbgnn_selltrace.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import recbole_gnn_amd as rbg

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = rbg._lib.lib
lib.mb_sell_trace_set.argtypes = [ctypes.c_void_p]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "r06_sell_wave_clock.jsonl"), "a")
uid, iid, nu, ni = rbg.synth.make(name)
n = nu + ni
g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
gen = torch.Generator().manual_seed(1)
uwd, iwd = torch.randn(nu, d, generator=gen).to(dev), torch.randn(ni, d, generator=gen).to(dev)
o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
REG = 32768
trace = torch.zeros(4 * REG * 16, dtype=torch.int64, device=dev)
assert lib.mb_sell_trace_set(ctypes.c_void_p(trace.data_ptr())) == 0
PH = ["entry->first header + entries requested", "->first batch issued (entries arrived)", "->last batch consumed", "reduction + epilogue",
      "hand-over to the next unit"]


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else 0.0


def run(label):
    for _ in range(5):
        rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L)
    torch.cuda.synchronize()
    trace.zero_()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L)
    b.record()
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(4, REG, 16)
    first = None
    for region, rname in ((0, "layer 1 (valued, row-major E0)"), (2, "layer 2 (compact)"), (3, "layer 3 (compact, mean epilogue)")):
        r = t[region]
        r = r[r[:, 1] > 0]
        if not len(r):
            continue
        st, en = r[:, 0].astype(np.float64) / 100.0, r[:, 1].astype(np.float64) / 100.0  # us (100 MHz)
        t0 = st.min()
        if first is None:
            first = t0
        ran = r[r[:, 2] > 0]
        life = (ran[:, 1] - ran[:, 0]).astype(np.float64) / 100.0
        cyc = ran[:, 4:9].astype(np.float64)
        tot = cyc.sum(1)
        rec = {"what": "sell_trace", "workload": name, "d": d, "form": label, "launch": rname, "propagation_us_traced": a.elapsed_time(b) * 1e3,
               "waves": int(len(r)), "waves_with_units": int(len(ran)), "units": int(ran[:, 2].sum()), "slots": int(ran[:, 3].sum()),
               "launch_start_after_first_launch_us": round(t0 - first, 2),
               "span_us": round(en.max() - t0, 2),
               "wave_start_us": {"p50": round(pct(st - t0, 50), 2), "p90": round(pct(st - t0, 90), 2), "p99": round(pct(st - t0, 99), 2), "max": round(float((st - t0).max()), 2)},
               "wave_end_us": {"p1": round(pct(en - t0, 1), 2), "p10": round(pct(en - t0, 10), 2), "p50": round(pct(en - t0, 50), 2), "p90": round(pct(en - t0, 90), 2),
                               "p99": round(pct(en - t0, 99), 2)},
               "wave_life_us": {"mean": round(float(life.mean()), 2), "p50": round(pct(life, 50), 2), "p99": round(pct(life, 99), 2)},
               "units_per_wave": round(float(ran[:, 2].mean()), 2),
               "cycles_per_wave": round(float(tot.mean()), 0),
               "phase_share": {PH[k]: round(float(cyc[:, k].sum() / tot.sum()), 3) for k in range(5)},
               "phase_cycles_per_unit": {PH[k]: round(float(cyc[:, k].sum() / ran[:, 2].sum()), 0) for k in range(5)},
               "cycles_per_wave_load_batch_phases": round(float((cyc[:, 1] + cyc[:, 2]).sum() / max(1, ran[:, 3].sum()) * 8), 1),
               "xcd_last_end_us": [round(float(en[r[:, 10] == x].max() - t0), 2) if (r[:, 10] == x).any() else None for x in range(8)],
               "xcd_waves": [int((r[:, 10] == x).sum()) for x in range(8)],
               # resident waves over time: how long the launch runs with fewer than half of its peak
               }
        grid = np.linspace(0, en.max() - t0, 201)
        live = np.array([((st - t0 <= x) & (en - t0 > x)).sum() for x in grid])
        rec["resident_waves_peak"] = int(live.max())
        rec["resident_waves_at_decile"] = [int(live[i]) for i in range(0, 201, 20)]
        print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()


run("one wave per unit, single-wave workgroups, wide rows as U units (r06)")
