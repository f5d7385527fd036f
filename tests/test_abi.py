"""The C-ABI library: loads without a GPU, exports every symbol include/rbgnn.h declares, and
follows its error convention.  No compute calls (CPU only)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "rbgnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rbg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rbg):
    lib = ctypes.CDLL(rbg.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rbgnn.h but not exported"
    assert set(syms) == set(rbg._lib.SIGNATURES), "python binding and header disagree"
    assert lib.rbg_abi_version() == 1


def test_no_torch_types_in_header():
    text = open(os.path.join(ROOT, "include", "rbgnn.h")).read()
    assert 'extern "C"' in text
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # declarations only, comments stripped
    for banned in ("torch", "at::", "Tensor", "std::", "#include <hip"):
        assert banned not in code, f"{banned!r} leaks into the C ABI"


def test_error_convention(rbg):
    lib = rbg._lib.lib
    out = ctypes.c_void_p()
    uid = np.array([1, 5], dtype=np.int64)
    iid = np.array([1, 1], dtype=np.int64)
    rc = lib.rbg_graph_create(ctypes.byref(out), 3, 3, 2, uid.ctypes.data, iid.ctypes.data, -1, 0)
    assert rc == rbg._lib.RBG_EINVAL and not out.value
    assert b"uid[1] = 5" in lib.rbg_last_error()
    assert lib.rbg_graph_create(None, 3, 3, 0, None, None, -1, 0) == rbg._lib.RBG_EINVAL
    assert lib.rbg_graph_create(ctypes.byref(out), -1, 3, 0, None, None, -1, 0) == rbg._lib.RBG_EINVAL
    assert lib.rbg_graph_create(ctypes.byref(out), 3, 3, 2, None, None, -1, 0) == rbg._lib.RBG_EINVAL
    with pytest.raises(rbg.RbgError) as ei:
        rbg.GraphHandle.from_interactions([7], [0], 3, 3)
    assert ei.value.code == rbg._lib.RBG_EINVAL
    assert lib.rbg_set_tuning(64, 8, -1) == rbg._lib.RBG_EINVAL  # wave_max < short_max
    assert lib.rbg_set_tuning(-1, -1, 8) == rbg._lib.RBG_EINVAL   # seg_len < 64
    lib.rbg_graph_destroy(None)  # no-op


def test_ops_refuse_host_graphs_and_missing_gpu(rbg):
    import torch
    lib = rbg._lib.lib
    g = rbg.GraphHandle.from_interactions([1], [1], 2, 2)
    assert not g.is_device
    # a device operator on a host graph is an error, never a silent CPU computation
    rc = lib.rbg_spmm_f32(g.ptr, ctypes.c_void_p(8), ctypes.c_void_p(16), 4, 0, None)
    assert rc == rbg._lib.RBG_ENODEV
    with pytest.raises((RuntimeError, TypeError)):
        rbg.ops.spmm_raw(g, torch.zeros(4, 4))
    # the column-slab plan (rbg_graph_attach_sell, r03) is a device structure: a host graph cannot carry one
    i32 = (ctypes.c_int32 * 2)(0, 0)
    assert lib.rbg_graph_attach_sell(g.ptr, 32, ctypes.c_void_p(8), 0, ctypes.c_void_p(8), i32, i32, ctypes.c_void_p(8)) == rbg._lib.RBG_ENODEV
    assert lib.rbg_graph_attach_sell(None, 32, None, 0, None, None, None, None) == rbg._lib.RBG_EINVAL
    assert lib.rbg_graph_sell_set_factors(g.ptr, ctypes.c_void_p(8)) == rbg._lib.RBG_EINVAL  # no plan attached
    assert lib.rbg_graph_has_sell(g.ptr, 64) == 0 and lib.rbg_graph_detach_sell(g.ptr) == 0
    assert not g.sell_eligible(64)
    with pytest.raises(ValueError):
        g.attach_sell(64)
    for key in ("sell", "sell_rowmajor", "sell_factored"):
        assert rbg.get_option(key) == 1
    assert rbg.get_option("sell_nt") == 0
    assert rbg.get_option("deterministic") == 0  # float atomics by default; 1 = ordered row scatters + fixed-point sums (csrc/ordered.h)
    rbg.set_option("deterministic", 1)
    assert rbg.get_option("deterministic") == 1
    rbg.set_option("deterministic", 0)
    for gone in ("sell_units_per_wave", "sell_depth", "sell_class_serial", "sweep"):  # measured negatives, moved out of the product in r05
        with pytest.raises(rbg.RbgError):
            rbg.set_option(gone, 1)
    if rbg.device_count() == 0:
        with pytest.raises(rbg.RbgError) as ei:
            rbg.GraphHandle.from_interactions([1], [1], 2, 2, device=0)
        assert ei.value.code == rbg._lib.RBG_ENODEV
        with pytest.raises(RuntimeError):
            rbg.LightGCN({"device": "cpu"}, rbg.InteractionDataset([1], [1], 2, 2))


def test_missing_extension_fails_loudly(rbg, tmp_path, monkeypatch):
    import importlib.util
    src = os.path.join(ROOT, "recbole-gnn_amd", "_lib.py")
    dst = tmp_path / "_lib_copy.py"
    dst.write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("_lib_copy", dst)
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError, match="no CPU fallback"):
        spec.loader.exec_module(mod)  # librbgnn.so is not next to the copy


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "recbole-gnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls it) starts two ranks through
    torch.distributed.run on 127.0.0.1; --launch-check stops after the rendezvous, so this runs without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": 2}
