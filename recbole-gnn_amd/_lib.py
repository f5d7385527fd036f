"""ctypes binding of librbgnn.so (include/rbgnn.h).  There is NO fallback: if the HIP library is
missing the import fails loudly, and every operator needs a GPU-resident graph."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RBGNN_LIB: a diagnostic build of the same sources, e.g. devtools/microbench/librbgnn_selltrace.so — never a fallback)
LIB_PATH = os.environ.get("RBGNN_LIB") or os.path.join(_HERE, "librbgnn.so")

RBG_OK = 0
RBG_EINVAL, RBG_ENOMEM, RBG_EHIP, RBG_ESHAPE, RBG_ENODEV, RBG_EUNSUPPORTED = -1, -2, -3, -4, -5, -6
GRAPH_DEFAULT, GRAPH_KEEP_HOST, GRAPH_BUILD_ON_HOST, GRAPH_NATURAL_ORDER, GRAPH_INPUTS_ON_DEVICE = 0, 1, 2, 4, 8
FWD_DEFAULT, FWD_KEEP_LAST_LAYER, FWD_LAYERS_SCRATCH = 0, 1, 2
BIGNN_CONV_ONLY, BIGNN_LEAKY_NORM = 0, 1
MAX_FUSED_LAYERS = 8

c_i64, c_i32, c_u32, c_int, c_f32, c_vp = (ctypes.c_int64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_int,
                                           ctypes.c_float, ctypes.c_void_p)
P = ctypes.POINTER

# name -> (restype, argtypes); every symbol include/rbgnn.h declares
SIGNATURES = {
    "rbg_abi_version": (c_int, []),
    "rbg_last_error": (ctypes.c_char_p, []),
    "rbg_device_count": (c_int, [P(c_int)]),
    "rbg_set_tuning": (c_int, [c_int, c_int, c_int]),
    "rbg_get_tuning": (c_int, [P(c_int), P(c_int), P(c_int)]),
    "rbg_set_option": (c_int, [ctypes.c_char_p, c_i64]),
    "rbg_get_option": (c_int, [ctypes.c_char_p, P(c_i64)]),
    "rbg_graph_create": (c_int, [P(c_vp), c_i64, c_i64, c_i64, c_vp, c_vp, c_int, c_u32]),
    "rbg_graph_create_masked": (c_int, [P(c_vp), c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_int, c_u32]),
    "rbg_graph_create_partitioned": (c_int, [P(c_vp), c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_u32]),
    "rbg_graph_create_csr": (c_int, [P(c_vp), c_i64, c_i64, c_vp, c_vp, c_vp, c_int, c_u32]),
    "rbg_graph_create_csr_classes": (c_int, [P(c_vp), c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_u32]),
    "rbg_graph_create_coo": (c_int, [P(c_vp), c_i64, c_i64, c_vp, c_vp, c_int, c_u32]),
    "rbg_norm_edges": (c_int, [c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "rbg_graph_info": (c_int, [c_vp, P(c_i64), P(c_i64), P(c_i64), P(c_int)]),
    "rbg_graph_bins": (c_int, [c_vp, c_int, P(c_i64), P(c_i64), P(c_i64), P(c_i64), P(c_i64)]),
    "rbg_spmm_kernel_name": (c_int, [c_vp, c_int, ctypes.c_char_p, c_int]),
    "rbg_graph_device_arrays": (c_int, [c_vp, P(c_vp), P(c_vp), P(c_vp)]),
    "rbg_graph_export_csr": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "rbg_graph_destroy": (None, [c_vp]),
    "rbg_graph_create_reweighted": (c_int, [P(c_vp), c_vp, c_vp]),
    "rbg_graph_transpose_map": (c_int, [c_vp, c_vp, c_vp]),
    "rbg_graph_attach_sell": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "rbg_graph_sell_set_factors": (c_int, [c_vp, c_vp]),
    "rbg_graph_plan_sell": (c_int, [c_vp, c_int, c_int]),
    "rbg_graph_sell_status": (c_int, [c_vp, ctypes.c_char_p, c_int]),
    "rbg_graph_sell_info": (c_int, [c_vp, P(c_int), P(c_int), P(c_i64), P(c_i32), P(c_int), P(c_int)]),
    "rbg_graph_sell_arrays": (c_int, [c_vp, P(c_vp), P(c_vp), P(c_vp), P(c_vp), P(c_vp)]),
    "rbg_graph_refresh_values": (c_int, [c_vp, c_vp]),
    "rbg_graph_detach_sell": (c_int, [c_vp]),
    "rbg_graph_has_sell": (c_int, [c_vp, c_int]),
    "rbg_lightgcn_forward_kernel_name": (c_int, [c_vp, c_int, c_u32, ctypes.c_char_p, c_int]),
    "rbg_spmm_f32": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "rbg_spmm_mean_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp]),
    "rbg_spmm_noise_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_vp]),
    "rbg_spmm_add_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "rbg_sign_noise_f32": (c_int, [c_vp, c_vp, c_i64, c_int, c_f32, c_vp, c_vp]),
    "rbg_once_mask_f32": (c_int, [c_vp, c_i64, c_i64, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "rbg_lightgcn_forward_f32": (c_int, [P(c_vp), c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_u32, c_vp]),
    "rbg_lightgcn_backward_f32": (c_int, [P(c_vp), c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "rbg_bignn_conv_f32": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_int,
                                   c_u32, c_f32, c_vp]),
    "rbg_bignn_dense_f32": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_u32, c_f32, c_vp]),
    "rbg_bpr_grad_f32": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "rbg_emb_reg_grad_f32": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_f32, c_vp, c_vp, c_vp]),
    "rbg_emb_reg_grad_nopow_f32": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "rbg_concat_bpr_begin_f32": (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]),
    "rbg_concat_bpr_scatter_f32": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_i64, c_f32, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rbg_adam_step_dev_f32": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "rbg_adam_step_dev_total_f32": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp]),
    "rbg_lightgcn_step_head_f32": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_vp]),
    "rbg_lightgcn_step_tail_f32": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_f32, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "rbg_adam_step_f32": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "rbg_score_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp]),
    "rbg_full_sort_topk_workspace": (c_int, [c_i64, c_i64, c_int, P(c_i64)]),
    "rbg_full_sort_topk_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "rbg_shard_ctx_create": (c_int, [P(c_vp), c_int]),
    "rbg_shard_ctx_destroy": (None, [c_vp]),
    "rbg_shard_layer_begin": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_vp]),
    "rbg_shard_layer_end": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "rbg_comm_unique_id": (c_int, [c_vp]),
    "rbg_comm_create": (c_int, [P(c_vp), c_int, c_int, c_vp, c_int]),
    "rbg_comm_destroy": (None, [c_vp]),
    "rbg_graph_create_sharded": (c_int, [P(c_vp), c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    "rbg_shard_destroy": (None, [c_vp]),
    "rbg_shard_status": (c_int, [c_vp, ctypes.c_char_p, c_int]),
    "rbg_spmm_sharded_f32": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "rbg_lightgcn_forward_sharded_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "rbg_ipc_alloc": (c_int, [P(c_vp), c_i64, c_int]),
    "rbg_ipc_free": (None, [c_vp]),
    "rbg_ipc_export": (c_int, [c_vp, c_vp]),
    "rbg_ipc_open": (c_int, [c_vp, P(c_vp), c_int]),
    "rbg_ipc_close": (c_int, [c_vp]),
    "rbg_ipc_signal": (c_int, [c_vp, ctypes.c_uint64, c_vp]),
    "rbg_ipc_wait": (c_int, [c_vp, c_int, ctypes.c_uint64, c_int, c_vp, c_vp]),
    "rbg_mean_f32": (c_int, [c_vp, c_int, c_i64, c_f32, c_vp, c_vp]),
    "rbg_gather_rows_f32": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_vp]),
    "rbg_bignn_layer_f32": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_f32, c_vp]),
    "rbg_bignn_backward_workspace": (c_int, [c_i64, c_int, c_int, P(c_i64)]),
    "rbg_bignn_backward_f32": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_f32,
                                       c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rbg_bignn_wgrad_workspace": (c_int, [c_i64, c_int, c_int, P(c_i64)]),
    "rbg_bignn_wgrad_f32": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rbg_infonce_workspace": (c_int, [c_i64, c_i64, c_int, P(c_i64)]),
    "rbg_infonce_f32": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rbg_infonce_masked_f32": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rbg_infonce_map_f32": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rbg_infonce_batch_f32": (c_int, [c_vp, c_vp, c_int, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rbg_lse_rows_workspace": (c_int, [c_i64, c_i64, c_int, P(c_i64)]),
    "rbg_lse_rows_f32": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_int, c_f32, c_f32, c_vp, c_vp, c_vp]),
    "rbg_lse_rows_backward_f32": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_int, c_f32, c_f32, c_vp, c_vp, c_vp,
                                          c_vp, c_vp, c_vp]),
}


class RbgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librbgnn error {code}: {msg}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C recbole-gnn_amd/csrc`). "
            "There is no CPU fallback for this engine.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, also loud
        fn.restype = res
        fn.argtypes = args
    if lib.rbg_abi_version() != 1:
        raise ImportError(f"librbgnn.so ABI {lib.rbg_abi_version()} != 1")
    return lib


lib = _load()


def check(rc):
    if rc != RBG_OK:
        raise RbgError(rc, lib.rbg_last_error().decode("utf-8", "replace"))
