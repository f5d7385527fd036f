"""recbole-gnn_amd — MI355X-native LightGCN / NGCF propagation engine behind RecBole-GNN's
GeneralGraphRecommender interface (get_norm_adj_mat / forward / full_sort_predict).

The package directory name carries a hyphen; import it as ``recbole_gnn_amd`` (the shim module at
the repository root) or with ``importlib.import_module("recbole-gnn_amd")``.
Importing fails loudly if librbgnn.so (the HIP extension) has not been built; there is no CPU path.
"""
from . import _lib, colsharded, driver, graph, hybrid, models, ops, sharded, sharded_train, synth, train  # noqa: F401
from ._lib import LIB_PATH, RbgError  # noqa: F401
from .graph import (GraphHandle, InteractionDataset, device_count, find_communities, get_option, get_tuning, norm_edges,  # noqa: F401
                    set_option, set_tuning)
from .models import NCL, NGCF, SGL, GeneralGraphRecommender, LightGCN, SimGCL, XSimGCL  # noqa: F401
from .ops import BiGNNConv, LightGCNConv, full_sort_topk, gather_rows, lightgcn_forward, score, spmm  # noqa: F401

from .train import (FusedBPRAdam, FusedNCLAdam, FusedNGCFAdam, FusedSGLAdam, FusedSimGCLAdam, FusedXSimGCLAdam, GraphedStep,  # noqa: F401,E402
                    fused_stepper)

__version__ = "0.1.0"
