import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def free_port():
    """A rendezvous port nothing listens on right now (the kernel picks it), for the multi-process tests."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rbg():
    import recbole_gnn_amd  # noqa: F401  (shim -> recbole-gnn_amd/)
    return recbole_gnn_amd


@pytest.fixture(scope="session")
def ref_inter():
    z = np.load(os.path.join(GOLDEN, "ref_test_inter.npz"))
    return z["uid"], z["iid"], int(z["n_users"]), int(z["n_items"])


@pytest.fixture(scope="session", params=["golden_ref_test.npz", "golden_toy.npz"])
def golden(request):
    z = np.load(os.path.join(GOLDEN, request.param))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu is running without a GPU")
    return torch.device("cuda:0")


# Known-answer graphs, shared by the oracle tests and the GPU parity tests.
def known_graphs():
    out = {}
    # one interaction (u=1, i=1): nodes {0:padU, 1:u1, 2:padI, 3:i1}; Â[1,3] = Â[3,1] = 1
    out["single"] = dict(uid=[1], iid=[1], n_users=2, n_items=2,
                         dense=np.array([[0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 0], [0, 1, 0, 0]], dtype=np.float64))
    # star: user 1 with items 1..m: weights 1/sqrt(m)
    m = 5
    dense = np.zeros((2 + m + 1, 2 + m + 1))
    for j in range(1, m + 1):
        dense[1, 2 + j] = dense[2 + j, 1] = 1.0 / np.sqrt(m)
    out["star"] = dict(uid=[1] * m, iid=list(range(1, m + 1)), n_users=2, n_items=m + 1, dense=dense)
    # 2x2 biclique: every weight 1/2
    dense = np.zeros((6, 6))
    for u in (1, 2):
        for i in (1, 2):
            dense[u, 3 + i] = dense[3 + i, u] = 0.5
    out["biclique"] = dict(uid=[1, 1, 2, 2], iid=[1, 2, 1, 2], n_users=3, n_items=3, dense=dense)
    # duplicate interaction (kept as two edges): deg(u1) = 3, deg(i1) = 2, deg(i2) = 1
    dense = np.zeros((5, 5))
    dense[1, 3] = dense[3, 1] = 2.0 / np.sqrt(3 * 2)
    dense[1, 4] = dense[4, 1] = 1.0 / np.sqrt(3 * 1)
    out["duplicate"] = dict(uid=[1, 1, 1], iid=[1, 1, 2], n_users=2, n_items=3, dense=dense)
    return out
