"""Shared test helpers (fixtures -> batches)."""
import os

import numpy as np
import torch

from cal_amd.data import Batch, Data

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ref_graph_file():
    return np.load(os.path.join(GOLDEN, "spmotif_ref_graphs.npz"))


def ref_graphs(ids=None):
    z = ref_graph_file()
    meta = [m.split(",") for m in z["meta"]]
    ids = range(len(meta)) if ids is None else ids
    return [Data(feat=torch.from_numpy(z[f"g{i}_feat"]),
                 edge_index=torch.from_numpy(z[f"g{i}_edge_index"]),
                 y=torch.from_numpy(z[f"g{i}_y"])) for i in ids]


def ref_batch(ids):
    return Batch.from_data_list(ref_graphs(ids))


def random_graph_batch(num_graphs=5, n_lo=3, n_hi=12, p=0.3, feat=6, seed=0,
                       self_loops=False, dtype=torch.float32, directed=False):
    """Small random graphs (optionally with explicit self loops / asymmetric edges)."""
    g = torch.Generator().manual_seed(seed)
    ds = []
    for _ in range(num_graphs):
        n = int(torch.randint(n_lo, n_hi + 1, (1,), generator=g))
        a = torch.rand(n, n, generator=g) < p
        if not directed:
            a = a | a.t()
        a.fill_diagonal_(False)
        if self_loops:
            a[0, 0] = True
        ei = a.nonzero().t().contiguous()
        ds.append(Data(x=torch.randn(n, feat, generator=g, dtype=dtype), edge_index=ei,
                       y=torch.randint(0, 4, (1,), generator=g)))
    return Batch.from_data_list(ds)


# Ten draws of tests/tools/fuzz_host.py / fuzz_engine.py kept as fixed cases: (seed, model, variant keywords,
# (hidden, layers, features, classes, graph sizes))
SWEEP_CASES = [
    (99001, "CausalGCN", {}, (128, 3, 10, 10, [3, 99, 23, 126, 129])),
    (99010, "CausalGCN", {"cat_or_add": "cat"}, (16, 3, 64, 10, [50, 64, 1, 2, 15])),
    (99012, "CausalGAT", {"cat_or_add": "cat"}, (80, 4, 1, 2, [44, 37, 64, 32, 64, 64, 51, 48, 51, 64, 1, 15, 29, 11, 1, 15])),
    (99022, "CausalGCN", {}, (48, 1, 139, 3, [25, 64, 58, 7, 29, 15, 50, 1, 32, 38, 20, 3, 15, 42, 38, 25])),
    (99041, "CausalGCN", {}, (32, 2, 139, 4, [11, 38, 53, 10, 62])),
    (99043, "CausalGIN", {}, (48, 0, 1, 3, [36, 32, 31, 57, 8, 39, 7, 45, 47, 3, 19, 15, 56, 2, 1, 47, 52])),
    (99047, "CausalGCN", {"without_edge_attention": True},
     (64, 3, 64, 4, [2, 26, 13, 10, 39, 35, 42, 64, 46, 63, 11, 21, 54, 36, 40, 47, 63])),
    (99049, "CausalGAT", {}, (64, 2, 10, 10, [7, 32, 29, 63, 18])),
    (99054, "CausalGIN", {}, (80, 3, 10, 3, [10, 64, 44, 53, 1, 1, 15, 34, 12, 64, 33, 32, 42, 2, 36, 29])),
    (99055, "CausalGAT", {"without_node_attention": True}, (128, 3, 37, 2, [32, 23, 3, 43, 21])),
]
