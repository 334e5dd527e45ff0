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
