"""Synthetic stand-ins for the benchmark configurations whose data files cannot be had here
(SURVEY.md section 8d, configs 3-5): the TU files (MUTAG, NCI1) are not in the reference tree and
there is no network, so the graphs below reproduce the published size statistics only.

* ``tu_like(kind="mutag")``: ~17.9 nodes, ~39.6 directed edges per graph, 7 node labels, 2 classes;
  features laid out like the reference's ``deg+odeg100`` expansion (datasets.py:16-18,
  feature_expansion.py:96-113): node-label one-hot, one normalised-degree column, one-hot degree
  over 0..100 -> F = 7 + 1 + 101 = 109.
* ``tu_like(kind="nci1")``: ~29.9 nodes, ~64.6 directed edges, 37 labels, 2 classes -> F = 139.
* ``ba_graphs``: Barabasi-Albert(m=2) graphs of ``n`` nodes (2*m*(n-m) directed edges), one-hot
  capped-degree features (featgen.py:21-28), the HBM stress shape of config 5.

All of it is seeded numpy; nothing here is on the measured path (batches are built before timing).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from .data import Data
from .spmotif import _ba_edges

_TU = {
    # kind: (mean nodes, extra ring-closing edges per node, node labels, classes)
    "mutag": (17.9, 0.21, 7, 2),
    "nci1": (29.9, 0.15, 37, 2),
}


def _directed(n: int, edges) -> np.ndarray:
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    both = np.concatenate([e, e[:, ::-1]], 0)
    order = np.lexsort((both[:, 1], both[:, 0]))         # grouped by source, like from_networkx
    return both[order].T.copy()


def tu_like(num_graphs: int, kind: str = "mutag", seed: int = 0) -> List[Data]:
    mean_n, ring, labels, classes = _TU[kind]
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num_graphs):
        n = int(max(4, round(rng.normal(mean_n, mean_n * 0.25))))
        # molecule-like: random tree with bounded branching + a few ring closures
        edges = set()
        deg = np.zeros(n, dtype=np.int64)
        for v in range(1, n):
            lo = max(0, v - 6)
            free = lo + np.flatnonzero(deg[lo:v] < 3)
            u = int(free[rng.integers(len(free))]) if len(free) else int(np.argmin(deg[:v]))
            edges.add((u, v))
            deg[u] += 1
            deg[v] += 1
        for _ in range(int(rng.poisson(ring * n))):
            u, v = (int(t) for t in rng.integers(n, size=2))
            if u != v and (min(u, v), max(u, v)) not in edges and deg[u] < 4 and deg[v] < 4:
                edges.add((min(u, v), max(u, v)))
                deg[u] += 1
                deg[v] += 1
        ei = _directed(n, sorted(edges))
        lab = rng.integers(labels, size=n)
        feat = np.zeros((n, labels + 1 + 101), dtype=np.float32)
        feat[np.arange(n), lab] = 1.0
        feat[:, labels] = deg / max(1.0, float(deg.max()))
        feat[np.arange(n), labels + 1 + np.minimum(deg, 100)] = 1.0
        out.append(Data(x=torch.from_numpy(feat), edge_index=torch.from_numpy(ei),
                        y=torch.tensor([int(rng.integers(classes))], dtype=torch.long)))
    return out


def ba_graphs(num_graphs: int, n: int = 5000, m: int = 2, max_degree: int = 10, classes: int = 4,
              seed: int = 0) -> List[Data]:
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num_graphs):
        _, edges = _ba_edges(n, m, rng)
        ei = _directed(n, edges)
        deg = np.bincount(ei[0], minlength=n)
        feat = np.zeros((n, max_degree), dtype=np.float32)
        feat[np.arange(n), np.minimum(deg, max_degree - 1)] = 1.0
        out.append(Data(feat=torch.from_numpy(feat), edge_index=torch.from_numpy(ei),
                        y=torch.tensor([int(rng.integers(classes))], dtype=torch.long)))
    return out
