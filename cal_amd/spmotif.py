"""SPMotif-style synthetic graphs, restated without networkx.

Follows the *distribution* of the reference generator chain
``utils.graph_dataset_generate`` (utils.py:59-89) ->
``gengraph.generate_graph`` (gengraph.py:51-78) ->
``synthetic_structsim.build_graph`` (synthetic_structsim.py:207-288) with

* context "tree": balanced ``node_num``-ary tree of height 2
  (utils.py:62-63: ``width_basis=2`` is the height, ``m=node_num`` the
  branching; synthetic_structsim.py:73-88),
* context "ba": Barabasi-Albert graph on ``node_num**2`` nodes, m=2
  (utils.py:62; synthetic_structsim.py:91-112),
* one motif (house 5 nodes / cycle-6 / 3x2 grid / diamond,
  synthetic_structsim.py:49-67,114-127,169-204) attached by one edge from the
  motif's first node to a uniformly chosen basis node
  (synthetic_structsim.py:243-245,266-268),
* ``int(noise * |E|)`` extra random edges between distinct non-adjacent nodes
  (gengraph.py:13-33),
* node feature = one-hot degree capped at ``max_degree - 1``
  (featgen.py:21-28),
* both directions of every undirected edge, grouped by source (what PyG's
  ``from_networkx`` yields, utils.py:55),

and the b-biased train mix of ``utils.dataset_bias_split`` (utils.py:123-159).
It uses numpy's Generator instead of networkx's RNG stream, so individual
graphs differ from the reference's; sizes and degree statistics match
(tests/test_data.py compares against reference-generated fixtures).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from .data import Data

CLASS_LIST = ["house", "cycle", "grid", "diamond"]      # utils.py:61

_MOTIFS: Dict[str, Tuple[int, List[Tuple[int, int]]]] = {
    # synthetic_structsim.py:169-195
    "house": (5, [(0, 1), (1, 2), (2, 3), (3, 0), (4, 0), (4, 1)]),
    # synthetic_structsim.py:49-67 with len_cycle=6 (gengraph.py:62)
    "cycle": (6, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 0)]),
    # synthetic_structsim.py:198-204: nx.grid_graph([3, 2]) relabelled
    "grid": (6, [(0, 1), (0, 3), (1, 2), (1, 4), (2, 5), (3, 4), (4, 5)]),
    # synthetic_structsim.py:114-127
    "diamond": (6, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 0), (5, 1), (4, 2)]),
}


def _tree_edges(r: int, height: int = 2) -> Tuple[int, List[Tuple[int, int]]]:
    n = sum(r ** h for h in range(height + 1))
    edges = [((c - 1) // r, c) for c in range(1, n)]
    return n, edges


def _ba_edges(n: int, m: int, rng: np.random.Generator) -> Tuple[int, List[Tuple[int, int]]]:
    edges = [(0, i) for i in range(1, m + 1)]            # star graph on m+1 nodes
    repeated = [0] * m + list(range(1, m + 1))
    for source in range(m + 1, n):
        targets = set()
        while len(targets) < m:
            targets.add(repeated[rng.integers(len(repeated))])
        for t in targets:
            edges.append((source, t))
        repeated.extend(targets)
        repeated.extend([source] * m)
    return n, edges


def make_graph(context: str, shape: str, node_num: int, rng: np.random.Generator,
               noise: float = 0.1, max_degree: int = 10, label: int = 0) -> Data:
    if context == "tree":
        n_basis, edges = _tree_edges(node_num, 2)
    elif context == "ba":
        n_basis, edges = _ba_edges(node_num ** 2, 2, rng)
    else:
        raise ValueError(context)
    n_s, motif = _MOTIFS[shape]
    plugin = int(rng.choice(n_basis))
    edges = list(edges) + [(a + n_basis, b + n_basis) for a, b in motif] + [(n_basis, plugin)]
    n = n_basis + n_s
    adj = [dict() for _ in range(n)]                      # insertion-ordered neighbours
    for a, b in edges:
        adj[a][b] = None
        adj[b][a] = None
    n_extra = int(len(edges) * noise)
    for _ in range(n_extra):
        while True:
            u = int(rng.integers(n))
            v = int(rng.integers(n))
            if u != v and v not in adj[u]:
                break
        adj[u][v] = None
        adj[v][u] = None
    src = np.fromiter((u for u in range(n) for _ in adj[u]), dtype=np.int64)
    dst = np.fromiter((v for u in range(n) for v in adj[u]), dtype=np.int64)
    deg = np.array([len(a) for a in adj], dtype=np.int64)
    feat = np.zeros((n, max_degree), dtype=np.float32)
    feat[np.arange(n), np.minimum(deg, max_degree - 1)] = 1.0
    return Data(feat=torch.from_numpy(feat),
                edge_index=torch.from_numpy(np.stack([src, dst])),
                y=torch.tensor([label], dtype=torch.long))


def generate_dataset(data_num: int, node_num: int = 7, noise: float = 0.1,
                     max_degree: int = 10, seed: int = 666):
    """utils.graph_dataset_generate: {context: {shape: [Data]*data_num}}."""
    rng = np.random.default_rng(seed)
    dataset = {"tree": {}, "ba": {}}
    for label, shape in enumerate(CLASS_LIST):
        tr, ba = [], []
        for _ in range(data_num):
            tr.append(make_graph("tree", shape, node_num, rng, noise, max_degree, label))
            ba.append(make_graph("ba", shape, node_num, rng, noise, max_degree, label))
        dataset["tree"][shape] = tr
        dataset["ba"][shape] = ba
    return dataset


def bias_split_counts(total: int, bias: float, num_classes: int = 4, split=(7, 1, 2)):
    """Per-class (tree, ba) counts for train/val/test of utils.dataset_bias_split
    (utils.py:130-146), including its float-truncation quirk
    (int(1400 * (1 - 0.9)) == 139)."""
    train_split, val_split, test_split = (float(s) / 10 for s in split)
    train_c = total * train_split / num_classes
    val_c = total * val_split / num_classes
    test_c = total * test_split / num_classes
    out = {}
    for shape in CLASS_LIST:
        b = bias if shape == "house" else 1 - bias
        out[shape] = dict(train=(int(train_c * b), int(train_c * (1 - b))),
                          val=(int(val_c * b), int(val_c * (1 - b))),
                          test=(int(test_c * 0.5), int(test_c * 0.5)))
    return out


def dataset_bias_split(dataset, bias: float, split=(7, 1, 2), total: int = 8000,
                       num_classes: int = 4, shuffle_seed: int = 666):
    """utils.dataset_bias_split (utils.py:123-159) -> train, val, test lists."""
    counts = bias_split_counts(total, bias, num_classes, split)
    tr_d, ba_d = dataset["tree"], dataset["ba"]
    train, val, test = [], [], []
    for shape in CLASS_LIST:
        (ttr, tba), (vtr, vba), (str_, sba) = (counts[shape]["train"], counts[shape]["val"],
                                               counts[shape]["test"])
        train += tr_d[shape][:ttr] + ba_d[shape][:tba]
        val += tr_d[shape][ttr:ttr + vtr] + ba_d[shape][tba:tba + vba]
        test += tr_d[shape][ttr + vtr:ttr + vtr + str_] + ba_d[shape][tba + vba:tba + vba + sba]
    rnd = np.random.default_rng(shuffle_seed)
    for lst in (train, val, test):
        perm = rnd.permutation(len(lst))
        lst[:] = [lst[i] for i in perm]
    return train, val, test


def train_mix(num_graphs: int, bias: float = 0.9, node_num: int = 7, noise: float = 0.1,
              max_degree: int = 10, seed: int = 666) -> List[Data]:
    """``num_graphs`` graphs drawn with the class/context proportions of the
    b-biased *training* split (1260/139 per class at b=0.9, total=8000), without
    materialising the full 16 000-graph dataset.  Used by bench.py."""
    rng = np.random.default_rng(seed)
    counts = bias_split_counts(8000, bias)
    cells, weights = [], []
    for label, shape in enumerate(CLASS_LIST):
        ttr, tba = counts[shape]["train"]
        cells += [("tree", shape, label), ("ba", shape, label)]
        weights += [ttr, tba]
    weights = np.asarray(weights, dtype=np.float64)
    picks = rng.choice(len(cells), size=num_graphs, p=weights / weights.sum())
    return [make_graph(cells[p][0], cells[p][1], node_num, rng, noise, max_degree, cells[p][2])
            for p in picks]
