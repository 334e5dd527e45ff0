"""TU graph-classification datasets for the real-data loop, restated without PyG:

* ``read_tu_data(folder, name)`` -- the TU text format (``<name>_A.txt`` 1-based ``i, j`` edge list,
  ``<name>_graph_indicator.txt``, ``<name>_graph_labels.txt``, optional ``<name>_node_labels.txt`` /
  ``<name>_node_attributes.txt``) as PyG's ``read_tu_data`` reads it for ``tu_dataset.py:72``: node labels shifted to
  start at 0 and one-hot encoded per column, node attributes in front of them, self loops removed, edges sorted and
  de-duplicated, graph labels mapped onto 0..C-1 in sorted order.
* ``expand_features`` -- the ``deg+odegN`` part of the reference's ``FeatureExpander`` (``feature_expansion.py:41-62,
  96-113``, selected by ``datasets.py:16-18``): a degree column and a one-hot degree capped at N, appended to the node
  features (a column of ones when the dataset has none).  The other switches of the feature string (``ak``, ``cent``,
  ``re*``, ``rand*``, ``groupd``) are off in every configuration the reference's ``opts.py`` produces for the CAL
  models and raise ``NotImplementedError`` here.
* ``TUDataset`` -- list-like (index by int / index tensor, ``num_features``, ``num_classes``, ``y``), what
  ``train_causal_real`` (``train_causal.py:63-76``) needs from ``TUDatasetExt``.
* ``k_fold`` -- ``utils.py:18-36``: stratified folds (sklearn ``StratifiedKFold(folds, shuffle=True,
  random_state=12345)``), validation fold = test fold (``epoch_select == 'test_max'``) or the previous one.

The data files themselves are not redistributable with this repository and there is no network in the build
environment: ``get_dataset`` reads them from ``<root>/<name>/raw`` when a user supplies them.
"""
from __future__ import annotations

import os
import re
from typing import List, Optional, Sequence

import numpy as np
import torch

from .data import Data


def _read_txt(path: str, dtype) -> np.ndarray:
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([dtype(t) for t in line.replace(",", " ").split()])
    return np.asarray(rows)


def read_tu_data(folder: str, name: str) -> List[Data]:
    def p(kind):
        return os.path.join(folder, "%s_%s.txt" % (name, kind))

    edges = _read_txt(p("A"), int).reshape(-1, 2) - 1                     # 1-based -> 0-based
    indicator = _read_txt(p("graph_indicator"), int).reshape(-1) - 1
    num_nodes = indicator.shape[0]
    y = _read_txt(p("graph_labels"), float).reshape(len(np.unique(indicator)), -1)
    x_parts = []
    if os.path.exists(p("node_attributes")):
        att = _read_txt(p("node_attributes"), float).reshape(num_nodes, -1)
        x_parts.append(torch.from_numpy(att.astype(np.float32)))
    if os.path.exists(p("node_labels")):
        lab = _read_txt(p("node_labels"), int).reshape(num_nodes, -1)
        lab = lab - lab.min(0, keepdims=True)
        for c in range(lab.shape[1]):
            x_parts.append(torch.nn.functional.one_hot(torch.from_numpy(lab[:, c]), int(lab[:, c].max()) + 1).float())
    x = torch.cat(x_parts, 1) if x_parts else None
    # graph labels: a single integer column becomes class ids 0..C-1 in sorted order
    if y.shape[1] == 1 and np.all(y == np.round(y)):
        _, inv = np.unique(y[:, 0].astype(np.int64), return_inverse=True)
        ys = torch.from_numpy(inv.astype(np.int64))
    else:
        ys = torch.from_numpy(y.astype(np.float32))
    # remove self loops, sort by (source, target), drop duplicates
    edges = edges[edges[:, 0] != edges[:, 1]]
    if len(edges):
        edges = np.unique(edges, axis=0)                                   # lexicographic order = coalesce
    node_ptr = np.concatenate([[0], np.cumsum(np.bincount(indicator, minlength=len(ys)))])
    e_graph = indicator[edges[:, 0]] if len(edges) else np.zeros(0, dtype=np.int64)
    order = np.argsort(e_graph, kind="stable")
    edges, e_graph = edges[order], e_graph[order]
    edge_ptr = np.concatenate([[0], np.cumsum(np.bincount(e_graph, minlength=len(ys)))])
    out = []
    for g in range(len(ys)):
        n0, n1 = int(node_ptr[g]), int(node_ptr[g + 1])
        ei = torch.from_numpy((edges[edge_ptr[g]:edge_ptr[g + 1]] - n0).T.copy()).long().reshape(2, -1)
        d = Data(x=x[n0:n1] if x is not None else None, edge_index=ei, y=ys[g].view(1))
        d.num_nodes = n1 - n0
        out.append(d)
    return out


def parse_feat_str(feat_str: str):
    """datasets.py:16-33 restricted to what the CAL configurations use: -> (degree: bool, onehot_maxdeg: int | None)."""
    degree = feat_str.find("deg") >= 0
    m = re.findall(r"odeg(\d+)", feat_str)
    onehot_maxdeg = int(m[0]) if m else None
    k = re.findall(r"an{0,1}k(\d+)", feat_str)
    unsupported = [(k and int(k[0]) > 0, "ak"), (re.findall(r"groupd(\d+)", feat_str), "groupd"),
                   (re.findall(r"re(\w+)", feat_str), "re*"), (re.findall(r"randa([\d\.]+)", feat_str), "randa"),
                   (re.findall(r"randd([\d\.]+)", feat_str), "randd"), (feat_str.find("cent") >= 0, "cent")]
    for on, what in unsupported:
        if on:
            raise NotImplementedError("feature string option %r is not part of the CAL configurations" % what)
    return degree, onehot_maxdeg


def expand_features(data: Data, degree: bool = True, onehot_maxdeg: Optional[int] = None) -> Data:
    """feature_expansion.py:41-62 with AK = 0, no centrality: x <- [x | deg | onehot(min(deg, maxdeg))]."""
    n = data.num_nodes
    x = data.x if data.x is not None else torch.ones(n, 1)
    row = data.edge_index[0]
    deg = torch.zeros(n, dtype=torch.float32).index_add_(0, row, torch.ones(row.numel()))        # degree(row, N)
    parts = [x]
    if degree:
        parts.append(deg.view(-1, 1))
    if onehot_maxdeg is not None and onehot_maxdeg > 0:
        capped = torch.clamp(deg, max=float(onehot_maxdeg)).long()
        parts.append(torch.nn.functional.one_hot(capped, onehot_maxdeg + 1).float())
    out = Data(x=torch.cat(parts, -1), edge_index=data.edge_index, y=data.y)
    out.num_nodes = n
    return out


class TUDataset:
    """What train_causal_real needs from TUDatasetExt: len, int / index-tensor indexing, num_features, num_classes, y."""

    def __init__(self, graphs: Sequence[Data], name: str = "TU"):
        self.graphs = list(graphs)
        self.name = name
        self.y = torch.cat([g.y.view(-1) for g in self.graphs]) if self.graphs else torch.zeros(0, dtype=torch.long)

    @property
    def num_features(self) -> int:
        return int(self.graphs[0].x.size(1))

    @property
    def num_classes(self) -> int:
        return int(self.y.max().item()) + 1

    def __len__(self) -> int:
        return len(self.graphs)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self.graphs[int(idx)]
        if torch.is_tensor(idx):
            if idx.dtype in (torch.bool, torch.uint8):
                idx = idx.nonzero().view(-1)
            idx = idx.tolist()
        return TUDataset([self.graphs[int(i)] for i in idx], self.name)

    def __iter__(self):
        return iter(self.graphs)

    def __repr__(self):
        return "%s(%d)" % (self.name, len(self))


def get_dataset(name: str, feat_str: str = "deg+odeg100", root: Optional[str] = None) -> TUDataset:
    """datasets.py:11-48: <root>/<name>/raw/<name>_*.txt -> feature-expanded dataset."""
    root = root if root else os.path.join(os.path.expanduser("~"), "pyG_data")
    raw = os.path.join(root, name, "raw")
    if not os.path.exists(os.path.join(raw, "%s_A.txt" % name)):
        raise FileNotFoundError("TU files for %r not found under %s (no download in this build: supply the raw text files)"
                                % (name, raw))
    degree, maxdeg = parse_feat_str(feat_str)
    graphs = [expand_features(g, degree, maxdeg) for g in read_tu_data(raw, name)]
    return TUDataset(graphs, name)


def k_fold(dataset, folds: int, epoch_select: str):
    """Stratified k-fold index sets with the contract of utils.py:18-36: returns (train, test, val) lists of ``folds``
    LongTensors.  Fold i's test set is the i-th stratified split (sklearn StratifiedKFold, shuffle, random_state 12345 --
    the reference's seed, so the folds are the same partition); its validation set is the test set itself for
    ``epoch_select == "test_max"``, otherwise the previous fold's test set (cyclically); the training set is every
    remaining graph, ascending."""
    from sklearn.model_selection import StratifiedKFold
    n = len(dataset)
    labels = dataset.y.view(-1).cpu().numpy()
    splitter = StratifiedKFold(folds, shuffle=True, random_state=12345)
    tests = [np.sort(held_out) for _, held_out in splitter.split(np.zeros(n), labels)]
    shift = 0 if epoch_select == "test_max" else 1
    vals = [tests[(i - shift) % folds] for i in range(folds)]
    everything = np.arange(n)
    trains = [np.setdiff1d(everything, np.union1d(tests[i], vals[i])) for i in range(folds)]
    as_t = lambda arrs: [torch.from_numpy(np.asarray(a, dtype=np.int64)) for a in arrs]
    return as_t(trains), as_t(tests), as_t(vals)
