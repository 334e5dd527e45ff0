"""TU text-format reader, deg+odegN feature expansion and stratified k-fold (cal_amd/tu.py; reference call sites
tu_dataset.py:72, datasets.py:11-48, feature_expansion.py:41-113, utils.py:18-36) on a tiny hand-written dataset."""
import os

import numpy as np
import pytest
import torch

from cal_amd import tu


def _write(tmp, name="TOY"):
    raw = os.path.join(tmp, name, "raw")
    os.makedirs(raw)
    # graph 1: triangle 1-2-3 (+ a self loop and a duplicate edge, both dropped); graph 2: path 4-5-6-7; graph 3: edge 8-9
    und = [(1, 2), (2, 3), (1, 3), (4, 5), (5, 6), (6, 7), (8, 9)]
    lines = ["%d, %d" % (a, b) for a, b in und] + ["%d, %d" % (b, a) for a, b in und] + ["2, 2", "1, 2"]
    open(os.path.join(raw, name + "_A.txt"), "w").write("\n".join(lines) + "\n")
    open(os.path.join(raw, name + "_graph_indicator.txt"), "w").write("\n".join(map(str, [1, 1, 1, 2, 2, 2, 2, 3, 3])) + "\n")
    open(os.path.join(raw, name + "_graph_labels.txt"), "w").write("1\n-1\n1\n")
    open(os.path.join(raw, name + "_node_labels.txt"), "w").write("\n".join(map(str, [3, 4, 3, 5, 5, 3, 4, 4, 3])) + "\n")
    return raw


def test_read_tu_data_and_feature_expansion(tmp_path):
    raw = _write(str(tmp_path))
    gs = tu.read_tu_data(raw, "TOY")
    assert [g.num_nodes for g in gs] == [3, 4, 2]
    assert torch.equal(torch.cat([g.y for g in gs]), torch.tensor([1, 0, 1]))        # {-1, 1} -> {0, 1}
    # triangle: 6 directed edges, sorted by (source, target), no self loop / duplicate
    assert gs[0].edge_index.tolist() == [[0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1]]
    assert gs[1].edge_index.tolist() == [[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2]]     # rebased to the graph's own nodes
    # node labels 3,4,5 -> one-hot over 3 classes
    assert gs[0].x.tolist() == [[1, 0, 0], [0, 1, 0], [1, 0, 0]]
    ds = tu.get_dataset("TOY", "deg+odeg4", root=str(tmp_path))
    assert ds.num_features == 3 + 1 + 5 and ds.num_classes == 2 and len(ds) == 3
    x = ds[1].x                                                                       # path: degrees 1, 2, 2, 1
    assert x[:, 3].tolist() == [1, 2, 2, 1]
    assert x[:, 4:].argmax(1).tolist() == [1, 2, 2, 1]
    capped = tu.expand_features(gs[0], degree=False, onehot_maxdeg=1)                 # triangle degree 2 capped at 1
    assert capped.x.shape == (3, 3 + 2) and capped.x[:, 4].tolist() == [1, 1, 1]
    with pytest.raises(NotImplementedError):
        tu.parse_feat_str("deg+ak3+reall")
    with pytest.raises(FileNotFoundError):
        tu.get_dataset("MUTAG", root=str(tmp_path))
    sub = ds[torch.tensor([2, 0])]
    assert len(sub) == 2 and sub[0].num_nodes == 2


def test_k_fold_is_stratified_and_partitions():
    from cal_amd.data import Data
    y = torch.tensor([0, 1] * 15)
    ds = tu.TUDataset([Data(x=torch.ones(2, 1), edge_index=torch.zeros(2, 0, dtype=torch.long), y=y[i].view(1)) for i in range(30)])
    for select in ("test_max", "val_max"):
        train, test, val = tu.k_fold(ds, 5, select)
        assert sorted(torch.cat(test).tolist()) == list(range(30))                    # the test folds partition the data
        for i in range(5):
            assert set(train[i].tolist()).isdisjoint(test[i].tolist()) and set(train[i].tolist()).isdisjoint(val[i].tolist())
            assert int(y[test[i]].sum()) == 3                                         # 3 of each class per fold
            if select == "test_max":
                assert torch.equal(val[i], test[i]) and len(train[i]) == 24
            else:
                assert torch.equal(val[i], test[i - 1]) and len(train[i]) == 18
