"""Drop-in for the reference's utils.py (imported as `utils` by main_syn.py:7 and `from utils import k_fold, num_graphs`
by train_causal.py:8): the symbols those call sites name, served by cal_amd's own implementations.

* ``num_graphs``               -- utils.py:12-16
* ``k_fold``                   -- utils.py:18-36 (cal_amd.tu.k_fold)
* ``graph_dataset_generate``   -- utils.py:59-89: {context: {shape: [Data] * data_num}}, saved to <save_path>/syn_dataset.pt;
                                  cal_amd.spmotif restates the generator distribution-wise (own RNG stream, no networkx)
* ``dataset_bias_split``       -- utils.py:123-159: returns (train, val, test, the) with the reference's per-class counts
* ``print_dataset_info``       -- utils.py:167-173: per-split (tree, ba) x class counts
"""
import os

import torch

from cal_amd import spmotif
from cal_amd.train_causal import num_graphs  # noqa: F401
from cal_amd.tu import k_fold  # noqa: F401

CLASS_LIST = list(spmotif.CLASS_LIST)


def graph_dataset_generate(args, save_path):
    dataset = spmotif.generate_dataset(args.data_num, node_num=args.node_num, noise=args.noise,
                                       max_degree=args.max_degree, seed=getattr(args, "seed", 666))
    if save_path:
        os.makedirs(save_path, exist_ok=True)
        path = os.path.join(save_path, "syn_dataset.pt")
        torch.save(dataset, path)
        print("save at:{}".format(path))
    return dataset


def dataset_bias_split(dataset, args, bias=None, split=None, total=20000):
    split = (7, 1, 2) if split is None else split
    train, val, test = spmotif.dataset_bias_split(dataset, bias, split=split, total=total,
                                                  num_classes=args.num_classes, shuffle_seed=getattr(args, "seed", 666))
    # `the`: the edge-count threshold separating tree- from BA-context graphs (mean of the first graph of every cell)
    firsts = [dataset[c][s][0] for s in CLASS_LIST for c in ("tree", "ba")]
    the = float(sum(g.num_edges for g in firsts)) / len(firsts)
    return train, val, test, the


def _group_counts(graphs, the):
    tree, ba = [0] * len(CLASS_LIST), [0] * len(CLASS_LIST)
    for g in graphs:
        (ba if g.num_edges > the else tree)[int(g.y.item())] += 1
    return tree, ba


def print_dataset_info(train_set, val_set, test_set, the, log=print):
    out = {}
    for title, graphs in (("Train", train_set), ("Val", val_set), ("Test", test_set)):
        tree, ba = _group_counts(graphs, the)
        out[title] = (tree, ba)
        log("%-5s total:%d | tree %s | ba %s | tree share %s" % (
            title, sum(tree) + sum(ba), dict(zip(CLASS_LIST, tree)), dict(zip(CLASS_LIST, ba)),
            ["%.1f%%" % (100.0 * t / max(1, t + b)) for t, b in zip(tree, ba)]))
    return out
