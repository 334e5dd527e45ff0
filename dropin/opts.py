"""Drop-in for the reference's opts.py (`import opts`, `from opts import setup_seed`: main_syn.py:3,5, main_real.py:3):
``parse_args`` with the reference's flag names and defaults (opts.py:8-69), ``setup_seed`` (opts.py:77-83),
``get_model`` (opts.py:85-119) and ``create_n_filter_triples`` (opts.py:121-...) for the one feature string CAL uses."""
import argparse
import random

import numpy as np
import torch

from model import CausalGAT, CausalGCN, CausalGIN, GATNet, GCNNet, GINNet


def _str2bool(x):
    return x.lower() == "true"


# (flag, type, default) -- opts.py:13-66
_FLAGS = [
    ("step_size", float, 0.001), ("min_lr", float, 1e-6), ("pretrain", int, 30), ("data_num", int, 2000),
    ("node_num", int, 15), ("max_degree", int, 10), ("feature_dim", int, -1), ("noise", float, 0.1),
    ("num_classes", int, 4), ("shape_num", int, 1), ("bias", float, 0.5), ("penalty_weight", float, 0.1),
    ("train_type", str, "base"), ("epochs", int, 100), ("batch_size", int, 128), ("the", int, 0),
    ("with_random", _str2bool, True), ("eval_random", _str2bool, False), ("normalize", _str2bool, False),
    ("save_model", _str2bool, False), ("inference", _str2bool, False), ("without_node_attention", _str2bool, False),
    ("without_edge_attention", _str2bool, False), ("k", int, 3), ("layers", int, 3), ("c", float, 0.5), ("o", float, 1.0),
    ("co", float, 0.5), ("harf_hidden", float, 0.5), ("cat_or_add", str, "add"), ("num_layers", int, 3), ("folds", int, 10),
    ("fc_num", str, "222"), ("data_root", str, "data"), ("save_dir", str, "debug"), ("dataset", str, "NCI1"),
    ("epoch_select", str, "test_max"), ("model", str, "GCN"), ("hidden", int, 128), ("seed", int, 666), ("lr", float, 0.001),
    ("lr_decay_factor", float, 0.5), ("lr_decay_step_size", int, 500), ("weight_decay", float, 0), ("global_pool", str, "sum"),
]


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    for name, typ, default in _FLAGS:
        parser.add_argument("--" + name, type=typ, default=default)
    args = parser.parse_args(argv)
    print_args(args)
    setup_seed(args.seed)
    return args


def print_args(args, str_num=80):
    for arg, val in vars(args).items():
        print(arg + "." * (str_num - len(arg) - len(str(val))) + str(val))
    print()


def setup_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


_MODELS = {"GCN": GCNNet, "GIN": GINNet, "GAT": GATNet, "CausalGCN": CausalGCN, "CausalGIN": CausalGIN, "CausalGAT": CausalGAT}


def get_model(args):
    cls = _MODELS.get(args.model)
    assert cls is not None, args.model
    if args.model.startswith("Causal"):
        return lambda num_features, num_classes: cls(num_features, num_classes, args)
    return lambda num_features, num_classes: cls(num_features, num_classes, args.hidden)


_NARROW_DEGREE = {"REDDIT-BINARY", "REDDIT-MULTI-5K", "REDDIT-MULTI-12K", "DD"}      # opts.py:130-136


def create_n_filter_triples(datasets, feat_strs=("deg+odeg100",), nets=("ResGCN",), reddit_odeg10=True, dd_odeg10_ak1=True, **_):
    """(dataset, feature string, net) triples of opts.py:121-139: the three REDDIT datasets and DD take the narrower one-hot
    degree ('odeg10'), DD also 'ak1' for 'ak3'."""
    out = []
    for d in datasets:
        for f in feat_strs:
            narrow = (reddit_odeg10 and d in _NARROW_DEGREE and d != "DD") or (dd_odeg10_ak1 and d == "DD")
            if narrow:
                f = f.replace("odeg100", "odeg10")
            if dd_odeg10_ak1 and d == "DD":
                f = f.replace("ak3", "ak1")
            for n in nets:
                out.append((d, f, n))
    return out
