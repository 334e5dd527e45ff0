"""The reference-named module surface (SURVEY.md 8b "module names a drop-in must provide"): with `dropin/` first on
sys.path every symbol the reference's entry scripts import resolves -- main_syn.py:1-7 (`train.train_baseline_syn`,
`train_causal.train_causal_syn`, `opts.setup_seed`, `opts`, `utils`), main_real.py:1-3 (`datasets.get_dataset`,
`train_causal.train_causal_real`, `opts`), opts.py:2 (`model.*`), model.py:7 (`gcn_conv.GCNConv`),
train_causal.py:8 (`utils.k_fold`, `utils.num_graphs`) -- and the host-side pieces behave like the reference's.
Runs in a child interpreter so the generic module names (`utils`, `datasets`, `train`) never leak into this process."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import sys, os, argparse
    root = sys.argv[1]
    sys.path[:0] = [os.path.join(root, "dropin"), root]
    # main_syn.py:1-7
    from train import train_baseline_syn
    from train_causal import train_causal_syn
    from opts import setup_seed
    import opts, utils
    # main_real.py:1-3
    from datasets import get_dataset
    from train_causal import train_causal_real
    # opts.py:2, model.py:7, train_causal.py:8
    from model import CausalGCN, CausalGIN, CausalGAT, GINNet, GCNNet, GATNet
    from gcn_conv import GCNConv
    from utils import k_fold, num_graphs
    for mod in (opts, utils):
        assert os.path.dirname(mod.__file__) == os.path.join(root, "dropin"), mod.__file__
    import datasets, train, train_causal, model, gcn_conv
    for mod in (datasets, train, train_causal, model, gcn_conv):
        assert os.path.dirname(mod.__file__) == os.path.join(root, "dropin"), mod.__file__
    for name in ("parse_args", "get_model", "setup_seed", "create_n_filter_triples", "print_args"):
        assert callable(getattr(opts, name)), name
    for name in ("graph_dataset_generate", "dataset_bias_split", "print_dataset_info", "k_fold", "num_graphs"):
        assert callable(getattr(utils, name)), name
    # the call sequence of main_syn.py:14-29 up to the training call, on a small dataset
    args = opts.parse_args(["--model", "CausalGCN", "--bias", "0.9", "--data_num", "40", "--node_num", "7", "--batch_size", "32"])
    assert args.bias == 0.9 and args.with_random is True and args.eval_random is False and args.layers == 3 and args.hidden == 128
    assert args.c == 0.5 and args.o == 1.0 and args.co == 0.5 and args.node_num == 7 and args.feature_dim == -1
    dataset = utils.graph_dataset_generate(args, None)
    train_set, val_set, test_set, the = utils.dataset_bias_split(dataset, args, bias=args.bias, split=[7, 1, 2], total=args.data_num * 4)
    groups = utils.print_dataset_info(train_set, val_set, test_set, the, log=lambda s: None)
    # utils.py:130-146 incl. its truncation: 28 per class -> int(28 * 0.9) + int(28 * 0.1) = 25 + 2; val 3 + 0; test 4 + 4
    assert (len(train_set), len(val_set), len(test_set)) == (108, 12, 32)
    tree, ba = groups["Train"]
    assert tree[0] == int(28 * 0.9) and ba[0] == int(28 * (1 - 0.9)) and tree[1] == int(28 * (1 - 0.9)) and ba[1] == int(28 * 0.9)
    model_func = opts.get_model(args)
    m = model_func(10, args.num_classes)
    assert isinstance(m, CausalGCN) and m.num_classes == 4 and sum(p.numel() for p in m.parameters()) == 138660
    try:
        train_baseline_syn(train_set, val_set, test_set, model_func=model_func, args=args)
        raise SystemExit("baseline training should be out of scope")
    except NotImplementedError:
        pass
    assert opts.create_n_filter_triples(["MUTAG"])[0] == ("MUTAG", "deg+odeg100", "ResGCN")
    # opts.py:130-136: DD and the three REDDIT sets (exact names) take the narrower one-hot degree, DD also ak3 -> ak1
    assert opts.create_n_filter_triples(["DD"], feat_strs=["deg+odeg100+ak3"])[0] == ("DD", "deg+odeg10+ak1", "ResGCN")
    assert opts.create_n_filter_triples(["REDDIT-BINARY"])[0][1] == "deg+odeg10"
    assert opts.create_n_filter_triples(["REDDIT-FOO"])[0][1] == "deg+odeg100"
    try:
        get_dataset("MUTAG", sparse=True, feat_str="deg+odeg100", root="/nonexistent")
        raise SystemExit("expected FileNotFoundError")
    except FileNotFoundError:
        pass
    print("DROPIN_OK")
""")


def test_reference_entry_script_imports_resolve_from_dropin():
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
