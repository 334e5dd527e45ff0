"""Drop-in for the reference's train.py (`from train import train_baseline_syn`, main_syn.py:1).  The baseline nets
(GCNNet / GINNet / GATNet, `--model GCN|GIN|GAT`) are outside the accelerated hot path: the name
resolves so main_syn.py imports, and says so when called."""


def train_baseline_syn(train_set, val_set, test_set, model_func=None, args=None):
    raise NotImplementedError("train_baseline_syn trains the non-causal baseline nets, which are outside the accelerated "
                              "hot path (SURVEY.md section 8); run --model CausalGCN | CausalGAT | CausalGIN")
