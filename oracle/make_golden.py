"""Generate tests/golden/*.npz -- run HERE only (needs /root/reference).

* ``spmotif_ref_graphs.npz``: graphs produced by importing the three reference
  modules that run in this container (gengraph.py, synthetic_structsim.py,
  featgen.py) with fixed seeds and converting them the way PyG's
  ``from_networkx`` does (utils.py:38-57).  These are *reference-generated
  inputs* (data, not source).
* ``causal_{gcn,gat}_batch8.npz``: outputs of the restated oracle
  (oracle/cal_oracle.py) on one fixed 8-graph batch with a fixed state dict and
  a fixed intervention permutation: logits, losses, a few gradients, and the
  parameters after one Adam step.  These are regression anchors for the
  restatement (the reference itself cannot run here: PARITY UNPINNED).

Usage: python oracle/make_golden.py
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def ref_graphs():
    sys.path.insert(0, "/root/reference")
    import featgen      # noqa: E402  (reference module, imported here only)
    import gengraph     # noqa: E402
    from cal_amd.data import from_networkx

    np.random.seed(666)
    random.seed(666)
    out = {}
    meta = []
    gid = 0
    for node_num in (7, 15):
        settings = {"ba": {"width_basis": node_num ** 2, "m": 2},
                    "tree": {"width_basis": 2, "m": node_num}}
        for context in ("tree", "ba"):
            for label, shape in enumerate(["house", "cycle", "grid", "diamond"]):
                reps = 3 if node_num == 7 else 1
                for _ in range(reps):
                    feature = featgen.ConstFeatureGen(None, max_degree=10)
                    G, _ = gengraph.generate_graph(
                        basis_type=context, shape=shape, nb_shapes=1,
                        width_basis=settings[context]["width_basis"],
                        feature_generator=feature, m=settings[context]["m"],
                        random_edges=0.1)
                    d = from_networkx(G)
                    out[f"g{gid}_edge_index"] = d.edge_index.numpy().astype(np.int64)
                    out[f"g{gid}_feat"] = d.feat.numpy().astype(np.float32)
                    out[f"g{gid}_y"] = np.array([label], dtype=np.int64)
                    meta.append((gid, node_num, context, shape))
                    gid += 1
    out["meta"] = np.array([f"{g},{n},{c},{s}" for g, n, c, s in meta])
    np.savez_compressed(os.path.join(GOLD, "spmotif_ref_graphs.npz"), **out)
    print("wrote", gid, "reference-generated graphs")
    return out, meta


def load_batch(out, ids):
    from cal_amd.data import Batch, Data
    ds = [Data(feat=torch.from_numpy(out[f"g{i}_feat"]),
               edge_index=torch.from_numpy(out[f"g{i}_edge_index"]),
               y=torch.from_numpy(out[f"g{i}_y"])) for i in ids]
    return Batch.from_data_list(ds)


def oracle_fixture(model, out, ids, fname):
    from oracle import cal_oracle as O
    b = load_batch(out, ids)
    torch.manual_seed(1234)
    sd = O.init_state(model, 10, 4, hidden=32, layers=2, heads=4)
    # make bias / BN affine non-trivial so the fixture exercises them
    g = torch.Generator().manual_seed(99)
    for k in list(sd):
        if k.endswith(".bias") or (("bn" in k) and k.endswith(".weight")):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    sd0 = {k: v.clone() for k, v in sd.items()}
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4])
    fx = {f"sd.{k}": v.numpy() for k, v in sd0.items()}
    fx["ids"] = np.array(ids)
    fx["perm"] = perm.numpy()
    # eval-mode logits
    sd_eval = {k: v.clone() for k, v in sd0.items()}
    ev = O.causal_forward(model, sd_eval, b.feat, b.edge_index, b.batch, perm=perm,
                          training=False, layers=2, heads=4)
    for n, t in zip(("c", "o", "co"), ev):
        fx[f"eval_logits_{n}"] = t.numpy()
    # one training step (GAT: attention dropout disabled so the step is deterministic)
    tr = O.CpuTrainer(model, {k: v.clone() for k, v in sd0.items()}, 4, lr=1e-3,
                      layers=2, heads=4, gat_dropout=0.0)
    loss, lc, lo, lco, logits = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
    for n, t in zip(("c", "o", "co"), logits):
        fx[f"train_logits_{n}"] = t.detach().numpy()
    fx["loss"] = np.array([loss.item(), lc.item(), lo.item(), lco.item()], dtype=np.float64)
    for k in tr.names:
        gk = tr.sd[k].grad
        fx[f"grad.{k}"] = (np.zeros(0, np.float32) if gk is None else gk.numpy())
        fx[f"post.{k}"] = tr.sd[k].detach().numpy()
    for k in tr.sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            fx[f"post.{k}"] = tr.sd[k].numpy()
    np.savez_compressed(os.path.join(GOLD, fname), **fx)
    print("wrote", fname, "loss", fx["loss"])


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    out, meta = ref_graphs()
    ids = [0, 4, 7, 10, 13, 16, 19, 22]     # node_num=7: mixed tree/ba, all 4 shapes
    oracle_fixture("CausalGCN", out, ids, "causal_gcn_batch8.npz")
    oracle_fixture("CausalGAT", out, ids, "causal_gat_batch8.npz")
