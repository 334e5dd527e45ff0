import argparse, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import spmotif, model as M
from cal_amd.data import Batch
from cal_amd.trainer import CausalTrainer
from cal_amd.train_causal import causal_loss
args = argparse.Namespace(layers=3, hidden=128, with_random=True, without_node_attention=False,
                          without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
gs = spmotif.train_mix(128, seed=1)
b = Batch.from_data_list(gs).to("cuda")
stage = sys.argv[1]
torch.manual_seed(0)
m = M.CausalGCN(10, 4, args).cuda().train()
if stage == "fwd":
    perm = torch.arange(128, device="cuda")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            b._plan = None
            with torch.no_grad(): out = m(b, perm=perm)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        b._plan = None
        with torch.no_grad(): out = m(b, perm=perm)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); print("fwd graph ok", out[0][0])
elif stage == "fwdbwd":
    tr = CausalTrainer(m, args, use_graph=True)
    cap = tr._capture(b)
    for _ in range(5): cap.graph.replay()
    torch.cuda.synchronize(); print("fwdbwd graph ok", cap.stats)
elif stage == "opt":
    tr = CausalTrainer(m, args, use_graph=True)
    tr._fwd_bwd(b, torch.arange(128, device="cuda"), tr.stats)
    tr._build_opt_graph()
    for _ in range(5): tr._opt_graph.replay()
    torch.cuda.synchronize(); print("opt graph ok")
elif stage == "full":
    tr = CausalTrainer(m, args, use_graph=True)
    tr.prepare(b)
    torch.cuda.synchronize(); print("prepared")
    for i in range(10):
        st = tr.step(b)
        torch.cuda.synchronize(); print(i, st.tolist())
