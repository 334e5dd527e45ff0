import argparse, sys, os, torch, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import spmotif, model as M
from cal_amd.data import Batch
from cal_amd.trainer import CausalTrainer
args = argparse.Namespace(layers=3, hidden=128, with_random=True, without_node_attention=False,
                          without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
gs = spmotif.train_mix(8 * 128, seed=666)
mode = sys.argv[1]
torch.manual_seed(0)
if mode == "each":
    i = int(sys.argv[2])
    b = Batch.from_data_list(gs[i*128:(i+1)*128]).to("cuda")
    m = M.CausalGCN(10, 4, args).cuda().train()
    tr = CausalTrainer(m, args, use_graph=True)
    tr.prepare(b)
    for _ in range(20): st = tr.step(b)
    torch.cuda.synchronize(); print("batch", i, "ok", st.tolist()[:2], b.batch.numel(), b.edge_index.shape)
elif mode == "dup":
    n = int(sys.argv[2])
    bs = [Batch.from_data_list(gs[:128]).to("cuda") for _ in range(n)]
    m = M.CausalGCN(10, 4, args).cuda().train()
    tr = CausalTrainer(m, args, use_graph=True)
    for b in bs: tr.prepare(b)
    torch.cuda.synchronize(); print("prepared", n)
    for i in range(40): st = tr.step(bs[i % n])
    torch.cuda.synchronize(); print("dup", n, "ok", st.tolist()[:2])
elif mode == "dup2":
    n = int(sys.argv[2]); rebuild = sys.argv[3] == "1"
    bs = [Batch.from_data_list(gs[:128]).to("cuda") for _ in range(n)]
    m = M.CausalGCN(10, 4, args).cuda().train()
    tr = CausalTrainer(m, args, use_graph=True, rebuild_plan=rebuild)
    for k, b in enumerate(bs):
        tr.prepare(b); torch.cuda.synchronize(); print("prepared", k, flush=True)
    for i in range(40):
        st = tr.step(bs[i % n]); torch.cuda.synchronize(); print("step", i, flush=True)
    print("dup2", n, rebuild, "ok", st.tolist()[:2])
