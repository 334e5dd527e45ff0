"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/cal_hip.h declares; modules keep the reference's
state-dict surface; GPU-only pieces fail loudly for CPU data."""
import argparse
import subprocess

import pytest
import torch

from cal_amd import _lib


def _args(**kw):
    d = dict(layers=3, hidden=128, with_random=True, without_node_attention=False,
             without_edge_attention=False, fc_num="222", cat_or_add="add")
    d.update(kw)
    return argparse.Namespace(**d)


def test_header_parses_and_library_exports_every_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 20
    for must in ("cal_plan_build", "cal_spmm_fwd", "cal_gcn_norm_bwd", "cal_edge_att_fwd",
                 "cal_node_att_split_bwd", "cal_add_pool_fwd", "cal_gat_fwd", "cal_gat_bwd"):
        assert must in protos
    h = _lib.lib()          # raises AttributeError if a declared symbol is not exported
    assert h.cal_version() >= 100
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(protos) <= exported
    # nothing but the cal_* API is exported from our sources
    assert all(s.startswith("cal_") or s.startswith("_") for s in exported)


def test_host_library_exports_the_operator_level_symbols():
    """libcalhost.so (plain C++, no HIP): the same names and prototypes as libcalhip.so for every operator-level entry."""
    h = _lib.host_lib()
    assert h.cal_version() >= 100
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.HOST_LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(_lib.HOST_SYMBOLS) <= exported and set(_lib.HOST_SYMBOLS) <= set(_lib.parse_header())
    assert all(s.startswith("cal_") or s.startswith("_") for s in exported)
    deps = subprocess.run(["ldd", _lib.HOST_LIB_PATH], capture_output=True, text=True).stdout
    assert "amdhip" not in deps and "torch" not in deps          # no HIP runtime, no torch: loads on any host


def test_gpu_only_pieces_refuse_cpu_tensors_and_devices_do_not_mix():
    """The step engine, the device-resident dataset and everything else without a host twin raise for CPU data; the
    host library is only ever chosen for CPU-resident tensors, never as a substitute for a missing libcalhip.so."""
    from cal_amd.model import CausalGCN
    from cal_amd.engine import StepEngine
    from cal_amd.device_data import DeviceDataset
    from tests.helpers import ref_graphs
    m = CausalGCN(10, 4, _args(layers=1, hidden=16))
    with pytest.raises(_lib.CalError):
        StepEngine(m)
    with pytest.raises(_lib.CalError):
        DeviceDataset(ref_graphs([0, 1]), device="cpu")
    with pytest.raises(_lib.CalError):
        _lib.call("cal_randperm", None, 4, 1, None, None, host=True)


@pytest.mark.parametrize("name", ["CausalGCN", "CausalGAT"])
def test_state_dict_surface_matches_reference_names(name):
    from cal_amd import model as M
    from oracle import cal_oracle as O
    m = getattr(M, name)(10, 4, _args())
    sd = m.state_dict()
    ref = O.init_state(name, 10, 4, hidden=128, layers=3, heads=4)
    assert set(sd) == set(ref)
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    assert m.num_classes == 4
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == (138660 if name == "CausalGCN" else n_params)
    # BN init (model.py:80-83) and glorot range (gcn_conv.py:40)
    assert torch.all(m.bn_feat.weight == 1) and torch.allclose(m.bn_feat.bias, torch.full((10,), 1e-4))
    a = (6.0 / (128 + 128)) ** 0.5
    assert m.context_convs.weight.abs().max() <= a and torch.all(m.context_convs.bias == 0)


def test_cat_variant_shapes():
    from cal_amd.model import CausalGCN
    m = CausalGCN(10, 4, _args(cat_or_add="cat"))
    assert m.fc1_co.weight.shape == (128, 256) and m.fc1_bn_co.weight.shape == (256,)


def test_gat_state_dict_conversion_from_pyg2_names():
    from cal_amd.gat_conv import GATConv
    g = GATConv(8, 2, heads=4)
    sd = {"lin_src.weight": torch.randn(8, 8), "att_src": torch.randn(1, 4, 2),
          "att_dst": torch.randn(1, 4, 2), "bias": torch.randn(8)}
    want_w = sd["lin_src.weight"].t().clone()
    want_att = torch.cat([sd["att_dst"], sd["att_src"]], -1)
    g.load_state_dict(dict(sd))
    assert torch.equal(g.weight, want_w) and torch.equal(g.att, want_att)


def test_intervention_gating_matches_reference():
    import random
    from cal_amd.model import CausalGAT, CausalGCN
    m = CausalGCN(10, 4, _args(with_random=False, layers=1, hidden=8))
    assert m.intervention_index(5, True).tolist() == [0, 1, 2, 3, 4]      # model.py:149
    g = CausalGAT(10, 4, _args(with_random=False, layers=1, hidden=8))
    random.seed(3)
    p = g.intervention_index(5, True).tolist()                              # model.py:435
    random.seed(3)
    l = list(range(5)); random.shuffle(l)
    assert p == l
    assert g.intervention_index(5, False).tolist() == [0, 1, 2, 3, 4]
