"""GraphPlan: per-mini-batch device structures consumed by the HIP kernels.

Built once per batch from ``edge_index`` / ``batch`` (what the reference hands
to every layer, model.py:87-89) by ``cal_plan_build`` / ``cal_graph_ptr``:

* CSR by destination (``rowptr_dst, nbr_dst, eid_dst``) -- forward aggregation,
* CSR by source (``rowptr_src, nbr_src, eid_src``) -- degree, transpose gather,
* ``row32 / col32`` -- int32 copy of ``edge_index`` for per-edge kernels,
* ``gptr`` -- node offsets of each graph (add-pool segments),
* the unweighted symmetric normalisation (``dis``, ``norm_e``), computed once and
  shared by every backbone layer (the reference recomputes it per layer,
  gcn_conv.py:79-89).

All int32 arrays live in one allocation.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib


def _stream() -> int:
    # the raw handle of torch's current stream on the current device: torch.cuda.current_stream() builds a Stream object through
    # four layers of device-index helpers (12 us a call, six calls per training step on the nn.Module surface)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _c(name: str, *args):
    """Call entry point ``name`` with tensors passed AS TENSORS: they become their data pointers, the library is
    chosen by where they live (CUDA -> libcalhip.so on torch's current stream, CPU -> libcalhost.so), and tensors of
    one call may not straddle devices."""
    dev, conv = None, []
    for a in args:
        if torch.is_tensor(a):
            d = a.device.type
            if dev is None:
                dev = d
            elif d != dev:
                raise _lib.CalError("%s: tensors on both %s and %s in one call" % (name, dev, d))
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    host = dev == "cpu"
    _lib.call(name, *conv, None if host else _stream(), host=host)


def _q(ref: torch.Tensor, name: str, *args) -> int:
    """Size query against the library that will serve ``ref``'s device."""
    return _lib.query(name, *args, host=not ref.is_cuda)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _al4(n: int) -> int:
    return (n + 3) // 4 * 4


class GraphPlan:
    def __init__(self, edge_index: torch.Tensor, num_nodes: int,
                 batch: Optional[torch.Tensor] = None, num_graphs: Optional[int] = None,
                 validate: bool = False):
        if edge_index.dtype != torch.long or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("edge_index must be int64 [2, E]")
        self.device = edge_index.device
        ei = edge_index.contiguous()
        self.edge_index = ei
        E, N = int(ei.size(1)), int(num_nodes)
        self.E, self.N = E, N
        B = 0
        if batch is not None:
            B = int(num_graphs) if num_graphs is not None else (int(batch.max().item()) + 1 if N else 0)
        self.B = B
        sizes = dict(rowptr_dst=N + 1, nbr_dst=E, eid_dst=E, rowptr_src=N + 1, nbr_src=E, eid_src=E,
                     row32=E, col32=E, work=4 * (N + 1) + 4 * E, status=1, gptr=B + 1)
        total = sum(_al4(max(v, 1)) for v in sizes.values())
        self._ints = torch.empty(total, dtype=torch.int32, device=self.device)
        off = 0
        for k, v in sizes.items():
            setattr(self, k, self._ints[off:off + v])
            off += _al4(max(v, 1))
        _c("cal_plan_build", ei, E, N, self.rowptr_dst, self.nbr_dst, self.eid_dst,
                  self.rowptr_src, self.nbr_src, self.eid_src, self.row32, self.col32,
                  self.work, self.status)
        self.batch = None
        if batch is not None:
            if batch.dtype != torch.long or batch.numel() != N:
                raise ValueError("batch must be int64 [N]")
            self.batch = batch.contiguous()
            _c("cal_graph_ptr", self.batch, N, B, self.gptr, self.status)
        self._unit: Dict[float, Tuple[torch.Tensor, torch.Tensor]] = {}
        if validate:
            self.check()

    def check(self):
        """Synchronising validation of the inputs (index range, sorted batch)."""
        st = int(self.status.item())
        if st & 1:
            raise IndexError("edge_index has entries outside [0, num_nodes)")
        if st & 2:
            raise ValueError("batch vector is not sorted / has ids outside [0, num_graphs)")

    def unit_norm(self, loop_w: float = 1.0):
        """(dis, norm_e) for edge_weight = None (gcn_conv.py:45-48 default of ones)."""
        key = float(loop_w)
        if key not in self._unit:
            dis = torch.empty(max(self.N, 1), dtype=torch.float32, device=self.device)
            norm = torch.empty(max(self.E, 1), dtype=torch.float32, device=self.device)
            _c("cal_gcn_norm_fwd", self.rowptr_src, self.eid_src, self.row32, self.col32,
                      None, key, self.N, self.E, dis, norm)
            self._unit[key] = (dis, norm)
        return self._unit[key]

    def ones(self):
        """(ones[N], ones[E]): unit coefficients for plain-sum aggregation (GINConv)."""
        if getattr(self, "_ones", None) is None:
            self._ones = (torch.ones(max(self.N, 1), dtype=torch.float32, device=self.device),
                          torch.ones(max(self.E, 1), dtype=torch.float32, device=self.device))
        return self._ones

    def pool_splits(self) -> int:
        if self.B == 0:
            return 1
        avg = self.N / max(self.B, 1)
        return max(1, min(64, int(avg // 256)))


def plan_of(data) -> GraphPlan:
    """GraphPlan cached on a batch object (cal_amd.data.Batch or a PyG Batch)."""
    plan = getattr(data, "_plan", None)
    ei = data.edge_index
    if plan is not None and plan.edge_index.data_ptr() == ei.data_ptr() and plan.E == ei.size(1) \
            and plan.device == ei.device:
        return plan
    x = data.x if getattr(data, "x", None) is not None else data.feat
    batch = getattr(data, "batch", None)
    n = int(x.size(0))
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long, device=ei.device)
        ng = 1
    else:
        ng = getattr(data, "num_graphs", None)
    plan = GraphPlan(ei, n, batch, ng)
    try:
        data._plan = plan
    except Exception:
        pass
    return plan
