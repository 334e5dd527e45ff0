"""Clean-room mini data layer: ``Data`` / ``Batch`` / ``DataLoader`` /
``from_networkx``.

The reference gets these from PyTorch-Geometric (``train_causal.py:4,13-15``,
``utils.py:4,55``), which is not vendored.  Only the surface the hot path
touches is provided (SURVEY.md section 8b "batch protocol"):

* ``data.x`` **or** ``data.feat`` ``[N, F]`` fp32 -- ``data.x`` is ``None`` when
  absent (``model.py:87``), ``data.edge_index`` ``[2, E]`` int64, ``data.batch``
  ``[N]`` int64 (sorted), ``data.y``, ``data.num_graphs``, ``data.to(device)``.
* ``DataLoader(dataset, batch_size, shuffle)`` with ``len(loader.dataset)``
  (``train_causal.py:194``).

A ``Batch`` additionally carries ``ptr`` ([B+1] node offsets) and lazily caches
the device ``GraphPlan`` (CSR structures) the HIP kernels consume.
"""
from __future__ import annotations

import operator
import random as _random
from typing import Iterable, List, Optional, Sequence

import torch

_NODE_KEYS = ("x", "feat")


# Small-graph packing (cal_engine_set_tiles): the step engine's per-graph kernels give every workgroup a 64-row tile; a batch
# whose graphs average <= 40 nodes (NCI1, MUTAG, ...) is grouped into tiles of CONSECUTIVE graphs -- each a block-diagonal
# graph of its own -- with these bounds (engine_gconv.hpp: 64-node variant, GC_TILE_GRAPHS)
TILE_NODES, TILE_EDGES, TILE_GRAPHS = 64, 1024, 8


def pack_tiles(node_sizes, edge_sizes):
    """Greedy grouping of consecutive graphs into tiles; returns the list of first-graph indices [T + 1] or None when
    packing does not apply (large graphs) or gains nothing."""
    nl = node_sizes.tolist() if hasattr(node_sizes, "tolist") else list(node_sizes)
    el = edge_sizes.tolist() if hasattr(edge_sizes, "tolist") else list(edge_sizes)
    B = len(nl)
    if B < 2 or sum(nl) > 40 * B:
        return None
    first, cn, ce, cg = [0], 0, 0, 0
    for i in range(B):
        a, b = nl[i], el[i]
        if a > TILE_NODES or b > TILE_EDGES:
            return None
        if cg and (cn + a > TILE_NODES or ce + b > TILE_EDGES or cg >= TILE_GRAPHS):
            first.append(i)
            cn = ce = cg = 0
        cn += a
        ce += b
        cg += 1
    first.append(B)
    return first if len(first) - 1 < B else None


def pack_order(node_sizes, edge_sizes):
    """Order of the graphs of a mini-batch that fills the tiles best (first-fit decreasing, ``cal_pack_order``): returns
    (order [B] positions, first [T + 1]) as Python lists, or None when packing does not apply.  A mini-batch is a set --
    the loss is a mean over it and the intervention permutation is random -- so the order inside it is the loader's to
    choose; ``pack_tiles`` of the reordered batch then finds (at least) these tiles."""
    import ctypes
    import numpy as np
    from . import _lib
    n = np.ascontiguousarray(node_sizes, dtype=np.int64)
    e = np.ascontiguousarray(edge_sizes, dtype=np.int64)
    B = int(n.shape[0])
    if B < 2 or int(n.sum()) > 40 * B:
        return None
    order = np.empty(B, dtype=np.int64)
    first = np.empty(B + 1, dtype=np.int64)
    T = _lib.query("cal_pack_order", n.ctypes.data_as(ctypes.c_void_p), e.ctypes.data_as(ctypes.c_void_p), B, TILE_NODES,
                   TILE_EDGES, TILE_GRAPHS, order.ctypes.data_as(ctypes.c_void_p), first.ctypes.data_as(ctypes.c_void_p))
    if T <= 0 or T >= B:
        return None
    return order, first[:T + 1]


class Data:
    def __init__(self, x=None, edge_index=None, y=None, feat=None, **kw):
        self.x = x
        self.feat = feat
        self.edge_index = edge_index
        self.y = y
        self.batch = None
        for k, v in kw.items():
            setattr(self, k, v)

    # -- PyG-style introspection -------------------------------------------
    @property
    def num_nodes(self) -> int:
        for k in _NODE_KEYS:
            v = getattr(self, k, None)
            if v is not None:
                return int(v.size(0))
        nn = getattr(self, "_num_nodes", None)
        if nn is not None:
            return nn
        return int(self.edge_index.max().item()) + 1 if self.edge_index.numel() else 0

    @num_nodes.setter
    def num_nodes(self, v):
        self._num_nodes = int(v)

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.size(1))

    @property
    def num_features(self) -> int:
        v = self.x if self.x is not None else self.feat
        return 0 if v is None else int(v.size(1))

    def _tensor_keys(self):
        return [k for k, v in self.__dict__.items() if torch.is_tensor(v)]

    def to(self, device, non_blocking: bool = False):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        for k, v in self.__dict__.items():
            if torch.is_tensor(v) and v.device != dev:         # (a device-resident loader's batch: nothing to move)
                self.__dict__[k] = v.to(dev, non_blocking=non_blocking)
        if getattr(self, "_plan", None) is not None and self._plan.device != dev:
            self._plan = None
        return self

    def __repr__(self):
        parts = [f"{k}={list(getattr(self, k).shape)}" for k in self._tensor_keys()]
        return f"{self.__class__.__name__}({', '.join(parts)})"


class Batch(Data):
    """Disjoint union of graphs (block-diagonal adjacency)."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ptr = None
        self.num_graphs = 0
        self._plan = None
        # largest node / edge count of a single member graph (host ints, 0 = unknown): lets the native step
        # engine pick its per-graph fused kernels without a device round trip
        self.max_nodes = 0
        self.max_edges = 0
        # layout facts of a collated batch (graph b owns nodes [ptr[b], ptr[b+1]) and the contiguous edge columns
        # [edge_ptr[b], edge_ptr[b+1]); no edge is a self loop): they select the engine's one-kernel CSR build
        self.edge_ptr = None
        self.no_self_loops = False
        # small-graph packing (pack_tiles): first graph / node offset / edge offset of every tile ([T + 1] each) and the
        # largest tile; None when the batch is not packed
        self.tile_ptr = self.tile_node_ptr = self.tile_edge_ptr = None
        self.tile_max_nodes = self.tile_max_edges = 0

    def set_tiles(self, first, node_off, edge_off):
        """Record a packing: ``first`` [T + 1] graph indices, ``node_off`` / ``edge_off`` [B + 1] offsets (host sequences)."""
        tn = [int(node_off[i]) for i in first]
        te = [int(edge_off[i]) for i in first]
        self.tile_ptr = torch.tensor(first, dtype=torch.long)
        self.tile_node_ptr = torch.tensor(tn, dtype=torch.long)
        self.tile_edge_ptr = torch.tensor(te, dtype=torch.long)
        self.tile_max_nodes = max(b - a for a, b in zip(tn[:-1], tn[1:]))
        self.tile_max_edges = max(b - a for a, b in zip(te[:-1], te[1:]))

    @staticmethod
    def from_data_list(data_list: Sequence[Data], pack: bool = False) -> "Batch":
        """``pack``: reorder the graphs so that runs of consecutive small graphs fill the engine's 64-node tiles
        (``pack_order``; the batch is the same set of graphs)."""
        if pack and len(data_list) > 1:
            po = pack_order([d.num_nodes for d in data_list], [int(d.edge_index.size(1)) for d in data_list])
            if po is not None:
                data_list = [data_list[int(i)] for i in po[0]]
        b = Batch()
        xs, feats, eis, ys, bvec, ptr = [], [], [], [], [], [0]
        off = 0
        for i, d in enumerate(data_list):
            n = d.num_nodes
            if d.x is not None:
                xs.append(d.x)
            if d.feat is not None:
                feats.append(d.feat)
            eis.append(d.edge_index + off)
            if d.y is not None:
                ys.append(d.y.view(-1))
            bvec.append(torch.full((n,), i, dtype=torch.long))
            off += n
            ptr.append(off)
        b.x = torch.cat(xs, 0) if xs else None
        b.feat = torch.cat(feats, 0) if feats else None
        b.edge_index = torch.cat(eis, 1) if eis else torch.zeros(2, 0, dtype=torch.long)
        b.y = torch.cat(ys, 0) if ys else None
        b.batch = torch.cat(bvec, 0) if bvec else torch.zeros(0, dtype=torch.long)
        b.ptr = torch.tensor(ptr, dtype=torch.long)
        b.num_graphs = len(data_list)
        b.max_nodes = max((int(d.num_nodes) for d in data_list), default=0)
        b.max_edges = max((int(d.edge_index.size(1)) for d in data_list), default=0)
        eptr = [0]
        for d in data_list:
            eptr.append(eptr[-1] + int(d.edge_index.size(1)))
        b.edge_ptr = torch.tensor(eptr, dtype=torch.long)
        b.no_self_loops = bool(b.edge_index.numel() == 0 or (b.edge_index[0] != b.edge_index[1]).all().item())
        if b.no_self_loops:
            first = pack_tiles([q - p for p, q in zip(ptr[:-1], ptr[1:])], [q - p for p, q in zip(eptr[:-1], eptr[1:])])
            if first is not None:
                b.set_tiles(first, ptr, eptr)
        return b

    @property
    def num_nodes(self) -> int:
        return int(self.batch.size(0))

    def to(self, device, non_blocking: bool = False):
        """A batch assembled by ``_HostConcat.collate`` lives in two staging buffers (one float, one int64: pinned when a GPU is
        present): it moves with TWO copies instead of one per attribute, and its tensors become views of the device copies."""
        st = getattr(self, "_staged", None)
        if st is None or torch.device(device).type == "cpu":
            return super().to(device, non_blocking=non_blocking)
        fbuf, ibuf, views, slot = st
        fd = fbuf.to(device, non_blocking=True)
        idv = ibuf.to(device, non_blocking=True)
        if slot is not None:
            slot.moved(device)
        for name, kind, a, b, shape in views:
            setattr(self, name, (fd if kind == "f" else idv)[a:b].view(shape))
        self._staged = None
        if getattr(self, "_plan", None) is not None and self._plan.device != torch.device(device):
            self._plan = None
        return self


# every attribute Batch.__init__ sets, at its default (the host collate fills instances without calling it)
_BATCH_DEFAULTS = dict(Batch().__dict__)


class _StageSlot:
    """One pinned (float, int64) buffer pair of the host loader's ring.  It is handed to a new batch only when the batch that
    last used it is gone or has been moved to the device; the ring is walked round-robin, so the copy that last read a slot
    was enqueued several batches ago and its event is waited for, not polled (hipEventQuery cost ~0.2 ms per call here)."""

    def __init__(self):
        self.fbuf = self.ibuf = None
        self.owner = None           # weakref to the Batch whose CPU tensors view the buffers
        self.event = None

    def held(self) -> bool:
        """Some tensor still views the staging memory: the batch itself, or a CPU view a caller took from it before moving the
        batch to the device (storage use counts; the owner weak reference alone when torch does not expose them)."""
        o = self.owner() if self.owner is not None else None
        if o is not None and getattr(o, "_staged", None) is not None:
            return True
        try:
            for buf in (self.fbuf, self.ibuf):
                if buf is not None and torch._C._storage_Use_Count(buf.untyped_storage()._cdata) > 2:   # (2 = this tensor + the query)
                    return True
        except Exception:
            pass
        return False

    def moved(self, device):
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(device))


class _HostConcat:
    """Host-resident concatenation of a static list of graphs and a vectorised collate over it: what ``DataLoader`` uses instead
    of ``Batch.from_data_list`` (one ``torch.cat`` per attribute over ~100 Python objects: ~1 ms per mini-batch of 128 SPMotif
    graphs, 4x the GPU step) when its dataset is a plain list of ``Data`` with the attributes the hot path reads
    (``x`` or ``feat``, ``edge_index``, one ``y`` per graph).  Same batches, bit for bit (``tests/test_data.py``)."""

    def __init__(self, graphs):
        import numpy as np
        self.ok = False
        if len(graphs) == 0 or not all(isinstance(d, Data) for d in graphs):
            return
        has_x = [d.x is not None for d in graphs]
        has_f = [d.feat is not None for d in graphs]
        if not ((all(has_x) and not any(has_f)) or (all(has_f) and not any(has_x))):
            return
        self.feat_is_x = all(has_x)
        key = "x" if self.feat_is_x else "feat"
        feats = [getattr(d, key) for d in graphs]
        if any(f.dim() != 2 or f.dtype != torch.float32 or f.size(1) != feats[0].size(1) or f.is_cuda for f in feats):
            return
        if any(d.y is None or d.y.numel() != 1 or d.y.dtype != torch.long for d in graphs):
            return
        if any(d.edge_index.dtype != torch.long or d.edge_index.dim() != 2 or d.edge_index.is_cuda for d in graphs):
            return
        self.X = np.ascontiguousarray(torch.cat(feats, 0).numpy())
        self.EI = np.ascontiguousarray(torch.cat([d.edge_index for d in graphs], 1).numpy())
        self.Y = np.ascontiguousarray(torch.cat([d.y.view(-1) for d in graphs]).numpy())
        self.nsz = np.asarray([f.size(0) for f in feats], dtype=np.int64)
        self.esz = np.asarray([int(d.edge_index.size(1)) for d in graphs], dtype=np.int64)
        self.node_ptr = np.ascontiguousarray(np.concatenate([[0], np.cumsum(self.nsz)]), dtype=np.int64)
        self.edge_ptr = np.ascontiguousarray(np.concatenate([[0], np.cumsum(self.esz)]), dtype=np.int64)
        self.F = int(self.X.shape[1])
        self.no_self_loops = bool(self.EI.shape[1] == 0 or (self.EI[0] != self.EI[1]).all())
        self.pin = torch.cuda.is_available()
        self.ring = [_StageSlot() for _ in range(8)] if self.pin else []
        self.ring_i = 0
        self.ok = True

    def _buffers(self, nf, ni):
        def cap(n):                 # pinned allocations cost ~1.5 ms each: a quarter of headroom, rounded up to a power of two
            c = 1 << 16
            while c < n + n // 4:
                c <<= 1
            return c
        for _ in range(len(self.ring)):
            slot = self.ring[self.ring_i]
            self.ring_i = (self.ring_i + 1) % len(self.ring)
            if slot.held():
                continue
            if slot.event is not None:
                slot.event.synchronize()
                slot.event = None
            if slot.fbuf is None or slot.fbuf.numel() < nf:
                slot.fbuf = torch.empty(cap(nf), dtype=torch.float32).pin_memory()
            if slot.ibuf is None or slot.ibuf.numel() < ni:
                slot.ibuf = torch.empty(cap(ni), dtype=torch.long).pin_memory()
            return slot.fbuf[:nf], slot.ibuf[:ni], slot
        return torch.empty(nf, dtype=torch.float32), torch.empty(ni, dtype=torch.long), None      # every slot held by a live batch: pageable

    def collate(self, idx) -> "Batch":
        import numpy as np
        import weakref
        from . import _lib
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        B = int(idx.shape[0])
        n, e = self.nsz[idx], self.esz[idx]
        noff = np.concatenate([[0], np.cumsum(n)])
        eoff = np.concatenate([[0], np.cumsum(e)])
        N, E, F = int(noff[-1]), int(eoff[-1]), self.F
        first = pack_tiles(n, e) if self.no_self_loops else None
        T1 = len(first) if first is not None else 0
        # int64 staging: [edge_index (2E) | batch (N) | y (B) | ptr (B + 1) | edge_ptr (B + 1) | tile first / node / edge offsets (3 T1)]
        o_b, o_y, o_p, o_e, o_t = 2 * E, 2 * E + N, 2 * E + N + B, 2 * E + N + 2 * B + 1, 2 * E + N + 3 * B + 2
        fbuf, ibuf, slot = self._buffers(N * F, o_t + 3 * T1)
        ia = ibuf.numpy()
        # host pointers only -> the host library (a CPU-only machine without libcalhip.so collates too)
        _lib.call("cal_collate_host", self.X.ctypes.data, self.EI.ctypes.data, int(self.EI.shape[1]), F, self.node_ptr.ctypes.data,
                  self.edge_ptr.ctypes.data, self.Y.ctypes.data, idx.ctypes.data, B, fbuf.data_ptr(), ibuf.data_ptr(), E,
                  ibuf.data_ptr() + 8 * o_b, ibuf.data_ptr() + 8 * o_y, host=True)
        ia[o_p:o_e] = noff
        ia[o_e:o_t] = eoff
        views = [("x" if self.feat_is_x else "feat", "f", 0, N * F, (N, F)), ("edge_index", "i", 0, 2 * E, (2, E)),
                 ("batch", "i", o_b, o_y, (N,)), ("y", "i", o_y, o_p, (B,)), ("ptr", "i", o_p, o_e, (B + 1,)),
                 ("edge_ptr", "i", o_e, o_t, (B + 1,))]
        # (the object is filled through its __dict__ in one go and the int64 views come from ONE split: Batch() + a slice and a view
        #  per attribute were 75 us of a 250 us collate, a third of what the GPU needs for the whole step)
        b = Batch.__new__(Batch)
        d = b.__dict__
        d.update(_BATCH_DEFAULTS)
        d["num_graphs"] = B
        d["max_nodes"] = int(n.max()) if B else 0
        d["max_edges"] = int(e.max()) if B else 0
        d["no_self_loops"] = self.no_self_loops
        sizes = [2 * E, N, B, B + 1, B + 1]
        if T1:
            fi = np.asarray(first, dtype=np.int64)
            tn, te = noff[fi], eoff[fi]
            ia[o_t:o_t + T1] = fi
            ia[o_t + T1:o_t + 2 * T1] = tn
            ia[o_t + 2 * T1:o_t + 3 * T1] = te
            views += [("tile_ptr", "i", o_t, o_t + T1, (T1,)), ("tile_node_ptr", "i", o_t + T1, o_t + 2 * T1, (T1,)),
                      ("tile_edge_ptr", "i", o_t + 2 * T1, o_t + 3 * T1, (T1,))]
            d["tile_max_nodes"], d["tile_max_edges"] = int((tn[1:] - tn[:-1]).max()), int((te[1:] - te[:-1]).max())
            sizes += [T1, T1, T1]
        parts = ibuf.split(sizes)
        d["x" if self.feat_is_x else "feat"] = fbuf.view(N, F)
        d["edge_index"] = parts[0].view(2, E)
        d["batch"], d["y"], d["ptr"], d["edge_ptr"] = parts[1], parts[2], parts[3], parts[4]
        if T1:
            d["tile_ptr"], d["tile_node_ptr"], d["tile_edge_ptr"] = parts[5], parts[6], parts[7]
        b._staged = (fbuf, ibuf, views, slot)
        if slot is not None:
            slot.owner = weakref.ref(b)
        return b


def shard_indices(n: int, shuffle: bool, rank: int, world_size: int, drop_last: bool,
                  generator: Optional[torch.Generator], seed: int, epoch: int) -> List[int]:
    """Indices of one epoch for replica ``rank`` of ``world_size``.

    Single replica: ``torch.randperm(n, generator)`` (``torch.utils.data.RandomSampler``) or ``range(n)``.
    Several replicas: every rank must see the SAME permutation and the SAME number of
    samples -- unequal shard lengths give ranks different step counts and the per-step gradient all-reduce
    hangs.  The permutation therefore comes from a generator seeded with ``seed + epoch`` on every rank (an
    explicit ``generator`` must then be seeded identically on all ranks), is padded by wrapping around to
    ``world_size * ceil(n / world_size)`` entries (``drop_last``: truncated to ``world_size * (n // world_size)``)
    as ``torch.utils.data.DistributedSampler`` does, and rank r takes entries r, r + world, ...
    """
    if world_size > 1 and shuffle and generator is None:
        generator = torch.Generator().manual_seed(int(seed) + int(epoch))
    idx = torch.randperm(n, generator=generator).tolist() if shuffle else list(range(n))
    if world_size > 1:
        if drop_last:
            idx = idx[:(n // world_size) * world_size]
        else:
            total = -(-n // world_size) * world_size
            while n and len(idx) < total:
                idx += idx[:total - len(idx)]
        idx = idx[rank::world_size]
    return idx


_CONCATS = []        # [(dataset list, per-graph signatures, _HostConcat)]: the last two static datasets a DataLoader was built over


_FP_PHASE = [0]
_VERSION_OF = operator.attrgetter("_version")


_NO_TENSOR = torch.zeros(0)          # stands in for a missing attribute (its version counter never moves)


def _graph_refs(dataset, idx=None):
    """(graphs, tensors): the graph OBJECTS at ``idx`` (all when None) and, three per graph, the tensor objects a collate reads
    from them (features, edge_index, y).  The cache keeps these references, so "is it still the same tensor" is an identity
    test on live objects (an address or ``data_ptr`` can be recycled by a replacement; a held object cannot)."""
    graphs = list(dataset) if idx is None else [dataset[i] for i in idx]
    tens = []
    add = tens.append
    for g in graphs:
        d = getattr(g, "__dict__", None) or {}
        x = d.get("x")
        if x is None:
            x = d.get("feat")
        ei, y = d.get("edge_index"), d.get("y")
        add(x if isinstance(x, torch.Tensor) else _NO_TENSOR)
        add(ei if isinstance(ei, torch.Tensor) else _NO_TENSOR)
        add(y if isinstance(y, torch.Tensor) else _NO_TENSOR)
    return graphs, tens


def _version_sum(tens) -> int:
    """Sum of the in-place version counters (torch bumps ``_version`` on every in-place write; the counters only grow, so an
    equal sum over the same objects means every one of them is unchanged)."""
    return sum(map(_VERSION_OF, tens))


def _graph_sig(dataset):
    graphs, tens = _graph_refs(dataset)
    vers = list(map(_VERSION_OF, tens))
    return (graphs, tens, sum(vers), vers)


def _unchanged(dataset, sig, full: bool = True) -> bool:
    """Has the list (or a graph in it) changed since ``sig`` was taken?  A replaced element, a re-assigned or edited label /
    feature / edge list and a transform applied in place all show: the graph and tensor OBJECTS are compared by identity, their
    in-place version counters by sum.  ``full`` (every NEW DataLoader over a cached list): all graphs, ~0.7 us each -- 4 ms for
    the reference's 5 596-graph training split, once per loader (round 5 compared a tuple of addresses and versions per graph:
    1.8-4 us each, 0.23 ms per step of a loop that builds a loader per 16-batch epoch, more than the step itself).
    ``full=False`` (a loader re-validating its cache at the start of each further epoch): lists of more than 512 graphs are
    sampled at ~512 positions whose phase rotates from call to call -- a transform over the whole list shows at once, an edit of
    ONE graph within a few epochs.  The reference's loader re-reads its dataset every batch (train_causal.py:13-15);
    ``clear_collate_cache()`` forces a rebuild."""
    graphs, tens, vsum, vers = sig
    n = len(dataset)
    if n != len(graphs):
        return False
    if full or n <= 512:
        g_now, t_now = _graph_refs(dataset)
        return all(map(operator.is_, g_now, graphs)) and all(map(operator.is_, t_now, tens)) and _version_sum(t_now) == vsum
    step = n // 509
    _FP_PHASE[0] = (_FP_PHASE[0] + 1) % step
    idx = list(range(_FP_PHASE[0], n, step)) + [0, n // 2, n - 1]
    g_now, t_now = _graph_refs(dataset, idx)
    t_then = [tens[3 * i + k] for i in idx for k in range(3)]
    return (all(map(operator.is_, g_now, [graphs[i] for i in idx])) and all(map(operator.is_, t_now, t_then))
            and _version_sum(t_now) == sum(vers[3 * i] + vers[3 * i + 1] + vers[3 * i + 2] for i in idx))


def clear_collate_cache():
    """Drop the cached host concatenations (they hold a copy of the last two datasets a DataLoader was built over)."""
    del _CONCATS[:]


def _concat_of(dataset, full: bool = True):
    """One ``_HostConcat`` per dataset list, kept across DataLoader objects (loops that build a new loader every epoch over the
    same list would pay the concatenation -- ~30 ms for 5 000 graphs -- sixteen steps apart); rebuilt when the list or a graph
    in it has changed since (``_unchanged``)."""
    for i, (ds, sigs, hc) in enumerate(_CONCATS):
        if ds is dataset:
            if _unchanged(dataset, sigs, full):
                return hc
            del _CONCATS[i]
            break
    hc = _HostConcat(dataset)
    _CONCATS.append((dataset, _graph_sig(dataset), hc))
    del _CONCATS[:-2]
    return hc


class DataLoader:
    """``DataLoader(dataset, batch_size, shuffle)`` (train_causal.py:13-15).

    Shuffling uses ``torch.randperm`` (as ``torch.utils.data.RandomSampler``
    does); ``rank``/``world_size`` give each data-parallel replica an equally
    long strided shard of every epoch's (shared) permutation (``shard_indices``).
    """

    def __init__(self, dataset, batch_size: int = 1, shuffle: bool = False,
                 rank: int = 0, world_size: int = 1, drop_last: bool = False,
                 generator: Optional[torch.Generator] = None, seed: int = 0):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.rank, self.world_size = rank, world_size
        self.drop_last = drop_last
        self.generator = generator
        self.seed, self.epoch = int(seed), 0
        self._concat = None         # _HostConcat of a static list dataset (built on the first epoch)

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def _indices(self) -> List[int]:
        return shard_indices(len(self.dataset), self.shuffle, self.rank, self.world_size, self.drop_last,
                             self.generator, self.seed, self.epoch)

    def _collate(self, chunk) -> Batch:
        if self._concat is None and isinstance(self.dataset, (list, tuple)):
            self._concat = _concat_of(self.dataset)
        if self._concat is not None and self._concat.ok:
            return self._concat.collate(chunk)
        return Batch.from_data_list([self.dataset[i] for i in chunk])

    def _shard_len(self) -> int:
        n, w = len(self.dataset), self.world_size
        if w <= 1:
            return n
        return n // w if self.drop_last else -(-n // w)

    def __len__(self) -> int:
        n = self._shard_len()
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self) -> Iterable[Batch]:
        idx = self._indices()
        self.epoch += 1
        if self._concat is not None:         # a further epoch of this loader: the cached concatenation is re-validated (sampled)
            self._concat = _concat_of(self.dataset, full=False)
        for s in range(0, len(idx), self.batch_size):
            chunk = idx[s:s + self.batch_size]
            if self.drop_last and len(chunk) < self.batch_size:
                return
            yield self._collate(chunk)


def from_networkx(G) -> Data:
    """PyG ``from_networkx`` (call site utils.py:55): relabel to 0..n-1, make
    directed (both directions of every undirected edge, grouped by source),
    stack node attributes into ``[N, ...]`` tensors."""
    import networkx as nx
    import numpy as np

    G = nx.convert_node_labels_to_integers(G)
    G = G.to_directed() if not nx.is_directed(G) else G
    edges = list(G.edges)
    edge_index = (torch.tensor(edges, dtype=torch.long).t().contiguous()
                  if edges else torch.zeros(2, 0, dtype=torch.long))
    data = Data(edge_index=edge_index)
    keys = set()
    for _, feat_dict in G.nodes(data=True):
        keys |= set(feat_dict.keys())
    for k in keys:
        vals = [G.nodes[i][k] for i in range(G.number_of_nodes())]
        setattr(data, str(k), torch.tensor(np.asarray(vals)))
    data.num_nodes = G.number_of_nodes()
    return data
