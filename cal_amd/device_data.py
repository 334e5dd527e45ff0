"""Device-resident dataset + on-GPU mini-batch assembly.

``DeviceDataset(list_of_Data)`` concatenates every graph once into HBM (8 000 SPMotif graphs are
~18 MB); ``DeviceLoader`` then yields ``Batch`` objects assembled by one ``cal_collate`` kernel per
step -- the host only draws the permutation and prefix-sums ``batch_size`` graph sizes -- instead of
the per-graph Python ``torch.cat`` of ``DataLoader`` / ``Batch.from_data_list``
(train_causal.py:13-15,171-174).  Results are bit-identical to the host collate.
"""
from __future__ import annotations

from typing import Iterable, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .data import Batch, pack_order, pack_tiles, shard_indices
from .plan import _p, _stream


class DeviceDataset:
    def __init__(self, graphs: Sequence, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.CalError("DeviceDataset lives in GPU memory (no CPU path)")
        feats = [g.x if g.x is not None else g.feat for g in graphs]
        self.feat_is_x = graphs[0].x is not None
        n = np.array([int(f.size(0)) for f in feats], dtype=np.int64)
        e = np.array([int(g.edge_index.size(1)) for g in graphs], dtype=np.int64)
        self.node_sizes, self.edge_sizes = n, e
        self.node_ptr_h = np.concatenate([[0], np.cumsum(n)])
        self.edge_ptr_h = np.concatenate([[0], np.cumsum(e)])
        self.G, self.F = len(graphs), int(feats[0].size(1))
        self.X = torch.cat(feats, 0).to(torch.float32).contiguous().to(dev)
        self.EI = torch.cat([g.edge_index for g in graphs], 1).contiguous().to(dev)       # local node ids
        self.Y = torch.cat([g.y.view(-1)[:1] for g in graphs]).to(torch.long).to(dev)
        self.node_ptr = torch.from_numpy(self.node_ptr_h).to(dev)
        self.edge_ptr = torch.from_numpy(self.edge_ptr_h).to(dev)
        self.device = dev
        self.no_self_loops = bool(self.EI.numel() == 0 or (self.EI[0] != self.EI[1]).all().item())
        self._pin = []          # ring of (pinned buffer, event): the async H2D copy of step k may still be
        self._pin_i = 0         # pending when the host prepares step k+1

    def __len__(self):
        return self.G

    def collate(self, idx, pack: bool = False) -> Batch:
        """Assemble the mini-batch of graphs ``idx`` (host int sequence / numpy / CPU tensor) on the GPU.  ``pack``: the
        graphs may be reordered inside the batch so that consecutive small graphs fill the engine's 64-node tiles
        (data.pack_order: same set of graphs, another order)."""
        idx = np.asarray(idx, dtype=np.int64)
        B = int(idx.shape[0])
        n, e = self.node_sizes[idx], self.edge_sizes[idx]
        first = None
        if pack and self.no_self_loops:
            po = pack_order(n, e)
            if po is not None:
                idx = idx[po[0]]
                n, e = self.node_sizes[idx], self.edge_sizes[idx]
                first = po[1].tolist()
        # small-graph packing: tiles of consecutive graphs for the engine's per-graph kernels (data.pack_tiles)
        if first is None and self.no_self_loops:
            first = pack_tiles(n, e)
        T1 = len(first) if first is not None else 0
        # one pinned staging buffer for [sel | node offsets | edge offsets | tile first graph | tile node off | tile edge off],
        # filled through its numpy view (no intermediate arrays / tensors)
        need = 3 * B + 2 + 3 * T1
        if not self._pin or self._pin[0][0].numel() < need:
            self._pin = []
            for _ in range(16):
                t = torch.empty(max(need, 4096), dtype=torch.long).pin_memory()
                self._pin.append([t, None, t.numpy()])
        slot = self._pin[self._pin_i]
        self._pin_i = (self._pin_i + 1) % len(self._pin)
        if slot[1] is not None:
            slot[1].synchronize()                     # the copy that last used this buffer has run
        host, hv = slot[0][:need], slot[2]
        hv[:B] = idx
        hv[B] = 0
        np.cumsum(n, out=hv[B + 1:2 * B + 1])
        hv[2 * B + 1] = 0
        np.cumsum(e, out=hv[2 * B + 2:3 * B + 2])
        noff, eoff = hv[B:2 * B + 1], hv[2 * B + 1:3 * B + 2]
        N, E = int(noff[-1]), int(eoff[-1])
        tn = te = None
        if T1:
            fi = np.asarray(first, dtype=np.int64)
            tn, te = noff[fi], eoff[fi]
            o = 3 * B + 2
            hv[o:o + T1] = fi
            hv[o + T1:o + 2 * T1] = tn
            hv[o + 2 * T1:o + 3 * T1] = te
        meta = host.to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        b = Batch()
        xo = torch.empty(N, self.F, dtype=torch.float32, device=self.device)
        b.edge_index = torch.empty(2, E, dtype=torch.long, device=self.device)
        b.batch = torch.empty(N, dtype=torch.long, device=self.device)
        b.y = torch.empty(B, dtype=torch.long, device=self.device)
        _lib.call("cal_collate", _p(self.X), _p(self.EI), int(self.EI.size(1)), self.F, _p(self.node_ptr),
                  _p(self.edge_ptr), _p(self.Y), meta[:B].data_ptr(), meta[B:2 * B + 1].data_ptr(),
                  meta[2 * B + 1:].data_ptr(), _p(xo), _p(b.edge_index), E, _p(b.batch), _p(b.y), B, _stream())
        if self.feat_is_x:
            b.x = xo
        else:
            b.feat = xo
        b.num_graphs = B
        b.order = idx                             # dataset index of every row of the batch, in batch order (packing permutes it)
        b.max_nodes = int(n.max()) if B else 0
        b.max_edges = int(e.max()) if B else 0
        b.ptr = meta[B:2 * B + 1]                 # node / edge offsets per graph, already on the device for cal_collate
        b.edge_ptr = meta[2 * B + 1:3 * B + 2]
        b.no_self_loops = self.no_self_loops
        if T1:
            o = 3 * B + 2
            b.tile_ptr, b.tile_node_ptr, b.tile_edge_ptr = meta[o:o + T1], meta[o + T1:o + 2 * T1], meta[o + 2 * T1:o + 3 * T1]
            b.tile_max_nodes, b.tile_max_edges = int((tn[1:] - tn[:-1]).max()), int((te[1:] - te[:-1]).max())
        b._meta = meta            # keep the device copy alive until the kernel has consumed it
        return b


class DeviceLoader:
    """``DataLoader(dataset, batch_size, shuffle)`` semantics (train_causal.py:13-15) over a
    DeviceDataset; ``rank``/``world_size`` shard every epoch's permutation for data parallelism."""

    def __init__(self, dataset: DeviceDataset, batch_size: int, shuffle: bool = False, rank: int = 0,
                 world_size: int = 1, drop_last: bool = False, generator: Optional[torch.Generator] = None,
                 seed: int = 0, pack: Optional[bool] = None):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), shuffle
        # small graphs (mean <= 40 nodes: NCI1, MUTAG, ...): order every mini-batch for the engine's 64-node tiles.  Packing
        # REORDERS the graphs inside a mini-batch (``Batch.order`` names them), which a shuffled loader may do freely (a
        # mini-batch is a set) but a shuffle=False loader may not by default: a consumer that maps output rows back to dataset
        # order would be silently permuted.  ``pack="small"`` asks for it by graph size alone (loops that only count hits).
        small = bool(dataset.G and dataset.node_sizes.mean() <= 40)
        self.pack = (small and shuffle) if pack is None else (small if pack == "small" else bool(pack))
        self.rank, self.world_size, self.drop_last, self.generator = rank, world_size, drop_last, generator
        self.seed, self.epoch = int(seed), 0

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def _indices(self):
        return np.asarray(shard_indices(len(self.dataset), self.shuffle, self.rank, self.world_size, self.drop_last,
                                        self.generator, self.seed, self.epoch), dtype=np.int64)

    def __len__(self):
        n, w = len(self.dataset), self.world_size
        if w > 1:
            n = n // w if self.drop_last else -(-n // w)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self) -> Iterable[Batch]:
        idx = self._indices()
        self.epoch += 1
        for s in range(0, len(idx), self.batch_size):
            chunk = idx[s:s + self.batch_size]
            if self.drop_last and len(chunk) < self.batch_size:
                return
            yield self.dataset.collate(chunk, pack=self.pack)
