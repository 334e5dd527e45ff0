"""One-shot gradient exchange over peer-mapped memory (SURVEY.md 8e "consider a hand-rolled one-shot xGMI P2P
all-reduce ... if RCCL launch latency threatens the 0.7 efficiency bar").

The data-parallel exchange of CAL's flat gradient bucket (555 KB at BASELINE config 2) is latency-bound: seven peers'
buckets are ~4 MB of reads per GPU.  ``P2PExchange`` gives every rank of ONE node a region that all ranks map (CUDA/HIP IPC
handles, exchanged once through the process group's object all-gather); the step then ends with ``cal_engine_p2p_adam`` --
one kernel that publishes the bucket, waits for the peers' flags, sums in rank order (bit-identical replicas) and applies
Adam with the 1 / world factor -- instead of [RCCL all-reduce node -> Adam kernel].  It is a plain kernel node: capturable,
no host involvement per step.  Opt-in: ``CausalTrainer(p2p_exchange=True)`` / ``CAL_AMD_P2P_EXCHANGE=1``.
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from . import _lib
from .plan import _stream


class P2PExchange:
    def __init__(self, engine, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("P2PExchange needs an initialised torch.distributed process group (one node)")
        self.engine = engine
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world > 8:
            raise ValueError("P2PExchange: at most 8 ranks (one node)")
        nbytes = _lib.query("cal_engine_p2p_region_bytes", engine._h)
        # its own allocation: an IPC handle names a whole device allocation
        self.region = torch.zeros(nbytes // 4, dtype=torch.float32, device=engine.device)
        from torch.multiprocessing.reductions import reduce_tensor
        fn, fargs = reduce_tensor(self.region)
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (fn, fargs, int(engine.device.index or 0)), group=group)
        self.peers, devs = [], []
        for r, (f, a, d) in enumerate(gathered):
            self.peers.append(self.region if r == self.rank else f(*a))        # maps the peer's allocation into this process
            devs.append(d)
        bases = (ctypes.c_void_p * self.world)(*[t.data_ptr() for t in self.peers])
        devarr = (ctypes.c_int64 * self.world)(*devs)
        _lib.call("cal_engine_p2p_bind", engine._h, bases, devarr, self.world, self.rank)
        engine.set_grad_scale(1.0 / self.world)
        torch.cuda.synchronize()
        dist.barrier(group=group)            # every region is zeroed and mapped before the first publish

    def adam(self):
        """After ``engine.train_step(adam=False, tick=True)``: exchange + Adam, one launch on the current stream."""
        _lib.call("cal_engine_p2p_adam", self.engine._h, _stream())
