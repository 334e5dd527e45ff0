"""One-shot gradient exchange over peer-mapped memory (SURVEY.md 8e "consider a hand-rolled one-shot xGMI P2P
all-reduce ... if RCCL launch latency threatens the 0.7 efficiency bar").

The data-parallel exchange of CAL's flat gradient bucket (555 KB at BASELINE config 2) is latency-bound: seven peers'
buckets are ~4 MB of reads per GPU.  ``P2PExchange`` gives every rank of ONE node a region that all ranks map: FINE-GRAINED
device memory from ``cal_p2p_alloc`` (``hipExtMallocWithFlags(hipDeviceMallocFinegrained)`` -- a kernel that polls a flag a
peer GPU writes and reads the peer's bucket while both kernels run needs that; ``torch.zeros`` is coarse-grained), shared
through 64-byte IPC handles (``cal_p2p_export`` / ``cal_p2p_open``) exchanged once through the process group's object
all-gather.  The step then ends with ``cal_engine_p2p_adam`` -- one kernel that publishes the bucket, waits for the peers'
flags, sums in rank order (bit-identical replicas) and applies Adam with the 1 / world factor -- instead of [RCCL all-reduce
node -> Adam kernel].  It is a plain kernel node: capturable, no host involvement per step.  If a peer's flag does not
arrive within the timeout, NO rank-local parameter is touched (in that launch or any later one) and ``status()`` turns 64;
``CausalTrainer.step`` reads it (a host-mapped word, no synchronisation) before every step and raises.
Opt-in: ``CausalTrainer(p2p_exchange=True)`` / ``CAL_AMD_P2P_EXCHANGE=1``.

What the one-GPU test (two ranks sharing device 0) can and cannot show: the protocol, the consensus and the arithmetic
(bit-identical to the all-reduce) -- not cross-device coherence over xGMI, which needs two GPUs (the driver's N > 1 runs
use the RCCL node by default for that reason).
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from . import _lib
from .plan import _stream


class P2PTimeout(RuntimeError):
    pass


class P2PExchange:
    def __init__(self, engine, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("P2PExchange needs an initialised torch.distributed process group (one node)")
        self.engine = engine
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world > 8:
            raise ValueError("P2PExchange: at most 8 ranks (one node)")
        self.group = group
        self.region = None
        self.mapped = []
        self._bind()

    def _bind(self):
        engine, group = self.engine, self.group
        nbytes = _lib.query("cal_engine_p2p_region_bytes", engine._h)
        with torch.cuda.device(engine.device):
            base = ctypes.c_void_p()
            _lib.call("cal_p2p_alloc", nbytes, ctypes.byref(base))          # its own allocation: an IPC handle names a whole one
            self.region = base.value
            handle = ctypes.create_string_buffer(64)
            _lib.call("cal_p2p_export", base, handle)
            gathered = [None] * self.world
            dist.all_gather_object(gathered, (bytes(handle.raw), int(engine.device.index or 0)), group=group)
            ptrs, devs = [], []
            for r, (hd, d) in enumerate(gathered):
                if r == self.rank:
                    ptrs.append(self.region)
                else:
                    m = ctypes.c_void_p()
                    _lib.call("cal_p2p_open", ctypes.create_string_buffer(hd, 64), ctypes.byref(m))     # maps the peer's region here
                    self.mapped.append(m.value)
                    ptrs.append(m.value)
                devs.append(d)
            bases = (ctypes.c_void_p * self.world)(*ptrs)
            devarr = (ctypes.c_int64 * self.world)(*devs)
            _lib.call("cal_engine_p2p_bind", engine._h, bases, devarr, self.world, self.rank)
        engine.set_grad_scale(1.0 / self.world)
        torch.cuda.synchronize()
        dist.barrier(group=group)            # every region is zeroed and mapped before the first publish

    def set_timeout(self, max_polls: int):
        """Bound of the in-kernel wait for the peers' flags, in polls (~1 us each; default 2^22).  Applies to launches enqueued
        or captured afterwards."""
        _lib.call("cal_engine_p2p_set_timeout", self.engine._h, int(max_polls))

    def status(self) -> int:
        """0, or 64 once an exchange has timed out (host-mapped word: no device synchronisation)."""
        return _lib.query("cal_engine_p2p_status", self.engine._h)

    def check(self):
        if self.status() != 0:
            raise P2PTimeout("one-shot peer-memory exchange: a peer's gradient bucket did not arrive within the timeout; the "
                             "parameters were left untouched from that step on (rank %d of %d)" % (self.rank, self.world))

    def adam(self):
        """After ``engine.train_step(adam=False, tick=True)``: exchange + Adam, one launch on the current stream."""
        _lib.call("cal_engine_p2p_adam", self.engine._h, _stream())

    def close(self):
        torch.cuda.synchronize()
        for m in self.mapped:
            _lib.call("cal_p2p_close", ctypes.c_void_p(m))
        self.mapped = []
        if self.region is not None:
            _lib.call("cal_p2p_free", ctypes.c_void_p(self.region))
            self.region = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
