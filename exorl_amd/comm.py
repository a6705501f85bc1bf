"""Process-wide RCCL communicator of libexorl_hip.so (exorl_comm_*, include/exorl_hip.h) for data-parallel agents.

The reference has no multi-GPU path; SURVEY 8e derives the exchanges its update implies (critic grads, sum|Q|, actor grads). Under
torch.distributed the default keeps the collectives with torch.distributed (all_reduce between exorl_agent_update_phase calls — RCCL
when the backend is nccl; the path the two-process tests execute). EXORL_DP_COMM=native attaches this communicator instead:
exorl_agent_update then enqueues the all-reduces between its phases itself — one host call per step, capturable into the step's
hipGraph; torch.distributed is only the bootstrap channel for the 128-byte id. OPT-IN because it is UNVERIFIED beyond one rank: this
pool gives one GPU per box, so RCCL has only ever run here as a 1-rank communicator. bench.py --gpus N tries it behind a
watchdog after it has a number from the default path.
"""
import ctypes as C
import os
import warnings

import torch

from . import _lib as L

ID_BYTES = 128
_cached = None


class Comm:
    def __init__(self, rank, nranks, id_bytes):
        self.lib = L.load()
        self.rank, self.nranks = rank, nranks
        buf = (C.c_char * ID_BYTES).from_buffer_copy(bytes(id_bytes))
        h = C.c_void_p()
        L.check(self.lib.exorl_comm_init(rank, nranks, buf, C.byref(h)))
        self.h = h

    @staticmethod
    def unique_id():
        buf = (C.c_char * ID_BYTES)()
        L.check(L.load().exorl_comm_unique_id(buf))
        return bytes(buf)

    def allreduce(self, t):
        assert t.dtype == torch.float32 and t.is_contiguous()
        L.check(self.lib.exorl_comm_allreduce(self.h, t.data_ptr(), t.numel(), L.current_stream()))

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h:
            self.lib.exorl_comm_destroy(h)


def native_comm(device):
    """The communicator spanning torch.distributed's ranks, created on first use (collective: every rank must get here), or None
    when the collectives stay with torch.distributed (not initialised, gloo backend, EXORL_DP_COMM != native, or RCCL refused)."""
    global _cached
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    if _cached is not None:
        return _cached or None
    if dist.get_backend() != 'nccl' or os.environ.get('EXORL_DP_COMM', 'torch') != 'native':
        _cached = False
        return None
    rank, world = dist.get_rank(), dist.get_world_size()
    idt = torch.zeros(ID_BYTES, dtype=torch.uint8, device=device)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(Comm.unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, 0)
    comm, err = None, ''
    try:
        with torch.cuda.device(device):
            comm = Comm(rank, world, idt.cpu().numpy().tobytes())
            probe = torch.full((4,), float(rank + 1), device=device)
            comm.allreduce(probe)
            torch.cuda.synchronize()
            if float(probe[0]) != world * (world + 1) / 2:
                raise L.ExorlError(f'probe all-reduce returned {float(probe[0])}')
    except L.ExorlError as e:
        comm, err = None, str(e)
    ok = torch.tensor([1.0 if comm is not None else 0.0], device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)              # all ranks take the same path
    if float(ok[0]) == 0.0:
        if rank == 0:
            warnings.warn(f'exorl_amd: native RCCL communicator unavailable ({err or "another rank failed"}); using torch.distributed')
        _cached = False
        return None
    _cached = comm
    return comm
