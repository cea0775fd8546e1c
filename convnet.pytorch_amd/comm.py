"""Direct-RCCL communicator of the data-parallel exchange step (C ABI: cn_comm_* in include/convnet_hip.h).

Replaces what the reference gets from ``nn.parallel.DistributedDataParallel`` (/root/reference
trainer.py:79-82) on the DATA path: bucket all-reduces, the construction-time parameter broadcast, the
BatchNorm buffer broadcast and the SyncBatchNorm reductions go straight to RCCL on HIP streams this
package controls.  ``torch.distributed`` keeps only the rendezvous role (its store / an object broadcast
ships rank 0's 128-byte RCCL unique id), exactly the part the reference delegates to ``--dist-init``.

On a host without devices (the gloo CPU tests) or with a non-RCCL process group there is no
communicator and the callers use ``torch.distributed`` collectives instead - stated, not silent:
``describe()`` names the transport and bench.py prints it.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check

_DEFAULT = None     # the communicator of the default data-parallel group (set by Trainer)


class RcclCommunicator(object):
    def __init__(self, process_group=None, device=None):
        L = _lib.load()
        self.pg = process_group
        self.rank = dist.get_rank(process_group)
        self.world = dist.get_world_size(process_group)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            check(L.cn_comm_unique_id(buf), 'cn_comm_unique_id')
        box = [buf.raw if self.rank == 0 else None]
        if self.world > 1:
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            dist.broadcast_object_list(box, src=src, group=process_group)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(L.cn_comm_init(ctypes.byref(self._h), box[0], self.rank, self.world), 'cn_comm_init')
        ver = ctypes.c_int(0)
        check(L.cn_comm_info(self._h, None, None, ctypes.byref(ver)), 'cn_comm_info')
        self.rccl_version = ver.value

    # -- collectives ---------------------------------------------------------------------------
    def allreduce_bucket(self, view, streams):
        """SUM all-reduce of a contiguous fp32 gradient slice on the communicator's stream, ordered after
        what is queued on `streams` (1 or 2 raw hipStream_t)."""
        a = streams[0]
        b = streams[1] if len(streams) > 1 else None
        check(_lib.load().cn_comm_allreduce_bucket(self._h, view.data_ptr(), view.numel(), a, b, len(streams)),
              'cn_comm_allreduce_bucket')

    def join(self, stream):
        check(_lib.load().cn_comm_join(self._h, stream), 'cn_comm_join')

    def allreduce_(self, t):
        """In-stream SUM all-reduce of a contiguous fp32 / fp64 tensor on torch's current stream."""
        code = {torch.float32: 0, torch.float64: 2}[t.dtype]
        check(_lib.load().cn_comm_allreduce(self._h, t.data_ptr(), t.numel(), code,
                                            torch.cuda.current_stream(t.device).cuda_stream), 'cn_comm_allreduce')
        return t

    def broadcast_(self, t, root=0):
        if not t.is_contiguous():
            raise _lib.ConvNetHipError('broadcast_ needs a contiguous tensor')
        check(_lib.load().cn_comm_broadcast(self._h, t.data_ptr(), t.numel() * t.element_size(), root,
                                            torch.cuda.current_stream(t.device).cuda_stream), 'cn_comm_broadcast')
        return t

    def destroy(self):
        if self._h:
            _lib.load().cn_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def describe(self):
        return 'direct RCCL %d (cn_comm_*), %d rank%s' % (self.rccl_version, self.world, '' if self.world == 1 else 's')


def wanted(device, process_group=None):
    """Direct RCCL is the transport whenever the ranks own HIP devices and the process group was brought up
    for RCCL (backend 'nccl'); CONVNET_AMD_COMM=torch forces the torch.distributed collectives (A/B)."""
    if torch.device(device).type != 'cuda' or _lib.is_emulated():
        return False
    if os.environ.get('CONVNET_AMD_COMM', 'rccl') == 'torch':
        return False
    return dist.get_backend(process_group) == 'nccl'


def create_default(device, process_group=None):
    global _DEFAULT
    if _DEFAULT is not None:
        _DEFAULT.destroy()
    comm, err = None, None
    try:
        comm = RcclCommunicator(process_group, device)
    except Exception as e:      # e.g. librccl missing / ncclCommInitRank refused on this node
        err = e
    if dist.get_world_size(process_group) > 1:
        # every rank must take the same transport: agree through the process group that is already up
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=torch.device(device))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=process_group)
        if int(ok.item()) == 0 and comm is not None:
            comm.destroy()
            comm = None
    if comm is None:
        import logging
        logging.warning('direct RCCL communicator unavailable (%s): the gradient exchange goes through '
                        'torch.distributed (%s) instead', err if err is not None else 'another rank failed',
                        dist.get_backend(process_group))
    _DEFAULT = comm
    return _DEFAULT


def default():
    return _DEFAULT


def destroy_default():
    global _DEFAULT
    if _DEFAULT is not None:
        _DEFAULT.destroy()
        _DEFAULT = None
