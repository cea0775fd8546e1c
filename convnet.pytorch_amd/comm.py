"""Direct-RCCL communicator of the data-parallel exchange step (C ABI: cn_comm_* in include/convnet_hip.h).

Replaces what the reference gets from ``nn.parallel.DistributedDataParallel`` (/root/reference
trainer.py:79-82) on the DATA path: bucket all-reduces, the construction-time parameter broadcast, the
BatchNorm buffer broadcast and the SyncBatchNorm reductions go straight to RCCL on HIP streams this
package controls.  ``torch.distributed`` keeps only the rendezvous role (its store / an object broadcast
ships rank 0's 128-byte RCCL unique id), exactly the part the reference delegates to ``--dist-init``.

On a host without devices (the gloo CPU tests) or with a non-RCCL process group there is no
communicator and the callers use ``torch.distributed`` collectives instead - stated, not silent:
``describe()`` names the transport and bench.py prints it.  With RCCL ranks there is no second transport to
fall back to: if the direct communicator cannot be built the job stops (``CONVNET_AMD_COMM=torch`` selects
the torch.distributed collectives explicitly, for A/B).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check

_DEFAULT = None     # the communicator of the default data-parallel group (set by Trainer)


def _agree(ok, what, err, process_group, device):
    """One MIN all-reduce over a status flag through the process group that is already up: every rank learns
    whether ALL ranks passed the phase, and every rank raises if one did not - nobody is left blocked in the next
    phase's collective (a rank that failed early used to meet its peers in mismatched collectives)."""
    if dist.get_world_size(process_group) > 1:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
        all_ok = int(flag.item()) == 1
    else:
        all_ok = ok
    if not all_ok:
        raise _lib.ConvNetHipError('direct-RCCL set-up failed in phase "%s" on %s: %s' % (
            what, 'this rank' if not ok else 'another rank', err if err is not None else 'see that rank'))


class RcclCommunicator(object):
    """Set-up in three phases, each closed by an agreement step (`_agree`), so that an asymmetric failure raises
    on every rank instead of hanging the job:
      1. every rank resolves librccl (cn_comm_load: no communicator, no collective);
      2. rank 0 draws the unique id and ALWAYS broadcasts - a 128-byte sentinel on failure;
      3. every rank enters ncclCommInitRank (cn_comm_init)."""

    def __init__(self, process_group=None, device=None):
        L = _lib.load()
        self.pg = process_group
        self.rank = dist.get_rank(process_group)
        self.world = dist.get_world_size(process_group)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self._h = ctypes.c_void_p()
        try:
            # phase 1
            rc = L.cn_comm_load()
            _agree(rc == 0, 'load librccl', _lib.last_error() if rc != 0 else None, process_group, self.device)
            # phase 2
            buf = ctypes.create_string_buffer(128)
            err = None
            if self.rank == 0 and L.cn_comm_unique_id(buf) != 0:
                err = _lib.last_error()
            box = [(buf.raw if err is None else b'') if self.rank == 0 else None]
            if self.world > 1:
                src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
                dist.broadcast_object_list(box, src=src, group=process_group)
            _agree(len(box[0]) == 128, 'unique id', err, process_group, self.device)
            # phase 3
            with torch.cuda.device(self.device):
                rc = L.cn_comm_init(ctypes.byref(self._h), box[0], self.rank, self.world)
            _agree(rc == 0, 'ncclCommInitRank', _lib.last_error() if rc != 0 else None, process_group, self.device)
            ver = ctypes.c_int(0)
            check(L.cn_comm_info(self._h, None, None, ctypes.byref(ver)), 'cn_comm_info')
            self.rccl_version = ver.value
        except Exception:
            # a rank whose own cn_comm_init succeeded still gets here when a PEER failed (the agreement step raises on
            # every rank): its communicator, stream and events must not outlive the failed set-up
            self.destroy()
            raise

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


def topology_summary(log_path):
    """What RCCL said about the communicators of this process (NCCL_DEBUG=INFO lines in `log_path`, written because the
    caller set NCCL_DEBUG / NCCL_DEBUG_SUBSYS=INIT,GRAPH / NCCL_DEBUG_FILE before the first RCCL call - bench.py does for
    N > 1): channel count, whether rings / trees were built, and the transports the peers are reached over (P2P/IPC over
    xGMI, SHM, NET).  Best effort: an unreadable or empty log gives {}."""
    import re
    try:
        with open(log_path, errors='replace') as f:
            text = f.read()
    except OSError:
        return {}
    out = {}
    m = re.findall(r'Channel (\d+)/(\d+)\s*:', text)
    if m:
        out['channels'] = max(int(b) for _, b in m)
    m = re.search(r'(\d+) coll channels', text)
    if m:
        out['coll_channels'] = int(m.group(1))
    if re.search(r'\bRing \d+\s*:', text) or 'Connected all rings' in text:
        out['rings'] = True
    if re.search(r'\bTrees? \[', text) or 'Connected all trees' in text:
        out['trees'] = True
    via = sorted(set(re.findall(r'via ([A-Za-z0-9/]+)', text)))
    if via:
        out['via'] = via[:6]
    m = re.search(r'RCCL version\s*:?\s*([0-9][^\s]*)', text) or re.search(r'NCCL version ([0-9][^\s]*)', text)
    if m:
        out['version_line'] = m.group(1)
    return out


def wanted(device, process_group=None):
    """Direct RCCL is the transport whenever the ranks own HIP devices and the process group was brought up
    for RCCL (backend 'nccl'); CONVNET_AMD_COMM=torch forces the torch.distributed collectives (A/B)."""
    if torch.device(device).type != 'cuda' or _lib.is_emulated():
        return False
    if os.environ.get('CONVNET_AMD_COMM', 'rccl') == 'torch':
        return False
    return dist.get_backend(process_group) == 'nccl'


def create_default(device, process_group=None):
    """The communicator of the default data-parallel group.  There is ONE transport per configuration and no
    fallback: with RCCL ranks (`wanted`) the direct communicator is built or the job stops with the reason on
    every rank; the torch.distributed collectives are used only where there is no RCCL to talk to (gloo CPU /
    emulator tests) or on explicit request (CONVNET_AMD_COMM=torch, an A/B switch)."""
    global _DEFAULT
    if _DEFAULT is not None:
        _DEFAULT.destroy()
        _DEFAULT = None
    _DEFAULT = RcclCommunicator(process_group, device)
    return _DEFAULT


def default():
    return _DEFAULT


def destroy_default():
    global _DEFAULT
    if _DEFAULT is not None:
        _DEFAULT.destroy()
        _DEFAULT = None
