"""OPT-IN one-shot all-reduce of small fp64 buffers through peer-mapped mailboxes (csrc/mailbox.hip, FACL_ONESHOT_SYNCBN=1).

The SyncBN reductions of the data-parallel step (facl_amd/dist.py) are 0.1-16 KB; as RCCL collectives each is latency-bound and, in
the default launch path, cuts the captured graph.  Here a reduction is one kernel launch on the current stream: capturable, no cut.
Bootstrap = one `all_gather_object` of the 64-byte IPC handles over the existing process group.  Rehearsed with several processes
on one GPU; across devices (xGMI) it is untested -- never the default.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib


class OneShotAllReduce:
    """``r = OneShotAllReduce(group, n_max); r(t)`` -- in-place SUM of an fp64 CUDA tensor (numel <= n_max) over the group,
    added in rank order.  Every rank must call with the same sizes in the same order.  ``r.check()`` raises if a peer ever
    failed to post in time (device error word; reads back = synchronises: call it outside the hot loop)."""

    def __init__(self, group=None, n_max=4608):
        lib = _lib.load_library()
        self.lib, self.group, self.n_max = lib, group, int(n_max)
        self.rank, self.R = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device())
        nbytes = lib.facl_mailbox_bytes(self.R, self.n_max)
        own = ctypes.c_void_p()
        handle = (ctypes.c_char * 64)()
        _lib.check(lib.facl_mailbox_alloc(nbytes, ctypes.byref(own), handle), "facl_mailbox_alloc")
        self._own = own
        handles = [None] * self.R
        dist.all_gather_object(handles, bytes(handle.raw), group=group)
        ptrs, self._opened = [], []
        for r, h in enumerate(handles):
            if r == self.rank:
                ptrs.append(own.value)
                continue
            p = ctypes.c_void_p()
            buf = (ctypes.c_char * 64).from_buffer_copy(h)
            _lib.check(lib.facl_mailbox_open(buf, ctypes.byref(p)), "facl_mailbox_open (peer %d)" % r)
            ptrs.append(p.value)
            self._opened.append(p)
        self.boxes = torch.tensor(ptrs, dtype=torch.int64, device=self.device)          # device array of R pointers
        self.seq = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        torch.cuda.synchronize()
        dist.barrier(group=group)                      # every mailbox exists, is zeroed and is opened before the first post

    def __call__(self, t):
        if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or t.numel() > self.n_max:
            raise RuntimeError("OneShotAllReduce takes contiguous fp64 CUDA tensors of at most %d elements" % self.n_max)
        _lib.check(self.lib.facl_mailbox_allreduce(_lib.ptr(t), _lib.ptr(t), t.numel(), self.n_max, _lib.ptr(self.boxes), self.rank,
                                                   self.R, _lib.ptr(self.seq), _lib.ptr(self.err), _lib.stream()),
                   "facl_mailbox_allreduce")
        return t

    def check(self):
        if int(self.err.item()) != 0:
            raise RuntimeError("one-shot all-reduce: a peer did not post within the time limit (results were poisoned with NaN)")

    def close(self):
        torch.cuda.synchronize()
        for p in self._opened:
            self.lib.facl_mailbox_close(p)
        self._opened = []
        if self._own is not None:
            self.lib.facl_mailbox_free(self._own)
            self._own = None
