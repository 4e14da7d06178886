"""Adam for the training step as ONE HIP launch over all parameter tensors (csrc/adam.hip).

Same update as ``torch.optim.Adam(params, lr, betas, eps)`` with weight_decay = 0, amsgrad = False, maximize = False
(cn3d_train_motion_GL.py:180: lr 3e-4, betas (0.5, 0.999), eps 1e-6); the step counter and the learning rate live on
the device, so a captured HIP graph advances its step counter by itself; learning-rate changes reach a replayed graph
through ``sync_lr()``.  Parameters without a gradient are skipped, like torch's."""
import ctypes

import torch

from . import _lib


class FusedAdam:
    MAX_TENSORS = 64

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        if len(self.params) > self.MAX_TENSORS:
            raise ValueError("FusedAdam handles at most %d tensors per step" % self.MAX_TENSORS)
        for p in self.params:
            _lib.require_cuda(p)
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("FusedAdam needs contiguous float32 CUDA parameters")
        dev = self.params[0].device
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.param_groups = [{"lr": float(lr), "betas": self.betas, "eps": self.eps, "params": self.params}]
        self._lr_host = None
        self._lr = torch.zeros(1, dtype=torch.float32, device=dev)
        self._step = torch.zeros(1, dtype=torch.float32, device=dev)
        self._consts = torch.zeros(2, dtype=torch.float32, device=dev)
        self.state = {p: {"exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)} for p in self.params}

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def sync_lr(self):
        """Host -> device copy of ``param_groups[0]["lr"]`` when a schedule changed it.  step() calls it; a captured
        HIP graph holds no such fill, so whoever REPLAYS a graph with this optimizer inside must call it before each
        replay (train_common.GraphedStep does): the fill lands on the replay stream, ahead of the graph."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:                              # host -> device only when the schedule changed it
            self._lr.fill_(lr)
            self._lr_host = lr

    @torch.no_grad()
    def step(self):
        lib = _lib.load_library()
        self.sync_lr()
        act = [p for p in self.params if p.grad is not None]
        if not act:
            return
        st = _lib.stream()
        _lib.check(lib.facl_adam_prep(_lib.ptr(self._lr), _lib.ptr(self._step), self.betas[0], self.betas[1],
                                      _lib.ptr(self._consts), st), "facl_adam_prep")
        nt = len(act)
        arr = ctypes.c_void_p * nt
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in act]
        P = arr(*[p.data_ptr() for p in act])
        G = arr(*[g.data_ptr() for g in grads])
        M = arr(*[self.state[p]["exp_avg"].data_ptr() for p in act])
        V = arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in act])
        N = (ctypes.c_int * nt)(*[p.numel() for p in act])
        _lib.check(lib.facl_adam_apply(nt, P, G, M, V, N, _lib.ptr(self._consts), self.betas[0], self.betas[1], self.eps, st),
                   "facl_adam_apply")

    def state_dict(self):
        """torch.optim.Adam's layout (state by parameter index), so either optimizer can resume the other's run."""
        step = self._step.detach().clone().cpu()[0]
        return {"state": {i: {"step": step.clone(), "exp_avg": self.state[p]["exp_avg"], "exp_avg_sq": self.state[p]["exp_avg_sq"]}
                          for i, p in enumerate(self.params)},
                "param_groups": [{"lr": self.param_groups[0]["lr"], "betas": self.betas, "eps": self.eps, "weight_decay": 0,
                                  "amsgrad": False, "params": list(range(len(self.params)))}]}

    def load_state_dict(self, sd):
        for i, p in enumerate(self.params):
            s = sd["state"].get(i)
            if s is None:
                continue
            self.state[p]["exp_avg"].copy_(s["exp_avg"])
            self.state[p]["exp_avg_sq"].copy_(s["exp_avg_sq"])
            self._step.fill_(float(s["step"]))
        self.param_groups[0]["lr"] = sd["param_groups"][0]["lr"]
