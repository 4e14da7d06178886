"""ctypes binding of libfacl_hip.so (the C ABI declared in include/facl_hip.h).

PyTorch is only plumbing here: tensors provide device memory (``data_ptr()``) and the current
HIP stream.  ``import torch`` MUST precede the dlopen so that the library binds to the HIP
runtime PyTorch already loaded (same SONAME libamdhip64.so.7) -- stream handles are only
meaningful inside one runtime.
"""
import ctypes
import os

import torch  # noqa: F401  (must be loaded before libfacl_hip.so, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_d = ctypes.c_double
c_l = ctypes.c_longlong

# name -> argtypes; every function returns int.  Kept in one table so tests can check that the
# library exports every symbol the header declares.
SIGNATURES = {
    "facl_version": [],
    "facl_fps_f32": [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "facl_fps_f64": [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "facl_fps_reorder": [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p],
    "facl_group": [c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p],
    "facl_group_clips": [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p],
    "facl_gather_rows": [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_p],
    "facl_scatter_rows": [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p],
    "facl_sinkhorn": [c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "facl_kmeans": [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "facl_adam_prep": [c_p, c_p, c_f, c_f, c_p, c_p],
    "facl_adam_apply": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_p],
    "facl_ws_bytes": [],
    "facl_bn_finalize": [c_p, c_i, c_d, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_absmax": [c_p, c_l, c_p, c_p],
    "facl_rows_act_amax": [c_p, c_l, c_i, c_p, c_p, c_p, c_p],
    "facl_bn_eval_consts": [c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p],
    "facl_sa_x_moments": [c_p, c_l, c_i, c_p, c_p, c_p],
    "facl_bn1_sums_from_moments": [c_p, c_d, c_i, c_p, c_p, c_p, c_p],
    "facl_sa_bn1_chain": [c_p, c_d, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_l1tab": [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_fwd2": [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_fwd3": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_fwd3_h3": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_fwd3_f16": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_pool": [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_eval": [c_p, c_l, c_i] + [c_p] * 12,
    "facl_rows_stats": [c_p, c_l, c_i, c_p, c_p, c_p],
    "facl_rows_bn_relu": [c_p, c_l, c_i, c_p, c_p, c_p, c_p],
    "facl_rows_segmax": [c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p],
    "facl_rows_bwd_stats": [c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p],
    "facl_rows_bwd_apply": [c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p],
    "facl_rows_bwd_apply_amax": [c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p],
    "facl_segmax_bwd_stats": [c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p],
    "facl_segmax_bwd_stats_ymax": [c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_i, c_p],
    "facl_segmax_bwd_apply": [c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p],
    "facl_segmax_bwd_apply_amax": [c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "facl_gemm_fwd": [c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p],
    "facl_gemm_fwd_segmax": [c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_gemm_dgrad": [c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_p],
    "facl_gemm_wgrad": [c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_i, c_p],
    "facl_gemm_fwd_f16": [c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p],
    "facl_gemm_fwd_segmax_f16": [c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_gemm_dgrad_f16": [c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_p],
    "facl_gemm_wgrad_f16": [c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_i, c_p],
    "facl_gemm_wgrad_pro": [c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p],
    "facl_gemm_wgrad_pro_x3": [c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p],
    "facl_gemm_wgrad_h3": [c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    "facl_gemm_rs_wgrad_slices": [c_l, c_i, c_i],
    "facl_gemm_rs_wgrad": [c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_gemm_rs_planes_bytes": [c_i, c_i, c_i],
    "facl_gemm_rs_planes": [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p],
    "facl_gemm_rs_planes_multi": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_p],
    "facl_gemm_rs_supported": [c_l, c_i, c_i],
    "facl_gemm_rs_fwd": [c_p, c_l, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_gemm_rs_dgrad": [c_p, c_l, c_i, c_p, c_i, c_p, c_i, c_p, c_p],
    "facl_gemm_rs_dgrad_bnstats": [c_p, c_l, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_gemm_fwd_x3": [c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p],
    "facl_gemm_fwd_segmax_x3": [c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_gemm_dgrad_x3": [c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_p],
    "facl_gemm_wgrad_x3": [c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_i, c_p],
    "facl_sa_fwd3_x3": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_contrast": [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p],
    "facl_contrast_pair": [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p],
    "facl_normalize_map": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p],
    "facl_scale_rows2": [c_p, c_p, c_l, c_l, c_i, c_p, c_p, c_p],
    "facl_build_views_f32": [c_p, c_l, c_i, c_p, c_p, c_p, c_i, c_p, c_p],
    "facl_build_views_f64": [c_p, c_l, c_i, c_p, c_p, c_p, c_i, c_p, c_p],
    "facl_sa_bwd0": [c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_bwd1": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_bwd_w3": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_bwd2": [c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "facl_sa_bwd_consts3": [c_p, c_p, c_p, c_p, c_d, c_p, c_p, c_p],
    "facl_sa_bwd_consts2": [c_p, c_p, c_d, c_p, c_p],
    "facl_bn_bwd_consts": [c_p, c_p, c_i, c_d, c_p, c_p, c_p, c_p],
    "facl_rows_center_wgrad": [c_p, c_p, c_l, c_i, c_p, c_p, c_p],
    "facl_viewmax_fwd": [c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "facl_viewmax_bwd": [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "facl_viewmax_bwd_add": [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "facl_sa_bwd_final": [c_p] * 13 + [c_i, c_d] + [c_p] * 9 + [c_p],
    "facl_viewmax_stack": [c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "facl_fc_bn_stats": [c_p, c_l, c_l, c_i, c_p, c_p, c_p],
    "facl_fc_bn_apply": [c_p, c_l, c_l, c_i, c_p, c_p, c_i, c_i, c_d, c_d, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p],
    "facl_fc_bn_bwd_stats": [c_p, c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_p],
    "facl_fc_bn_bwd_apply": [c_p, c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_d, c_d, c_p, c_p, c_p, c_p, c_p],
    "facl_col_sums": [c_p, c_l, c_i, c_p, c_p],
    "facl_contrast_pair_sum": [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p],
    "facl_gemm_wgrad_acc": [c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_i, c_p],
    "facl_mailbox_bytes": [c_i, c_i],
    "facl_mailbox_alloc": [c_l, c_p, c_p],
    "facl_mailbox_open": [c_p, c_p],
    "facl_mailbox_close": [c_p],
    "facl_mailbox_free": [c_p],
    "facl_mailbox_allreduce": [c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p],
}
RESTYPE_I64 = {"facl_ws_bytes", "facl_gemm_rs_planes_bytes", "facl_mailbox_bytes"}


def lib_path():
    # FACL_LIB: load another build of the same ABI (kernel experiments: A/B timing, parity of a candidate build)
    return os.environ.get("FACL_LIB") or os.path.join(_HERE, "libfacl_hip.so")


def load_library():
    """dlopen the HIP library; raises RuntimeError (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: facl_amd has no CPU/eager fallback. Build it with "
            "`python -m facl_amd.build` (hipcc --offload-arch=gfx950).")
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = ctypes.c_longlong if name in RESTYPE_I64 else ctypes.c_int
    _LIB = lib
    return lib


_ERR = {-1: "unsupported shape", -2: "NULL pointer", -3: "misaligned pointer", -4: "unsupported configuration"}


def check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, f"hipError {rc}")
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")


AMAX_WORDS = 2048          # include/facl_hip.h: FACL_AMAX_WORDS


def amax_buffers(n, device, zero=True):
    """(n, AMAX_WORDS) int32: n operand-maximum buffers of the fp16x3 GEMMs (csrc/common.h).  Producers either STORE a bound
    into every slot (facl_bn_finalize) or RAISE slots with atomics (facl_sa_pool, facl_absmax, the BatchNorm-backward row
    kernels): the latter need zeros to start from -- `zero=False` when every row is stored or zeroed by a kernel of the step
    itself (facl_bn_finalize's `zamax`), which spares the fill launch."""
    if zero:
        return torch.zeros((n, AMAX_WORDS), dtype=torch.int32, device=device)
    return empty((n, AMAX_WORDS), dtype=torch.int32, device=device)


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


# ---- side stream: leaf work beside the critical chain (OPT-IN, measured slower) -------------------------------------------------
# Parts of the step are chains of launch-latency-bound kernels that occupy a few CUs each (the FC head / loss block: ~20 us
# GEMMs of one workgroup per CU, single-workgroup finalisations).  Work that nothing downstream of the chain waits for --
# the weight / bias gradients of the FC head, F.normalize + mapping -- can go to ONE side stream per device (a parallel branch
# of the captured graph), joined before its results are handed on.  MEASURED (round 4, same box, gpurun_out/r5c_ab.log):
# 3.225 ms per step with the three branches against 3.11 ms without -- a fork / join inside a HIP graph costs far more than
# the ~60 us of small kernels it takes off the chain (cross-queue dependencies instead of in-order dispatch).  Default off;
# FACL_SIDE_STREAM=1 switches the branches on.
SIDE_STREAM = os.environ.get("FACL_SIDE_STREAM", "0") != "0"
_SIDE = {}


class fork:
    """``with fork() as f: ...`` -- the launches inside go to the device's side stream, which first waits for everything
    issued so far on the current stream; ``f.join(*tensors)`` makes the current stream wait for the side stream (once per
    fork; tensors allocated inside are marked as used by the current stream).  ``fork(enabled=False)`` is a no-op."""

    def __init__(self, enabled=True):
        self.on = bool(enabled) and SIDE_STREAM
        self.joined = False

    def __enter__(self):
        if self.on:
            self.main = torch.cuda.current_stream()
            key = (self.main.device.index,)
            if key not in _SIDE:
                _SIDE[key] = torch.cuda.Stream(device=self.main.device)
            self.side = _SIDE[key]
            if self.side.cuda_stream == self.main.cuda_stream:       # already inside a fork: stay there
                self.on = False
                return self
            self.side.wait_stream(self.main)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self._ctx.__exit__(*exc)
            if exc[0] is not None:                                    # never leave a dangling branch behind an exception
                self.main.wait_stream(self.side)
                self.joined = True
        return False

    def join(self, *tensors):
        if self.on and not self.joined:
            self.main.wait_stream(self.side)
            self.joined = True
            for t in tensors:
                if t is not None:
                    t.record_stream(self.main)


_PENDING = []          # forks whose join was left to the caller of the model (facl_amd.cn3d_model_conbag: lazy_code)


def join_pending():
    while _PENDING:
        f, ts = _PENDING.pop()
        f.join(*ts)


# ---- debug switch: poisoned scratch ------------------------------------------------------------------------------------
# Every output / scratch tensor the host layer hands to a kernel is `torch.empty` (the kernels write every element
# they later read).  With POISON on (FACL_POISON=1, or `with poisoned():`) those tensors and the partial-sum workspace are
# filled with NaN / 0xFF bytes first, so a partial row or an output element that a launch leaves unwritten for some
# ragged shape shows up as NaN in the results instead of silently reading whatever the allocator recycled.
POISON = os.environ.get("FACL_POISON", "0") not in ("", "0")


class poisoned:
    def __enter__(self):
        global POISON
        self.prev, POISON = POISON, True
        return self

    def __exit__(self, *exc):
        global POISON
        POISON = self.prev
        return False


def _poison(t):
    if POISON and t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        else:
            t.view(torch.uint8).fill_(0xFF)
    return t


def empty(*size, **kw):
    return _poison(torch.empty(*size, **kw))


def empty_like(t, **kw):
    return _poison(torch.empty_like(t, **kw))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("facl_amd ops run on the GPU only (tensor is on %s); there is no CPU path" % t.device)


# ---- debug taps: integer side outputs of a forward that no caller of the reference's interface ever sees ----------------------
# When TAPS is a dict (tests: tests/helpers.routing_taps), forward passes drop the argmax tensors of their max-pools here
# ("sa_arg": max over the K neighbours, "seg_arg": my_max_pool over the S centroids, "view_arg": the view maximum), so a
# reference evaluation can route its gradients through the same positions (tie-proof gradient parity).
TAPS = None


def tap(name, t):
    if TAPS is not None:
        TAPS[name] = t.detach().clone()


def tap_relu(name, y=None, scale=None, shift=None, a=None):
    """ReLU decisions of a layer as a bool tensor: `a > 0` where the activation exists, else the sign of the kernels' own
    fma(scale, y, shift) (BatchNorm folded into two constants per channel) evaluated exactly: the product of two fp32
    numbers and the sum are exact enough in fp64 that the sign equals the sign of the fp32 FMA."""
    if TAPS is not None:
        TAPS[name] = (a > 0) if a is not None else ((y.double() * scale.double() + shift.double()) > 0)


# ---- optional in-step kernel timing (bench.py's roofline section) ----------------------------------------------------
# When TIMING is a dict, `timed(label)` brackets the launches issued inside the `with` block with HIP events on the
# launch stream (eager execution only: events cannot be recorded inside a graph replay) and appends the event pair to
# TIMING[label] (created on first use, so bench.py sees every heavy entry point of the step without a fixed list).
TIMING = None


class timed:
    def __init__(self, label):
        self.on = TIMING is not None
        self.label = label

    def __enter__(self):
        if self.on:
            import torch
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.on:
            import torch
            self.e1.record(torch.cuda.current_stream())
            TIMING.setdefault(self.label, []).append((self.e0, self.e1))
        return False


def timing_ms(label):
    """Average milliseconds of the recorded brackets of `label` (synchronises)."""
    evs = TIMING.get(label)
    if not evs:
        return None
    evs[-1][1].synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / len(evs)


def timing_table():
    """{label: (average ms per bracket, number of brackets)} for everything recorded so far (synchronises)."""
    import torch
    torch.cuda.synchronize()
    return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in TIMING.items() if v}
