"""Data-parallel plumbing: one process per GPU over RCCL (torch.distributed backend "nccl" on ROCm).

The reference is single-process (nn.DataParallel pinned to one device, cn3d_train_motion_GL.py:33,176);
its only collective helper, ``concat_all_gather`` (cn3d_model_conbag.py:559-570), is unreachable.  Here
clips are sharded across ranks and three exchanges make the R-rank step equal the 1-rank step at the
global batch:
  1. embeddings all-gather (autograd-aware) before the losses  -> cross-GPU negatives,
  2. SyncBN: all-reduce of the (sum, sumsq) / (dbeta, dgamma) fp64 buffers between kernel passes,
  3. gradient all-reduce (one flat 9.4 MB bucket), averaged.
"""
import datetime
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return 0, world if dist.is_initialized() else 1
    rank = int(os.environ["RANK"])
    if backend is None:
        # "nccl" IS RCCL on ROCm.  FACL_DIST_BACKEND=gloo rehearses the N>1 path on a single-GPU box.
        backend = os.environ.get("FACL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    # A mismatched or missing collective must RAISE, not wait: the default process-group timeout is 10 minutes, longer than
    # any launcher's patience.  With the watchdog's asynchronous error handling a timed-out RCCL call tears the rank down
    # (non-zero exit), the launcher then stops the other ranks.  FACL_DIST_TIMEOUT_S overrides (first-iteration RCCL
    # set-up of 8 ranks takes seconds, a 3 ms step never legitimately waits two minutes).
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    full_graph_env()
    timeout = datetime.timedelta(seconds=float(os.environ.get("FACL_DIST_TIMEOUT_S", "120")))
    dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world


def full_graph_env():
    """OPT-IN FACL_DP_GRAPH=full only (call before the process group exists).  Observed once in ~10 one-rank rehearsals of that mode
    (gpurun_out/r04_evidence.log, round 4): the process group's watchdog thread queried an event `last recorded in a capturing
    stream` (hipErrorCapturedEvent) and tore the process down with SIGABRT.  Two precautions, neither proven sufficient: no
    re-use of the process group's events through its cache, and (train_common.GraphedStep) a pause between the eager warm-up and
    the capture so that the watchdog has retired every eager collective before the first captured one is issued."""
    if os.environ.get("FACL_DP_GRAPH", "segments") == "full":
        os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")


class CaptureAborted(RuntimeError):
    """Raised on a rank that learns at a vote that ANOTHER rank's graph capture failed (its own was fine so far)."""


def vote(ok, group=None):
    """Agreement across ranks: MIN all-reduce of a success flag.  Returns True iff every rank voted True.  Used only while
    a step is being captured / validated (never inside the replayed step), so its host round trip costs nothing later.
    RCCL reduces device tensors, gloo host tensors."""
    if not (dist.is_available() and dist.is_initialized()):
        return bool(ok)
    on_dev = dist.get_backend(group) == "nccl"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class GraphSegments:
    """HIP-graph capture of a step that CONTAINS collectives: the capture is cut at every collective, so the step becomes
    [graph 0] collective [graph 1] collective ... [graph n] -- kernel segments replayed as graphs (no per-launch host cost),
    collectives issued eagerly between them on the same stream (the ordinary, everywhere-supported way to call RCCL).

    Why: the eager step is host-bound (160 launches through Python: 5.4 ms against 3.4 ms for the one-graph replay on one
    GPU), and capturing RCCL calls INSIDE a graph cannot be rehearsed on a one-GPU box.  All segments allocate from one
    private memory pool and are replayed in capture order, so every tensor a collective touches keeps its address; a
    collective therefore must not allocate its own output (callers allocate before the cut).  During the capture pass the
    collectives really execute (on garbage: no kernel of the step has run), which keeps the ranks in lockstep.

    Cuts happen wherever the step's Python runs -- including the autograd engine's worker thread -- hence the relaxed
    capture mode (a capture begun on one thread is ended on another).

    Failure protocol (a capture that fails on ONE rank must not leave the others replaying segments against its eager
    launches: mismatched collectives = a hang).  While capturing, every cut and the end of the capture carry one extra
    tiny collective, a `vote`: a rank whose segment raised votes "failed" exactly once (train_common.GraphedStep does, from
    its handler), which pairs with the others' next vote; they raise CaptureAborted there.  Up to that vote every rank has
    issued the same collectives, so all ranks leave the capture at the same point of the collective sequence and can
    continue TOGETHER on eager launches.  A failure inside a collective itself cannot be voted on: the process-group
    timeout (init_from_env) bounds it."""

    def __init__(self):
        self.pool = torch.cuda.graph_pool_handle()
        self.items = []                 # CUDAGraph objects and zero-argument callables, in replay order
        self.cur = None
        self.voting = dist.is_available() and dist.is_initialized()
        self.ncuts = 0
        # test hooks: "<rank>:<cut>" -- that rank's capture raises (FAIL) or the process dies (EXIT) when it reaches that cut
        self._inject = {}
        for key in ("FACL_TEST_CAPTURE_FAIL", "FACL_TEST_CAPTURE_EXIT"):
            v = os.environ.get(key)
            if v and self.voting:
                r, c = (int(x) for x in v.split(":"))
                if r == dist.get_rank():
                    self._inject[key] = c

    def begin(self):
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool, capture_error_mode="relaxed")

    def _close(self):
        self.cur.capture_end()
        self.items.append(self.cur)
        self.cur = None

    def _agree(self):
        if self.voting and not vote(True):
            raise CaptureAborted("graph capture failed on another rank (after %d collectives)" % self.ncuts)

    def cut(self, fn):
        if self._inject.get("FACL_TEST_CAPTURE_EXIT") == self.ncuts:
            os._exit(17)
        if self._inject.get("FACL_TEST_CAPTURE_FAIL") == self.ncuts:
            raise RuntimeError("injected capture failure at cut %d (FACL_TEST_CAPTURE_FAIL)" % self.ncuts)
        self._close()
        self._agree()
        fn()
        self.items.append(fn)
        self.ncuts += 1
        self.begin()

    def end(self):
        self._close()
        self._agree()

    def abort(self):
        if self.cur is not None:
            try:
                self.cur.capture_end()
            except Exception:
                pass
            self.cur = None

    def replay(self):
        with torch.no_grad():           # the collectives were recorded inside autograd Functions / hooks (no-grad contexts)
            for it in self.items:
                if isinstance(it, torch.cuda.CUDAGraph):
                    it.replay()
                else:
                    it()

    @property
    def n_graphs(self):
        return sum(isinstance(it, torch.cuda.CUDAGraph) for it in self.items)


class CaptureFailed(RuntimeError):
    """The segmented capture failed -- on EVERY rank, at the same point of the collective sequence (see GraphSegments)."""


def run_capture(rec, body, rank=0):
    """Run `body()` with `rec` recording (cut at every collective).  Returns body's result, or raises CaptureFailed on all
    ranks together: the rank whose segment raised votes "failed" once, which pairs with the next vote of the others."""
    set_recorder(rec)
    try:
        rec.begin()
        out = body()
        rec.end()
        return out
    except CaptureAborted as e:                  # learnt at a vote: the failure is already agreed on
        rec.abort()
        raise CaptureFailed(str(e)) from e
    except Exception as e:                       # this rank's segment raised: tell the others at their next vote
        rec.abort()
        if rec.voting:
            vote(False)
        raise CaptureFailed("graph-segment capture failed on rank %d after %d collectives: %s: %s"
                            % (rank, rec.ncuts, type(e).__name__, e)) from e
    finally:
        set_recorder(None)


_RECORDER = None       # the GraphSegments being captured, or None (eager)


def set_recorder(rec):
    global _RECORDER
    _RECORDER = rec


def collective(fn):
    """Every collective of this module goes through here: run now (eager) or cut the capture around it."""
    if _RECORDER is None:
        fn()
    else:
        _RECORDER.cut(fn)


def make_bn_reduce_fn(group=None):
    """In-place SUM all-reduce of an fp64 statistics buffer (SyncBN hook of the BN passes).

    OPT-IN ``FACL_ONESHOT_SYNCBN=1``: the reduction is ONE kernel launch on the current stream through peer-mapped mailboxes
    (facl_amd/mailbox.py, csrc/mailbox.hip) instead of a collective of the process group: no graph cut, no RCCL latency; sums are
    added in rank order (bit-identical on every rank).  Rehearsed with processes sharing one GPU only -- never the default."""
    if not is_distributed():
        return None
    oneshot = None
    if os.environ.get("FACL_ONESHOT_SYNCBN", "0") not in ("", "0"):
        from .mailbox import OneShotAllReduce
        oneshot = OneShotAllReduce(group, n_max=4608)        # the largest SyncBN buffer of the step: B2_V = 4608 doubles

    def reduce_fn(t):
        if oneshot is not None and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() <= oneshot.n_max:
            return oneshot(t)
        collective(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group))
        return t

    reduce_fn.world_size = dist.get_world_size(group)        # equal shards: global counts = local * world_size
    reduce_fn.oneshot = oneshot
    return reduce_fn


def _use_reduce_scatter(group=None):
    """Decided ONCE per process group, before any collective is issued: the backward of the embeddings all-gather is a
    reduce-scatter on RCCL (each rank needs only the sum of ITS rows: 1/R of the all-reduce traffic) and an all-reduce
    + slice on gloo (the CPU rehearsal backend has no reduce_scatter_tensor).  Every rank evaluates the same
    expression on the same backend, so the ranks cannot diverge; a failing collective raises and the rank exits."""
    key = id(group)
    if key not in _RS_DECISION:
        _RS_DECISION[key] = dist.get_backend(group) == "nccl" and hasattr(dist, "reduce_scatter_tensor")
    return _RS_DECISION[key]


_RS_DECISION = {}


# ---- pure layout functions of the embeddings exchange (no collective inside: tests/test_host_logic_cpu.py simulates
# R in {2,4,8} ranks with lists of tensors and holds the three of them to the single-process gradient)
def gathered_to_view_major(buf, G, R, Bl, C):
    """`all_gather_into_tensor` output (rank-major: rank r's (G*Bl, C) view-major rows at [r*G*Bl, (r+1)*G*Bl)) ->
    the global view-major rows (row = g*(R*Bl) + r*Bl + b), i.e. what one process holding all R*Bl clips would hold."""
    return buf.view(R, G, Bl, C).permute(1, 0, 2, 3).reshape(G * R * Bl, C)


def view_major_to_rank_chunks(g, G, R, Bl, C):
    """Gradient w.r.t. the global view-major rows -> the contiguous (R*G*Bl, C) input of `reduce_scatter_tensor`:
    chunk r (rows [r*G*Bl, (r+1)*G*Bl)) holds, in rank r's local view-major order, this rank's contribution to the
    gradient of rank r's rows.  Exact inverse of `gathered_to_view_major`."""
    return g.reshape(G, R, Bl, C).permute(1, 0, 2, 3).contiguous().view(R * G * Bl, C)


def local_rows_of(g_sum, G, R, r, Bl, C):
    """The all_reduce + slice form: rank r's (G*Bl, C) rows of the rank-summed global gradient."""
    return g_sum.reshape(G, R, Bl, C)[:, r].reshape(G * Bl, C)


class _AllGatherViewMajor(torch.autograd.Function):
    """(G*B_l, C) view-major local rows -> (G*R*B_l, C) view-major global rows (row = g*(R*B_l) + r*B_l + b).

    Forward semantics follow ``concat_all_gather`` (cn3d_model_conbag.py:559-570: gather -> cat on dim 0), plus the
    re-layout that the view-major row order needs (SURVEY hard part 5): ONE all_gather_into_tensor into a rank-major
    (R,G,B_l,C) buffer, then one permuted copy.  Backward = sum over ranks of the gathered gradient restricted to the
    local rows (the single-process loss back-propagates through the negatives)."""

    @staticmethod
    def forward(ctx, x, G, group):
        R = dist.get_world_size(group)
        r = dist.get_rank(group)
        Bl, C = x.shape[0] // G, x.shape[1]
        buf = torch.empty((R * G * Bl, C), dtype=x.dtype, device=x.device)      # concatenation along dim 0: rank-major
        xc = x.contiguous()
        collective(lambda: dist.all_gather_into_tensor(buf, xc, group=group))
        ctx.meta = (G, R, r, Bl, C, group)
        return gathered_to_view_major(buf, G, R, Bl, C)

    @staticmethod
    def backward(ctx, g):
        G, R, r, Bl, C, group = ctx.meta
        if _use_reduce_scatter(group):
            gin = view_major_to_rank_chunks(g, G, R, Bl, C)
            out = torch.empty((G * Bl, C), dtype=g.dtype, device=g.device)
            collective(lambda: dist.reduce_scatter_tensor(out, gin, op=dist.ReduceOp.SUM, group=group))
            return out, None, None
        g = g.contiguous()
        collective(lambda: dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group))
        return local_rows_of(g, G, R, r, Bl, C), None, None


def all_gather_view_major(x, G, group=None):
    if not is_distributed():
        return x
    return _AllGatherViewMajor.apply(x, G, group)


def allreduce_gradients(params, group=None):
    """Average the gradients across ranks through ONE flat fp32 bucket (2.36 M params = 9.4 MB:
    latency-bound on xGMI, so a single collective beats per-tensor calls)."""
    if not is_distributed():
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    collective(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group))
    flat.div_(dist.get_world_size(group))
    views, o = [], 0
    for g in grads:
        views.append(flat[o:o + g.numel()].view_as(g))
        o += g.numel()
    if hasattr(torch, "_foreach_copy_"):
        torch._foreach_copy_(grads, views)                   # one multi-tensor kernel instead of ~50 small copies
    else:
        for g, v in zip(grads, views):
            g.copy_(v)


class GradSync:
    """Gradient averaging with the big bucket overlapped with the rest of the backward.

    99 % of the 2.36 M parameters sit in the tail (net3DV_3 / netR_FC), whose gradients are complete before the
    set-abstraction backward (2 ms of kernels) starts.  Post-accumulate hooks count the tail parameters; when the
    last one has its gradient the flat bucket is all-reduced ASYNCHRONOUSLY (RCCL runs it on its own stream) and
    `finish()` -- called after backward -- reduces the small remainder, waits, scales and copies back.  The first
    step runs fully synchronous and records which parameters receive gradients at all (e.g. `mapping.weight` does not).
    """

    def __init__(self, named_params, group=None, early_prefixes=("net3DV_3.", "netR_FC.")):
        self.group = group
        self.named = [(k, p) for k, p in named_params if p.requires_grad]
        self.early_prefixes = early_prefixes
        self.early = None              # parameters of the overlapped bucket (known after the first step)
        self.late = None
        self.pending = 0
        self.work = None
        self.flat = None
        self.handles = []

    def _arm(self):
        self.pending = len(self.early)
        self.work, self.flat = None, None

    def reset(self):
        """After a step that did not run to its end (a failed graph capture): forget the half-counted hooks."""
        if self.early is not None:
            self._arm()

    def _hook(self, _p):
        self.pending -= 1
        if self.pending == 0:
            grads = [p.grad for p in self.early]
            self.flat = flat = torch.cat([g.reshape(-1) for g in grads])
            self.work = holder = {}          # the async handle of THIS launch (a replayed segment list re-runs the closure)

            def launch(flat=flat, holder=holder):
                holder["w"] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            collective(launch)

    @staticmethod
    def _scatter_back(flat, grads, world):
        flat.div_(world)
        views, o = [], 0
        for g in grads:
            views.append(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        if hasattr(torch, "_foreach_copy_"):
            torch._foreach_copy_(grads, views)
        else:
            for g, v in zip(grads, views):
                g.copy_(v)

    def finish(self):
        """Call after loss.backward(): every parameter gradient is the average over the ranks afterwards."""
        if not is_distributed():
            return
        world = dist.get_world_size(self.group)
        if self.early is None:                                   # first step: synchronous, learn the buckets
            allreduce_gradients([p for _, p in self.named], self.group)
            with_grad = [(k, p) for k, p in self.named if p.grad is not None]
            self.early = [p for k, p in with_grad if k.startswith(self.early_prefixes)]
            self.late = [p for k, p in with_grad if not k.startswith(self.early_prefixes)]
            if self.early and hasattr(self.early[0], "register_post_accumulate_grad_hook"):
                self.handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.early]
            else:
                self.late, self.early = self.early + self.late, []
            self._arm()
            return
        if self.late:
            allreduce_gradients(self.late, self.group)
        if self.early:
            if self.work is None:                                # a hook did not fire (unexpected): stay correct
                allreduce_gradients(self.early, self.group)
            else:
                collective(lambda holder=self.work: holder["w"].wait())
                self._scatter_back(self.flat, [p.grad for p in self.early], world)
        self._arm()
