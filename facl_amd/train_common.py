"""Shared body of the two training entries (the reference ships two scripts that differ in three
lines: default ``branch_choose``, data root and checkpoint file name -- cn3d_train_apperance_GL.py:135,161,341)."""
import argparse
import logging
import os
import random
import time

import numpy as np
import torch

from . import cn3d_model_conbag as MODELL
from . import dist as fdist
from .utils_my import contrastive_losses_stacked, group_points_3DV, knn_radius_group


def build_parser(default_branch):
    """Flags of cn3d_train_motion_GL.py:77-135 (names, types and defaults unchanged) plus the additions
    marked NEW: synthetic data (the NTU files are not redistributable), real grouping parameters
    (the reference overrides knn_K / ball_radius with literals inside the grouper, utils_my.py:260-261)."""
    p = argparse.ArgumentParser(description="Training")
    p.add_argument('--batchSize', type=int, default=64, help='input batch size')
    p.add_argument('--nepoch', type=int, default=100, help='number of epochs to train for')
    p.add_argument('--INPUT_FEATURE_NUM', type=int, default=4, help='number of input point features')
    p.add_argument('--temperal_num', type=int, default=3, help='number of input point features')
    p.add_argument('--pooling', type=str, default='concatenation')
    p.add_argument('--dataset', type=str, default='ntu60')
    p.add_argument('--weight_decay', type=float, default=0.0008, help='weight decay (SGD only)')
    p.add_argument('--learning_rate', type=float, default=0.0003, help='learning rate at t=0')
    p.add_argument('--momentum', type=float, default=0.9, help='momentum (SGD only)')
    p.add_argument('--workers', type=int, default=0, help='number of data loading workers')
    p.add_argument('--root_path', type=str, default='../ntu/ntu60_new2/raw/', help='preprocess folder')
    p.add_argument('--depth_path', type=str, default='', help='raw_depth_png')
    p.add_argument('--save_root_dir', type=str, default='../ntu/ntu60_new2/model/', help='output folder')
    p.add_argument('--model', type=str, default='', help='model name for training resume')
    p.add_argument('--optimizer', type=str, default='', help='optimizer name for training resume')
    p.add_argument('--ngpu', type=int, default=1, help='# GPUs')
    p.add_argument('--main_gpu', type=int, default=0, help='main GPU id')
    p.add_argument('--emb_dims', type=int, default=1024)
    p.add_argument('--k', type=int, default=20)
    p.add_argument('--dropout', type=float, default=0.05)
    p.add_argument('--learning_rate_decay', type=float, default=1e-7)
    p.add_argument('--size', type=str, default='full')
    p.add_argument('--SAMPLE_NUM', type=int, default=512, help='number of sample points')
    p.add_argument('--Num_Class', type=int, default=512, help='number of outputs')
    p.add_argument('--knn_K', type=int, default=64, help='K for knn search')
    p.add_argument('--sample_num_level1', type=int, default=64, help='number of first layer groups')
    p.add_argument('--sample_num_level2', type=int, default=64, help='number of second layer groups')
    p.add_argument('--ball_radius', type=float, default=0.16, help='square of radius for ball query in level 1')
    p.add_argument('--ball_radius2', type=float, default=0.25, help='square of radius for ball query in level 2')
    p.add_argument('--ex_feature', type=int, default=7)
    p.add_argument('--save_feature_dir', type=str, default='../ntu/ntu60_new2/features/')
    p.add_argument('--save_label_dir', type=str, default='../ntu/ntu60_new2/labels/')
    p.add_argument('--branch_choose', type=str, default=default_branch)
    # NEW
    p.add_argument('--synthetic', type=int, default=1,
                   help='NEW: 1 = iid U[-0.5,0.5) clouds (no NTU data here); 2 = synthetic RAW clips (the four (rows, 8) clouds the '
                        'loader reads per video) through the GPU view construction (facl_amd.views.build_views = the dataset '
                        "class's get_data_train, cn3D_data_set.py:285-350): needs --num_crop 10 --SAMPLE_NUM 512 --INPUT_FEATURE_NUM 4")
    p.add_argument('--view_rng', type=str, default='numpy', choices=('numpy', 'device'),
                   help='NEW (--synthetic 2): numpy = draw the view construction\'s random numbers on the host in the reference\'s '
                        'NumPy order; device = draw them on the GPU (same distributions, another stream)')
    p.add_argument('--num_crop', type=int, default=10, help='NEW: views per clip (literal 10 at :189)')
    p.add_argument('--steps_per_epoch', type=int, default=8, help='NEW: synthetic iterations per epoch')
    p.add_argument('--group_radius', type=float, default=None,
                   help='NEW: r^2 of the grouper (default: the reference literals 0.06 at N=512, 0.16 otherwise)')
    p.add_argument('--log_file', type=str, default='', help='NEW: log path (reference: ../ntu/ntu60_new2/30_0425.log)')
    p.add_argument('--swa_if', type=int, default=0, help='NEW: 1 = add 0.6 * the SwAV term (literal swa_if = 0 at :238)')
    p.add_argument('--cld_if', type=int, default=0, help='NEW: 1 = add the CLD k-means term (literal cld_if = 0 at :319)')
    p.add_argument('--precision', type=str, default='f32', choices=('f32', 'x3b', 'x3'),
                   help='NEW: arithmetic of the dense contractions (facl_amd.tail.precision): f32 = fp32-grade (default); '
                        'x3b = three bf16 products in the backward GEMMs only (features / loss unchanged); x3 = everywhere')
    p.add_argument('--graph', type=int, default=1,
                   help='NEW: 1 = replay the iteration as HIP graph(s) (train_common.GraphedStep; the eager loop is host-bound at '
                        'this step time); needs the default loss (swa_if = cld_if = 0).  0 = eager launches')
    p.add_argument('--fps_reorder', type=int, default=0,
                   help='NEW: 1 = FPS-reorder every view on the GPU before grouping (cn3D_data_set.py:665-672; the '
                        'reference assumes FPS-ordered clouds but its live loader never calls it)')
    return p


def synthetic_batch(B, G, N, D, device, generator=None):
    """(B,G,N,D) float32 iid U[-0.5,0.5), the bench / parity input distribution."""
    return torch.rand(B, G, N, D, device=device, generator=generator) - 0.5


def appearance_batch(B, G, N, D, device, generator=None):
    """(B,G,N,D) float32 appearance-style synthetic clips (BASELINE configs[2]; SURVEY hard part 7).

    What the appearance branch feeds the same model (cn3D_data_set.py:125-137, generate_NTU.py:249-264): per clip ONE
    voxelised body surface -- 30 mm voxels normalised by the body height, i.e. coordinates on a ~1/64 grid with y in
    [-0.5,0.5], a narrower x and a thin depth relief, 4th channel an appearance value in [-0.5,0.5] -- and every view
    = N rows drawn WITH replacement from it (get_data_train, :287-318), so clouds contain duplicated rows (exact
    distance ties in the kNN) and grid-aligned neighbours.  Views g >= 1 are jittered (sigma 0.01, clip 0.05, :767-778),
    odd views x-mirrored (:708-713); view 0 is the raw resample and keeps its exact duplicates."""
    P0 = max(64, int(0.6 * N))
    r = lambda *shape: torch.rand(*shape, device=device, generator=generator)
    bx = (r(B, P0) - 0.5) * 0.44
    by = r(B, P0) - 0.5
    bz = 0.08 * torch.sin(6.0 * bx) * torch.cos(4.0 * by) + (r(B, P0) - 0.5) * 0.04
    base = torch.stack((bx, by, bz), dim=-1)
    base = torch.round(base * 64.0) / 64.0                                     # voxel grid
    if D == 4:
        app = torch.round((r(B, P0, 1) - 0.5) * 16.0) / 16.0
        base = torch.cat((base, app), dim=-1)
    idx = torch.randint(0, P0, (B, G, N), device=device, generator=generator)
    out = torch.gather(base.unsqueeze(1).expand(B, G, P0, D), 2, idx.unsqueeze(-1).expand(B, G, N, D)).clone()
    if G > 1:
        noise = (0.01 * torch.randn(B, G - 1, N, 3, device=device, generator=generator)).clamp_(-0.05, 0.05)
        out[:, 1:, :, :3] += noise
        out[:, 1::2, :, 0] *= -1.0
    return out.float()


class ContrastiveStep:
    """One training iteration = the loop body of cn3d_train_motion_GL.py:224-335."""

    def __init__(self, netR, optimizer, opt, num_crop, group_radius=None, fps_reorder=False, swa_if=0, cld_if=0):
        self.netR, self.optimizer, self.opt, self.G = netR, optimizer, opt, num_crop
        self.swa_if, self.cld_if = int(swa_if), int(cld_if)
        self.swav_state = None
        self.epoch = 0
        self.r2 = group_radius
        self.fps_reorder = fps_reorder
        self._one = None
        self.rank = torch.distributed.get_rank() if fdist.is_distributed() else 0
        self.grad_sync = fdist.GradSync(list(netR.named_parameters())) if fdist.is_distributed() else None

    def group(self, data1):
        opt = self.opt
        if self.r2 is None and opt.SAMPLE_NUM == 512:
            return group_points_3DV(data1, opt)                                   # :230 (K=64, r^2=0.06 literals)
        r2 = self.r2 if self.r2 is not None else 0.16                             # group_points_3DV_2048's literal
        opt.INPUT_FEATURE_NUM = data1.shape[-1]
        return knn_radius_group(data1, opt.sample_num_level1, opt.knn_K, r2)     # (M,N,D) or clip-major (B,G,N,D)

    def __call__(self, out_points, epoch=0, order=None):
        netR, G = self.netR, self.G
        if order is None:
            order = np.arange(0, G, 1)
            np.random.shuffle(order)                                               # :297-298
        if not torch.is_tensor(order):
            order = torch.as_tensor(np.asarray(order), dtype=torch.long).to(out_points.device)
        self.epoch = epoch
        return self.run(out_points, order)

    def run(self, out_points, order):
        """Device-only body (no host round trips): this is what GraphedStep captures into a HIP graph.
        `out_points`: the loader's clip-major (B,G,N,D) batch, or -- 3-dimensional -- the view-major (G*B,N,D) rows that
        facl_amd.views.build_views writes directly (the permute + reshape of :226 already done)."""
        netR, G = self.netR, self.G
        if out_points.dim() == 3:
            M_, N, D = out_points.shape
            B = M_ // G
            data1 = out_points if out_points.dtype == torch.float32 else out_points.float()
            if self.fps_reorder:
                from .fps import fps_sample_data
                data1 = fps_sample_data(data1, self.opt.sample_num_level1,
                                        start_idx=torch.zeros(data1.shape[0], dtype=torch.int32, device=data1.device))
            xt, yt = self.group(data1)
            return self._encode_and_step(xt, yt, B, order)
        B, G_, N, D = out_points.shape
        if self.fps_reorder or out_points.dtype != torch.float32 or (self.r2 is None and self.opt.SAMPLE_NUM == 512):
            data1 = out_points.permute(1, 0, 2, 3).reshape(-1, N, D).float()      # :226-228 (view-major rows)
            if self.fps_reorder:                                                   # FPS picks first (start index 0)
                from .fps import fps_sample_data
                data1 = fps_sample_data(data1, self.opt.sample_num_level1,
                                        start_idx=torch.zeros(data1.shape[0], dtype=torch.int32, device=data1.device))
            xt, yt = self.group(data1)
        else:
            xt, yt = self.group(out_points)        # clip-major batch: the grouping kernel reads view-major in place
        return self._encode_and_step(xt, yt, B, order)

    def _encode_and_step(self, xt, yt, B, order):
        netR, G = self.netR, self.G
        # x_nor / code feed the SwAV / CLD terms only: without them F.normalize + mapping run beside the loss block
        netR.lazy_code = not (self.swa_if or self.cld_if) and not fdist.is_distributed()
        x, code, x_nor, x_global = netR(xt, yt, 1)                                 # :234
        x_keys = fdist.all_gather_view_major(x, G)
        off = self.rank * B
        # global (:265-287) + circle (:290-316) losses: similarity GEMMs + one HIP kernel each (csrc/loss.hip)
        from .tail import precision as _precision
        with _precision(getattr(netR, "precision", "f32")):    # the similarity GEMMs follow the model's arithmetic
            loss_c, loss_circle, loss = contrastive_losses_stacked(G, netR._stacked, order, x_keys=None if x_keys is x else x_keys,
                                                                   clip_offset=off, with_sum=True)
        # loss = loss_circle + loss_c (:329; swa, CLD terms are 0 ...): the fp32 sum comes out of the loss launch itself
        if self.swa_if:                                                            # ... unless switched on: :239-263
            from . import swav_cld
            if self.swav_state is None:
                self.swav_state = swav_cld.SwavState(B, G, x_nor.shape[1])
            self.swav_state.maybe_create(self.epoch, x.device)
            loss = loss + 0.6 * swav_cld.swav_loss(code, x_nor, netR.mapping.weight, self.swav_state)
        if self.cld_if:                                                            # :319-326
            from . import swav_cld
            loss = loss + swav_cld.cld_loss(x_nor, B, G)
        self.optimizer.zero_grad(set_to_none=True)
        if self._one is None or self._one.device != loss.device:
            self._one = torch.ones((), dtype=torch.float32, device=loss.device)    # the seed of backward(): no fill launch per step
        loss.backward(self._one)
        if self.grad_sync is not None:
            self.grad_sync.finish()                                                # tail bucket overlapped with the SA backward
        from . import _lib as _flib
        _flib.join_pending()                                                       # the side-stream branch of x_nor / code
        self.optimizer.step()
        return loss, loss_c, loss_circle


class GraphCaptureFailed(RuntimeError):
    """The step could not be captured (or its replay did not reproduce the eager step).  Model, BatchNorm buffers and
    optimizer state are back at their values from before the attempt.  Under data parallelism EVERY rank raises it at the
    same point of the collective sequence (facl_amd/dist.py: GraphSegments' failure protocol), so all ranks may continue
    together on eager launches; any other exception out of GraphedStep under world > 1 is not agreed on and must end the rank."""


_CAPTURE_STREAMS = {}


def _capture_stream(dev):
    """One side stream per device for warm-up + capture (the scratch workspace of the HIP passes is keyed by stream:
    a fresh stream per GraphedStep would pin another workspace each time)."""
    key = (dev.type, dev.index)
    if key not in _CAPTURE_STREAMS:
        _CAPTURE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _CAPTURE_STREAMS[key]


class GraphedStep:
    """The whole training iteration (grouping -> forward -> losses -> backward -> Adam) captured once into a HIP
    graph and replayed: ~250 kernel launches per step collapse into one graph launch, so the step is no longer bound
    by host launch latency.  Under data parallelism the capture is cut at every collective and the kernel segments between
    them replay as graphs (collectives stay eager: facl_amd/dist.py: GraphSegments); the optimizer must keep its step counter on the device
    (facl_amd.optim.FusedAdam, or torch.optim.Adam(capturable=True, lr=<tensor>)).  Learning-rate changes between
    replays: FusedAdam's device-side lr is refreshed before every replay (`sync_lr`); with torch's capturable Adam only
    a TENSOR lr updated in place is seen by the graph.

    Failure: GraphCaptureFailed, with the training state restored (the three warm-up calls are REAL optimizer steps).  Under
    data parallelism the segmented replay is additionally VALIDATED before it is trusted: one eager step and one replayed
    step from the same state must give the same loss on every rank (`validate`, default on when distributed)."""

    def __init__(self, step, example_points, G, restore=False, validate=None):
        """`restore`: put parameters, BatchNorm buffers and optimizer state back to what they were before the three
        warm-up steps, so that a training run continues exactly where an eager run would be."""
        self.step, self.G = step, G
        dev = example_points.device
        self.points = example_points.clone()
        self.order = torch.arange(G, dtype=torch.long, device=dev)
        self.graph = self.segments = None
        distributed = fdist.is_distributed()
        validate = distributed if validate is None else validate
        snap = self._snapshot()
        try:
            self._capture(distributed, dev)
            if validate and (self.segments is not None or getattr(self, "full_dp_graph", False)):
                self._validate()
        except Exception:
            self.graph = self.segments = None
            torch.cuda.synchronize()
            self._restore(snap)                          # three real optimizer steps (and maybe more) ran on `example_points`
            step.optimizer.zero_grad(set_to_none=True)
            if step.grad_sync is not None:
                step.grad_sync.reset()
            raise
        if restore:
            self._restore(snap)

    # ---- training state = parameters + BatchNorm buffers (+ their host-side counters) + optimizer state
    def _snapshot(self):
        step = self.step
        net = {k: v.detach().clone() for k, v in step.netR.state_dict().items()}
        opt = None
        if hasattr(step.optimizer, "_step"):             # FusedAdam.state_dict() shares its moment tensors: deep copy
            sd = step.optimizer.state_dict()
            opt = {"state": {i: {k: v.clone() for k, v in st.items()} for i, st in sd["state"].items()},
                   "param_groups": sd["param_groups"]}
        return net, opt, [(m, m.steps) for m in self._bn_modules()]

    def _restore(self, snap):
        net, opt, steps = snap
        with torch.no_grad():                            # in place: a captured graph holds the addresses of these tensors
            cur = self.step.netR.state_dict()
            for k, v in net.items():
                cur[k].copy_(v)
        for m, n in steps:
            m.steps = n
        if opt is not None:
            self.step.optimizer.load_state_dict(opt)

    def _capture(self, distributed, dev):
        step = self.step
        s = _capture_stream(dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                       # warm-up on a side stream (allocator + lazy inits)
            for _ in range(3):
                step.run(self.points, self.order)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # The capture pass runs the Python forward (no kernel executes), so the host-side num_batches_tracked counters
        # would run ahead of the running-statistics updates: restore them afterwards.  NOTE: the 3 warm-up calls above
        # are REAL optimizer steps on `example_points` (they also settle the allocator and Adam's lazy state).
        saved = [(m, m.steps) for m in self._bn_modules()]
        if distributed and os.environ.get("FACL_DP_GRAPH", "segments") == "full":
            # OPT-IN experiment (never the default): the collectives are captured INSIDE one graph (torch's RCCL process group
            # joins a stream capture the way it does on NCCL), so the data-parallel step has no cut at all.  Rehearsed with a
            # 1-rank RCCL group only (bench.py --rehearse-dp 1); with N > 1 ranks it has never run: the validation below
            # (eager step == replayed step, voted across ranks) is what stands between a wrong replay and the measurement.
            # (facl_amd/dist.py: full_graph_env -- the watchdog must have retired the warm-up's eager collectives before a
            # captured one is issued: it polls every ~100 ms)
            import time
            time.sleep(0.5)
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph, stream=s):
                    self.out = step.run(self.points, self.order)
            except Exception as e:
                ok = False
                err = "%s: %s" % (type(e).__name__, e)
            else:
                ok, err = True, ""
            if not fdist.vote(ok):
                raise GraphCaptureFailed("full-graph capture of the data-parallel step failed on some rank%s" % (" (here: %s)" % err if err else ""))
            self.graph = graph
            self.full_dp_graph = True
        elif distributed:
            # data parallel: the capture is cut at every collective (facl_amd/dist.py: GraphSegments) -- kernel segments
            # replay as graphs, the collectives in between are ordinary eager RCCL calls on the same stream
            rec = fdist.GraphSegments()
            s.wait_stream(torch.cuda.current_stream())
            try:
                with torch.cuda.stream(s):
                    self.out = fdist.run_capture(rec, lambda: step.run(self.points, self.order), step.rank)
            except fdist.CaptureFailed as e:             # agreed on by every rank (facl_amd/dist.py: run_capture)
                raise GraphCaptureFailed(str(e)) from e
            self.segments = rec
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
        else:
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph, stream=s):
                    self.out = step.run(self.points, self.order)
            except Exception as e:
                raise GraphCaptureFailed("graph capture failed: %s: %s" % (type(e).__name__, e)) from e
            self.graph = graph
        for m, n in saved:
            m.steps = n

    def _validate(self):
        """One eager step and one replayed step from the same state must agree (they run the same kernels in the same
        order: the losses are equal to the last bits) -- on EVERY rank, or nobody replays.  What this catches is what a
        one-GPU rehearsal cannot: an ordering lost between the collective library's stream and the next graph segment."""
        step = self.step
        snap = self._snapshot()
        loss_e = float(step.run(self.points, self.order)[0])
        self._restore(snap)
        if self.segments is not None:
            self.segments.replay()
        else:
            self.graph.replay()
        loss_g = float(self.out[0])
        self._restore(snap)
        ok = loss_e == loss_e and abs(loss_e - loss_g) <= 1e-5 * abs(loss_e)
        if not fdist.vote(ok):
            raise GraphCaptureFailed("replayed graph segments do not reproduce the eager step (this rank: eager loss %r, "
                                     "replayed %r)" % (loss_e, loss_g))

    def _bn_modules(self):
        return [m for m in self.step.netR.modules() if hasattr(m, "count_batch")]

    def __call__(self, out_points, epoch=0, order=None):
        if order is None:
            order = np.arange(0, self.G, 1)
            np.random.shuffle(order)
        self.points.copy_(out_points, non_blocking=True)
        self.order.copy_(torch.as_tensor(np.asarray(order), dtype=torch.long), non_blocking=True)
        self.step.epoch = epoch
        sync = getattr(self.step.optimizer, "sync_lr", None)
        if sync is not None:
            sync()                                       # a StepLR change made since the capture reaches k_adam_prep
        if self.segments is not None:
            self.segments.replay()
        else:
            self.graph.replay()
        for m in self._bn_modules():                     # host-side num_batches_tracked (netR_FC.1 counts twice)
            m.count_batch()
        self.step.netR.netR_FC[1].count_batch()
        return self.out


def lr_for_epoch(base_lr, epoch, step_size=4, gamma=0.7):
    """StepLR(4, 0.7) stepped with the explicit epoch every iteration (:181,:333) = closed form."""
    return base_lr * gamma ** (epoch // step_size)


def run(default_branch, ckpt_pattern, args=None):
    opt = build_parser(default_branch).parse_args(args)
    print(opt)
    local = int(os.environ.get("LOCAL_RANK", opt.main_gpu))
    torch.cuda.set_device(local)               # before the process group: RCCL binds its communicator to the current device
    device = torch.device("cuda", local)
    rank, world = fdist.init_from_env()

    opt.manualSeed = 1
    random.seed(opt.manualSeed)
    torch.manual_seed(opt.manualSeed)
    np.random.seed(opt.manualSeed)          # shared circle-loss permutation across ranks
    os.makedirs(opt.save_root_dir, exist_ok=True)
    if opt.log_file:
        logging.basicConfig(format='%(asctime)s %(message)s', datefmt='%Y/%m/%d %H:%M:%S',
                            filename=opt.log_file, level=logging.INFO)
    logging.info('======================================================')

    num_crop = opt.num_crop
    netR = MODELL.PointNet_Plus(opt, gost=num_crop).to(device)
    netR.precision = opt.precision
    netR.bn_reduce_fn = fdist.make_bn_reduce_fn()
    # cn3d_train_motion_GL.py:180: the same update as torch.optim.Adam, all tensors in one HIP launch (facl_amd/optim.py)
    from .optim import FusedAdam
    optimizer = FusedAdam(netR.parameters(), lr=opt.learning_rate, betas=(0.5, 0.999), eps=1e-06)
    step = ContrastiveStep(netR, optimizer, opt, num_crop, opt.group_radius, bool(opt.fps_reorder), opt.swa_if, opt.cld_if)
    gen = torch.Generator(device=device)
    gen.manual_seed(1000 + rank)
    view_rng = np.random.RandomState(2000 + rank)         # --synthetic 2: the generator the view construction draws from

    run_step = step
    for epoch in range(0, opt.nepoch):
        netR.train()
        for g in optimizer.param_groups:
            g["lr"] = lr_for_epoch(opt.learning_rate, epoch)
        loss_sigma, t0 = 0.0, time.time()
        for i in range(opt.steps_per_epoch):
            if opt.synthetic == 2:
                # the loop body from the loader's output on (:224-228): raw clips -> the 10 augmented views of every clip,
                # built on the GPU in one launch, view-major float32 (facl_amd/views.py; draws in the reference's NumPy order)
                if (num_crop, opt.SAMPLE_NUM, opt.INPUT_FEATURE_NUM) != (10, 512, 4):
                    raise RuntimeError("--synthetic 2 builds the reference's 10 views of 512 points x 4 channels: "
                                       "use --num_crop 10 --SAMPLE_NUM 512 --INPUT_FEATURE_NUM 4")
                from .views import build_views, synthetic_raw_clip
                base = ((epoch * opt.steps_per_epoch + i) * world + rank) * opt.batchSize
                clips = [synthetic_raw_clip(base + b) for b in range(opt.batchSize)]
                # --view_rng numpy: the reference's NumPy stream (a seed reproduces its views; ~0.2 ms of host draws per clip);
                # device: the same distributions drawn by a torch generator on the GPU (no per-clip host work)
                out_points = build_views(clips, view_rng, device, device_rng=gen if opt.view_rng == "device" else None)
            elif opt.synthetic == 1:
                out_points = synthetic_batch(opt.batchSize, num_crop, opt.SAMPLE_NUM, opt.INPUT_FEATURE_NUM, device, gen)
            else:
                raise RuntimeError("only --synthetic 1 / 2 are supported: the NTU dataset file pipeline "
                                   "(cn3D_data_set.py: video lists, .npy loading) is outside this repository's scope")
            if run_step is step and opt.graph and not (opt.swa_if or opt.cld_if):
                try:                                     # capture on the first batch; state restored: same trajectory as eager
                    run_step = GraphedStep(step, out_points, num_crop, restore=True)
                except GraphCaptureFailed as e:
                    # training state is back at its pre-capture values and, under data parallelism, EVERY rank is here
                    # (GraphSegments' votes): all ranks continue together on eager launches.  Anything else raised under
                    # world > 1 is not agreed on across ranks: it propagates, the rank exits non-zero and the launcher
                    # stops the others (no rank is left replaying segments against another's eager collectives).
                    print("graph capture failed (%s); running eager" % e)
                    opt.graph = 0
            loss, _, _ = run_step(out_points, epoch)
            torch.cuda.synchronize()
            lv = loss.item()
            if lv != lv or lv in (float("inf"), float("-inf")):
                # inputs / a checkpoint with NaN or inf (the arithmetic itself has no range limit any more: DESIGN 3.0):
                # stop here instead of training on garbage
                raise FloatingPointError("non-finite loss %r at epoch %d, iteration %d" % (lv, epoch, i))
            loss_sigma += lv
        clips = opt.batchSize * opt.steps_per_epoch * world / (time.time() - t0)
        logging.info('{} --epoch{} ==Average loss:{}'.format('Valid', epoch, loss_sigma / (i + 1)))
        if rank == 0:
            print('epoch:', epoch, 'loss mode is :', 1, '--loss:', loss_sigma / (i + 1), '| clips/s: %.1f' % clips)
            if epoch % 5 == 0:
                torch.save(netR.state_dict(), ckpt_pattern % (opt.save_root_dir, epoch))
    return netR
