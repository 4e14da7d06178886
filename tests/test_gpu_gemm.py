"""GPU parity of the fp32 MFMA GEMMs (forward with fused prologue/epilogue, dgrad, split-K wgrad) vs fp64."""
import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _poisoned_scratch():
    """Every test of this module runs with NaN-poisoned scratch (facl_amd._lib.poisoned): the partial-sum workspace and
    every output / scratch tensor the host layer allocates are filled with NaN bytes before the launches, so a partial
    row or output element left unwritten at a ragged shape fails the comparison instead of reading recycled memory."""
    from facl_amd import _lib
    with _lib.poisoned():
        yield


def _ws():
    from facl_amd.sa_mlp import _Workspace
    return _Workspace.get(torch.device(DEV))


@pytest.mark.parametrize("M,K,N,pro,ctr", [(256, 64, 128, False, False), (4096, 256, 256, True, True),
                                           (1000, 512, 384, True, False), (32, 1024, 1024, False, False),
                                           (768, 1024, 512, False, False), (130, 36, 200, True, True),
                                           (800, 1000, 1024, False, False), (80, 260, 500, True, True)])   # last two: in-workgroup split-K, ragged stages
def test_gemm_fwd(M, K, N, pro, ctr):
    from facl_amd import _lib
    lib = _lib.load_library()
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    a = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    ps = torch.rand(K, device=DEV, generator=g) + 0.5 if pro else None
    pt = torch.randn(K, device=DEV, generator=g) * 0.3 if pro else None
    cen = torch.randn(M, 3, device=DEV, generator=g) if ctr else None
    Wc = torch.randn(N, 3, device=DEV, generator=g) if ctr else None
    y = _lib.empty(M, N, device=DEV)
    sums = _lib.empty(N, 2, dtype=torch.float64, device=DEV)
    p = _lib.ptr
    _lib.check(lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), p(ps), p(pt), p(cen), p(Wc), 3, p(y), p(sums), p(_ws()),
                                 _lib.stream()), "gemm_fwd")
    a64 = a.double()
    if pro:
        a64 = torch.relu(a64 * ps.double() + pt.double())
    ref = a64 @ W.double().t() + b.double()
    if ctr:
        ref = ref + cen.double() @ Wc.double().t()
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert rel_err(sums[:, 0].cpu().numpy(), ref.sum(0).cpu().numpy()) < 1e-5 * max(1.0, float(ref.abs().sum(0).max() / ref.sum(0).abs().max()))
    assert rel_err(sums[:, 1].cpu().numpy(), (ref * ref).sum(0).cpu().numpy()) < 1e-5


@pytest.mark.parametrize("M,N,K,ldw,off", [(4096, 256, 256, 256, 0), (1000, 384, 520, 523 + 1, 4), (32, 512, 1024, 1024, 0),
                                           (49152 // 8, 1024, 512, 512, 0), (800, 1024, 1024, 1024, 0), (80, 260, 100, 100, 0)])
def test_gemm_dgrad(M, N, K, ldw, off):
    from facl_amd import _lib
    lib = _lib.load_library()
    g = torch.Generator(device=DEV).manual_seed(M + N)
    dy = torch.randn(M, N, device=DEV, generator=g)
    Wfull = torch.randn(N, ldw, device=DEV, generator=g) / N ** 0.5
    W = Wfull[:, off:off + K]
    da = _lib.empty(M, K, device=DEV)
    _lib.check(lib.facl_gemm_dgrad(_lib.ptr(dy), M, N, W.data_ptr(), ldw, K, _lib.ptr(da), _lib.stream()), "dgrad")
    ref = dy.double() @ W.double()
    assert rel_err(da.cpu().numpy(), ref.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("M,N,K,nz", [(4096, 256, 256, 4), (49152 // 4, 512, 256, 12), (1000, 128, 384, 3), (32, 1024, 1024, 1),
                                      (768, 512, 1024, 2), (800, 1024, 1024, 4), (300, 256, 192, 2)])
def test_gemm_wgrad(M, N, K, nz):
    from facl_amd import _lib
    lib = _lib.load_library()
    g = torch.Generator(device=DEV).manual_seed(M + K)
    dy = torch.randn(M, N, device=DEV, generator=g)
    a = torch.randn(M, K, device=DEV, generator=g)
    dW = _lib.empty(N, K, device=DEV)
    slices = _lib.empty(nz * N * K, device=DEV)
    _lib.check(lib.facl_gemm_wgrad(_lib.ptr(dy), _lib.ptr(a), M, N, K, K, _lib.ptr(dW), _lib.ptr(slices), nz,
                                   _lib.stream()), "wgrad")
    ref = dy.double().t() @ a.double()
    assert rel_err(dW.cpu().numpy(), ref.cpu().numpy()) < 3e-6


_SBK_H3_CHILD = r"""
import sys, torch
sys.path.insert(0, ".")
from facl_amd import _lib
lib = _lib.load_library(); p = _lib.ptr; dev = torch.device("cuda:0")
from facl_amd.sa_mlp import _Workspace
ws = _Workspace.get(dev)
worst = 0.0
def rel(x, ref): return float((x.double() - ref).abs().max() / ref.abs().max())
g = torch.Generator(device=dev).manual_seed(3)
for M, K, N, mag_a, mag_w in ((800, 1024, 1024, 1.0, 0.03), (800, 1000, 512, 3e-6, 40.0), (80, 260, 500, 2e4, 1e-5), (768, 512, 768, 1.0, 1.0)):
    a = torch.randn(M, K, device=dev, generator=g) * mag_a
    a[:, : K // 2] *= 1e-3                                   # stages of very different magnitude inside one contraction
    W = torch.randn(N, K, device=dev, generator=g) * mag_w
    b = torch.randn(N, device=dev, generator=g) * mag_a * mag_w
    y = torch.full((M, N), float("nan"), device=dev)
    _lib.check(lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), None, None, None, None, 0, p(y), None, p(ws), _lib.stream()), "fwd")
    worst = max(worst, rel(y, a.double() @ W.double().t() + b.double()))
    dy = torch.randn(M, N, device=dev, generator=g) * 1e-4
    da = torch.full((M, K), float("nan"), device=dev)
    _lib.check(lib.facl_gemm_dgrad(p(dy), M, N, p(W), K, K, p(da), _lib.stream()), "dgrad")
    worst = max(worst, rel(da, dy.double() @ W.double()))
    dW = torch.full((N, K), float("nan"), device=dev); sl = torch.empty(N * K, device=dev)
    _lib.check(lib.facl_gemm_wgrad(p(dy), p(a), M, N, K, K, p(dW), p(sl), 1, _lib.stream()), "wgrad")
    worst = max(worst, rel(dW, dy.double().t() @ a.double()))
print("WORST %.3e" % worst)
"""


def test_few_row_gemms_optin_self_scaled_fp16x3_is_fp32_grade():
    """OPT-IN FACL_SBK_H3=1 (k_gemm_sbk NP = 4: every 64 x 32 stage scaled by the power of two of its own maximum): forward, dgrad
    and weight gradient of few-row shapes -- ragged stages, operands of very different and of tiny / huge magnitude -- stay at
    fp32-GEMM level against fp64.  The switch is read once per process: a child process."""
    import os, subprocess, sys
    env = dict(os.environ, FACL_SBK_H3="1")
    r = subprocess.run([sys.executable, "-c", _SBK_H3_CHILD], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    worst = float(r.stdout.strip().splitlines()[-1].split()[1])
    print("self-scaled fp16x3 worst relative error", worst)
    assert worst < 3e-6


@pytest.mark.parametrize("K,N,ctr", [(256, 256, True), (256, 512, False), (512, 1024, False)])
def test_tail_layers_headline_size_vs_torch_fp64(K, N, ctr):
    """The three net3DV_3 layers at the headline row count (B*T*S = 49,152 centroid rows) through the Python wrappers the
    encoder uses (forward + fused BN statistics [+ centre term], dgrad, split-K wgrad with the production slice count)
    against fp64 matmuls on the GPU; the error must stay at fp32-GEMM level (rocBLAS sgemm sits at 3-4e-7 here)."""
    from facl_amd import tail
    M = 49152
    g = torch.Generator(device=DEV).manual_seed(K + N)
    a = torch.relu(torch.randn(M, K, device=DEV, generator=g))            # post-ReLU activations: half zeros
    W = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    cen = torch.randn(M, 3, device=DEV, generator=g) if ctr else None
    Wc = torch.randn(N, 3, device=DEV, generator=g).contiguous() if ctr else None
    dy = torch.randn(M, N, device=DEV, generator=g) * 1e-3
    y, sums = tail.gemm_fwd(a, W, b, want_stats=True, centers=cen, Wc=Wc)
    ref = a.double() @ W.double().t() + b.double()
    if ctr:
        ref = ref + cen.double() @ Wc.double().t()
    assert float((y.double() - ref).norm() / ref.norm()) < 6e-7
    assert float((sums[:, 0] - ref.sum(0)).abs().max() / ref.abs().sum(0).max()) < 1e-6
    assert float(((sums[:, 1] - (ref * ref).sum(0)).abs() / (ref * ref).sum(0)).max()) < 1e-6
    del ref, y
    da = tail.gemm_dgrad(dy, W)
    ref = dy.double() @ W.double()
    assert float((da.double() - ref).norm() / ref.norm()) < 6e-7
    del ref, da
    dW = tail.gemm_wgrad(dy, a)
    ref = dy.double().t() @ a.double()
    assert float((dW.double() - ref).norm() / ref.norm()) < 2e-6          # 49,152-term sums, slices added in fp32


def test_gemm_fwd_segmax_fused_epilogue():
    """facl_gemm_fwd_segmax: y, BN sums, and per (64-row block, column) max of sgn*y with the FIRST argmax (ties included)."""
    from facl_amd import _lib
    lib = _lib.load_library()
    M, K, N = 64 * 512, 64, 1024                                        # 256 x 8 tiles of 128x128: the fused kernel applies
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.randn(M, K, device=DEV, generator=g)
    a[64:128] = a[64:65]                                                # a whole block of identical rows: exact ties
    W = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    sgn = torch.where(torch.rand(N, device=DEV, generator=g) < 0.5, -1.0, 1.0)
    y = _lib.empty(M, N, device=DEV)
    sums = _lib.empty(N, 2, dtype=torch.float64, device=DEV)
    ymax = _lib.empty(M // 64, N, device=DEV)
    arg = _lib.empty(M // 64, N, dtype=torch.int32, device=DEV)
    p = _lib.ptr
    _lib.check(lib.facl_gemm_fwd_segmax(p(a), M, K, p(W), K, N, p(b), p(sgn), p(y), p(sums), p(ymax), p(arg), p(_ws()),
                                        _lib.stream()), "gemm_fwd_segmax")
    ref = a.double() @ W.double().t() + b.double()
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert rel_err(sums[:, 1].cpu().numpy(), (ref * ref).sum(0).cpu().numpy()) < 1e-5
    sy = (y * sgn).view(M // 64, 64, N)                                 # the kernel's own fp32 y: max / argmax are exact
    want, _ = sy.max(dim=1)
    assert torch.equal(ymax, want)
    first = (sy == want.unsqueeze(1)).int().argmax(dim=1).int()          # first row attaining the max
    assert torch.equal(arg, first)
    assert int(arg[1].max()) == 0                                        # the all-ties block resolves to row 0
    # too small for the 128x128-tile kernel: the entry point says so and launches nothing
    assert lib.facl_gemm_fwd_segmax(p(a), 128, K, p(W), K, 256, p(b), p(sgn), p(y), None, p(ymax), p(arg), p(_ws()),
                                    _lib.stream()) == -4


@pytest.mark.parametrize("M,K,N", [(4096, 256, 512), (800, 1024, 1024), (1000, 104, 200)])
def test_gemm_x3_twins_within_their_bound(M, K, N):
    """The opt-in "bf16x3" entries (two bf16 pieces per operand, three products): forward / dgrad / wgrad against fp64.
    Bound stated in include/facl_hip.h: <= 3 * 2^-16 per product; measured ~1e-5 of the result's scale."""
    from facl_amd import _lib
    lib = _lib.load_library()
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    a = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    dy = torch.randn(M, N, device=DEV, generator=g)
    p, st = _lib.ptr, _lib.stream()
    y, y6 = _lib.empty(M, N, device=DEV), _lib.empty(M, N, device=DEV)
    sums = _lib.empty(N, 2, dtype=torch.float64, device=DEV)
    _lib.check(lib.facl_gemm_fwd_x3(p(a), M, K, p(W), K, N, p(b), None, None, None, None, 0, p(y), p(sums), p(_ws()), st), "fwd_x3")
    _lib.check(lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), None, None, None, None, 0, p(y6), None, p(_ws()), st), "fwd")
    ref = a.double() @ W.double().t() + b.double()
    e3, e6 = rel_err(y.cpu().numpy(), ref.cpu().numpy()), rel_err(y6.cpu().numpy(), ref.cpu().numpy())
    assert e6 < 2e-6 and e6 < e3 < 5e-5, (e3, e6)                      # really the three-product arithmetic, inside its bound
    assert rel_err(sums[:, 1].cpu().numpy(), (ref * ref).sum(0).cpu().numpy()) < 5e-5
    da = _lib.empty(M, K, device=DEV)
    _lib.check(lib.facl_gemm_dgrad_x3(p(dy), M, N, p(W), K, K, p(da), st), "dgrad_x3")
    assert rel_err(da.cpu().numpy(), (dy.double() @ W.double()).cpu().numpy()) < 5e-5
    nz = 4
    dW = _lib.empty(N, K, device=DEV)
    sl = _lib.empty(nz * N * K, device=DEV)
    _lib.check(lib.facl_gemm_wgrad_x3(p(dy), p(a), M, N, K, K, p(dW), p(sl), nz, st), "wgrad_x3")
    assert rel_err(dW.cpu().numpy(), (dy.double().t() @ a.double()).cpu().numpy()) < 5e-5


def _rs_planes(lib, W, transposed, Wc=None, half=0):
    from facl_amd import _lib
    N, K = W.shape
    nb = lib.facl_gemm_rs_planes_bytes(K if transposed else N, N if transposed else K, 1 if Wc is not None else 0)
    planes = _lib.empty(nb, dtype=torch.uint8, device=DEV)
    _lib.check(lib.facl_gemm_rs_planes(_lib.ptr(W), W.stride(0), N, K, int(transposed), _lib.ptr(Wc), 3, half, _lib.ptr(planes),
                                       _lib.stream()), "rs_planes")
    return planes


@pytest.mark.parametrize("M,K,N,pro,ctr,seg", [(4096, 256, 256, False, True, False), (4096, 256, 512, True, False, False),
                                               (6144, 512, 1024, True, False, True), (4000, 64, 256, True, True, False),
                                               (2048 + 96, 128, 512, False, False, False), (49152 // 4, 512, 1024, True, False, True)])
def test_gemm_rs_fwd_equals_gemm_fwd_and_fp64(M, K, N, pro, ctr, seg):
    """Row-streamed forward (csrc/gemm_rs.hip: pre-split weight planes through an LDS-DMA ring, activation rows through
    per-wave slots, BN + ReLU prologue, centre k-step, fused statistics / my_max_pool) vs facl_gemm_fwd[_segmax] on the same
    inputs -- same arithmetic, so the outputs must agree to the last bit -- and vs an fp64 product.  Ragged row counts
    (M % 256 != 0, M % 32 != 0) included."""
    from facl_amd import _lib
    lib = _lib.load_library()
    assert lib.facl_gemm_rs_supported(M, K, N) == 1
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    a = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    ps = torch.rand(K, device=DEV, generator=g) + 0.5 if pro else None
    pt = torch.randn(K, device=DEV, generator=g) * 0.3 if pro else None
    cen = torch.randn(M, 3, device=DEV, generator=g) if ctr else None
    Wc = torch.randn(N, 3, device=DEV, generator=g).contiguous() if ctr else None
    sgn = torch.randn(N, device=DEV, generator=g) if seg else None
    p = _lib.ptr
    planes = _rs_planes(lib, W, False, Wc)
    y, sums = _lib.empty(M, N, device=DEV), _lib.empty(N, 2, dtype=torch.float64, device=DEV)
    ymax = _lib.empty(M // 64, N, device=DEV) if seg else None
    arg = _lib.empty(M // 64, N, dtype=torch.int32, device=DEV) if seg else None
    _lib.check(lib.facl_gemm_rs_fwd(p(a), M, K, p(planes), 0, None, N, p(b), p(ps), p(pt), p(cen), p(y), p(sums), p(sgn), p(ymax), p(arg),
                                    p(_ws()), _lib.stream()), "rs_fwd")
    # reference 1: fp64
    a64 = a.double()
    if pro:
        a64 = torch.relu(a64 * ps.double() + pt.double())
    ref = a64 @ W.double().t() + b.double()
    if ctr:
        ref = ref + cen.double() @ Wc.double().t()
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert rel_err(sums[:, 1].cpu().numpy(), (ref * ref).sum(0).cpu().numpy()) < 1e-5
    assert rel_err(sums[:, 0].cpu().numpy(), ref.sum(0).cpu().numpy()) < 1e-5 * max(1.0, float(ref.abs().sum(0).max() / ref.sum(0).abs().max()))
    # fused my_max_pool: exactly the first maximum of sign(sgn) * y over each block of 64 rows of the kernel's own y
    if seg:
        sy = (y * torch.where(sgn < 0, -1.0, 1.0)).view(M // 64, 64, N)
        assert torch.equal(ymax, sy.max(dim=1).values)
        first = (sy == ymax.unsqueeze(1)).float().argmax(dim=1)
        assert torch.equal(arg.long(), first)
    # reference 2: the LDS-staged kernel, same arithmetic -> bit-equal (the centre term is an epilogue FMA there and a
    # k-step here: with it only to rounding)
    y2, sums2 = _lib.empty(M, N, device=DEV), _lib.empty(N, 2, dtype=torch.float64, device=DEV)
    _lib.check(lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), p(ps), p(pt), p(cen), p(Wc), 3, p(y2), p(sums2), p(_ws()),
                                 _lib.stream()), "gemm_fwd")
    if not ctr:
        assert torch.equal(y, y2)
    else:
        assert rel_err(y.cpu().numpy(), y2.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("M,N,K", [(8192, 256, 256), (8100, 512, 256), (6144, 1024, 512)])
def test_gemm_rs_dgrad_equals_gemm_dgrad_and_fp64(M, N, K):
    from facl_amd import _lib
    lib = _lib.load_library()
    assert lib.facl_gemm_rs_supported(M, N, K) == 1
    g = torch.Generator(device=DEV).manual_seed(M + N)
    dy = torch.randn(M, N, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) / N ** 0.5
    planes = _rs_planes(lib, W, True)
    da, da2 = _lib.empty(M, K, device=DEV), _lib.empty(M, K, device=DEV)
    _lib.check(lib.facl_gemm_rs_dgrad(_lib.ptr(dy), M, N, _lib.ptr(planes), 0, None, K, _lib.ptr(da), _lib.stream()), "rs_dgrad")
    _lib.check(lib.facl_gemm_dgrad(_lib.ptr(dy), M, N, _lib.ptr(W), K, K, _lib.ptr(da2), _lib.stream()), "dgrad")
    ref = dy.double() @ W.double()
    assert rel_err(da.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert torch.equal(da, da2)


@pytest.mark.parametrize("M,N,K", [(8192, 512, 256), (8100, 1024, 512)])
def test_gemm_rs_dgrad_bnstats_equals_the_rows_pass(M, N, K):
    """facl_gemm_rs_dgrad_bnstats: da identical to facl_gemm_rs_dgrad, and the fused BatchNorm-backward column sums equal
    facl_rows_bwd_stats(da, y, bnc) (fp64 accumulation there, fp32 over 32 rows then fp64 here) and an fp64 evaluation."""
    from facl_amd import _lib
    lib = _lib.load_library()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    dy = torch.randn(M, N, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) / N ** 0.5
    y = torch.randn(M, K, device=DEV, generator=g)
    bnc = torch.stack((torch.randn(K, device=DEV, generator=g) * 0.1, torch.rand(K, device=DEV, generator=g) + 0.5,
                       torch.randn(K, device=DEV, generator=g), torch.randn(K, device=DEV, generator=g) * 0.3,
                       torch.zeros(K, device=DEV))).contiguous()                      # mean | invstd | scale | shift | (unused)
    planes = _rs_planes(lib, W, True)
    p = _lib.ptr
    da, da2 = _lib.empty(M, K, device=DEV), _lib.empty(M, K, device=DEV)
    sums, sums2 = _lib.empty(K, 2, dtype=torch.float64, device=DEV), _lib.empty(K, 2, dtype=torch.float64, device=DEV)
    _lib.check(lib.facl_gemm_rs_dgrad_bnstats(p(dy), M, N, p(planes), 0, None, K, p(da), p(y), p(bnc), p(sums), p(_ws()), _lib.stream()), "bnstats")
    _lib.check(lib.facl_gemm_rs_dgrad(p(dy), M, N, p(planes), 0, None, K, p(da2), _lib.stream()), "dgrad")
    assert torch.equal(da, da2)
    _lib.check(lib.facl_rows_bwd_stats(p(da), p(y), M, K, p(bnc), p(sums2), p(_ws()), _lib.stream()), "rows_bwd_stats")
    d = torch.where(bnc[2] * y + bnc[3] > 0, da, torch.zeros_like(da)).double()
    yhat = ((y - bnc[0]) * bnc[1]).double()
    ref = torch.stack((d.sum(0), (d * yhat).sum(0)), 1)
    scale = torch.stack((d.abs().sum(0), (d * yhat).abs().sum(0)), 1).clamp_min(1e-30)   # sums with cancellation: bound by the magnitudes
    assert float(((sums - ref).abs() / scale).max()) < 2e-6
    assert float(((sums - sums2).abs() / scale).max()) < 2e-6


@pytest.mark.parametrize("M,N,K,pro", [(8192, 1024, 512, True), (6000, 1024, 512, False), (4096 + 40, 2048, 256, True)])
def test_gemm_rs_wgrad_vs_fp64_and_staged_kernel(M, N, K, pro):
    """Register-streamed weight gradient (k_wgrad_rs: dy^T fragments loaded in operand shape, activation recomputed and
    split once per stage by the workgroup, fragment-ordered LDS planes) vs fp64 and vs facl_gemm_wgrad on a materialised
    activation; ragged row counts (M % 32 != 0) and the slice boundaries included."""
    from facl_amd import _lib
    lib = _lib.load_library()
    nz = lib.facl_gemm_rs_wgrad_slices(M, N, K)
    assert nz >= 1
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    dy = torch.randn(M, N, device=DEV, generator=g)
    y = torch.randn(M, K, device=DEV, generator=g)
    ps = torch.rand(K, device=DEV, generator=g) + 0.5 if pro else None
    pt = torch.randn(K, device=DEV, generator=g) * 0.3 if pro else None
    dW, sl = _lib.empty(N, K, device=DEV), _lib.empty(nz * N * K, device=DEV)
    p = _lib.ptr
    _lib.check(lib.facl_gemm_rs_wgrad(p(dy), p(y), M, N, K, p(ps), p(pt), None, None, p(dW), p(sl), _lib.stream()), "rs_wgrad")
    a = torch.relu(y * ps + pt) if pro else y
    a64 = torch.relu(y.double() * ps.double() + pt.double()) if pro else y.double()
    ref = dy.double().t() @ a64
    assert rel_err(dW.cpu().numpy(), ref.cpu().numpy()) < 3e-6
    dW2, sl2 = _lib.empty(N, K, device=DEV), _lib.empty(8 * N * K, device=DEV)
    _lib.check(lib.facl_gemm_wgrad(p(dy), p(a), M, N, K, K, p(dW2), p(sl2), 8, _lib.stream()), "wgrad")
    assert rel_err(dW.cpu().numpy(), dW2.cpu().numpy()) < 2e-6
    assert lib.facl_gemm_rs_wgrad_slices(49152, 512, 256) == 0          # too few output blocks: left to the staged kernel


def _act_amax(lib, a, ps=None, pt=None, cen=None):
    """The FACL_AMAX_WORDS buffer facl_gemm_rs_fwd (half = 1) takes its row operand's power-of-two scale from: the measured
    maximum of f(a) (facl_rows_act_amax with a prologue, facl_absmax without) and of the centre coordinates."""
    from facl_amd import _lib
    p = _lib.ptr
    amax = _lib.amax_buffers(1, a.device)[0]
    if ps is not None:
        _lib.check(lib.facl_rows_act_amax(p(a), a.shape[0], a.shape[1], p(ps), p(pt), p(amax), _lib.stream()), "rows_act_amax")
    else:
        _lib.check(lib.facl_absmax(p(a), a.numel(), p(amax), _lib.stream()), "absmax")
    if cen is not None:
        _lib.check(lib.facl_absmax(p(cen), cen.numel(), p(amax), _lib.stream()), "absmax")
    return amax


@pytest.mark.parametrize("M,K,N,pro,ctr,seg", [(4096, 256, 256, False, True, False), (4000, 256, 512, True, False, False),
                                               (6144, 512, 1024, True, False, True)])
def test_gemm_rs_fwd_fp16x3_is_fp32_grade(M, K, N, pro, ctr, seg):
    """The forward arithmetic of the row-streamed GEMM: fp16x3 (csrc/common.h: operands pre-scaled by exact powers of two,
    split into two fp16 planes, three products per multiply-add, fp32 accumulation).  Held to the SAME 2e-6 bound against an
    fp64 product as the bf16x6 kernels, compared with them and with torch's fp32 matmul on the same inputs (the measured
    errors are printed), on activation-like inputs that include tiny values (fp16 subnormal second pieces) and exact zeros."""
    from facl_amd import _lib
    lib = _lib.load_library()
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    a = torch.randn(M, K, device=DEV, generator=g) * 1.5
    a[:, ::7] *= 1e-3                                                   # a band of tiny activations
    a[:, 3::11] = 0.0
    W = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    ps = torch.rand(K, device=DEV, generator=g) + 0.5 if pro else None
    pt = torch.randn(K, device=DEV, generator=g) * 0.3 if pro else None
    cen = torch.rand(M, 3, device=DEV, generator=g) - 0.5 if ctr else None
    Wc = torch.randn(N, 3, device=DEV, generator=g).contiguous() if ctr else None
    sgn = torch.randn(N, device=DEV, generator=g) if seg else None
    p = _lib.ptr
    outs = {}
    amax = _act_amax(lib, a, ps, pt, cen)
    for half in (1, 0):
        planes = _rs_planes(lib, W, False, Wc, half)
        y, sums = _lib.empty(M, N, device=DEV), _lib.empty(N, 2, dtype=torch.float64, device=DEV)
        ymax = _lib.empty(M // 64, N, device=DEV) if seg else None
        arg = _lib.empty(M // 64, N, dtype=torch.int32, device=DEV) if seg else None
        _lib.check(lib.facl_gemm_rs_fwd(p(a), M, K, p(planes), half, p(amax) if half else None, N, p(b), p(ps), p(pt), p(cen), p(y),
                                        p(sums), p(sgn), p(ymax), p(arg), p(_ws()), _lib.stream()), "rs_fwd")
        outs[half] = (y, sums, ymax, arg)
    a64 = a.double()
    if pro:
        a64 = torch.relu(a64 * ps.double() + pt.double())
    ref = a64 @ W.double().t() + b.double()
    if ctr:
        ref = ref + cen.double() @ Wc.double().t()
    a32 = torch.relu(a * ps + pt) if pro else a
    t32 = a32 @ W.t() + b + (cen @ Wc.t() if ctr else 0.0)
    e_h3, e_x6, e_t = (rel_err(v.cpu().numpy(), ref.cpu().numpy()) for v in (outs[1][0], outs[0][0], t32))
    print(f"fp16x3 {e_h3:.2e}   bf16x6 {e_x6:.2e}   torch fp32 {e_t:.2e}")
    assert e_h3 < 2e-6 and e_h3 < 2.0 * max(e_x6, e_t)
    y, sums, ymax, arg = outs[1]
    assert rel_err(sums[:, 1].cpu().numpy(), (ref * ref).sum(0).cpu().numpy()) < 1e-5
    if seg:
        sy = (y * torch.where(sgn < 0, -1.0, 1.0)).view(M // 64, 64, N)
        assert torch.equal(ymax, sy.max(dim=1).values)
        assert torch.equal(arg.long(), (sy == ymax.unsqueeze(1)).float().argmax(dim=1))


def _bn_consts(K, g):
    return torch.stack((torch.randn(K, device=DEV, generator=g) * 0.1, torch.rand(K, device=DEV, generator=g) + 0.5,
                        torch.randn(K, device=DEV, generator=g), torch.randn(K, device=DEV, generator=g) * 0.3,
                        torch.zeros(K, device=DEV))).contiguous()                     # mean | invstd | scale | shift | (unused)


def _clear_of_the_relu_edge(y, bnc):
    """Entries whose scale*y + shift is within rounding of 0 would open or close the ReLU depending on fma vs mul+add; the
    reference expression below is not the kernel's instruction sequence, so move them well inside the open side."""
    t = bnc[2].double() * y.double() + bnc[3].double()
    return torch.where(t.abs() < 1e-4, ((1.0 - bnc[3]) / bnc[2]).expand_as(y), y).contiguous()


@pytest.mark.parametrize("mag", [1.0, 3e-9, 7e5])
@pytest.mark.parametrize("M,N,K", [(8100, 1024, 512), (8192, 256, 256)])
def test_gemm_rs_backward_fp16x3_dynamic_scale(M, N, K, mag):
    """Backward fp16x3: the gradient operand's power-of-two scale is chosen on the device from max|dy|, which the kernel that
    WRITES dy maintains (facl_rows_bwd_apply_amax).  Gradients of magnitude 3e-9 .. 7e5 with a heavy tail (a few entries
    1000x the rest) and a band of exact zeros: the published maximum is exact, dgrad (+ fused BatchNorm-backward sums) and
    the weight gradient meet the bf16x6 kernels' fp64 bounds."""
    from facl_amd import _lib
    lib = _lib.load_library()
    p = _lib.ptr
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    # dy comes out of the BatchNorm-backward rows pass, as in the model
    dout = torch.randn(M, N, device=DEV, generator=g) * mag
    dout[::97, ::13] *= 1000.0
    dout[:, 5::17] = 0.0
    yprev = torch.randn(M, N, device=DEV, generator=g)
    bnc_n = _bn_consts(N, g)
    kk = torch.randn(2, N, device=DEV, generator=g) * (1e-3 * mag)
    dy, dy2 = _lib.empty(M, N, device=DEV), _lib.empty(M, N, device=DEV)
    amax = torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device=DEV)
    _lib.check(lib.facl_rows_bwd_apply_amax(p(dout), p(yprev), M, N, p(bnc_n), p(kk), p(dy), p(amax), _lib.stream()), "apply_amax")
    _lib.check(lib.facl_rows_bwd_apply(p(dout), p(yprev), M, N, p(bnc_n), p(kk), p(dy2), _lib.stream()), "apply")
    assert torch.equal(dy, dy2)
    assert float(amax.view(torch.float32).max()) == float(dy.abs().max())
    # ---- dgrad (+ BN-backward sums)
    W = torch.randn(N, K, device=DEV, generator=g) / N ** 0.5
    bnc = _bn_consts(K, g)
    y = _clear_of_the_relu_edge(torch.randn(M, K, device=DEV, generator=g), bnc)
    ref = dy.double() @ W.double()
    das = {}
    for half in (1, 0):
        planes = _rs_planes(lib, W, True, None, half)
        da, da2 = _lib.empty(M, K, device=DEV), _lib.empty(M, K, device=DEV)
        sums = _lib.empty(K, 2, dtype=torch.float64, device=DEV)
        am = p(amax) if half else None
        _lib.check(lib.facl_gemm_rs_dgrad(p(dy), M, N, p(planes), half, am, K, p(da), _lib.stream()), "dgrad")
        _lib.check(lib.facl_gemm_rs_dgrad_bnstats(p(dy), M, N, p(planes), half, am, K, p(da2), p(y), p(bnc), p(sums), p(_ws()),
                                                  _lib.stream()), "bnstats")
        assert torch.equal(da, da2)
        das[half] = (rel_err(da.cpu().numpy(), ref.cpu().numpy()), sums)
    print(f"dgrad fp16x3 {das[1][0]:.2e}   bf16x6 {das[0][0]:.2e}")
    assert das[1][0] < 2e-6 and das[1][0] < 2.0 * das[0][0]
    d = torch.where(bnc[2] * y + bnc[3] > 0, ref, torch.zeros_like(ref))
    yhat = ((y - bnc[0]) * bnc[1]).double()
    refs = torch.stack((d.sum(0), (d * yhat).sum(0)), 1)
    scale = torch.stack((d.abs().sum(0), (d * yhat).abs().sum(0)), 1).clamp_min(1e-300)
    assert float(((das[1][1] - refs).abs() / scale).max()) < 2e-6
    assert lib.facl_gemm_rs_dgrad(p(dy), M, N, p(planes), 1, None, K, p(da), _lib.stream()) == -2      # FACL_E_NULL: fp16x3 needs the maximum
    # ---- weight gradient dW (N,K2) = dy^T relu(bn(y2)) where the register-streamed kernel serves the shape
    K2 = 512
    nz = lib.facl_gemm_rs_wgrad_slices(M, N, K2)
    if nz >= 1:
        y2 = torch.randn(M, K2, device=DEV, generator=g) * 3.0
        ps, pt = torch.rand(K2, device=DEV, generator=g) + 0.5, torch.randn(K2, device=DEV, generator=g) * 0.3
        refw = dy.double().t() @ torch.relu(y2.double() * ps.double() + pt.double())
        errs = {}
        for half in (1, 0):
            dW, sl = _lib.empty(N, K2, device=DEV), _lib.empty(nz * N * K2, device=DEV)
            _lib.check(lib.facl_gemm_rs_wgrad(p(dy), p(y2), M, N, K2, p(ps), p(pt), p(amax) if half else None,
                                              p(_act_amax(lib, y2, ps, pt)) if half else None, p(dW), p(sl), _lib.stream()), "rs_wgrad")
            errs[half] = rel_err(dW.cpu().numpy(), refw.cpu().numpy())
        print(f"wgrad fp16x3 {errs[1]:.2e}   bf16x6 {errs[0]:.2e}")
        assert errs[1] < 3e-6 and errs[1] < 2.0 * errs[0]
    # ---- the narrower layers' weight gradient on the LDS-staged kernel (facl_gemm_wgrad_h3), with and without the prologue
    for K3, pro in ((256, True), (256, False)):
        y3 = torch.randn(M, K3, device=DEV, generator=g) * 3.0
        ps, pt = torch.rand(K3, device=DEV, generator=g) + 0.5, torch.randn(K3, device=DEV, generator=g) * 0.3
        a64 = torch.relu(y3.double() * ps.double() + pt.double()) if pro else y3.double()
        refw = dy.double().t() @ a64
        nz = 256 // ((N // 128) * (K3 // 128)) + 1                     # at least one resident round of 128x128 workgroups
        dW, sl = _lib.empty(N, K3, device=DEV), _lib.empty(nz * N * K3, device=DEV)
        rc = lib.facl_gemm_wgrad_h3(p(dy), p(y3), M, N, K3, K3, p(ps) if pro else None, p(pt) if pro else None, p(amax),
                                    p(_act_amax(lib, y3, ps if pro else None, pt if pro else None)), p(dW), p(sl), nz, _lib.stream())
        _lib.check(rc, "wgrad_h3")
        e = rel_err(dW.cpu().numpy(), refw.cpu().numpy())
        print(f"wgrad (staged kernel) fp16x3 {e:.2e}  pro={pro}")
        assert e < 3e-6


def test_segmax_bwd_apply_amax_publishes_the_exact_maximum():
    from facl_amd import _lib
    lib = _lib.load_library()
    p = _lib.ptr
    g = torch.Generator(device=DEV).manual_seed(5)
    Mc, S, C = 96, 64, 1024
    y = torch.randn(Mc * S, C, device=DEV, generator=g)
    xpre = torch.randn(Mc, C, device=DEV, generator=g)
    dxpre = torch.randn(Mc, C, device=DEV, generator=g) * 1e-4
    arg = torch.randint(0, S, (Mc, C), device=DEV, generator=g, dtype=torch.int32)
    bnc = _bn_consts(C, g)
    kk = torch.randn(2, C, device=DEV, generator=g) * 1e-7
    dy, dy2 = _lib.empty(Mc * S, C, device=DEV), _lib.empty(Mc * S, C, device=DEV)
    amax = torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device=DEV)
    _lib.check(lib.facl_segmax_bwd_apply_amax(p(dxpre), p(xpre), p(y), p(arg), Mc, S, C, p(bnc), p(kk), p(dy), p(amax),
                                              _lib.stream()), "segmax_apply_amax")
    _lib.check(lib.facl_segmax_bwd_apply(p(dxpre), p(xpre), p(y), p(arg), Mc, S, C, p(bnc), p(kk), p(dy2), _lib.stream()), "segmax_apply")
    assert torch.equal(dy, dy2)
    assert float(amax.view(torch.float32).max()) == float(dy.abs().max())


def test_segmax_bwd_stats_from_kept_maxima_equals_the_gather():
    """facl_segmax_bwd_stats_ymax: y at the argmax is sign(gamma) * ymax exactly, so the BatchNorm-backward sums taken from the
    (M, C) maxima the forward kept are bit-equal to the ones gathered from y through arg (negative gammas and ties included)."""
    from facl_amd import _lib
    lib = _lib.load_library()
    p = _lib.ptr
    g = torch.Generator(device=DEV).manual_seed(11)
    for Mc, S, C in ((96, 64, 1024), (7, 64, 256)):
        y = torch.randn(Mc * S, C, device=DEV, generator=g)
        y[: 3 * S] = y[: 3 * S].round()                                    # exact ties inside the first blocks
        bnc = _bn_consts(C, g)
        sgn = torch.where(torch.rand(C, device=DEV, generator=g) < 0.4, -1.0, 1.0)
        bnc[4] = sgn
        bnc[2] = bnc[2].abs() * sgn                                        # scale carries gamma's sign
        sy = (y * sgn).view(Mc, S, C)
        ymax, arg = sy.max(dim=1)
        arg = (sy == ymax[:, None, :]).int().argmax(dim=1).to(torch.int32)          # first maximum
        xpre = torch.relu(bnc[2].abs() * ymax + bnc[3]).contiguous()
        dxpre = torch.randn(Mc, C, device=DEV, generator=g)
        s1 = _lib.empty(C, 2, dtype=torch.float64, device=DEV)
        s2 = _lib.empty(C, 2, dtype=torch.float64, device=DEV)
        _lib.check(lib.facl_segmax_bwd_stats(p(dxpre), p(xpre), p(y), p(arg.contiguous()), Mc, S, C, p(bnc), p(s1), p(_ws()),
                                             _lib.stream()), "segmax_bwd_stats")
        za = torch.full((5,), -1, dtype=torch.int32, device=DEV)
        _lib.check(lib.facl_segmax_bwd_stats_ymax(p(dxpre), p(xpre), p(ymax.contiguous()), Mc, C, p(bnc), p(s2), p(_ws()),
                                                  p(za), 4, _lib.stream()), "segmax_bwd_stats_ymax")
        assert torch.equal(s1, s2)
        assert za.tolist() == [0, 0, 0, 0, -1]                               # the words it is asked to zero, and only those
        assert float(s1.abs().sum()) > 0


@pytest.mark.parametrize("sw,sa", [(2.0 ** -12, 1.0), (2.0 ** 6, 1.0), (1.0, 2.0 ** -9), (1.0, 2.0 ** 7), (2.0 ** 20, 2.0 ** -20),
                                   (2.0 ** -25, 2.0 ** 30)])
@pytest.mark.parametrize("pro", [False, True])
def test_fp16x3_forward_is_scale_equivariant(sw, sa, pro):
    """fp16x3 has NO range contract (rounds 1-3: activations 2^4, weights 2^8 fixed, i.e. |a| < 4094, |w| < 255 or NaN, and
    15-18 bits for uniformly tiny tensors): every operand is scaled by the power of two of its own maximum / bound
    (csrc/common.h).  Scaling W or a by powers of two from 2^-25 to 2^30 scales the exact result by the same factor: the GEMM
    must stay within 2e-6 of the fp64 product at every scale (with the BatchNorm + ReLU prologue the scale sits on the
    prologue constants, as a tiny / huge gamma would put it there)."""
    from facl_amd import _lib
    lib = _lib.load_library()
    p = _lib.ptr
    M, K, N = 4096, 256, 512
    g = torch.Generator(device=DEV).manual_seed(17)
    a = torch.randn(M, K, device=DEV, generator=g)
    a[:, ::7] *= 1e-3
    W = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5 * sw).contiguous()
    W[::5] *= 2.0 ** -6                                               # column tiles of different magnitude: one scale per 32 columns
    b = torch.randn(N, device=DEV, generator=g) * (sw * sa)
    if pro:
        ps, pt = (torch.rand(K, device=DEV, generator=g) + 0.5) * sa, torch.randn(K, device=DEV, generator=g) * (0.3 * sa)
    else:
        a, ps, pt = (a * sa).contiguous(), None, None
    planes = _rs_planes(lib, W, False, None, 1)
    y = _lib.empty(M, N, device=DEV)
    _lib.check(lib.facl_gemm_rs_fwd(p(a), M, K, p(planes), 1, p(_act_amax(lib, a, ps, pt)), N, p(b), p(ps), p(pt), None, p(y), None,
                                    None, None, None, p(_ws()), _lib.stream()), "rs_fwd")
    a64 = torch.relu(a.double() * ps.double() + pt.double()) if pro else a.double()
    ref = a64 @ W.double().t() + b.double()
    e = rel_err(y.cpu().numpy(), ref.cpu().numpy())
    print(f"W x {sw:.1e}, a x {sa:.1e}, prologue {pro}: {e:.2e}")
    assert torch.isfinite(y).all() and e < 2e-6


def test_fp16x3_large_weights_and_activations_are_finite_and_correct():
    """The numbers the old contract excluded: a |w| = 300 weight, a 5000.0 activation (5000 * 2^4 overflowed fp16), and a
    bound that is 2^10 too generous (Samuelson's inequality at 3 M positions is) -- finite, 2e-6 of fp64.  Poisoned inputs
    stay loud: a NaN activation gives a NaN output row, nothing else."""
    from facl_amd import _lib
    lib = _lib.load_library()
    p = _lib.ptr
    M, K, N = 2048, 64, 256
    g = torch.Generator(device=DEV).manual_seed(11)
    a = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.zeros(N, device=DEV)
    a[5, 7] = 5000.0
    W[3, 9] = 300.0

    def run(amax):
        planes = _rs_planes(lib, W, False, None, 1)
        y = _lib.empty(M, N, device=DEV)
        _lib.check(lib.facl_gemm_rs_fwd(p(a), M, K, p(planes), 1, p(amax), N, p(b), None, None, None, p(y), None, None, None, None,
                                        p(_ws()), _lib.stream()), "rs_fwd")
        return y
    ref = a.double() @ W.double().t()
    amax = _act_amax(lib, a)
    y = run(amax)
    assert torch.isfinite(y).all() and rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    # a bound 2^10 above the true maximum of ordinary (outlier-free) activations -- what Samuelson's inequality gives at 3 M
    # positions -- costs nothing measurable: the elements still sit within the 16 octaves that keep 22 bits
    a[5, 7] = 1.0
    ref = a.double() @ W.double().t()
    loose = (_act_amax(lib, a).view(torch.float32) * 1024.0).view(torch.int32).contiguous()
    y = run(loose)
    assert torch.isfinite(y).all() and rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert lib.facl_gemm_rs_fwd(p(a), M, K, p(_rs_planes(lib, W, False, None, 1)), 1, None, N, p(b), None, None, None, p(y), None, None,
                                None, None, p(_ws()), _lib.stream()) == -2       # FACL_E_NULL: fp16x3 needs the operand's maximum
    a[9, 3] = float("nan")
    y = run(_act_amax(lib, a))
    assert not torch.isfinite(y[9]).any() and torch.isfinite(y[:9]).all() and torch.isfinite(y[10:]).all()


def test_eval_mode_net3dv3_propagates_a_nan_like_torch():
    """relu / max-pool of the row-streamed path must not swallow a NaN (fmaxf(NaN, 0) = 0; `v > best` skips NaN) --
    torch.relu and MaxPool2d propagate it.  One poisoned centroid row through net3DV_3 in eval mode: that cloud's x_pre is
    non-finite, every other cloud's is finite and unchanged."""
    from types import SimpleNamespace
    from facl_amd.cn3d_model_conbag import PointNet_Plus
    from oracle.weights import formula_state_dict
    opt = SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64, sample_num_level2=64,
                          INPUT_FEATURE_NUM=4, Num_Class=512, batchSize=8, pooling="concatenation", SAMPLE_NUM=512)
    net = PointNet_Plus(opt, gost=4)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in formula_state_dict(4).items()})
    net = net.to(DEV).eval()
    g = torch.Generator(device=DEV).manual_seed(3)
    M, S = 32, 64
    pooled = torch.rand(M * S, 256, device=DEV, generator=g)
    centers = torch.rand(M * S, 3, device=DEV, generator=g) - 0.5
    from facl_amd import tail
    assert tail.net3dv3_supported(M * S, (256, 256, 512, 1024), S, "f32")
    with torch.no_grad():
        clean = tail.net3dv3(pooled, centers, net.net3DV_3, False, S)
        pooled[5 * S + 17, 100] = float("nan")
        bad = tail.net3dv3(pooled, centers, net.net3DV_3, False, S)
    assert torch.isfinite(clean).all()
    assert not torch.isfinite(bad[5]).any()
    keep = [i for i in range(M) if i != 5]
    assert torch.isfinite(bad[keep]).all()


