"""instrumented k_gemm_rs (scratch/lib_exp_rs.so): where a wave's loop time goes (s_memtime stamps)"""
import os, sys, ctypes, torch
sys.path.insert(0, ".")
os.environ["FACL_LIB"] = os.path.abspath("scratch/lib_exp_rs.so")
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace
from facl_amd import tail
lib = _lib.load_library()
raw = ctypes.CDLL(os.environ["FACL_LIB"])
DEV = torch.device("cuda:0")
p = _lib.ptr
ws = _Workspace.get(DEV)
M = 49152
def amax_of(t):
    b = torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device=DEV)
    _lib.check(lib.facl_absmax(p(t), t.numel(), p(b), _lib.stream()), "absmax")
    return b
def report(tag, n):
    buf = (ctypes.c_ulonglong * 12)()
    torch.cuda.synchronize()
    raw.facl_dbg_read(buf)
    v = list(buf)
    cnt = max(v[7], 1)
    names = ["mfma even", "wait vmcnt(4)+barrier", "mfma odd", "wait vmcnt(0)", "barrier", "read_stage+planes", "loop total"]
    tot = v[6] / cnt
    print(tag, "waves sampled", cnt, " loop cycles per wave %.0f" % tot)
    for nm, x in zip(names[:6], v[:6]):
        print("   %-24s %8.0f  %.3f" % (nm, x / cnt, x / cnt / tot))
    print("   prologue (entry -> loop)  %8.0f   last stage %8.0f   epilogue %8.0f" % (v[8] / cnt, v[9] / cnt, v[10] / cnt))
    raw.facl_dbg_zero()
torch.manual_seed(0)
for K, N in ((512, 1024), (1024, 512), (256, 512)):
    a = torch.randn(M, K, device=DEV).abs(); W = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV)
    ps = torch.rand(K, device=DEV) + 0.5; pt = torch.randn(K, device=DEV) * 0.3
    pl = tail.rs_planes(W, False, half=True)
    am = amax_of(a * 2)
    y, sums, _, _ = tail._rs_fwd(a, pl, N, b, None, None, False, None, ws, True, am)
    raw.facl_dbg_zero()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        tail._rs_fwd(a, pl, N, b, None, None, False, None, ws, True, am)
    e1.record(); torch.cuda.synchronize()
    print("fwd %dx%dx%d no PRO: %.4f ms" % (M, K, N, e0.elapsed_time(e1) / 10))
    report("  ", 10)
