import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FACL_LIB"] = os.path.abspath("scratch/lib_dbg.so")
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace
lib = _lib.load_library()
dev = torch.device("cuda:0")
D = 3; nunits = 5; P = nunits * 64
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
x = R(P, D) * 0.3
y2f, dz2f = R(nunits * 4096), R(nunits * 4096)
W1, b1, W2 = R(64, D) * 0.3, R(64) * 0.1, R(64, 64) * 0.1
l1tab = torch.empty(64, 8, device=dev)
p = _lib.ptr; st = _lib.stream()
_lib.check(lib.facl_sa_l1tab(p(W1), p(b1), D, None, None, p(l1tab), st), "l1tab")
bw2 = torch.rand(4, 64, device=dev, generator=g)
ws = _Workspace.get(dev)
wsd = ws.view(torch.float64)
def rowmap(r, h): return (r & 3) + 8 * (r >> 2) + 4 * h
for u in (0, 1):
    o2 = torch.zeros(4608, dtype=torch.float64, device=dev)
    xs = x[u * 64:(u + 1) * 64].contiguous()
    rc = lib.facl_sa_bwd2(p(dz2f[u * 4096:]), p(y2f[u * 4096:]), p(xs), 1, D, p(bw2), p(W2), p(l1tab), p(o2), p(ws), st)
    torch.cuda.synchronize()
    v = wsd[8 * 4608: 8 * 4608 + 4096].cpu().view(2, 2, 16, 64)     # [ct][ct1][r][lane]
    lt = l1tab.cpu().double()
    xd = xs.cpu().double()
    nbad = 0
    for ct in range(2):
        for ct1 in range(2):
            for r in range(16):
                for lane in range(64):
                    h, q = lane >> 5, lane & 31
                    pp = 32 * ct + rowmap(r, h); c1 = 32 * ct1 + q
                    ref = lt[c1, 4] + (lt[c1, :3] * xd[pp]).sum()
                    if abs(float(v[ct, ct1, r, lane]) - float(ref)) > 1e-5:
                        if nbad < 12:
                            print(f"u{u} BAD v ct={ct} ct1={ct1} r={r} lane={lane} (p={pp}, c1={c1}): got {float(v[ct,ct1,r,lane]):.6f} want {float(ref):.6f}")
                        nbad += 1
    print("unit", u, "bad v count", nbad)

lib_b = ctypes.CDLL(os.path.abspath("scratch/lib_base.so"))
for name, argtypes in _lib.SIGNATURES.items():
    fn = getattr(lib_b, name); fn.argtypes = argtypes
    fn.restype = ctypes.c_longlong if name in _lib.RESTYPE_I64 else ctypes.c_int
lib_n = ctypes.CDLL(os.path.abspath("facl_amd/libfacl_hip.so"))
for name, argtypes in _lib.SIGNATURES.items():
    fn = getattr(lib_n, name); fn.argtypes = argtypes
    fn.restype = ctypes.c_longlong if name in _lib.RESTYPE_I64 else ctypes.c_int
for u in range(nunits):
    outs = []
    for L in (lib, lib_n, lib_b):
        o2 = torch.zeros(4608, dtype=torch.float64, device=dev)
        xs = x[u * 64:(u + 1) * 64]
        rc = L.facl_sa_bwd2(p(dz2f[u * 4096:]), p(y2f[u * 4096:]), xs.data_ptr(), 1, D, p(bw2), p(W2), p(l1tab), p(o2), p(ws), st)
        torch.cuda.synchronize(); outs.append(o2.cpu())
    a, n, b = outs
    print(f"unit {u}: dbg-vs-old {float((a-b).norm()/b.norm()):.3e}   new-vs-old {float((n-b).norm()/b.norm()):.3e}")
    # da1 check on the dbg build (last call order: dbg ran first, so rerun dbg)
    o2 = torch.zeros(4608, dtype=torch.float64, device=dev)
    lib.facl_sa_bwd2(p(dz2f[u * 4096:]), p(y2f[u * 4096:]), xs.data_ptr(), 1, D, p(bw2), p(W2), p(l1tab), p(o2), p(ws), st)
    torch.cuda.synchronize()
    da = wsd[9 * 4608: 9 * 4608 + 4096].cpu().view(2, 2, 16, 64)
    # expected da1 = dy2 @ W2 with dy2 from the fragment layout
    z = dz2f[u * 4096:(u + 1) * 4096].cpu().double().view(2, 2, 4, 64, 4)   # [ct][rt][r4][lane][e]
    y = y2f[u * 4096:(u + 1) * 4096].cpu().double().view(2, 2, 4, 64, 4)
    bw = bw2.cpu().double()
    dy = torch.zeros(64, 64, dtype=torch.float64)
    for ct in range(2):
        for rt in range(2):
            for r4 in range(4):
                for lane in range(64):
                    h, q = lane >> 5, lane & 31
                    for e in range(4):
                        c = 32 * rt + 8 * r4 + 4 * h + e
                        pp = 32 * ct + q
                        dy[pp, c] = bw[0, c] * z[ct, rt, r4, lane, e] + bw[2, c] * (y[ct, rt, r4, lane, e] - bw[3, c]) + bw[1, c]
    ref = dy @ W2.cpu().double()
    nbad = 0
    for ct in range(2):
        for ct1 in range(2):
            for r in range(16):
                for lane in range(64):
                    h, q = lane >> 5, lane & 31
                    pp = 32 * ct + rowmap(r, h); c1 = 32 * ct1 + q
                    if abs(float(da[ct, ct1, r, lane]) - float(ref[pp, c1])) > 1e-4 * float(ref.abs().max()):
                        if nbad < 6:
                            print(f"   BAD da1 ct={ct} ct1={ct1} r={r} lane={lane}: got {float(da[ct,ct1,r,lane]):.6f} want {float(ref[pp,c1]):.6f}")
                        nbad += 1
    print("   bad da1", nbad)
