import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace
lib = _lib.load_library()
lib_b = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, argtypes in _lib.SIGNATURES.items():
    fn = getattr(lib_b, name); fn.argtypes = argtypes
    fn.restype = ctypes.c_longlong if name in _lib.RESTYPE_I64 else ctypes.c_int
dev = torch.device("cuda:0")
for D in (3, 4):
  for nunits in (1, 5, 2048):
    P = nunits * 64
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = R(P, D) * 0.3
    y2f, dz2f = R(nunits * 4096), R(nunits * 4096)
    W1, b1, W2 = R(64, D) * 0.3, R(64) * 0.1, R(64, 64) * 0.1
    l1tab = torch.empty(64, 8, device=dev)
    _lib.check(lib.facl_sa_l1tab(_lib.ptr(W1), _lib.ptr(b1), D, None, None, _lib.ptr(l1tab), _lib.stream()), "l1tab")
    bw2 = torch.rand(4, 64, device=dev, generator=g)
    ws = _Workspace.get(dev)
    p = _lib.ptr; st = _lib.stream()
    outs = []
    for L in (lib, lib_b):
        o2 = torch.zeros(4608, dtype=torch.float64, device=dev)
        rc = L.facl_sa_bwd2(p(dz2f), p(y2f), p(x), nunits, D, p(bw2), p(W2), p(l1tab), p(o2), p(ws), st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        outs.append(o2.cpu())
    a, b = outs
    dw_a, dw_b = a[:4096].view(64, 64), b[:4096].view(64, 64)
    r_a, r_b = a[4096:].view(8, 64), b[4096:].view(8, 64)
    print(f"D={D} nunits={nunits}: dW2 rel {float((dw_a-dw_b).norm()/dw_b.norm()):.3e}   R1 rel {float((r_a-r_b).norm()/r_b.norm()):.3e}")
    if (r_a - r_b).norm() / r_b.norm() > 1e-4:
        for d in range(8):
            print("  R1 row", d, "new", r_a[d, :4].tolist(), "old", r_b[d, :4].tolist())
    if (dw_a - dw_b).norm() / dw_b.norm() > 1e-4:
        print("  dW2 new", dw_a[:2, :4].tolist(), "old", dw_b[:2, :4].tolist())

print("---- per-unit isolation, D=3")
D = 3; nunits = 5; P = nunits * 64
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
x = R(P, D) * 0.3
y2f, dz2f = R(nunits * 4096), R(nunits * 4096)
W1, b1, W2 = R(64, D) * 0.3, R(64) * 0.1, R(64, 64) * 0.1
l1tab = torch.empty(64, 8, device=dev)
_lib.check(lib.facl_sa_l1tab(_lib.ptr(W1), _lib.ptr(b1), D, None, None, _lib.ptr(l1tab), _lib.stream()), "l1tab")
bw2 = torch.rand(4, 64, device=dev, generator=g)
for u in range(nunits):
    outs = []
    for L in (lib, lib_b):
        o2 = torch.zeros(4608, dtype=torch.float64, device=dev)
        xs = x[u * 64:(u + 1) * 64]
        rc = L.facl_sa_bwd2(p(dz2f[u * 4096:]), p(y2f[u * 4096:]), xs.data_ptr(), 1, D, p(bw2), p(W2), p(l1tab), p(o2), p(ws), st)
        torch.cuda.synchronize(); outs.append(o2.cpu())
    a, b = outs
    d = (a - b).abs()
    i = int(d.argmax())
    print(f"unit {u} (x ptr % 16 = {xs.data_ptr() % 16}): rel {float((a-b).norm()/b.norm()):.3e}  max diff at {i} ({'dW2 c2=%d c1=%d' % (i // 64, i % 64) if i < 4096 else 'R1 d=%d c1=%d' % ((i - 4096) // 64, (i - 4096) % 64)})  new {float(a[i]):.6f} old {float(b[i]):.6f}")
    bad = (d[:4096].view(64, 64) > 1e-4 * b[:4096].abs().max()).nonzero()
    print("   bad dW2 entries:", bad.shape[0], "cols:", sorted(set(bad[:, 1].tolist()))[:20], "rows:", sorted(set(bad[:, 0].tolist()))[:20])
