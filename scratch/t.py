import torch, ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libt.so"))
x = torch.arange(1000, device="cuda", dtype=torch.float32); y = torch.ones(1000, device="cuda")
s = torch.cuda.current_stream().cuda_stream
rc = lib.t_axpy(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_float(2.0), ctypes.c_int(1000), ctypes.c_void_p(s))
torch.cuda.synchronize(); print("rc", rc, y[:5], (y - (2*x+1)).abs().max().item())
a = torch.randn(32,32,device="cuda"); b = torch.randn(32,32,device="cuda"); d = torch.zeros(32,32,device="cuda")
rc = lib.t_mfma(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(s))
torch.cuda.synchronize(); ref = (a.double()@b.double()).float(); print("mfma rc", rc, (d-ref).abs().max().item())
maps = open("/proc/self/maps").read(); print([l.split()[-1] for l in maps.splitlines() if "amdhip" in l][:3])
print(torch.cuda.get_device_name(0), os.cpu_count())
