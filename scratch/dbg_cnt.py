import ctypes, os, sys, subprocess
sys.path.insert(0, '/root/repo')
import torch
os.environ['FACL_LIB'] = '/root/repo/facl_amd/libfacl_hip_t1.so'
from facl_amd import _lib
lib = _lib.load_library()
raw = ctypes.CDLL(os.environ['FACL_LIB'])
raw.facl_debug_counters.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
sys.argv = ['x', '--only', 'bwd1']
import runpy
raw.facl_debug_counters(None, 1)
runpy.run_path('/root/repo/tools/microbench_sa.py', run_name='__main__')
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
raw.facl_debug_counters(out, 0)
n = out[4]
print('n', n, 'scatter acc %.0f rest(barrierA+write+B) %.0f | dense wait(load+A+B) %.0f work %.0f (cycles per wave-launch)' % (out[0]/n, out[1]/n, out[2]/n, out[3]/n))
