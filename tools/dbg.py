import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
print("torch", torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))
import __graft_entry__ as g
pkg = g.load_package()
L = pkg.capi.lib()
maps = open('/proc/self/maps').read()
libs = sorted(set(l.split()[-1] for l in maps.splitlines() if 'amdhip' in l or 'hsa-runtime' in l))
print(libs)
hip = ctypes.CDLL("libamdhip64.so")
n = ctypes.c_int(-1)
print("hipGetDeviceCount rc", hip.hipGetDeviceCount(ctypes.byref(n)), n.value)
h = ctypes.c_void_p()
print("ctx rc", L.ismhip_ctx_create(0, None, ctypes.byref(h)))
