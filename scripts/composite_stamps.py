"""In-kernel cycle stamps of the composite kernel's phases (build with PIVP_EXTRA_FLAGS=-DPIVP_CP_STAMPS):
staging | group softmax | first pixel | second pixel | end, for block (1, 3)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib
lib = _lib.load(); raw = ctypes.CDLL(lib._name) if hasattr(lib, '_name') else None
dev='cuda:0'; B=32
st = torch.cuda.current_stream().cuda_stream
R = lambda *s: torch.randn(*s, device=dev)
img=R(B,3,64,64); lg=R(B,11,4096); l0=R(B,3,4096); kern=torch.rand(B,250,device=dev); out=torch.empty(B,3,4096,device=dev); masks=torch.empty(B,11,4096,device=dev)
for _ in range(3):
    rc = lib.pivp_composite(img.data_ptr(), lg.data_ptr(), l0.data_ptr(), kern.data_ptr(), out.data_ptr(), masks.data_ptr(), B, 64, 64, 10, 0, 0, st)
torch.cuda.synchronize()
import glob
so = ctypes.CDLL(glob.glob('physical-interaction-video-prediction_amd/libpivp_hip.so')[0])
buf = (ctypes.c_longlong * 8)()
print('rc', so.pivp_debug_cp_stamps(buf))
v = list(buf)
print('stamps', [v[i] - v[0] for i in range(6)])
