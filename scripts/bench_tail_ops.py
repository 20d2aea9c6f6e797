"""Microbenchmark of the non-ConvLSTM kernels of one timestep at the B=32 shapes of config 2 (hipEvent timing through
the per-op C ABI, random operands).  Usage: python scripts/bench_tail_ops.py [B] [iters]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.load()
dev = 'cuda:0'
st = torch.cuda.current_stream().cuda_stream
rs = np.random.RandomState(0)
R = lambda *s: torch.from_numpy(rs.randn(*s).astype(np.float32)).to(dev)
E = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
ops = {}

def op(name, nbytes):
    def deco(f):
        ops[name] = (f, nbytes)
        return f
    return deco

# enc0: frame -> 32x32x32
img = R(B, 3, 64, 64); w0 = R(75, 32); b0 = R(32); e0 = E(B, 32, 32, 32)
@op('conv_enc0', img.numel() * 4 + e0.numel() * 4)
def _(): return lib.pivp_conv_enc0(img.data_ptr(), w0.data_ptr(), b0.data_ptr(), e0.data_ptr(), B, 64, 64, st)

# LayerNorm of a 32x32x32 and of the 64x64x64 map (separate statistics pass)
def ln_case(name, C, H):
    n = C * H * H
    x = R(B, n); g = R(n); be = R(n); o = E(B, n)
    sc = E(lib.pivp_layernorm_scratch_floats(B, n))
    @op(name, 3 * x.numel() * 4)
    def _(): return lib.pivp_layernorm(x.data_ptr(), g.data_ptr(), be.data_ptr(), o.data_ptr(), sc.data_ptr(), B, n, C, C, 1e-6, 0, st)
ln_case('layernorm 32x32x32', 32, 32)
ln_case('layernorm 64x64x64', 64, 64)

# enc1 / enc2: 3x3 stride-2 convs
def conv_case(name, cin, cout, H):
    x = R(B, H, H, cin); w = R(9 * cin * cout); b = R(cout); o = E(B, H // 2, H // 2, cout)
    @op(name, (x.numel() + o.numel() + w.numel()) * 4)
    def _(): return lib.pivp_conv3x3s2(x.data_ptr(), cin, cin, w.data_ptr(), b.data_ptr(), o.data_ptr(), cout, cout, 1, B, H, H, st)
conv_case('enc1 conv3x3s2 32->32 @32', 32, 32, 32)
conv_case('enc2 conv3x3s2 64->64 @16', 64, 64, 16)
conv_case('enc6 dgrad = conv3x3s2 64->64 @64', 64, 64, 64)
conv_case('enc5 dgrad = conv3x3s2 96->96 @32', 96, 96, 32)
conv_case('enc4 dgrad = conv3x3s2 128->128 @16', 128, 128, 16)

def deconv_case(name, cin, cout, H):
    x = R(B, H, H, cin); w = R(9 * cin * cout); b = R(cout); o = E(B, 2 * H, 2 * H, cout)
    @op(name, (x.numel() + o.numel() + w.numel()) * 4)
    def _(): return lib.pivp_deconv3x3s2(x.data_ptr(), cin, cin, w.data_ptr(), b.data_ptr(), o.data_ptr(), cout, cout, 1, B, H, H, st)
deconv_case('enc4 deconv 128->128 @8', 128, 128, 8)
deconv_case('enc5 deconv 96->96 @16', 96, 96, 16)
deconv_case('enc6 deconv 64->64 @32', 64, 64, 32)

# enc3 + state predictor
e2 = R(B, 64, 64); act = R(B, 5); sta = R(B, 5); w3 = R(74 * 64); b3 = R(64); wcs = R(5, 10); bcs = R(5); e3 = E(B, 64, 64); so = E(B, 5)
@op('enc3_state', (e2.numel() + e3.numel()) * 4)
def _(): return lib.pivp_enc3_state(e2.data_ptr(), act.data_ptr(), sta.data_ptr(), w3.data_ptr(), b3.data_ptr(), wcs.data_ptr(), bcs.data_ptr(),
                                    e3.data_ptr(), so.data_ptr(), B, 64, 1, st)

# heads (CDNA: 11 mask planes + 3 enc7 planes)
e6 = R(B, 4096, 64); wm = R(64, 11); bm = R(11); we = R(64, 3); bee = R(3)
lg = E(B, 11, 4096); e7 = E(B, 3, 4096); l0 = E(B, 3, 4096)
@op('heads_1x1', (e6.numel() + lg.numel() + 2 * e7.numel()) * 4)
def _(): return lib.pivp_heads(e6.data_ptr(), wm.data_ptr(), bm.data_ptr(), we.data_ptr(), bee.data_ptr(), lg.data_ptr(), e7.data_ptr(), l0.data_ptr(),
                               B, 4096, 10, 0, st)

# CDNA kernel generator
h5 = R(B, 8192); wk = R(8192 * 256); bk = R(256); sck = E(lib.pivp_linear_scratch_floats(B, 8192)); kern = E(B, 250)
@op('cdna_kernels', (wk.numel() + h5.numel()) * 4)
def _(): return lib.pivp_cdna_kernels(h5.data_ptr(), wk.data_ptr(), bk.data_ptr(), sck.data_ptr(), kern.data_ptr(), B, 8192, 10, st)

# composite
kern.copy_(torch.rand_like(kern)); out = E(B, 3, 4096); masks = E(B, 11, 4096)
@op('composite cdna (+masks)', (img.numel() + lg.numel() + l0.numel() + out.numel() + masks.numel()) * 4)
def _(): return lib.pivp_composite(img.data_ptr(), lg.data_ptr(), l0.data_ptr(), kern.data_ptr(), out.data_ptr(), masks.data_ptr(), B, 64, 64, 10, 0, 0, st)
@op('composite cdna', (img.numel() + lg.numel() + l0.numel() + out.numel()) * 4)
def _(): return lib.pivp_composite(img.data_ptr(), lg.data_ptr(), l0.data_ptr(), kern.data_ptr(), out.data_ptr(), None, B, 64, 64, 10, 0, 0, st)

only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
for name, (f, nb) in ops.items():
    if only and not any(o in name for o in only):
        continue
    rc = f(); assert rc == 0, (name, rc)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0_ = torch.cuda.Event(enable_timing=True); e1_ = torch.cuda.Event(enable_timing=True)
        e0_.record()
        for _ in range(iters):
            f()
        e1_.record(); torch.cuda.synchronize()
        ts.append(e0_.elapsed_time(e1_) / iters * 1e3)
    us = float(np.median(ts))
    print('%-32s %8.1f us   %7.1f MB   %6.2f TB/s' % (name, us, nb / 1e6, nb / us / 1e6))
