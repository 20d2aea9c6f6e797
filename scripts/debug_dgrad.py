"""pivp_conv5x5_f32 (the fp32 ConvLSTM data gradient's kernel: plain 5x5 stride-1 conv on packed weights) against torch's conv2d on the GPU, at the sweep's shapes."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
import pivp_amd  # noqa: F401
from pivp_amd import _lib
lib = _lib.load()
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = torch.cuda.current_stream().cuda_stream
for name, cin, cout, H in [('lstm1', 128, 64, 32), ('lstm3', 256, 96, 16), ('lstm4', 256, 128, 16), ('lstm5', 512, 192, 8), ('lstm6', 256, 192, 16), ('lstm7', 128, 128, 32),
                           ('lstm7 dx only', 128, 96, 32)]:
    torch.manual_seed(0)
    x = torch.randn(B, cin, H, H, device=dev)
    W = torch.randn(cout, cin, 5, 5, device=dev) / np.sqrt(25 * cin)
    ref = F.conv2d(x.double(), W.double(), padding=2).float()
    packed = W.permute(2, 3, 1, 0).reshape(25, cin // 32, 32, cout).permute(0, 1, 3, 2).contiguous()      # [tap][ci/32][n][ci%32]
    xn = x.permute(0, 2, 3, 1).contiguous()
    out = torch.full((B * H * H, cout), float('nan'), device=dev)
    rc = lib.pivp_conv5x5_f32(xn.data_ptr(), cin, cin, packed.data_ptr(), out.data_ptr(), cout, B, H, H, st)
    torch.cuda.synchronize()
    got = out.view(B, H, H, cout).permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    bad = (~torch.isfinite(got)).sum().item()
    print('%-14s B %d cin %d cout %d map %d: rc %d, max |err| / max |ref| %.2e, non-finite %d' % (name, B, cin, cout, H, rc, err, bad))
    if err > 1e-4 or bad:
        e = (got - ref).abs()
        idx = torch.nonzero(e > 1e-3 * ref.abs().max())
        print('   first bad elements (b, c, y, x):', idx[:8].tolist(), ' count', idx.shape[0], 'of', e.numel())
