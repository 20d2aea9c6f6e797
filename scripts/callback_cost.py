import sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import pivp_amd
rs = np.random.RandomState(0)
B, T = 32, 10
x = [torch.from_numpy(rs.random_sample((T, B, 3, 64, 64)).astype(np.float32)).cuda(),
     torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).cuda(),
     torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).cuda()]
np.random.seed(0)
m = pivp_amd.Model(10, prefix='x', keep_activations=True)
opt = pivp_amd.Adam().setup(m)
def step(cb):
    m.reset_state()
    m(x, 0); m.cleargrads()
    m.backward(on_group=cb) if cb else m.backward()
    opt.step(m)
for name, cb in (('no callback', None), ('no-op callback', lambda g: None), ('no callback', None), ('no-op callback', lambda g: None)):
    for _ in range(3): step(cb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(cb)
    torch.cuda.synchronize(); print(name, '%.3f ms' % ((time.perf_counter() - t0) / 10 * 1e3))
