"""What announcing gradient groups costs a train step on ONE rank (no real communication): plain backward, backward with a no-op listener
(the model's stream joins the side stream at every announcement), and GradAllReduce's overlapped path on a 1-rank group (only the
collective's stream waits: pivp_plan_set_group_join(0) + pivp_plan_group_wait)."""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, '.')
import pivp_amd
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29731')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
rs = np.random.RandomState(0)
B, T = 32, 10
x = [torch.from_numpy(rs.random_sample((T, B, 3, 64, 64)).astype(np.float32)).cuda(),
     torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).cuda(),
     torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).cuda()]
np.random.seed(0)
m = pivp_amd.Model(10, prefix='x', keep_activations=True)
opt = pivp_amd.Adam().setup(m)
dp = pivp_amd.GradAllReduce()
dp_join = pivp_amd.GradAllReduce(defer_side_wait=False)
def step(kind):
    m.reset_state()
    m(x, 0); m.cleargrads()
    if kind == 'plain': m.backward()
    elif kind == 'listener': m.backward(on_group=lambda g: None)
    elif kind.startswith('overlapped all-reduce, main'): dp_join.backward_and_allreduce(m, force_overlap=True)
    else: dp.backward_and_allreduce(m, force_overlap=True)
    opt.step(m)
for kind in ('plain', 'listener', 'overlapped all-reduce, main stream joins', 'overlapped all-reduce, only the collective waits') * 2:
    for _ in range(3): step(kind)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(kind)
    torch.cuda.synchronize(); print('%-52s %.3f ms' % (kind, (time.perf_counter() - t0) / 10 * 1e3))
dist.destroy_process_group()
