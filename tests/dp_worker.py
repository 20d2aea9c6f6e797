"""Worker of tests/test_gpu_train.py::test_two_process_data_parallel_step: one rank of a 2-process data-parallel train step.
Both ranks share cuda:0 (RCCL refuses two ranks on one device, so the collective backend is gloo, which stages device tensors
through the host) -- everything else is the product path: Model, the backward sweep with its side stream and gradient-group
callbacks, GradAllReduce.backward_and_allreduce, the Chainer-rule Adam step.  Launched by torch.distributed.run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    import pivp_amd
    from oracle import restatement as R
    torch.cuda.set_device(0)
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
    imgs, acts, stas = R.synthetic_batch(4, 5)                       # global batch 4 -> 2 per rank
    mine = pivp_amd.shard_batch([imgs, acts, stas], rank, world)
    m = pivp_amd.Model(10, prefix='dp', keep_activations=True, device='cuda:0')
    m.load_state_dict_reference(P)
    dp = pivp_amd.GradAllReduce()
    opt = pivp_amd.Adam(alpha=0.001).setup(m, data_parallel=dp)
    with pivp_amd.using_config('train', True):
        loss = float(opt.update(m, [np.ascontiguousarray(a) for a in mine], 0))
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), loss=loss, issued=np.array(dp.issued),
             params=m._flat_params.cpu().numpy(), grads=m._flat_grads.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
