"""Does the bf16 precision mode (BASELINE.json config 3's arithmetic) train like fp32?  The same CDNA 64x64 model (fixture seed 1), the
same batches (oracle `moving_batch`, seeds 1000 + step), the same Adam, feed-self, once per precision; the training loss is logged every
50 steps and both results are evaluated with the fp32 rollout on held-out sequences (seed 7).  Needs the MI355X."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import oracle.restatement as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--workers', type=int, default=12)
    ap.add_argument('--out', default='gpurun_out/r03/train_curve_bf16_vs_fp32.txt')
    args = ap.parse_args()
    T, S = 10, 64
    import concurrent.futures as cf
    import multiprocessing as mp
    pool = cf.ProcessPoolExecutor(args.workers, mp_context=mp.get_context('fork'))      # forked before anything initialises the GPU
    batches = [pool.submit(R.moving_batch, args.batch, T, S, S, 1000 + i) for i in range(args.steps)]
    held = R.moving_batch(args.batch, T, S, S, seed=7)
    import torch
    import pivp_amd
    assert torch.cuda.is_available()
    P0 = R.init_params(seed=1, dtype=np.float32, scale=1.0, num_masks=10, model_type='CDNA', height=S, width=S)
    curves, finals = {}, {}
    for precision in ('fp32', 'bf16'):
        m = pivp_amd.Model(10, prefix='curve', keep_activations=True, scheduled_sampling_k=-1.0, is_cdna=True, precision=precision)
        m.load_state_dict_reference(P0)
        opt = pivp_amd.Adam(alpha=1e-3).setup(m)
        losses = []
        t0 = time.time()
        with pivp_amd.using_config('train', True):
            for it in range(args.steps):
                x = batches[it].result()
                m.reset_state()
                loss = opt.update(m, list(x), it)
                losses.append(loss)
                if it % 200 == 0:
                    print('%s step %5d  loss %.6f  (%.0f s)' % (precision, it, float(loss), time.time() - t0), flush=True)
        curves[precision] = torch.stack([l.reshape(()) for l in losses]).cpu().numpy()
        ev = pivp_amd.Model(10, prefix='curve', is_cdna=True)                              # fp32 rollout for both
        ev.load_state_dict_reference(m.state_dict_reference())
        with pivp_amd.using_config('train', False):
            hl = float(ev(list(held), 0))
        gen = torch.stack(ev.gen_images).cpu().numpy()
        finals[precision] = (hl, float(((held[0][1:] - gen) ** 2).mean()))
    pool.shutdown()
    copy_err = float(((held[0][1:] - held[0][:-1]) ** 2).mean())
    lines = ['CDNA 64x64, B = %d, T = 10, Adam 1e-3, feed-self, %d steps, identical batches and initialisation' % (args.batch, args.steps),
             'training loss, mean over windows of 100 steps:', '  steps        fp32      bf16     bf16/fp32']
    for a in range(0, args.steps, max(100, args.steps // 20)):
        b = min(args.steps, a + 100)
        f, h = float(curves['fp32'][a:b].mean()), float(curves['bf16'][a:b].mean())
        lines.append('  %5d-%-5d %.6f  %.6f  %.3f' % (a, b, f, h, h / f))
    lines.append('held-out (seed 7), fp32 rollout of the trained weights: loss fp32-trained %.6f, bf16-trained %.6f; mse(pred, next) %.6f / %.6f; '
                 'mse(prev, next) %.6f' % (finals['fp32'][0], finals['bf16'][0], finals['fp32'][1], finals['bf16'][1], copy_err))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines), flush=True)


if __name__ == '__main__':
    main()
