#!/usr/bin/env python
"""Benchmark of the hot path: predicted frames/s of the 10-frame 64x64x3 CDNA rollout (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU through torch.distributed.run; every rank runs the
same per-GPU workload (weak scaling; the forward rollout shards over the batch with no data-path
collective: SURVEY.md 8e "inference rollout: replicas only").

A "step" is one Model.__call__ (TM:620-764) over one synthetic batch already resident in HBM:
B sequences x (T-1) predicted frames, feed-self after the context frames as predict_model.py:126-128.
Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel (ConvLSTM gate conv, fp32 MFMA implicit GEMM), algorithmic flops of
                 its launches / their HIP-event time measured on the launch stream in a second,
                 instrumented pass over the same K steps (events perturb the clean timing slightly,
                 so `value` comes from the un-instrumented pass)
  cpu_baseline : the CPU restatement of the reference path (oracle/torch_restatement.py, fp32, all
                 host threads) on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32-input MFMA dense peak (= vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak of the same guide (the headline 5 PFLOP/s figure includes 2:1 sparsity)


def _pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/), or None.
    PMC counters cannot be collected inside this process; the passes are re-run per round on the same command."""
    try:
        best = None
        pdir = os.path.join(ROOT, 'profiles')
        for rnd in sorted(os.listdir(pdir)):
            f = os.path.join(pdir, rnd, 'pmc_traffic.json')
            if os.path.exists(f):
                best = f
        if best is None:
            return None
        with open(best) as fh:
            return round(json.load(fh)['hbm_bytes_per_launch'])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='sequences per GPU (config 2: 32)')
    ap.add_argument('--seq-len', type=int, default=10)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--model', default='CDNA', choices=['CDNA', 'STP', 'DNA'])
    ap.add_argument('--mode', default='rollout', choices=['rollout', 'train'],
                    help='rollout: Model.__call__ forward (predict_model.py:126-128); train: optimizer.update = forward + '
                         'BPTT backward + gradient all-reduce + Adam (train_model.py:950)')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16', 'bf16x3'],
                    help='fp32: the parity path and the headline metric (config 2). bf16: ConvLSTM gate convolutions with bf16 operands, '
                         'fp32 accumulation (config 3); reports its per-pixel error instead of meeting the 1e-4 gate')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import pivp_amd

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device(dev))

    B, T, S = args.batch, args.seq_len, args.size
    nm = 1 if args.model == 'DNA' else 10
    np.random.seed(1234 + rank)
    train = args.mode == 'train'
    model = pivp_amd.Model(nm, is_cdna=args.model == 'CDNA', is_stp=args.model == 'STP', is_dna=args.model == 'DNA',
                           prefix='bench', device=dev, keep_activations=train, precision=args.precision)
    rs = np.random.RandomState(rank)
    images = torch.from_numpy(rs.random_sample((T, B, 3, S, S)).astype(np.float32)).to(dev)
    actions = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).to(dev)
    states = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).to(dev)

    opt = None
    if train:
        dp = pivp_amd.GradAllReduce() if world > 1 else None
        opt = pivp_amd.Adam(alpha=0.001).setup(model, data_parallel=dp)   # TM:860-861
        if world > 1:                                                        # identical replicas: broadcast rank 0's init
            with pivp_amd.using_config('train', False):
                model([images, actions, states], 0)
            dist.broadcast(model._flat_params, src=0)

    def step():
        model.reset_state()
        if train:
            return opt.update(model, [images, actions, states], 0)          # schedsamp_k = -1: feed-self, deterministic
        return model([images, actions, states], 0)

    def barrier():
        if dist is not None:
            dist.barrier()

    with pivp_amd.using_config('train', False):
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        loss_val = float(loss)

        roofline = None
        if not args.no_roofline and rank != 0 and train and world > 1:
            for _ in range(args.steps):        # rank 0's instrumented pass below calls the gradient all-reduce: every rank must take part
                step()
            torch.cuda.synchronize()
        if not args.no_roofline and rank == 0:
            plan = model._active
            lib = plan.lib
            lib.pivp_plan_set_profiling(plan.h, 1)
            ms_tot = np.zeros(7); n_tot = np.zeros(7, dtype=np.int64); flops = np.zeros(7)
            t_prof = 0.0
            for _ in range(args.steps):
                tp0 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                t_prof += time.perf_counter() - tp0
                ms = (ctypes.c_double * 7)(); n = (ctypes.c_int * 7)(); fl = (ctypes.c_double * 7)()
                rc = lib.pivp_plan_profile_read(plan.h, ms, n, fl)
                assert rc == 0, rc
                ms_tot += np.array(ms[:]); n_tot += np.array(n[:]); flops += np.array(fl[:])
            lib.pivp_plan_set_profiling(plan.h, 0)
            total_flops = float(flops.sum())
            total_s = float(ms_tot.sum()) * 1e-3
            achieved = total_flops / total_s / 1e12
            bf16 = args.precision != 'fp32'
            # bf16x3 executes three bf16 MFMAs per algorithmic product: its ceiling in algorithmic flops is a third of the bf16 peak
            peak = (PEAK_BF16_MFMA_TFLOPS / (3.0 if args.precision == 'bf16x3' else 1.0)) if bf16 else PEAK_FP32_MFMA_TFLOPS
            roofline = {
                'bound': 'mfma',
                'kernel': ('convlstm_bf16_kernel<NCH> (ConvLSTM 5x5 gate conv, %s, + fused gates)' %
                           ('bf16 operands' if args.precision == 'bf16' else 'fp32 operands as 2 bf16 pieces, 3 MFMAs per product') if bf16 else
                           'igemm_f32_kernel<WM,WN,4,true> (ConvLSTM 5x5 gate conv + fused gates)'),
                'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(achieved / peak, 4),
                'traffic': None if bf16 else _pmc_traffic(),
                'launches': int(n_tot.sum()), 'avg_launch_us': round(total_s / max(1, int(n_tot.sum())) * 1e6, 2),
                'algorithmic_gflop_per_launch': round(total_flops / max(1, int(n_tot.sum())) / 1e9, 3),
                'per_layer_tflops': {('lstm%d' % (i + 1)): round(float(flops[i] / (ms_tot[i] * 1e-3) / 1e12), 2)
                                     for i in range(7) if ms_tot[i] > 0},
                'share_of_step_time': round(total_s / args.steps / (elapsed / args.steps), 3),
                'ms_per_step_with_events': round(t_prof / args.steps * 1e3, 3),
            }

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import restatement as R
        from oracle.torch_restatement import TorchModel
        # The same workload as the GPU leg (batch, frame size, T).  16 threads: PyTorch-CPU on this model is fastest there on the GPU box's host
        # (scripts/cpu_thread_scan.py: 65 / 307 / 59 frames/s at 1 / 16 / 64 threads for B = 2; 128 threads, the default, gave 23)
        cpu_threads = min(16, torch.get_num_threads())
        torch.set_num_threads(cpu_threads)
        cb, ct = B, T
        P = R.init_params(seed=1, dtype=np.float32, scale=1.0, num_masks=nm, model_type=args.model, height=S, width=S)
        ci, ca, cs = R.synthetic_batch(cb, ct, S, S)
        tm = TorchModel(nm, is_cdna=args.model == 'CDNA', is_stp=args.model == 'STP', is_dna=args.model == 'DNA',
                        params=P, dtype=torch.float32)
        tm.train = False
        with torch.no_grad():
            tm([ci, ca, cs], 0); tm.reset_state()      # warm-up
            reps, c0 = 0, time.perf_counter()
            while time.perf_counter() - c0 < args.cpu_seconds:
                tm([ci, ca, cs], 0); tm.reset_state()
                reps += 1
            cel = time.perf_counter() - c0
        if train:      # same bounded sample, but forward + autograd backward + Chainer-rule Adam
            from oracle.torch_restatement import chainer_adam_step
            tmt = TorchModel(nm, is_cdna=True, params=P, dtype=torch.float32, requires_grad=True)
            Pm = {k: v.detach().numpy() for k, v in tmt.p.items()}
            Mm = {k: np.zeros_like(v) for k, v in Pm.items()}; Vm = {k: np.zeros_like(v) for k, v in Pm.items()}
            reps, c0 = 0, time.perf_counter()
            while time.perf_counter() - c0 < args.cpu_seconds:
                for v in tmt.p.values():
                    v.grad = None
                l = tmt([ci, ca, cs], 0); l.backward(); tmt.reset_state()
                with torch.no_grad():
                    chainer_adam_step(Pm, {k: v.grad.numpy() for k, v in tmt.p.items()}, Mm, Vm, reps + 1)
                reps += 1
            cel = time.perf_counter() - c0
        cpu_baseline = {'value': round(cb * (ct - 1) * reps / cel, 2), 'unit': 'predicted frames/s',
                        'cores': cpu_threads, 'kind': 'port',
                        'sample': '%d %s of B=%d T=%d %dx%d %s, fp32 PyTorch-CPU restatement of the reference path '
                                  '(oracle/torch_restatement.py), %.1f s' % (reps, 'train steps (fwd+bwd+Adam)' if train else 'rollouts',
                                                                              cb, ct, S, S, args.model, cel)}

    if rank == 0:
        frames = world * B * (T - 1) * args.steps
        out = {
            'metric': 'predicted frames/sec (64x64x3, 10-step rollout)',
            'value': round(frames / elapsed, 1),
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'bf16': 'bf16 ConvLSTM operands, f32 accumulate and elsewhere',
                      'bf16x3': 'f32 operands of the ConvLSTM forward as 2 bf16 pieces (3 bf16 MFMAs per product), f32 accumulate and elsewhere'}[args.precision],
            'data': 'synthetic',
            'config': {'workload': '%s %s, batch %d/GPU, %d-frame %dx%dx3 sequences, action-conditioned, num_masks=%d, '
                                   'random-init weights' % (args.model, 'train step (optimizer.update: forward + BPTT backward + grad '
                                   'all-reduce + Adam, schedsamp_k=-1)' if train else 'rollout forward (Model.__call__, feed-self)',
                                   B, T, S, S, nm),
                       'global_batch': world * B, 'frames_per_step': world * B * (T - 1),
                       'parallelism': ('dp%d (RCCL all-reduce of the flat gradient)' if train else 'replicas x%d') % world,
                       'loss': loss_val},
            'roofline': roofline,
            'cpu_baseline': cpu_baseline,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
