#!/usr/bin/env python
"""Benchmark of the hot path: predicted frames/s of the 10-frame 64x64x3 CDNA rollout (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
One process per GPU.  When the driver launches the ranks itself (torch.distributed.run: WORLD_SIZE / RANK / LOCAL_RANK in the
environment) this file is a rank.  Started plainly with --gpus N > 1 it is the LAUNCHER: before anything touches the GPU it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a child
process, relays rank 0's JSON line and exits with the child's code.  A world size different from --gpus is an error, never a
silent 1-GPU run.  Every rank runs the same per-GPU workload (weak scaling).

Two legs, both in the ONE JSON line rank 0 prints:
  value / ms_per_step : the rollout, Model.__call__ forward, feed-self (predict_model.py:126-128).  It shards over the batch with
                        no data-path collective (SURVEY.md 8e "inference rollout: replicas only").
  train               : optimizer.update (train_model.py:950) = forward + BPTT backward + gradient all-reduce over RCCL, overlapped
                        with the backward sweep + Adam: ms_per_step, frames_per_s (whole job), rccl_ranks, and
                        allreduce_ms_exposed = that step time minus the time of the same step with the collective switched off.
  train_bf16          : the same leg in the bf16 precision mode (fp32 runs only): BASELINE.json's config 3 is bf16 data parallel, so at
                        --gpus 8 (global batch 256) this object is config 3.

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


def _pmc_traffic(bf16=False):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/), or None.
    PMC counters cannot be collected inside this process; the passes are re-run per round on the same command."""
    try:
        best = None
        pdir = os.path.join(ROOT, 'profiles')
        name = 'pmc_traffic_bf16.json' if bf16 else 'pmc_traffic.json'
        for rnd in sorted(os.listdir(pdir)):
            f = os.path.join(pdir, rnd, name)
            if os.path.exists(f):
                best = f
        if best is None:
            return None
        with open(best) as fh:
            return round(json.load(fh)['hbm_bytes_per_launch'])
    except Exception:
        return None


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--config', type=int, default=None, choices=[2, 3, 4, 5],
                    help="preset of a BASELINE.json config, numbered from 1 as VERDICT / DESIGN do (config N = `configs[N-1]`; per-GPU batch 32 throughout): 2 = the default "
                         "(CDNA 64x64 T=10 fp32 rollout), 3 = --precision bf16 --mode train (at --gpus 8: global batch 256), 4 = --model STP, "
                         "5 = --size 128 --seq-len 20.  Explicit flags given after it still win.")
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='sequences per GPU (config 2: 32)')
    ap.add_argument('--seq-len', type=int, default=10)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--model', default='CDNA', choices=['CDNA', 'STP', 'DNA'])
    ap.add_argument('--mode', default='rollout', choices=['rollout', 'train'],
                    help='rollout (default): `value` is the rollout and the train step is reported in `train`; train: only the '
                         'train step runs and `value` is its frames/s (profiling runs)')
    ap.add_argument('--no-train', action='store_true', help='rollout mode: skip the train leg')
    ap.add_argument('--no-bf16x6', action='store_true', help='fp32 rollout runs: skip the additional three-piece rollout leg (`rollout_bf16x6`)')
    ap.add_argument('--no-bf16-train', action='store_true', help='fp32 runs: skip the additional bf16 train leg (`train_bf16`, config 3\'s arithmetic)')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16', 'bf16x3', 'bf16x6', 'fp16x3'],
                    help='fp32: the parity path and the headline metric (config 2). bf16: ConvLSTM gate convolutions with bf16 operands, '
                         'fp32 accumulation (config 3); reports its per-pixel error instead of meeting the 1e-4 gate')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='collective backend of the ranks: nccl = RCCL over xGMI (the product); gloo stages device tensors through the host')
    ap.add_argument('--share-gpu', action='store_true',
                    help='rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo: RCCL refuses two ranks on one device). '
                         'Real kernels and the real data-parallel step; the numbers are NOT a multi-GPU measurement.')
    ap.add_argument('--dry', action='store_true',
                    help='host logic only: gloo on CPU, stub kernels (HostStubModel); exercises the launcher, the rank bookkeeping, the '
                         'barrier / max-over-ranks timing and the overlapped all-reduce without a GPU.  Not a measurement.')
    return ap


# init_process_group / every collective: a hung RCCL bootstrap must end the run, not burn the box's lease.  240 s, not less: on a fresh box the
# first `import torch` of N concurrent ranks takes 1-2 minutes, and the ranks reach the rendezvous that far apart at worst.
RENDEZVOUS_TIMEOUT_S = int(os.environ.get('PIVP_RENDEZVOUS_TIMEOUT', '240'))


def _launch_timeout(args):
    """Wall-clock bound of the whole N-rank child (PIVP_BENCH_TIMEOUT overrides): imports + rendezvous + every leg of the run.
    The default run is ~1 min per rank set on an MI355X; the bound is generous and only there so that a hang exits non-zero."""
    env = os.environ.get('PIVP_BENCH_TIMEOUT')
    if env:
        return float(env)
    return 600.0 + 4.0 * (args.steps + args.warmup) * max(1.0, (args.size / 64.0) ** 2 * args.seq_len / 10.0 * args.batch / 32.0)


def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher around us: start the N ranks as a child process tree.  Nothing in this process has touched
    the GPU (torch is not even imported yet), and the child is started with subprocess, never exec'd over us.  torchrun picks the
    rendezvous port itself (--standalone: a c10d store on a free port; no window in which another job can take a port we chose),
    the child is bounded in time, and every failure ends in a non-zero exit code with a one-line reason."""
    import signal
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))   # the ranks share the host's cores
    limit = _launch_timeout(args)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)          # the process group we started (torchrun + its ranks), nothing else
        except OSError:
            pass
        stdout, _ = proc.communicate()
        sys.stderr.write((stdout or '')[-4000:])
        sys.stderr.write('bench.py: the %d-rank run did not finish within %.0f s (hung rendezvous or collective?); killed\n' % (args.gpus, limit))
        return 124
    line = None
    for ln in stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + '\n')
    if proc.returncode != 0 or line is None:
        sys.stderr.write('bench.py: the %d-rank run failed (exit code %d%s)\n' % (
            args.gpus, proc.returncode, '' if line is not None or proc.returncode else ', no JSON line from rank 0'))
        return proc.returncode or 1
    got = json.loads(line).get('n_gpus')
    if got != args.gpus:
        sys.stderr.write('bench.py: asked for %d ranks, the run reports %r\n' % (args.gpus, got))
        return 1
    print(line)
    return 0


def timed(step, steps, warmup, sync, barrier):
    """W untimed warm-up steps, then exactly K steps between barrier + device synchronisation on both sides."""
    out = None
    for _ in range(warmup):
        out = step()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync(); barrier()
    return time.perf_counter() - t0, out


CONFIG_PRESETS = {      # BASELINE.json `configs[i]` -> the flags that select it (per-GPU batch 32; config 3 / 5 are quoted at 8 GPUs: --gpus 8)
    2: ([], '1xMI355X: batch=32, 64x64x3, 10-frame CDNA, action-conditioned, fp32'),
    3: (['--precision', 'bf16', '--mode', 'train'], '8xMI355X DP: global batch=256, 64x64x3, 10-frame CDNA, RCCL all-reduce over xGMI, bf16'),
    4: (['--model', 'STP'], 'STP variant (spatial-transformer predictor) in place of CDNA kernels, batch=32, 1 GPU'),
    5: (['--size', '128', '--seq-len', '20'], '128x128x3 frames, 20-step rollout, num_masks=10, 8xMI355X DP (bandwidth-bound stress)'),
}


def parse_args(argv):
    """--config N expands to its preset flags IN FRONT of the command line, so explicit flags override the preset."""
    ap = build_parser()
    first = ap.parse_args(argv)
    if first.config is None:
        return first
    args = ap.parse_args(CONFIG_PRESETS[first.config][0] + list(argv))
    return args


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    world = int(env_world or '1')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a run of a different size' % (args.gpus, world))

    import numpy as np
    import torch
    import pivp_amd

    dry = args.dry
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
        if args.share_gpu:
            if args.backend != 'gloo' and world > 1:
                raise SystemExit('bench.py: --share-gpu needs --backend gloo')
            local_rank = 0
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit('bench.py: rank %d has no GPU (%d visible)' % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
    dev = 'cpu' if dry else 'cuda:%d' % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import datetime
        backend = 'gloo' if (dry or args.backend == 'gloo') else 'nccl'
        try:
            tmo = datetime.timedelta(seconds=RENDEZVOUS_TIMEOUT_S)
            if backend == 'gloo':
                dist.init_process_group('gloo', timeout=tmo)
            else:
                dist.init_process_group('nccl', device_id=torch.device(dev), timeout=tmo)
            if dist.get_world_size() != args.gpus:
                raise RuntimeError('process group has %d ranks, --gpus %d' % (dist.get_world_size(), args.gpus))
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                         # the first collective builds the communicator: fail HERE, with the rank named
            if int(probe.item()) != world:
                raise RuntimeError('first all-reduce summed to %r over %d ranks' % (probe.item(), world))
        except BaseException as e:                         # noqa: B902
            sys.stderr.write('bench.py: rank %d/%d (local %d, device %s, backend %s, MASTER %s:%s) could not join the process group: %s: %s\n' % (
                rank, world, local_rank, dev, backend, os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'), type(e).__name__, e))
            sys.stderr.flush()
            os._exit(3)                                    # no destructor of a half-built communicator may block the exit

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    B, T, S = args.batch, args.seq_len, args.size
    nm = 1 if args.model == 'DNA' else 10
    np.random.seed(1234 + rank)
    kinds = dict(is_cdna=args.model == 'CDNA', is_stp=args.model == 'STP', is_dna=args.model == 'DNA')
    rs = np.random.RandomState(rank)
    if not dry:
        images = torch.from_numpy(rs.random_sample((T, B, 3, S, S)).astype(np.float32)).to(dev)
        actions = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).to(dev)
        states = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).to(dev)
    do_rollout = args.mode == 'rollout'
    do_train = args.mode == 'train' or not args.no_train

    elapsed = loss_val = None
    roofline = None
    model = None
    with pivp_amd.using_config('train', False):
        # ---- leg 1: the rollout (`value`) --------------------------------------------------------------------
        if do_rollout and not dry:
            model = pivp_amd.Model(nm, prefix='bench', device=dev, keep_activations=False, precision=args.precision, **kinds)

            def rollout_step():
                model.reset_state()
                return model([images, actions, states], 0)
            elapsed, loss = timed(rollout_step, args.steps, args.warmup, sync, barrier)
            loss_val = float(loss)
            elapsed = max_over_ranks(elapsed)
        elif do_rollout:
            elapsed, _ = timed(lambda: time.sleep(0.001), args.steps, args.warmup, sync, barrier)
            elapsed = max_over_ranks(elapsed); loss_val = 0.0

        # ---- leg 1b (fp32 rollout runs): the same rollout with the gate convolutions on the bf16 / fp16 matrix cores at fp32 grade: every fp32 operand
        # as three bf16 pieces and six MFMAs per product (417 TFLOP/s ceiling), or as two fp16 pieces and three MFMAs (833).  ADDITIONAL objects, never
        # `value`; `max_l2_vs_f32_rollout` is the largest per-pixel L2 between their frames and the fp32 rollout's on this run's input (the gates against
        # the float64 oracle are tests/test_gpu_trained.py's).
        split_objs = {}
        if do_rollout and not dry and args.precision == 'fp32' and not args.no_bf16x6 and world == 1:       # (one rank only: a leg that fails on one
            # rank alone would leave the others in a barrier; the multi-GPU runs measure `value` and the data-parallel train steps)
            for mode, what in (('bf16x6', '3 bf16 pieces, 6 bf16 MFMAs per product'), ('fp16x3', '2 fp16 pieces (weights packed times a per-tensor power of two), 3 fp16 MFMAs per product')):
                try:
                    m6 = pivp_amd.Model(nm, prefix='bench', device=dev, keep_activations=False, precision=mode, **kinds)

                    def x6_step():
                        m6.reset_state()
                        return m6([images, actions, states], 0)
                    x6_step()
                    m6._flat_params.copy_(model._flat_params)            # the fp32 leg's weights (parameters are lazily sized: after one call)
                    t6, _ = timed(x6_step, args.steps, args.warmup, sync, barrier)
                    t6 = max_over_ranks(t6)
                    rollout_step()
                    d = torch.stack(m6.gen_images).double() - torch.stack(model.gen_images).double()
                    l2_steps = [float(v) for v in d.pow(2).sum(dim=2).sqrt().flatten(1).max(dim=1).values]
                    obj = {'ms_per_step': round(t6 / args.steps * 1e3, 3), 'frames_per_s': round(world * B * (T - 1) * args.steps / t6, 1),
                           'max_l2_vs_f32_rollout': max(l2_steps),
                           # per predicted frame: the first is one pass through the network; with RANDOM-INIT weights (this run's) any two fp32-grade
                           # evaluations then drift apart by a factor per fed-back step -- the gate against float64 on trained weights
                           # is tests/test_gpu_trained.py::test_bf16x6_mode_is_fp32_grade_on_trained_weights
                           'max_l2_vs_f32_rollout_per_step': [float('%.3g' % v) for v in l2_steps],
                           'dtype': 'f32 operands of the ConvLSTM forward as %s (layers on 8-wide maps: the f32 kernel)' % what}
                    if rank == 0 and not args.no_roofline:
                        r6 = roofline_pass(args, m6, x6_step, t6, np, torch, precision=mode)
                        obj.update({'achieved_tflops': r6['achieved'], 'peak_tflops': r6['peak'], 'frac': r6['frac'], 'per_layer_tflops': r6['per_layer_tflops'],
                                    'layers_in_this_arithmetic': r6['layers'],
                                    # (nominal peak at 2.4 GHz; measured: on real operands 16-bit MFMA loops hold 1.81-1.85 GHz at ~1270 W)
                                    'peak_note': 'nominal (2.4 GHz); these loops are power-limited at ~0.76 of it on real operands: profiles/r04/clock_power_operand_values.txt'})
                    split_objs[mode] = obj
                    del m6
                except Exception as e:
                    sys.stderr.write('bench.py: rollout_%s leg failed (%s: %s)\n' % (mode, type(e).__name__, e))

        # ---- leg 2: the data-parallel train step (the run's precision; in the default fp32 run also config 3's bf16 arithmetic) ----------
        train_obj = train_bf16_obj = train_x6_obj = None
        tmodel = opt = None
        dp = pivp_amd.GradAllReduce() if (do_train and world > 1) else None

        def train_leg(precision):
            """-> (result object, model, optimizer, seconds for K steps with the collective, last loss)"""
            use_dp = [True]
            if dry:
                sys.path.insert(0, os.path.join(ROOT, 'tests'))
                from host_stub import HostStubModel          # the test double lives with the tests, not in the product package
                tm = HostStubModel(sizes=(1 << 16, 1 << 14, 1 << 15, 1 << 15, 1 << 16, 1 << 14), value=float(rank + 1))
                op = None
                ngrad = sum(tm.sizes)

                def train_step():
                    tm.cleargrads()
                    if dp is not None and use_dp[0]:
                        dp.backward_and_allreduce(tm)
                    else:
                        tm.backward()
                    return 0.0
            else:
                tm = pivp_amd.Model(nm, prefix='bench', device=dev, keep_activations=True, precision=precision, **kinds)
                op = pivp_amd.Adam(alpha=0.001).setup(tm, data_parallel=dp)             # TM:860-861
                tm([images, actions, states], 0)                                        # parameters are lazily sized: one forward first
                if world > 1:                                                           # identical replicas: rank 0's initialisation
                    dist.broadcast(tm._flat_params, src=0)
                ngrad = int(tm._flat_params.numel())

                def train_step():
                    tm.reset_state()
                    op._dp = dp if use_dp[0] else None
                    return op.update(tm, [images, actions, states], 0)                  # schedsamp_k = -1: feed-self, deterministic
            t_with, tloss = timed(train_step, args.steps, args.warmup, sync, barrier)
            t_with = max_over_ranks(t_with)
            algo = dp.last_algo if dp is not None else None
            payload_bytes = dp.last_payload_bytes if dp is not None else None
            t_without = None
            compare = None
            if world > 1:
                if precision == 'bf16' and dp.algo == 'auto' and not dry:
                    # SURVEY.md 5's "measured comparison": the same step with the other schedule of the bf16 payload (the default, all-links
                    # reduce-scatter + all-gather with an fp32 local sum, against one ring all-reduce that sums in bf16)
                    compare = {algo: round(t_with / args.steps * 1e3, 3)}
                    other = 'allreduce' if algo == 'rs_ag' else 'rs_ag'
                    dp.algo = other
                    try:
                        t_other, _ = timed(train_step, args.steps, max(1, args.warmup // 2), sync, barrier)
                        compare[other] = round(max_over_ranks(t_other) / args.steps * 1e3, 3)
                    finally:
                        dp.algo = 'auto'
                # the same step with the collective switched off (replicas drift apart: timing only, run last)
                use_dp[0] = False
                t_without, _ = timed(train_step, args.steps, max(1, args.warmup // 2), sync, barrier)
                t_without = max_over_ranks(t_without)
                use_dp[0] = True
                if op is not None:
                    op._dp = dp
            obj = {
                'ms_per_step': round(t_with / args.steps * 1e3, 3),
                'frames_per_s': round(world * B * (T - 1) * args.steps / t_with, 1),
                'rccl_ranks': dist.get_world_size() if dist is not None else 1,
                'backend': (dist.get_backend() if dist is not None else None),
                'ms_per_step_without_allreduce': None if t_without is None else round(t_without / args.steps * 1e3, 3),
                'allreduce_ms_exposed': 0.0 if t_without is None else round(max(0.0, t_with - t_without) / args.steps * 1e3, 3),
                # what one rank hands to the all-reduce per step: the fp32 flat gradient buffer, or its bf16 image in the bf16 precision
                # mode (parallel.GradAllReduce(payload='auto'))
                # what one rank hands to the collective per step (GradAllReduce.last_payload_bytes); with one rank nothing travels: the figure
                # is what a data-parallel rank of this precision would send
                'gradient_bytes_per_step': payload_bytes if payload_bytes else (2 if precision == 'bf16' else 4) * ngrad,
                'gradient_payload': 'bf16 (summed over the ranks in fp32, one rounding; fp32 flat gradient buffer and Adam)' if precision == 'bf16' else 'fp32',
                'allreduce': 'none (1 rank)' if world == 1 else '6 gradient groups, SUM, issued from inside the backward sweep of t = 0 on a side stream',
                'allreduce_algo': algo if world > 1 else 'none (1 rank; a data-parallel run of this precision uses %s)' % ('rs_ag' if precision == 'bf16' else 'allreduce'),
                'allreduce_algo_ms_per_step': compare,
                'dtype': {'fp32': 'f32', 'bf16': 'bf16 ConvLSTM / enc5 / enc6 operands, f32 accumulate, gradients and optimizer',
                          'bf16x3': 'f32 as 2 bf16 pieces in the ConvLSTM forward and data gradients',
                          'bf16x6': 'f32 as 3 bf16 pieces in the ConvLSTM gate convolutions, their data and weight gradients (6 bf16 MFMAs per product: fp32-grade), f32 elsewhere',
                          'fp16x3': 'f32 as 2 fp16 pieces (3 fp16 MFMAs per product, f32 accumulate) in the ConvLSTM gate convolutions, their data and weight gradients (dG scaled by a power of two) and enc5 / enc6; f32 elsewhere'}[precision],
                'workload': 'optimizer.update (TM:950): forward + BPTT backward + gradient all-reduce + Adam, schedsamp_k=-1, batch %d/GPU' % B,
                'loss': float(tloss),
            }
            return obj, tm, op, t_with, float(tloss)

        if do_train:
            train_obj, tmodel, opt, t_with, tloss = train_leg(args.precision)
            if args.mode == 'train':
                elapsed, loss_val = t_with, tloss
            elif args.precision == 'fp32' and not dry and not args.no_bf16_train:
                # BASELINE.json config 3 names bf16 for the data-parallel configuration: at --gpus 8 this object IS config 3 (global batch 256)
                train_bf16_obj, m16, _, _, _ = train_leg('bf16')
                del m16
            if args.mode != 'train' and args.precision == 'fp32' and not dry and not args.no_bf16x6 and world == 1:
                # the fp32-grade train step on the bf16 matrix cores: gate convolutions and their data gradients as six bf16 MFMAs per product
                # (three pieces per fp32 operand), weight gradients and everything else fp32.  An additional object: `train` stays the fp32 kernels'.
                train_x6_obj = {}
                for mode in ('bf16x6', 'fp16x3'):       # (fp16x3: its forward, and its data gradients with dG scaled by a power of two)
                    try:
                        train_x6_obj[mode], m6t, _, _, _ = train_leg(mode)
                        del m6t
                    except Exception as e:
                        sys.stderr.write('bench.py: train_%s leg failed (%s: %s)\n' % (mode, type(e).__name__, e))

        # ---- the dominant kernel against its roofline, HIP events on the launch stream, second pass over the same K steps ----------
        pmodel = model if do_rollout else tmodel
        if not args.no_roofline and not dry:
            def prof_step():
                pmodel.reset_state()
                if do_rollout:
                    return pmodel([images, actions, states], 0)
                return opt.update(pmodel, [images, actions, states], 0)
            if rank != 0 and not do_rollout and world > 1:
                for _ in range(args.steps):        # rank 0's instrumented pass calls the gradient all-reduce: every rank must take part
                    prof_step()
                sync()
            if rank == 0:
                roofline = roofline_pass(args, pmodel, prof_step, elapsed, np, torch)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
        try:
            cpu_baseline = cpu_baseline_leg(args, B, T, S, nm, np, torch, train=args.mode == 'train')
        except Exception as e:                             # the GPU legs are measured: a failing CPU leg must not discard them
            sys.stderr.write('bench.py: cpu_baseline leg failed (%s: %s); reporting cpu_baseline = null\n' % (type(e).__name__, e))

    if rank == 0:
        frames = world * B * (T - 1) * args.steps
        train_mode = args.mode == 'train'
        preset = ''
        if args.config is not None:     # name the BASELINE.json config this run is (one rank's share of it when it is quoted at 8 GPUs)
            share = '' if args.config not in (3, 5) or world == 8 else "; this run: %d of its 8 ranks' shares" % world
            preset = 'BASELINE.json config %d = configs[%d] (%s%s): ' % (args.config, args.config - 1, CONFIG_PRESETS[args.config][1], share)
        out = {
            'metric': 'predicted frames/sec (%dx%dx3, %d-step rollout)' % (S, S, T),
            'value': round(frames / elapsed, 1) if not dry else 0.0,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'bf16': 'bf16 ConvLSTM operands, f32 accumulate and elsewhere',
                      'bf16x3': 'f32 operands of the ConvLSTM forward as 2 bf16 pieces (3 bf16 MFMAs per product), f32 accumulate and elsewhere',
                      'bf16x6': 'f32 operands of the ConvLSTM forward as 3 bf16 pieces (6 bf16 MFMAs per product: fp32-grade products), f32 accumulate and elsewhere',
                      'fp16x3': 'f32 operands of the ConvLSTM forward (train mode: and of its data / weight gradients) as 2 fp16 pieces (3 fp16 MFMAs per product: 22-bit operands), f32 accumulate and elsewhere'}[args.precision],
            'data': ('synthetic' if not args.share_gpu else 'synthetic; REHEARSAL: %d ranks share one GPU over gloo, not a multi-GPU measurement' % world)
                    if not dry else 'none: --dry run of the host logic on CPU (gloo, stub kernels); NOT a measurement',
            'config': {'workload': preset + '%s %s, batch %d/GPU, %d-frame %dx%dx3 sequences, action-conditioned, num_masks=%d, '
                                   'random-init weights' % (args.model, 'train step (optimizer.update: forward + BPTT backward + grad '
                                   'all-reduce + Adam, schedsamp_k=-1)' if train_mode else 'rollout forward (Model.__call__, feed-self)',
                                   B, T, S, S, nm),
                       'global_batch': world * B, 'frames_per_step': world * B * (T - 1),
                       'parallelism': ('dp%d (RCCL all-reduce of the flat gradient)' if train_mode else 'replicas x%d') % world,
                       'loss': loss_val},
            'roofline': roofline,
            'cpu_baseline': cpu_baseline,
        }
        for mode, obj in split_objs.items():
            out['rollout_' + mode] = obj
        if not train_mode:
            out['train'] = train_obj
            if train_bf16_obj is not None:
                out['train_bf16'] = train_bf16_obj
            for mode, obj in (train_x6_obj or {}).items():
                out['train_' + mode] = obj
        if dry:
            out['dry'] = True
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        barrier()                      # rank 0 was still measuring (roofline pass): every rank leaves the group together
        dist.destroy_process_group()


def roofline_pass(args, model, step, elapsed, np, torch, precision=None):
    precision = precision or args.precision
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
    layers = list(range(7))
    if precision in ('bf16x6', 'fp16x3'):      # layers on maps that are not a multiple of 16 wide run the fp32 kernel in this mode: not part of its fraction
        widths = [args.size // 2, args.size // 2, args.size // 4, args.size // 4, args.size // 8, args.size // 4, args.size // 2]
        layers = [i for i in range(7) if widths[i] % 16 == 0 or (precision == 'fp16x3' and widths[i] % 8 == 0 and args.batch % 2 == 0)]     # (fp16 pieces: 8-wide maps too)
    total_flops = float(flops[layers].sum())
    total_s = float(ms_tot[layers].sum()) * 1e-3
    achieved = total_flops / total_s / 1e12
    bf16 = precision != 'fp32'
    # bf16x3 / bf16x6 execute three / six bf16 MFMAs per algorithmic product: their ceiling in algorithmic flops is a third / a sixth of the bf16 peak
    peak = (PEAK_BF16_MFMA_TFLOPS / {'bf16x3': 3.0, 'bf16x6': 6.0, 'fp16x3': 3.0}.get(precision, 1.0)) if bf16 else PEAK_FP32_MFMA_TFLOPS
    return {
        'layers': ['lstm%d' % (i + 1) for i in layers],
        'bound': 'mfma',
        'kernel': ('convlstm_bf16_kernel<NCH> (ConvLSTM 5x5 gate conv, %s, + fused gates)' %
                   {'bf16': 'bf16 operands', 'bf16x3': 'fp32 operands as 2 bf16 pieces, 3 MFMAs per product',
                    'bf16x6': 'fp32 operands as 3 bf16 pieces, 6 MFMAs per product', 'fp16x3': 'fp32 operands as 2 fp16 pieces, 3 MFMAs per product'}[precision] if bf16 else
                   'igemm_f32_kernel<WM,WN,4,true> (ConvLSTM 5x5 gate conv + fused gates)'),
        'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
        'frac': round(achieved / peak, 4),
        'traffic': _pmc_traffic(bf16),
        'traffic_source': 'HBM bytes per launch from the committed rocprofv3 --pmc passes of this command (profiles/*/pmc_traffic%s.json: '
                          'FETCH_SIZE x2 per the gfx950 note + WRITE_SIZE); not measured in this run' % ('_bf16' if bf16 else ''),
        'launches': int(n_tot[layers].sum()), 'avg_launch_us': round(total_s / max(1, int(n_tot[layers].sum())) * 1e6, 2),
        'algorithmic_gflop_per_launch': round(total_flops / max(1, int(n_tot[layers].sum())) / 1e9, 3),
        'per_layer_tflops': {('lstm%d' % (i + 1)): round(float(flops[i] / (ms_tot[i] * 1e-3) / 1e12), 2)
                             for i in layers if ms_tot[i] > 0},
        'share_of_step_time': round(total_s / args.steps / (elapsed / args.steps), 3),
        'ms_per_step_with_events': round(t_prof / args.steps * 1e3, 3),
    }


def cpu_baseline_leg(args, B, T, S, nm, np, torch, train):
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
    roll_per_rep = cel / max(1, reps)
    if train:      # same bounded sample, but forward + autograd backward + Chainer-rule Adam
        from oracle.torch_restatement import chainer_adam_step
        tmt = TorchModel(nm, is_cdna=args.model == 'CDNA', is_stp=args.model == 'STP', is_dna=args.model == 'DNA',
                         params=P, dtype=torch.float32, requires_grad=True)
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
    value = round(cb * (ct - 1) * reps / cel, 2)
    one_thread = None
    try:        # "plus a 1-thread run" -- bounded (~8 s): the same workload when one rollout fits, else a B = 2 sample of it
        ob = cb if roll_per_rep * cpu_threads * 0.4 < 8.0 else 2      # one thread is ~6x slower than 16 on this model (scripts/cpu_thread_scan.py)
        oi, oa, os_ = (ci, ca, cs) if ob == cb else R.synthetic_batch(ob, ct, S, S)
        torch.set_num_threads(1)
        with torch.no_grad():
            r1, c1 = 0, time.perf_counter()
            while r1 == 0 or time.perf_counter() - c1 < 8.0:
                tm([oi, oa, os_], 0); tm.reset_state()
                r1 += 1
            e1 = time.perf_counter() - c1
        one_thread = {'value': round(ob * (ct - 1) * r1 / e1, 2), 'unit': 'predicted frames/s', 'cores': 1,
                      'sample': '%d rollouts of B=%d T=%d %dx%d %s, %.1f s%s' % (r1, ob, ct, S, S, args.model, e1,
                                                                                  ' (rollout forward only)' if train else '')}
    finally:
        torch.set_num_threads(cpu_threads)
    return {'value': value, 'unit': 'predicted frames/s',
            'cores': cpu_threads, 'host_cores': os.cpu_count(), 'one_thread': one_thread, 'kind': 'port',
            'sample': '%d %s of B=%d T=%d %dx%d %s, fp32 PyTorch-CPU restatement of the reference path '
                      '(oracle/torch_restatement.py), %.1f s on %d threads (the measured optimum of PyTorch-CPU on this model; the host has %d)' % (reps, 'train steps (fwd+bwd+Adam)' if train else 'rollouts',
                                                                  cb, ct, S, S, args.model, cel, cpu_threads, os.cpu_count() or 0)}


if __name__ == '__main__':
    main()
