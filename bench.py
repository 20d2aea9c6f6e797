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

The run is a sequence of LEGS, least risky first (rollout: no data-path collective; fp32 train step: one all-reduce per gradient group; bf16 train
step: all-to-all + all-gather per group; then rank 0's roofline and CPU passes).  Every leg is entered by all ranks together: set-up without collectives,
one MIN all-reduce of an ok flag, the measured body, a second agreement.  A Python-side failure on any rank drops that leg on every rank and the run goes
on; a hang is ended by each rank's watchdog thread at the leg's deadline / the run's budget (450 s from the job's start: inside the driver's 600 s), and a
SIGTERM from the launcher (another rank died) is taken by the same thread.  In every case rank 0 prints the ONE line built from the legs that DID finish,
with `incomplete` naming the others and `incomplete_reason` saying why; no field is ever made up, and without a finished `value` leg there is no line.
Exit code 0 iff the leg that carries `value` finished on all ranks.

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
    ap.add_argument('--main-priority', default='auto', choices=['auto', 'on', 'off'],
                    help='wave priority 3 for the backward sweep\'s kernels (auto: the library\'s rule -- on unless a gradient listener is registered, i.e. on at 1 GPU)')
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
# Every rank is done -- line printed, process gone -- this many seconds after the JOB started (the launcher's start; under an outer torchrun, that
# launcher's).  The driver gives `bench.py` 600 s: a run that cannot finish says what it has measured inside that limit instead of being killed mute.
BUDGET_S = float(os.environ.get('PIVP_BENCH_BUDGET', '450'))
MAX_IMPORT_ALLOWANCE_S = 240.0   # a job start older than this is not believed (see _job_start): imports + rendezvous of N ranks on a fresh box
LAUNCH_GRACE_S = 30.0           # the launcher waits this much longer than the ranks' own budget: 480 s for the default arguments


def _launch_timeout(args):
    """Wall-clock bound of the whole N-rank child (PIVP_BENCH_TIMEOUT overrides): imports + rendezvous + every leg of the run.  The default run
    takes ~1 min; the ranks watch the same budget themselves (`Run.deadline`) and skip legs or stop with their partial line before this fires."""
    env = os.environ.get('PIVP_BENCH_TIMEOUT')
    if env:
        return float(env)
    return BUDGET_S + LAUNCH_GRACE_S


def _proc_start_epoch(pid):
    """wall-clock time at which process `pid` started (Linux /proc), or None"""
    try:
        with open('/proc/%d/stat' % pid) as f:
            ticks = int(f.read().rsplit(')', 1)[1].split()[19])          # field 22, `starttime`, in clock ticks since boot
        with open('/proc/uptime') as f:
            up = float(f.read().split()[0])
        t = time.time() - (up - ticks / float(os.sysconf('SC_CLK_TCK')))
        return t if 0.0 <= time.time() - t < 3600.0 else None
    except Exception:
        return None


def _parent_is_torchrun(pid):
    """is process `pid` a torch.distributed launcher (torchrun / `python -m torch.distributed.run|launch`)?  Only such a parent's age is the job's age: a
    long-lived shell, a driver that exports WORLD_SIZE, a PyTorchJob entrypoint or an elastic agent restarting workers can be hours old."""
    try:
        with open('/proc/%d/cmdline' % pid, 'rb') as f:
            words = f.read().decode('utf-8', 'replace').split('\0')
    except Exception:
        return False
    joined = ' '.join(words)
    return ('torch.distributed.run' in joined or 'torch.distributed.launch' in joined or
            any(os.path.basename(w) == 'torchrun' for w in words[:3]))


def _job_start():
    """When the job this process belongs to started: the launcher's clock (PIVP_BENCH_T0), else -- a rank under somebody else's torchrun -- the
    start of that launcher (its `import torch` can take minutes on a fresh box and counts against the caller's limit), else this process's own
    start.  Whatever the source, the start is never taken to be older than MAX_IMPORT_ALLOWANCE_S: a stale clock (a parent that is not the
    launcher, a PIVP_BENCH_T0 inherited from an earlier job) must not put `Run.deadline` in the past before the first leg has run.
    Returns (t0, source)."""
    now = time.time()
    env = os.environ.get('PIVP_BENCH_T0')
    t, src = None, None
    if env:
        try:
            t, src = float(env), 'PIVP_BENCH_T0'
        except ValueError:
            t = None
    if t is None and os.environ.get('WORLD_SIZE') is not None and _parent_is_torchrun(os.getppid()):
        t = _proc_start_epoch(os.getppid())
        src = 'start of the parent launcher (pid %d)' % os.getppid()
    if t is None:
        t, src = _proc_start_epoch(os.getpid()), 'start of this process'
    if t is None:
        t, src = now, 'now'
    if now - t > MAX_IMPORT_ALLOWANCE_S or t > now:
        t, src = now - min(MAX_IMPORT_ALLOWANCE_S, max(0.0, now - (_proc_start_epoch(os.getpid()) or now))), src + ', clamped (stale clock)'
    return t, src


def _assemble_from_partial(path, reason):
    """the line rank 0 would have printed, from the side file it keeps (finished legs only), or None"""
    try:
        with open(path) as f:
            st = json.load(f)
    except Exception:
        return None
    line = st.get('line') or {}
    if 'value' not in line:
        return None
    line['incomplete'] = [l for l in st.get('planned', []) if l not in st.get('finished', [])]
    why = dict(st.get('skipped') or {})
    why['run'] = reason
    line['incomplete_reason'] = why
    return line


def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher around us: start the N ranks as a child process tree.  Nothing in this process has touched
    the GPU (torch is not even imported yet), and the child is started with subprocess, never exec'd over us.  torchrun picks the
    rendezvous port itself (--standalone: a c10d store on a free port; no window in which another job can take a port we chose),
    the child is bounded in time, and every failure ends in a one-line reason.  What the run HAS measured is never lost: rank 0 keeps
    the finished legs in a side file (PIVP_BENCH_PARTIAL) and prints its line itself when its budget runs out or the launcher ends it;
    if the child still dies without a line, the line is assembled here from the side file, with `incomplete` naming the missing legs.
    Exit code 0 iff the leg that carries `value` finished on all ranks."""
    import signal
    import subprocess
    import tempfile
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))   # the ranks share the host's cores
    env.setdefault('PIVP_BENCH_T0', repr(time.time()))
    fd, partial = tempfile.mkstemp(prefix='pivp_bench_partial_', suffix='.json')
    os.close(fd)
    env['PIVP_BENCH_PARTIAL'] = partial
    limit = _launch_timeout(args)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    reason = None
    try:
        stdout, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)          # the process group we started (torchrun + its ranks), nothing else
        except OSError:
            pass
        stdout, _ = proc.communicate()
        reason = 'the %d-rank run did not finish within %.0f s (hung rendezvous or collective?); killed' % (args.gpus, limit)
        sys.stderr.write((stdout or '')[-4000:])
        sys.stderr.write('bench.py: %s\n' % reason)
    line = None
    for ln in (stdout or '').splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        elif reason is None:
            sys.stderr.write(ln + '\n')
    rc = 124 if reason else proc.returncode
    if rc != 0 and reason is None:
        reason = 'the %d-rank run failed (exit code %d)' % (args.gpus, rc)
        sys.stderr.write('bench.py: %s\n' % reason)
    parsed = None
    if line is not None:
        try:
            parsed = json.loads(line)
        except ValueError:
            parsed = None
    if parsed is None:                                   # rank 0 never printed: what it had finished is in its side file
        parsed = _assemble_from_partial(partial, reason or 'no JSON line from rank 0')
        if parsed is not None:
            sys.stderr.write('bench.py: rank 0 printed no line; assembled from its finished legs (%s)\n' % partial)
    try:
        os.unlink(partial)
    except OSError:
        pass
    if parsed is None:
        if rc == 0:
            sys.stderr.write('bench.py: the %d-rank run ended without a JSON line from rank 0\n' % args.gpus)
        return rc or 1
    if parsed.get('n_gpus') != args.gpus:
        sys.stderr.write('bench.py: asked for %d ranks, the run reports %r\n' % (args.gpus, parsed.get('n_gpus')))
        return 1
    print(json.dumps(parsed))
    value_leg = 'train' if args.mode == 'train' else 'rollout'
    return 0 if value_leg not in parsed.get('incomplete', []) else (rc or 1)


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


class Run(object):
    """What this rank has measured so far, and the clock it runs against.  The contract line can be assembled from it at ANY time: at the normal
    end, when the budget runs out in the middle of a leg (a hung collective cannot be recovered in-process: the watchdog thread prints what is
    finished and ends the process), or when the launcher ends the job (SIGTERM after another rank died)."""

    def __init__(self, rank, world, value_leg):
        import threading
        self.rank, self.world, self.value_leg = rank, world, value_leg
        self.t0, self.t0_source = _job_start()
        self.deadline = self.t0 + BUDGET_S
        if world > 1:
            sys.stderr.write('bench.py: rank %d: job start = %s, %.0f s ago; %.0f s of the %.0f s budget left\n'
                             % (rank, self.t0_source, time.time() - self.t0, self.deadline - time.time(), BUDGET_S))
        self.line = {}              # the contract line, as far as it is measured (rank 0)
        self.planned = []           # the legs of this run, in order
        self.finished = []
        self.skipped = {}           # leg -> why it is not in the line
        self.leg = None
        self.leg_deadline = None
        self.value_done = False     # the leg that carries `value` has finished on ALL ranks
        self.closed = False
        self.emitted = False
        self.lock = threading.Lock()
        self.partial_path = os.environ.get('PIVP_BENCH_PARTIAL') if rank == 0 else None

    def remaining(self):
        return self.deadline - time.time()

    def begin(self, leg, budget_s):
        self.leg = leg
        self.leg_deadline = min(self.deadline, time.time() + budget_s)

    def end(self, leg, ok):
        self.leg = None
        self.leg_deadline = None
        if ok:
            self.finished.append(leg)
            if leg == self.value_leg:
                self.value_done = True
        self.save_partial()

    def save_partial(self):
        if self.partial_path is None:
            return
        try:
            tmp = self.partial_path + '.tmp'
            with open(tmp, 'w') as f:
                json.dump({'line': self.line, 'planned': self.planned, 'finished': self.finished, 'skipped': self.skipped}, f)
            os.replace(tmp, self.partial_path)
        except Exception:
            pass

    def emit(self, reason=None):
        """rank 0: print the ONE line (once).  Only measured fields: without a finished `value` leg nothing is printed."""
        with self.lock:
            if self.emitted or self.rank != 0:
                return
            self.emitted = True
            if 'value' not in self.line:
                return
            out = dict(self.line)
            out['incomplete'] = [l for l in self.planned if l not in self.finished]
            why = dict(self.skipped)
            if reason:
                why['run'] = reason
            if why:
                out['incomplete_reason'] = why
            sys.stdout.write(json.dumps(out) + '\n')
            sys.stdout.flush()

    def abort(self, reason):
        """from the watchdog thread: say why, print what is finished (rank 0), leave.  os._exit: the main thread may sit in a collective forever."""
        sys.stderr.write('bench.py: rank %d/%d: %s%s\n' % (self.rank, self.world, reason, '' if self.leg is None else ' (in leg %r)' % self.leg))
        sys.stderr.flush()
        self.emit(reason + ('' if self.leg is None else ' (in leg %r)' % self.leg))
        os._exit(0 if self.value_done else 124)


def _start_watchdog(run):
    """SIGTERM is blocked in every thread (the mask is inherited by threads created later) and taken synchronously here: a Python signal handler
    would never run while the main thread sits in a collective or a device synchronisation."""
    import signal
    import threading
    signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})

    def loop():
        while not run.closed:
            got = None
            try:
                got = signal.sigtimedwait({signal.SIGTERM}, 0.25)
            except (InterruptedError, OSError):
                time.sleep(0.25)
            now = time.time()
            if got is not None:
                run.abort('terminated by the launcher (SIGTERM: another rank failed, or the caller\'s limit)')
            if run.closed:
                return
            if now > run.deadline:
                run.abort('the budget of %.0f s from the job\'s start is used up' % BUDGET_S)
            ld = run.leg_deadline
            if ld is not None and now > ld:
                run.abort('the leg did not finish within its time (hung collective or peer?)')
    th = threading.Thread(target=loop, name='pivp-bench-watchdog', daemon=True)
    th.start()
    return th


def _inject(leg, where, rank):
    """test hook (tests/test_bench_launcher.py): PIVP_BENCH_INJECT=<leg>:<rank>:<raise|hang|die>:<setup|run>"""
    spec = os.environ.get('PIVP_BENCH_INJECT')
    if not spec:
        return
    l, r, kind, w = spec.split(':')
    if l == leg and int(r) == rank and w == where:
        if kind == 'hang':
            time.sleep(1e6)
        if kind == 'die':                       # a rank that is gone without a word (segfault, OOM kill)
            os._exit(17)
        raise RuntimeError('injected failure in %s of leg %s on rank %d' % (where, leg, rank))


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
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # before torch / HIP: this pool's host driver only supports dmabuf IPC (ranks started by an outer torchrun too)
    run = Run(rank, world, 'train' if args.mode == 'train' else 'rollout')
    _start_watchdog(run)

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

    def agree(ok, err, leg, where):
        """True iff EVERY rank says ok (one MIN all-reduce of a flag); otherwise every rank learns who failed and why, and the leg is dropped on
        all of them together -- no rank walks into a collective its peers will never join."""
        if dist is None:
            if not ok:
                run.skipped[leg] = '%s failed: %s' % (where, err)
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 1:
            return True
        errs = [None] * world
        dist.all_gather_object(errs, None if ok else str(err))
        run.skipped[leg] = '%s failed on ' % where + '; '.join('rank %d: %s' % (r, e) for r, e in enumerate(errs) if e is not None)
        return False

    def run_leg(name, setup, body, need_s=0.0, budget_s=120.0):
        """One leg on all ranks together: [enough of the budget left?] -> setup() (no collectives: allocation, model construction, first call) ->
        agreement -> body(ctx) (collectives allowed) -> agreement.  A Python-side failure on any rank drops the leg on every rank and the run goes on;
        a leg that hangs is ended by the watchdog at its deadline with the finished legs printed.  -> body's result or None."""
        go = run.remaining() > need_s + 5.0
        if not agree(go, 'only %.0f s of the %.0f s budget left, the leg needs ~%.0f s' % (max(0.0, run.remaining()), BUDGET_S, need_s), name, 'budget check'):
            if rank == 0:
                sys.stderr.write('bench.py: leg %s skipped: %s\n' % (name, run.skipped.get(name)))
            run.save_partial()
            return None
        run.begin(name, float(os.environ.get('PIVP_BENCH_LEG_BUDGET') or max(budget_s, 4.0 * need_s)))
        res = ctx = err = None
        try:
            _inject(name, 'setup', rank)
            ctx = setup() if setup is not None else None
        except Exception as e:
            err = '%s: %s' % (type(e).__name__, e)
        ok = agree(err is None, err, name, 'setup')
        if ok:
            try:
                _inject(name, 'run', rank)
                res = body(ctx)
            except Exception as e:
                err = '%s: %s' % (type(e).__name__, e)
            ok = agree(err is None, err, name, 'run')
        if not ok:
            sys.stderr.write('bench.py: rank %d: leg %s dropped: %s\n' % (rank, name, run.skipped.get(name)))
            res = None
        run.end(name, ok)
        return res

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
    train_mode = args.mode == 'train'
    fp32_run = args.precision == 'fp32'
    split_modes = (('bf16x6', '3 bf16 pieces, 6 bf16 MFMAs per product'),
                   ('fp16x3', '2 fp16 pieces (weights packed times a per-tensor power of two), 3 fp16 MFMAs per product'))
    do_split = do_rollout and not dry and fp32_run and not args.no_bf16x6 and world == 1     # (additional objects of the 1-GPU line only)
    do_bf16_train = do_train and not train_mode and fp32_run and not args.no_bf16_train and (not dry or world > 1)
    do_x6_train = do_train and not train_mode and fp32_run and not dry and not args.no_bf16x6 and world == 1
    # the legs of this run, least risky first: the replicas' rollout (no data-path collective), the fp32 train step (one all-reduce per gradient
    # group), the bf16 train step (all-to-all + all-gather per group), then the passes that only rank 0 measures
    if do_rollout:
        run.planned.append('rollout')
        if do_split:
            run.planned += ['rollout_' + m for m, _ in split_modes]
    if do_train:
        run.planned.append('train')
        if do_bf16_train:
            run.planned.append('train_bf16')
        if do_x6_train:
            run.planned += ['train_' + m for m, _ in split_modes]
    if not args.no_roofline and not dry:
        run.planned.append('roofline')
    if world == 1 and not args.no_cpu_baseline and not dry:
        run.planned.append('cpu_baseline')

    def base_line(elapsed, loss_val):
        """the contract fields, as soon as the leg that carries `value` has finished on all ranks"""
        frames = world * B * (T - 1) * args.steps
        preset = ''
        if args.config is not None:     # name the BASELINE.json config this run is (one rank's share of it when it is quoted at 8 GPUs)
            share = '' if args.config not in (3, 5) or world == 8 else "; this run: %d of its 8 ranks' shares" % world
            preset = 'BASELINE.json config %d = configs[%d] (%s%s): ' % (args.config, args.config - 1, CONFIG_PRESETS[args.config][1], share)
        run.line.update({
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
                       'loss': loss_val,
                       # objects a reader of an N > 1 record should not look for: they are measured by the 1-GPU line only (rank 0's host cores / one GPU's legs)
                       **({'single_rank_only': ['cpu_baseline', 'rollout_bf16x6', 'rollout_fp16x3', 'train_bf16x6', 'train_fp16x3']} if world > 1 else {})},
            'roofline': None,
            'cpu_baseline': None,
        })
        if dry:
            run.line['dry'] = True
        elif pivp_amd._lib.build_flags():       # an instrumented / timing-only build of the library (PIVP_EXTRA_FLAGS): say so in the line
            run.line['library_build_flags'] = pivp_amd._lib.build_flags()

    st = {'model': None, 'elapsed': None, 'tmodel': None, 'opt': None, 't_ref': 1.0}     # what later legs need from earlier ones
    with pivp_amd.using_config('train', False):
        # ---- leg 1: the rollout (`value`) --------------------------------------------------------------------
        if do_rollout:
            def rollout_setup():
                if dry:
                    return lambda: time.sleep(0.001)
                st['model'] = model = pivp_amd.Model(nm, prefix='bench', device=dev, keep_activations=False, precision=args.precision, **kinds)

                def rollout_step():
                    model.reset_state()
                    return model([images, actions, states], 0)
                rollout_step()                              # parameters are lazily sized, the plan is built: fail in setup, not in the timed loop
                return rollout_step

            def rollout_body(step):
                w0 = time.perf_counter()
                elapsed, loss = timed(step, args.steps, args.warmup, sync, barrier)
                st['elapsed'] = elapsed = max_over_ranks(elapsed)
                st['t_ref'] = max(0.05, time.perf_counter() - w0)
                st['rollout_step'] = step
                base_line(elapsed, 0.0 if dry else float(loss))
                return elapsed
            run_leg('rollout', rollout_setup, rollout_body, budget_s=180.0)

        # ---- leg 1b (fp32 rollout runs): the same rollout with the gate convolutions on the bf16 / fp16 matrix cores at fp32 grade: every fp32 operand
        # as three bf16 pieces and six MFMAs per product (417 TFLOP/s ceiling), or as two fp16 pieces and three MFMAs (833).  ADDITIONAL objects, never
        # `value`; `max_l2_vs_f32_rollout` is the largest per-pixel L2 between their frames and the fp32 rollout's on this run's input; the gate against
        # the float64 oracle on TRAINED weights is `max_l2_vs_float64_trained` (the committed cdna_b32_t10_trained fixture, when the tree has it).
        if do_split and 'rollout' in run.finished:
            model = st['model']
            for mode, what in split_modes:
                def split_setup(mode=mode):
                    m6 = pivp_amd.Model(nm, prefix='bench', device=dev, keep_activations=False, precision=mode, **kinds)

                    def x6_step():
                        m6.reset_state()
                        return m6([images, actions, states], 0)
                    x6_step()
                    m6._flat_params.copy_(model._flat_params)            # the fp32 leg's weights (parameters are lazily sized: after one call)
                    return m6, x6_step

                def split_body(ctx, mode=mode, what=what):
                    m6, x6_step = ctx
                    t6, _ = timed(x6_step, args.steps, args.warmup, sync, barrier)
                    t6 = max_over_ranks(t6)
                    st['rollout_step']()
                    d = torch.stack(m6.gen_images).double() - torch.stack(model.gen_images).double()
                    l2_steps = [float(v) for v in d.pow(2).sum(dim=2).sqrt().flatten(1).max(dim=1).values]
                    obj = {'ms_per_step': round(t6 / args.steps * 1e3, 3), 'frames_per_s': round(world * B * (T - 1) * args.steps / t6, 1),
                           # per predicted frame: the first is one pass through the network; with RANDOM-INIT weights (this run's) any two fp32-grade
                           # evaluations then drift apart by a factor per fed-back step: NOT the parity figure -- that is the next field
                           'max_l2_vs_f32_rollout': max(l2_steps),
                           'max_l2_vs_f32_rollout_per_step': [float('%.3g' % v) for v in l2_steps],
                           'dtype': 'f32 operands of the ConvLSTM forward as %s' % what}
                    obj.update(trained_fixture_distance(mode, dev, np, torch, pivp_amd))
                    if not args.no_roofline:
                        r6 = roofline_pass(args, m6, x6_step, t6, np, torch, precision=mode)
                        obj.update({'achieved_tflops': r6['achieved'], 'peak_tflops': r6['peak'], 'frac': r6['frac'], 'per_layer_tflops': r6['per_layer_tflops'],
                                    'layers_in_this_arithmetic': r6['layers'],
                                    # (nominal peak at 2.4 GHz; measured: on real operands 16-bit MFMA loops hold 1.81-1.85 GHz at ~1270 W)
                                    'peak_note': 'nominal (2.4 GHz); these loops are power-limited at ~0.76 of it on real operands: profiles/r04/clock_power_operand_values.txt'})
                    run.line['rollout_' + mode] = obj
                    return obj
                run_leg('rollout_' + mode, split_setup, split_body, need_s=4.0 * st['t_ref'])

        # ---- leg 2: the data-parallel train step (the run's precision; in the default fp32 run also config 3's bf16 arithmetic) ----------
        dp = pivp_amd.GradAllReduce() if (do_train and world > 1) else None

        def train_setup(precision):
            """no collectives in here: a failure on one rank is agreed on before any peer enters one"""
            if dry:
                sys.path.insert(0, os.path.join(ROOT, 'tests'))
                from host_stub import HostStubModel          # the test double lives with the tests, not in the product package
                tm = HostStubModel(sizes=(1 << 16, 1 << 14, 1 << 15, 1 << 15, 1 << 16, 1 << 14), value=float(rank + 1), precision=precision)
                op = None
            else:
                tm = pivp_amd.Model(nm, prefix='bench', device=dev, keep_activations=True, precision=precision,
                                    main_priority={'auto': None, 'on': True, 'off': False}[args.main_priority], **kinds)
                op = pivp_amd.Adam(alpha=0.001).setup(tm, data_parallel=dp)             # TM:860-861
                tm([images, actions, states], 0)                                        # parameters are lazily sized: one forward first
            if dp is not None:
                dp.prepare(tm, probe=False)              # the collective stream and every staging buffer, outside the first sweep
            return tm, op

        def train_body(ctx, precision):
            """-> the leg's object; st['tmodel'] / st['opt'] / st['t_with'] for the passes behind it"""
            tm, op = ctx
            use_dp = [True]
            if dry:
                ngrad = sum(tm.sizes)

                def train_step():
                    tm.cleargrads()
                    if dp is not None and use_dp[0]:
                        dp.backward_and_allreduce(tm)
                    else:
                        tm.backward()
                    return 0.0
            else:
                if world > 1:                            # identical replicas: rank 0's initialisation (and the precision modes' weight packs dropped)
                    tm.broadcast_params(src=0)
                ngrad = int(tm._flat_params.numel())

                def train_step():
                    tm.reset_state()
                    op._dp = dp if use_dp[0] else None
                    return op.update(tm, [images, actions, states], 0)                  # schedsamp_k = -1: feed-self, deterministic
            probed = dp.prepare(tm, probe=True) if dp is not None else []   # one untimed collective of each kind this schedule uses
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
                        dp.prepare(tm, probe=True)
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
            st['tmodel'], st['opt'], st['t_with'], st['tloss'] = tm, op, t_with, float(tloss)
            return {
                'ms_per_step': round(t_with / args.steps * 1e3, 3),
                'frames_per_s': round(world * B * (T - 1) * args.steps / t_with, 1),
                'rccl_ranks': dist.get_world_size() if dist is not None else 1,
                'backend': (dist.get_backend() if dist is not None else None),
                'ms_per_step_without_allreduce': None if t_without is None else round(t_without / args.steps * 1e3, 3),
                'allreduce_ms_exposed': 0.0 if t_without is None else round(max(0.0, t_with - t_without) / args.steps * 1e3, 3),
                # what one rank hands to the collective per step (GradAllReduce.last_payload_bytes): the fp32 flat gradient buffer, or its bf16
                # image in the bf16 precision mode; with one rank nothing travels: the figure is what a data-parallel rank would send
                'gradient_bytes_per_step': payload_bytes if payload_bytes else (2 if precision == 'bf16' else 4) * ngrad,
                'gradient_payload': 'bf16 (summed over the ranks in fp32, one rounding; fp32 flat gradient buffer and Adam)' if precision == 'bf16' else 'fp32',
                'allreduce': 'none (1 rank)' if world == 1 else '6 gradient groups, SUM, issued from inside the backward sweep of t = 0 on a side stream',
                'allreduce_algo': algo if world > 1 else 'none (1 rank; a data-parallel run of this precision uses %s)' % ('rs_ag' if precision == 'bf16' else 'allreduce'),
                'allreduce_algo_ms_per_step': compare,
                'collectives_probed_before_timing': probed,
                'dtype': {'fp32': 'f32', 'bf16': 'bf16 ConvLSTM / enc5 / enc6 operands, f32 accumulate, gradients and optimizer',
                          'bf16x3': 'f32 as 2 bf16 pieces in the ConvLSTM forward and data gradients',
                          'bf16x6': 'f32 as 3 bf16 pieces in the ConvLSTM gate convolutions, their data and weight gradients (6 bf16 MFMAs per product: fp32-grade), f32 elsewhere',
                          'fp16x3': 'f32 as 2 fp16 pieces (3 fp16 MFMAs per product, f32 accumulate) in the ConvLSTM gate convolutions, their data and weight gradients (dG scaled by a power of two) and enc5 / enc6; f32 elsewhere'}[precision],
                'workload': 'optimizer.update (TM:950): forward + BPTT backward + gradient all-reduce + Adam, schedsamp_k=-1, batch %d/GPU' % B,
                'loss': float(tloss),
            }

        def train_leg(name, precision, keep=False):
            need = 14.0 * st['t_ref'] * (1.0 if world == 1 else 2.5)      # a train step is ~3.4 rollouts; at N > 1 up to three timed passes
            obj = run_leg(name, lambda: train_setup(precision), lambda ctx: train_body(ctx, precision), need_s=need)
            if obj is not None:
                if train_mode:
                    base_line(st['t_with'], st['tloss'])
                else:
                    run.line[name] = obj
            if not keep:
                st['tmodel'] = st['opt'] = None
            return obj

        if do_train:
            train_leg('train', args.precision, keep=train_mode)
            if do_bf16_train:
                # BASELINE.json config 3 names bf16 for the data-parallel configuration: at --gpus 8 this object IS config 3 (global batch 256)
                train_leg('train_bf16', 'bf16')
            if do_x6_train:
                # the fp32-grade train steps on the 16-bit matrix cores (additional objects: `train` stays the fp32 kernels')
                for mode, _ in split_modes:
                    train_leg('train_' + mode, mode)

        # ---- the dominant kernel against its roofline, HIP events on the launch stream, second pass over the same K steps ----------
        if 'roofline' in run.planned and run.value_done:
            pmodel = st['model'] if do_rollout else st['tmodel']
            opt = st['opt']

            def prof_step():
                pmodel.reset_state()
                if do_rollout:
                    return pmodel([images, actions, states], 0)
                return opt.update(pmodel, [images, actions, states], 0)

            def roofline_body(_):
                if rank != 0 and not do_rollout and world > 1:
                    for _i in range(args.steps):   # rank 0's instrumented pass calls the gradient all-reduce: every rank must take part
                        prof_step()
                    sync()
                if rank == 0:
                    run.line['roofline'] = roofline_pass(args, pmodel, prof_step, st['elapsed'] if do_rollout else st['t_with'], np, torch)
                return True
            run_leg('roofline', None, roofline_body, need_s=3.0 * st['t_ref'])

    if 'cpu_baseline' in run.planned and run.value_done:
        def cpu_body(_):
            run.line['cpu_baseline'] = cpu_baseline_leg(args, B, T, S, nm, np, torch, train=train_mode)
            return True
        run_leg('cpu_baseline', None, cpu_body, need_s=args.cpu_seconds * (2.0 if train_mode else 1.0) + 25.0)     # the GPU legs are measured: a failing CPU leg must not discard them

    run.closed = True
    run.leg_deadline = None
    run.emit()
    for leg, why in run.skipped.items():
        sys.stderr.write('bench.py: rank %d: leg %s is not in the line: %s\n' % (rank, leg, why))
    if dist is not None:
        try:
            barrier()                  # rank 0 was still measuring: every rank leaves the group together
            dist.destroy_process_group()
        except Exception as e:
            sys.stderr.write('bench.py: rank %d: leaving the process group failed (%s: %s)\n' % (rank, type(e).__name__, e))
    sys.stdout.flush()
    if not run.value_done:
        sys.exit(1)


def trained_fixture_distance(mode, dev, np, torch, pivp_amd):
    """{'max_l2_vs_float64_trained': ...}: this precision mode's rollout of the committed TRAINED config-2 model (tests/golden/cdna_b32_t10_trained.npz:
    float64-oracle pixels; weights = float32(init) + committed int8 deltas; inputs from the oracle module's seeded generators, as the fixture defines them)
    against the float64 oracle -- the figure the 1e-4 gate is about, on weights where fp32-grade evaluations do not drift apart.  A 1-second run in
    which the oracle side is the CHECKER (its generators and the committed frames), as in tests/test_gpu_trained.py.  {} without the fixture."""
    try:
        gold = os.path.join(ROOT, 'tests', 'golden')
        fx = os.path.join(gold, 'cdna_b32_t10_trained.npz')
        if not os.path.exists(fx):
            return {}
        if gold not in sys.path:
            sys.path.insert(0, gold)
        import trained_weights as TW
        from oracle import restatement as R
        g = np.load(fx)
        T, B, size, nm = int(g['seq_len']), int(g['batch']), int(g['size']), int(g['num_masks'])
        P = TW.load_trained(str(g['weights']), R.init_params(seed=1, dtype=np.float32, scale=1.0, num_masks=nm, model_type='CDNA', height=size, width=size))
        imgs, acts, stas = R.moving_batch(B, T, size, size, seed=int(g['data_seed']))
        m = pivp_amd.Model(nm, prefix='fixture', device=dev, keep_activations=False, precision=mode)
        m.load_state_dict_reference(P)
        with pivp_amd.using_config('train', False):
            m([imgs, acts, stas], 0)
        gen = torch.stack(m.gen_images).cpu().numpy()
        pix = np.ascontiguousarray(gen.transpose(0, 1, 3, 4, 2)).reshape(-1, 3)[::int(g['pixel_stride'])]
        l2 = np.sqrt(((pix.astype(np.float64) - g['gen_pixels']) ** 2).sum(axis=1))
        return {'max_l2_vs_float64_trained': float('%.3g' % l2.max()),
                'max_l2_vs_float64_trained_note': 'the parity figure: the trained config-2 model of tests/golden (cdna_b32_t10_trained, B=32 T=10), this mode\'s frames '
                                                  'against the float64 oracle\'s on %d sampled pixels of all 9 steps; gate 1e-4; the fp32 oracle itself: %.3g'
                                                  % (l2.size, float(g['fp32_oracle_pixels_l2'].max()))}
    except Exception as e:
        sys.stderr.write('bench.py: trained-fixture distance of %s not taken (%s: %s)\n' % (mode, type(e).__name__, e))
        return {}


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
    if precision in ('bf16x6', 'fp16x3'):      # layers whose map these kernels' tiles do not serve run the fp32 kernel in this mode: not part of its fraction
        widths = [args.size // 2, args.size // 2, args.size // 4, args.size // 4, args.size // 8, args.size // 4, args.size // 2]
        layers = [i for i in range(7) if widths[i] % 16 == 0 or (widths[i] % 8 == 0 and args.batch % 2 == 0)]     # (8-wide maps: tiles of two images, an even batch)
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
