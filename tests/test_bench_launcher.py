"""bench.py as its own launcher: `python bench.py --gpus N` must start N ranks itself, before anything touches
a GPU, and must never report a run of a different size.  Exercised here with --dry (gloo on CPU, stub kernels): the launcher, the
rank bookkeeping, the barrier / max-over-ranks timing and the overlapped all-reduce are the real code; only the kernels are stubs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(kw)
    return env


def test_bench_launches_its_own_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry', '--steps', '2', '--warmup', '1'],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                               # ONE JSON line, printed by rank 0 and relayed by the launcher
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['dry'] is True and out['value'] == 0.0
    assert out['config']['global_batch'] == 64 and out['scaling'] == 'weak'
    tr = out['train']
    assert tr['rccl_ranks'] == 2 and tr['backend'] == 'gloo'
    assert tr['ms_per_step'] > 0 and tr['ms_per_step_without_allreduce'] > 0 and tr['allreduce_ms_exposed'] >= 0
    for key in ('metric', 'unit', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'vs_baseline', 'dtype', 'data', 'roofline', 'cpu_baseline'):
        assert key in out
    # legs in the order least risky first, all finished; the collectives of each schedule were probed before the timed steps
    assert out['incomplete'] == [] and 'incomplete_reason' not in out
    assert tr['collectives_probed_before_timing'] == ['all_reduce']
    tb = out['train_bf16']                               # config 3's payload and schedule over gloo: all-to-all + fp32 local sum + all-gather
    assert tb['rccl_ranks'] == 2 and tb['allreduce_algo'] == 'rs_ag' and tb['collectives_probed_before_timing'] == ['all_to_all_single', 'all_gather_into_tensor']


def test_bench_refuses_a_world_of_the_wrong_size():
    # a launcher that started 1 rank while the command line says 2 (the round-1 failure: it printed n_gpus = 1)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry', '--steps', '1', '--warmup', '0'],
                       env=_env(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert 'refusing' in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]


def test_bench_single_rank_dry_reports_both_legs():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry', '--steps', '2', '--warmup', '1'],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][0])
    assert out['n_gpus'] == 1 and out['train']['rccl_ranks'] == 1 and out['train']['allreduce_ms_exposed'] == 0.0


def test_eight_ranks_dry():
    """The driver's largest run (--gpus 8): launcher, rendezvous on a port torchrun picks, 8 ranks' bookkeeping, the overlapped
    all-reduce over 8 gloo ranks and the one JSON line.  Host logic only (stub kernels)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--dry', '--steps', '2', '--warmup', '1'],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['config']['global_batch'] == 256 and out['config']['frames_per_step'] == 8 * 32 * 9
    assert out['train']['rccl_ranks'] == 8 and out['train']['backend'] == 'gloo'
    # an N > 1 record names the objects only the 1-GPU line carries
    assert 'cpu_baseline' in out['config']['single_rank_only'] and out.get('cpu_baseline') is None


def test_launcher_times_out_with_a_reason():
    # a child that cannot finish inside the bound is killed (its own process group only) and the launcher exits non-zero with one line
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry', '--steps', '2', '--warmup', '1'],
                       env=_env(PIVP_BENCH_TIMEOUT='0.5'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 124
    assert 'did not finish within' in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]


def test_default_limits_sit_inside_the_drivers():
    """The driver gives `bench.py` 600 s.  With the default arguments the ranks' own budget and the launcher's limit are below it."""
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(['--gpus', '8'])
    assert bench.BUDGET_S <= 450.0 and bench._launch_timeout(args) <= 480.0
    assert bench._launch_timeout(args) > bench.BUDGET_S          # the ranks speak first, the launcher's kill is the backstop


def _line(p):
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    return json.loads(lines[0])


def _run(extra_env, launcher='own', timeout=300):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry', '--steps', '2', '--warmup', '1']
    if launcher == 'outer':          # the driver's way: bench.py is a rank under somebody else's torchrun
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', '29641'] + cmd[1:]
    return subprocess.run(cmd, env=_env(**extra_env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT)


def test_a_failing_train_leg_on_one_rank_keeps_the_measured_legs():
    """an exception in train_leg('bf16') on rank 1 must not discard rank 0's rollout `value` nor the fp32 train leg.  The failure
    is agreed on by all ranks (one MIN all-reduce of an ok flag) BEFORE anybody enters the leg's collectives; the leg is dropped everywhere, the
    line says which and why, exit code 0."""
    p = _run({'PIVP_BENCH_INJECT': 'train_bf16:1:raise:setup'})
    assert p.returncode == 0, p.stderr[-2000:]
    out = _line(p)
    assert out['n_gpus'] == 2 and out['incomplete'] == ['train_bf16'] and 'train_bf16' not in out
    assert 'rank 1' in out['incomplete_reason']['train_bf16'] and 'injected failure' in out['incomplete_reason']['train_bf16']
    assert out['train']['rccl_ranks'] == 2 and out['ms_per_step'] > 0


def test_a_hung_leg_ends_with_the_line_and_a_reason_inside_the_budget():
    """rank 1 never returns from the bf16 train leg: rank 0 blocks in its collective.  At the leg's deadline every rank's watchdog thread ends its
    process; rank 0 first prints the line of the finished legs with the reason.  Exit code 0: `value` was measured on all ranks."""
    import time
    t0 = time.time()
    p = _run({'PIVP_BENCH_INJECT': 'train_bf16:1:hang:run', 'PIVP_BENCH_LEG_BUDGET': '6'})
    assert time.time() - t0 < 120
    assert p.returncode == 0, p.stderr[-2000:]
    out = _line(p)
    assert out['incomplete'] == ['train_bf16'] and 'train' in out and out['n_gpus'] == 2
    assert 'did not finish within its time' in out['incomplete_reason']['run'] and "'train_bf16'" in out['incomplete_reason']['run']


def test_the_whole_budget_bounds_the_run():
    p = _run({'PIVP_BENCH_INJECT': 'train:0:hang:setup', 'PIVP_BENCH_BUDGET': '25'})
    assert p.returncode == 0, p.stderr[-2000:]
    out = _line(p)
    assert out['incomplete'] == ['train', 'train_bf16'] and 'budget of 25 s' in out['incomplete_reason']['run']


def test_a_rank_that_dies_does_not_take_the_line_with_it():
    """A rank gone without a word (segfault, OOM kill): torchrun ends the others with SIGTERM.  Rank 0 takes the signal in its watchdog thread
    (the main thread sits in a collective) and prints what it has -- under our launcher AND under the driver's own torchrun."""
    for launcher in ('own', 'outer'):
        p = _run({'PIVP_BENCH_INJECT': 'train_bf16:1:die:run'}, launcher=launcher)
        out = _line(p)
        assert out['incomplete'] == ['train_bf16'] and 'train' in out and out['value'] == 0.0 and out['n_gpus'] == 2, launcher
        assert 'SIGTERM' in out['incomplete_reason']['run']
        if launcher == 'own':
            assert p.returncode == 0, p.stderr[-2000:]      # `value` (the rollout leg) finished on all ranks


def test_no_value_no_line():
    """Nothing is fabricated: when the leg that carries `value` fails there is no line and the exit code is not 0."""
    p = _run({'PIVP_BENCH_INJECT': 'rollout:1:raise:setup'})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert 'injected failure' in p.stderr


def test_a_stale_parent_does_not_spend_the_budget_before_the_first_leg():
    """a rank whose parent is NOT a torch launcher (a long-lived shell or agent that exports WORLD_SIZE) must not take that parent's
    age for the job's; and an inherited, hours-old PIVP_BENCH_T0 is clamped.  Either way the deadline lies in the future when the run starts."""
    sys.path.insert(0, ROOT)
    import time
    import bench
    saved = {k: os.environ.get(k) for k in ('WORLD_SIZE', 'PIVP_BENCH_T0')}
    try:
        os.environ['WORLD_SIZE'] = '1'
        os.environ.pop('PIVP_BENCH_T0', None)
        assert not bench._parent_is_torchrun(os.getppid())            # pytest's parent is a shell / runner, not torchrun
        t0, src = bench._job_start()
        assert 'parent' not in src and time.time() - t0 <= bench.MAX_IMPORT_ALLOWANCE_S + 1.0
        os.environ['PIVP_BENCH_T0'] = repr(time.time() - 3000.0)
        t0, src = bench._job_start()
        assert 'clamped' in src and t0 + bench.BUDGET_S > time.time() + 100.0
        os.environ['PIVP_BENCH_T0'] = repr(time.time() - 5.0)         # a fresh launcher clock is believed as it is
        t0, src = bench._job_start()
        assert src == 'PIVP_BENCH_T0' and 4.0 < time.time() - t0 < 7.0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
