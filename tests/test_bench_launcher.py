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


def test_launcher_times_out_with_a_reason():
    # a child that cannot finish inside the bound is killed (its own process group only) and the launcher exits non-zero with one line
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry', '--steps', '2', '--warmup', '1'],
                       env=_env(PIVP_BENCH_TIMEOUT='0.5'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 124
    assert 'did not finish within' in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
