"""The CPU side of SURVEY.md 8(d): the restatement of the reference path (oracle/torch_restatement.py, fp32 PyTorch-CPU: the same op classes as
Chainer's im2col + BLAS) timed on the host cores of the box it runs on -- config 1 (B = 2) and the config 2 shape (B = 32), rollout and train step,
on all host threads and on one.  Bounded samples (about `--seconds` each).  Labelled "CPU restatement of the reference path", never "Chainer"."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R
from oracle.torch_restatement import TorchModel, chainer_adam_step


def rollout_rate(B, T, seconds):
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
    ci, ca, cs = R.synthetic_batch(B, T)
    tm = TorchModel(10, is_cdna=True, params=P, dtype=torch.float32); tm.train = False
    with torch.no_grad():
        tm([ci, ca, cs], 0); tm.reset_state()
        reps, c0 = 0, time.perf_counter()
        while time.perf_counter() - c0 < seconds or reps == 0:
            tm([ci, ca, cs], 0); tm.reset_state(); reps += 1
    return B * (T - 1) * reps / (time.perf_counter() - c0), reps


def train_rate(B, T, seconds):
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
    ci, ca, cs = R.synthetic_batch(B, T)
    tm = TorchModel(10, is_cdna=True, params=P, dtype=torch.float32, requires_grad=True)
    Pm = {k: v.detach().numpy() for k, v in tm.p.items()}
    Mm = {k: np.zeros_like(v) for k, v in Pm.items()}; Vm = {k: np.zeros_like(v) for k, v in Pm.items()}
    reps, c0 = 0, time.perf_counter()
    while time.perf_counter() - c0 < seconds or reps == 0:
        for v in tm.p.values():
            v.grad = None
        l = tm([ci, ca, cs], 0); l.backward(); tm.reset_state()
        with torch.no_grad():
            chainer_adam_step(Pm, {k: v.grad.numpy() for k, v in tm.p.items()}, Mm, Vm, reps + 1)
        reps += 1
    return B * (T - 1) * reps / (time.perf_counter() - c0), reps


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=12.0)
    a = ap.parse_args()
    allt = torch.get_num_threads()
    print('host: os.cpu_count() = %s, torch threads = %d' % (os.cpu_count(), allt), flush=True)
    for threads in (allt, 1):
        torch.set_num_threads(threads)
        for B in (2, 32):
            if threads == 1 and B == 32:
                continue                       # 288 frames on one thread: minutes per call
            r, n = rollout_rate(B, 10, a.seconds)
            print('threads %3d  B = %2d  rollout     %8.2f predicted frames/s  (%d calls)' % (threads, B, r, n), flush=True)
            r, n = train_rate(B, 10, a.seconds)
            print('threads %3d  B = %2d  train step  %8.2f predicted frames/s  (%d calls)' % (threads, B, r, n), flush=True)
