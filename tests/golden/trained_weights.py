"""Trained weights of the golden fixtures: a committed int8 file of (trained - initial) per parameter tensor.

`tests/golden/train_weights.py` (run on the MI355X) trains the model from `R.init_params(seed=1, scale=1.0)` on `R.moving_batch`
sequences and stores, per tensor, q = round((W_trained - W_init) / scale) as int8 with scale = std(W_trained - W_init) / 4.  The weights
the fixtures are about are DEFINED as  W = float32(W_init) + float32(q) * float32(scale)  (elementwise IEEE float32, so the same
bytes on every machine): they are inputs, like the frames, and both the float64 / float32 oracles (make_golden.py) and the HIP path
(tests/test_gpu_trained.py) load them through `load_trained`.  The file holds numbers only: no source, no reference data."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def quantize(w_init, w_trained):
    """-> (q int8, scale float32) per the rule above; an untouched tensor gives scale 0."""
    d = np.asarray(w_trained, np.float32) - np.asarray(w_init, np.float32)
    sd = float(d.std())
    if sd == 0.0:
        return np.zeros(d.shape, np.int8), np.float32(0.0)
    scale = np.float32(sd / 4.0)
    q = np.clip(np.rint(d / scale), -127, 127).astype(np.int8)
    return q, scale


def reconstruct(w_init, q, scale):
    return (np.asarray(w_init, np.float32) + q.astype(np.float32) * np.float32(scale)).astype(np.float32)


def load_trained(name, init_params):
    """name: file stem under tests/golden (e.g. 'trained_stp64_q8'); init_params: dict key -> float32 array of the initial weights
    (R.init_params(seed=1, dtype=np.float32, scale=1.0, ...)).  Returns an OrderedDict-like dict of float32 arrays."""
    f = np.load(os.path.join(HERE, name + '.npz'))
    out = type(init_params)()
    for key, w0 in init_params.items():
        k = key.replace('/', '.')
        out[key] = reconstruct(w0, f['q:' + k], f['scale:' + k])
    return out
