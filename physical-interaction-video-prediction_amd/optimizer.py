"""Chainer-2 style Adam for the MI355X model (reference: `optimizers.Adam(alpha=learning_rate)`, `optimizer.setup(model)`,
`optimizer.update(model, [imgs, acts, stas], itr)` at train_model.py:860-861 and :950).

update() = forward (loss) -> cleargrads -> backward -> [gradient all-reduce when data-parallel] -> Adam step, the
sequence of chainer.Optimizer.update(lossfun, *args).  The step itself is one HIP launch over the model's flat parameter
buffer with Chainer's epsilon placement (eps added to the UNcorrected sqrt(v))."""
import math

import numpy as np
import torch

from . import _lib


class Adam(object):
    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.alpha, self.beta1, self.beta2, self.eps = alpha, beta1, beta2, eps
        self.t = 0
        self.target = None
        self._m = None
        self._v = None
        self._dp = None

    def setup(self, model, data_parallel=None):
        """`data_parallel`: an optional parallel.GradAllReduce; its world size rescales the summed gradients."""
        self.target = model
        self._dp = data_parallel
        return self

    @property
    def lr(self):
        """AdamRule.lr: alpha * sqrt(1 - beta2^t) / (1 - beta1^t)."""
        fix1 = 1.0 - math.pow(self.beta1, self.t)
        fix2 = 1.0 - math.pow(self.beta2, self.t)
        return self.alpha * math.sqrt(fix2) / fix1

    def _state(self, model):
        if self._m is None or self._m.numel() != model._flat_params.numel():
            self._m = torch.zeros_like(model._flat_params)
            self._v = torch.zeros_like(model._flat_params)
        return self._m, self._v

    def step(self, model=None):
        """Apply one Adam step from the gradients currently held by the model."""
        model = model or self.target
        g = model._ensure_grads()
        m, v = self._state(model)
        self.t += 1
        gscale = 1.0
        if self._dp is not None:
            gscale = 1.0 / self._dp.world_size
        lib = _lib.load()
        with torch.cuda.device(model._flat_params.device):      # launches need the model's device to be the current one
            _lib.check(lib.pivp_adam_step(model._flat_params.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                          model._flat_params.numel(), self.lr, self.beta1, self.beta2, self.eps, gscale,
                                          model._stream()), 'pivp_adam_step')
        if hasattr(model, 'params_changed'):
            model.params_changed()      # (a raw-pointer write: torch's version counter does not see it)

    def update(self, lossfun, *args):
        """chainer.Optimizer.update(lossfun, *args): loss = lossfun(*args); cleargrads; backward; update."""
        model = self.target
        loss = lossfun(*args)
        model.cleargrads()
        if self._dp is not None:
            self._dp.backward_and_allreduce(model)   # all-reduce of each gradient group overlapped with the rest of the sweep
        else:
            model.backward()
        self.step(model)
        return loss

    # state-<epoch> files of the reference (train_model.py:1037) hold t and per-parameter m, v
    def state_dict_reference(self):
        from . import checkpoint as ckpt
        model = self.target
        out = {'t': np.asarray(self.t)}
        self._state(model)
        if self._m is not None:
            for key, shape in model._shapes().items():
                o, n = model._offsets[key]
                out[key + '/m'] = ckpt.from_internal(key, self._m[o:o + n].cpu().numpy(), shape)
                out[key + '/v'] = ckpt.from_internal(key, self._v[o:o + n].cpu().numpy(), shape)
        return out
