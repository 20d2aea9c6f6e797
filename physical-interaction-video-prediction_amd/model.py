"""Host-side mirror of the reference's `Model` (src/models/train_model.py:478-764, "TM").

Same constructor, call, `reset_state()` and attribute surface:
    model = Model(num_masks, is_cdna=True, is_dna=False, is_stp=False, use_state=True,
                  scheduled_sampling_k=-1, num_frame_before_prediction=2, prefix=None)
    loss = model([images, actions, states], iter_num)      # time-major, NCHW float32 frames in [0,1]
    model.gen_images, model.psnr_all, model.summaries, model.conv_res, model.loss
    model.reset_state()
All arithmetic runs in the gfx950 HIP library behind include/pivp_hip.h; PyTorch supplies device
memory and the stream only.  There is no CPU fallback: without the built library or without a
GPU the call raises.
"""
from __future__ import annotations

import contextlib
import ctypes
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from . import checkpoint as ckpt


class _Config(object):
    """Stand-in for chainer.config: only the `train` flag matters on this path (TM:649)."""
    train = True


config = _Config()


@contextlib.contextmanager
def using_config(name, value):
    """chainer.using_config equivalent (used as `with using_config('train', False)`, predict_model.py:126)."""
    old = getattr(config, name)
    setattr(config, name, value)
    try:
        yield
    finally:
        setattr(config, name, old)


LSTM_SIZES = OrderedDict(lstm1=(32, 32), lstm2=(32, 32), lstm3=(32, 64), lstm4=(64, 64),
                         lstm5=(64, 128), lstm6=(128, 64), lstm7=(96, 32))   # name -> (x channels, C); TM:509-515


def reference_param_shapes(num_masks, model_type, use_state, height, width):
    """Key -> shape in the reference's Chainer npz layout (TM:499-542)."""
    h2, w2, h4, w4, h8, w8 = height // 2, width // 2, height // 4, width // 4, height // 8, width // 8
    s = OrderedDict()
    s['enc0/W'] = (32, 3, 5, 5); s['enc0/b'] = (32,)
    s['enc1/W'] = (32, 32, 3, 3); s['enc1/b'] = (32,)
    s['enc2/W'] = (64, 64, 3, 3); s['enc2/b'] = (64,)
    s['enc3/W'] = (64, 64 + (10 if use_state else 0), 1, 1); s['enc3/b'] = (64,)
    s['enc4/W'] = (128, 128, 3, 3); s['enc4/b'] = (128,)
    s['enc5/W'] = (96, 96, 3, 3); s['enc5/b'] = (96,)
    s['enc6/W'] = (64, 64, 3, 3); s['enc6/b'] = (64,)
    for name, (cx, c) in LSTM_SIZES.items():
        s[name + '/conv/W'] = (4 * c, cx + c, 5, 5)
        s[name + '/conv/b'] = (4 * c,)
    ln = OrderedDict(norm_enc0=32 * h2 * w2, hidden1=32 * h2 * w2, hidden2=32 * h2 * w2, hidden3=64 * h4 * w4,
                     hidden4=64 * h4 * w4, hidden5=128 * h8 * w8, hidden6=64 * h4 * w4, hidden7=32 * h2 * w2,
                     norm_enc6=64 * height * width)
    for name, n in ln.items():
        s[name + '/norm/gamma'] = (n,)
        s[name + '/norm/beta'] = (n,)
    s['masks/W'] = (64, num_masks + 1, 1, 1); s['masks/b'] = (num_masks + 1,)
    s['current_state/W'] = (5, 10); s['current_state/b'] = (5,)
    if model_type == 'CDNA':
        s['model/enc7/W'] = (64, 3, 1, 1); s['model/enc7/b'] = (3,)
        s['model/cdna_kerns/W'] = (25 * num_masks, 128 * h8 * w8); s['model/cdna_kerns/b'] = (25 * num_masks,)
    elif model_type == 'STP':
        s['model/enc7/W'] = (64, 3, 1, 1); s['model/enc7/b'] = (3,)
        s['model/stp_input/W'] = (100, 128 * h8 * w8); s['model/stp_input/b'] = (100,)
        s['model/identity_params/W'] = (6, 100); s['model/identity_params/b'] = (6,)
    else:
        s['model/enc7/W'] = (64, 25, 1, 1); s['model/enc7/b'] = (25,)
    return s


def default_init(shapes, rng=None):
    """Chainer 2 defaults: LeCunNormal W (std sqrt(1/fan_in)), zero bias, LN gamma 1 / beta 0."""
    rng = np.random if rng is None else rng
    out = OrderedDict()
    for key, shape in shapes.items():
        if key.endswith('/W'):
            fan_in = shape[1] if len(shape) == 2 else shape[1] * shape[2] * shape[3]
            out[key] = (rng.standard_normal(shape) * math.sqrt(1.0 / fan_in)).astype(np.float32)
        elif key.endswith('/gamma'):
            out[key] = np.ones(shape, np.float32)
        else:
            out[key] = np.zeros(shape, np.float32)
    return out


def scheduled_sampling_masks(batch_size, seq_len, context_frames, k, iter_num, rng=None):
    """Per-step ground-truth selection of the reference's scheduled sampling (TM:654-656, TM:73-122).

    Returns uint8 [T-1][B]; one `shuffle(arange(B))` is drawn from NumPy's global RNG per step that
    the reference would call scheduled_sample for (t >= context_frames), in the same order."""
    rng = np.random if rng is None else rng
    ngt = int(np.int32(np.round(np.float32(batch_size) * (k / (k + np.exp(iter_num / k))))))
    mask = np.zeros((seq_len - 1, batch_size), np.uint8)
    for t in range(seq_len - 1):
        if t > context_frames - 1:
            idx = np.arange(int(batch_size))
            rng.shuffle(idx)
            mask[t, idx[:ngt]] = 1
    return mask


class _Plan(object):
    def __init__(self, lib, cfg):
        self.lib = lib
        self.cfg = cfg
        h = ctypes.c_void_p()
        _lib.check(lib.pivp_plan_create(ctypes.byref(cfg), ctypes.byref(h)), 'pivp_plan_create')
        self.h = h
        self.names = [lib.pivp_param_name(h, i).decode() for i in range(lib.pivp_param_count(h))]
        self.numel = [lib.pivp_param_numel(h, i) for i in range(len(self.names))]
        self.workspace = None

    def __del__(self):
        try:
            if self.h:
                self.lib.pivp_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Model(object):
    """MI355X drop-in for the reference's `Model` (TM:478-764)."""

    def __init__(self, num_masks, is_cdna=True, is_dna=False, is_stp=False, use_state=True,
                 scheduled_sampling_k=-1, num_frame_before_prediction=2, prefix=None,
                 device='cuda:0', ln_eps=1e-6, stp_border='clamp', keep_activations=False, precision='fp32', main_priority=None):
        if is_cdna:                      # TM:531-542, precedence cdna > stp > dna
            self.model_type = 'CDNA'
        elif is_stp:
            self.model_type = 'STP'
        elif is_dna:
            self.model_type = 'DNA'
        else:
            raise ValueError("No network specified")
        self.num_masks = num_masks
        self.use_state = use_state
        self.scheduled_sampling_k = scheduled_sampling_k
        # scheduled_sample draws from NumPy's GLOBAL RNG in the reference (TM:94); None keeps that.  Data-parallel training gives
        # every rank its own stream here (train.py: RandomState(seed + 1 + rank)), or all ranks would draw the same shuffle
        # for their shards.
        self.sampling_rng = None
        self.num_frame_before_prediction = num_frame_before_prediction
        self.prefix = prefix
        self.device = torch.device(device)
        self.ln_eps = float(ln_eps)
        if stp_border not in ('clamp', 'zeros'):
            raise ValueError("stp_border must be 'clamp' or 'zeros'")
        self.stp_border = stp_border
        self.keep_activations = bool(keep_activations)
        # 'fp32' is the parity path (per-pixel L2 < 1e-4 vs the reference).  'bf16' (BASELINE.json config 3) rounds the operands of
        # the seven ConvLSTM gate convolutions, of their data / weight gradients and of the enc5 / enc6 transposed convs to bf16 -- fp32 accumulation, gates, state, every
        # other op, the parameters, the gradients and Adam stay fp32 -- and reports, not gates, its error.
        # 'bf16x3' is the split mode: the forward gate convolutions (and the enc5 / enc6 transposed convs) take every fp32 operand as two bf16 pieces and form a product on
        # three bf16 MFMAs (16 bits of product mantissa); the rollout stays inside the 1e-4 gate; backward and everything else fp32.
        # 'bf16x6': three bf16 pieces per operand and the six significant products (fp32-grade results on the bf16 matrix cores) in the forward gate
        # convolutions of every layer whose map is a multiple of 16 wide; everything else, and the whole backward pass, fp32.
        # 'fp16x3': the forward gate convolutions with every operand as two fp16 pieces (weights pre-scaled by a per-tensor power of two), three MFMAs per product: 22-bit operands,
        # still fp32-grade (its truncation is a quarter of fp32's own rounding error); the data gradients of the sweep take the same form with dG staged times a power of two from its largest value (gradients lie below fp16's normal range).
        if precision not in ('fp32', 'bf16', 'bf16x3', 'bf16x6', 'fp16x3'):
            raise ValueError("precision must be 'fp32', 'bf16', 'bf16x3', 'bf16x6' or 'fp16x3'")
        self.precision = precision
        self.main_priority = main_priority     # None / False / True: include/pivp_hip.h, pivp_plan_set_main_priority
        self._ref_pending = None       # reference-layout arrays loaded before the first call
        self._params = None            # name -> view into _flat_params (internal layout)
        self._flat_params = None
        self._flat_grads = None
        self._grads = None
        self._offsets = None
        self._gt_mask = None
        self._hw = None
        self._plans = {}
        self._active = None
        self._results = None
        self._gen = None
        self.gen_states = None
        self.loss = 0.0
        self.psnr_all = 0.0
        self.gen_images = []

    # ---- parameters ---------------------------------------------------------------------
    def _shapes(self):
        H, W = self._hw
        return reference_param_shapes(self.num_masks, self.model_type, self.use_state, H, W)

    def _require_gpu(self):
        if not torch.cuda.is_available():
            raise RuntimeError('no MI355X visible: this path has no CPU fallback (torch.cuda.is_available() is False)')
        return _lib.load()

    def _ensure_params(self, H, W):
        if self._params is not None:
            if self._hw != (H, W):
                # TM:186-208 / quirk 3: lazily sized LN gamma/beta weld a model to one frame size
                raise ValueError('model was sized for %dx%d frames, got %dx%d' % (self._hw + (H, W)))
            return
        self._hw = (H, W)
        shapes = self._shapes()
        ref = self._ref_pending if self._ref_pending is not None else default_init(shapes)
        self._ref_pending = None
        self._upload(ref, shapes)

    def _upload(self, ref, shapes):
        """Parameters live in ONE flat device buffer (internal layouts, 256-B aligned slices): Adam is a single
        launch over it and the data-parallel gradient all-reduce a single collective on its twin `_flat_grads`."""
        host = OrderedDict()
        for key, shape in shapes.items():
            if key not in ref:
                raise KeyError('checkpoint is missing %r' % key)
            a = np.asarray(ref[key])
            if tuple(a.shape) != tuple(shape):
                raise ValueError('%s: expected shape %s, got %s' % (key, shape, a.shape))
            host[key] = ckpt.to_internal(key, a)
        # Flat layout: the parameters (and so their gradients) are laid out in the order in which the backward sweep finishes
        # them at t = 0 (pivp_param_group: 6 groups, heads first, enc0 last), so that each group is ONE contiguous slice the
        # data-parallel all-reduce can send while the sweep is still working on the next group.
        lib = _lib.load()                                       # host-only query: works without a GPU
        group = {key: int(lib.pivp_param_group_by_name(key.encode())) for key in host}
        order = sorted(host, key=lambda k: group[k])            # stable: checkpoint order inside a group
        offsets, off = OrderedDict(), 0
        bounds = []
        for key in order:
            if not bounds or bounds[-1][0] != group[key]:
                bounds.append([group[key], off, off])
            offsets[key] = (off, host[key].size)
            off += (host[key].size + 63) // 64 * 64
            bounds[-1][2] = off
        self._group_ranges = [(a, b) for _, a, b in bounds]     # [(start, end)] floats, one per gradient group in sweep order
        flat = np.zeros(off, np.float32)
        for key, v in host.items():
            o, n = offsets[key]
            flat[o:o + n] = v
        new_flat = torch.from_numpy(flat).to(self.device)
        if self._flat_params is not None and self._flat_params.numel() == new_flat.numel():
            self._flat_params.copy_(new_flat)          # keep addresses stable for plans and optimizers
        else:
            self._flat_params = new_flat
            self._flat_grads = None
        self._offsets = offsets
        self._params = OrderedDict((k, self._flat_params[o:o + n]) for k, (o, n) in offsets.items())
        for plan in self._plans.values():
            self._bind(plan)

    def _ensure_grads(self):
        if self._flat_grads is None:
            self._flat_grads = torch.zeros_like(self._flat_params)
            self._grads = OrderedDict((k, self._flat_grads[o:o + n]) for k, (o, n) in self._offsets.items())
            for plan in self._plans.values():
                self._bind(plan)
        return self._flat_grads

    def load_state_dict_reference(self, ref):
        """Load arrays in the reference's Chainer-npz layout (chainer.serializers.load_npz target)."""
        ref = {k: np.asarray(v) for k, v in ref.items()}
        if self._hw is None:
            self._ref_pending = ref
        else:
            self._require_gpu()
            self._upload(ref, self._shapes())

    def state_dict_reference(self):
        """Parameters in the reference's Chainer-npz layout (what save_npz would write)."""
        if self._params is None:
            if self._ref_pending is not None:
                return OrderedDict(self._ref_pending)
            raise RuntimeError('parameters are lazily sized from the first input (as in the reference); call the model first')
        out = OrderedDict()
        for key, shape in self._shapes().items():
            out[key] = ckpt.from_internal(key, self._params[key].cpu().numpy(), shape)
        return out

    def count_params(self):
        return sum(int(np.prod(s)) for s in self._shapes().values())

    def params_changed(self):
        """Tell the model that its parameters were written behind torch's back: a collective (`dist.broadcast` / `all_reduce` leave
        `Tensor._version` untouched), a ctypes kernel, a raw-pointer copy.  The precision modes keep bf16 / fp16 packs of the weights
        across calls (pivp_plan_set_pack_cache); they are keyed by torch's in-place version counter and this epoch, so every writer
        torch cannot see must call this (include/pivp_hip.h: pivp_plan_params_changed).  `Adam.step` and `broadcast_params` do."""
        self._params_epoch = getattr(self, '_params_epoch', 0) + 1

    def broadcast_params(self, src=0, group=None):
        """Data-parallel replicas start identical: rank `src`'s parameters to every rank, and the weight packs of the
        precision modes invalidated (the collective writes through the raw pointer).  Parameters are lazily sized: call the model once first."""
        import torch.distributed as dist
        if self._flat_params is None:
            raise RuntimeError('call the model first (parameters are lazily sized)')
        dist.broadcast(self._flat_params, src=src, group=group)
        self.params_changed()

    # ---- plans --------------------------------------------------------------------------
    def _bind(self, plan):
        lib = plan.lib
        for i, (name, n) in enumerate(zip(plan.names, plan.numel)):
            t = self._params[name]
            if t.numel() != n:
                raise ValueError('%s: internal size %d != library size %d' % (name, t.numel(), n))
            _lib.check(lib.pivp_plan_set_param(plan.h, i, t.data_ptr()), 'pivp_plan_set_param(%s)' % name)
            if self._grads is not None:
                _lib.check(lib.pivp_plan_set_grad(plan.h, i, self._grads[name].data_ptr()), 'pivp_plan_set_grad(%s)' % name)

    def _plan_for(self, B, T, H, W):
        key = (B, T, H, W, self.keep_activations)
        plan = self._plans.get(key)
        if plan is None:
            lib = self._require_gpu()
            cfg = _lib.PivpConfig(batch=B, seq_len=T, height=H, width=W, num_masks=self.num_masks,
                                  model_type={'CDNA': 0, 'STP': 1, 'DNA': 2}[self.model_type],
                                  use_state=1 if self.use_state else 0,
                                  context_frames=self.num_frame_before_prediction,
                                  keep_activations=1 if self.keep_activations else 0,
                                  ln_eps=self.ln_eps, stp_zero_border=1 if self.stp_border == 'zeros' else 0)
            plan = _Plan(lib, cfg)
            _lib.check(lib.pivp_plan_set_pack_cache(plan.h, 1), 'pivp_plan_set_pack_cache')
            if self.main_priority is not None:      # None: the library's rule (wave priority 3 for the sweep's kernels unless a gradient listener is registered)
                _lib.check(lib.pivp_plan_set_main_priority(plan.h, 1 if self.main_priority else 0), 'pivp_plan_set_main_priority')
            if self.precision != 'fp32':
                _lib.check(lib.pivp_plan_set_precision(plan.h, {'bf16': 1, 'bf16x3': 2, 'bf16x6': 3, 'fp16x3': 4}[self.precision]),
                           'pivp_plan_set_precision(%s)' % self.precision)
            nbytes = lib.pivp_plan_workspace_bytes(plan.h)
            plan.workspace = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=self.device)
            base = plan.workspace.data_ptr()
            aligned = (base + 255) // 256 * 256
            _lib.check(lib.pivp_plan_set_workspace(plan.h, aligned, nbytes), 'pivp_plan_set_workspace')
            self._bind(plan)
            self._plans[key] = plan
            self._reset_plan(plan)
        return plan

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _reset_plan(self, plan):
        with torch.cuda.device(self.device):
            _lib.check(plan.lib.pivp_reset_state(plan.h, self._stream()), 'pivp_reset_state')

    # ---- reference surface ----------------------------------------------------------------
    def reset_state(self):
        """TM:604-618: clears loss/PSNR/summaries/conv_res and the seven ConvLSTM (c, h) pairs."""
        self.loss = 0.0
        self.psnr_all = 0.0
        self._results = None
        if self._active is not None:
            self._reset_plan(self._active)

    def to_gpu(self, device=None):
        if device is not None:
            self.device = torch.device('cuda:%d' % device if isinstance(device, int) else device)
        return self

    def _as_device(self, a, shape_tail):
        if isinstance(a, (list, tuple)):
            a = torch.stack([x if torch.is_tensor(x) else torch.from_numpy(np.asarray(x, dtype=np.float32)) for x in a])
        elif not torch.is_tensor(a):
            a = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))
        a = a.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(a.shape[-len(shape_tail):]) != tuple(shape_tail) and shape_tail:
            raise ValueError('unexpected trailing shape %s' % (tuple(a.shape),))
        return a

    def __call__(self, x, iter_num=-1.0):
        """TM:620-764.  x = [images (T,B,3,H,W), actions (T,B,5), states (T,B,5)], time-major."""
        if len(x) > 1:
            images, actions, states = x
        else:
            images, actions, states = x[0], None, None
        if actions is None or states is None:
            # the reference dereferences states[0] (TM:646) and fails the same way
            raise TypeError("'NoneType' object is not subscriptable")
        with torch.cuda.device(self.device) if torch.cuda.is_available() else contextlib.nullcontext():
            self._require_gpu()
            images = self._as_device(images, ())
            if images.dim() != 5 or images.shape[2] != 3:
                raise ValueError('images must be time-major (T, B, 3, H, W)')
            T, B, _, H, W = images.shape
            actions = self._as_device(actions, (5,))
            states = self._as_device(states, (5,))
            self._ensure_params(H, W)
            plan = self._plan_for(B, T, H, W)
            self._active = plan
            ctx = self.num_frame_before_prediction
            # feed-self when not training or k == -1 (TM:649-657)
            gt_ptr = None
            self._gt_mask = None
            if config.train and self.scheduled_sampling_k != -1:
                mask = scheduled_sampling_masks(B, T, ctx, self.scheduled_sampling_k, iter_num, rng=self.sampling_rng)
                self._gt_mask = torch.from_numpy(mask).to(self.device)
                gt_ptr = self._gt_mask.data_ptr()
            gen = torch.empty((T - 1, B, 3, H, W), dtype=torch.float32, device=self.device)
            gen_states = torch.empty((T - 1, B, 5), dtype=torch.float32, device=self.device)
            nf = T - ctx
            results = torch.empty(2 + 3 * nf, dtype=torch.float32, device=self.device)
            # the precision modes' weight packs are kept across calls while the parameters are untouched: torch counts in-place writes to the flat
            # buffer and its views (_version), the optimizer's own kernel reports itself (_params_epoch)
            pkey = (self._flat_params.data_ptr(), self._flat_params._version, getattr(self, '_params_epoch', 0))
            if getattr(plan, 'packed_key', None) != pkey:
                _lib.check(plan.lib.pivp_plan_params_changed(plan.h), 'pivp_plan_params_changed')
                plan.packed_key = pkey
            _lib.check(plan.lib.pivp_rollout_forward(plan.h, images.data_ptr(), actions.data_ptr(), states.data_ptr(),
                                                     gt_ptr, gen.data_ptr(), gen_states.data_ptr(), results.data_ptr(),
                                                     self._stream()), 'pivp_rollout_forward')
            self._inputs = (images, actions, states)   # keep alive until the stream has consumed them
            self._gen = gen
            self._gen_states = gen_states
            self._results = results
            self._nf = nf
            self.gen_images = [gen[t] for t in range(T - 1)]
            self.gen_states = [gen_states[t] for t in range(T - 1)]
            self.loss = results[0]
            self.psnr_all = results[1]
        return self.loss

    # ---- training surface (Chainer: model.cleargrads(); loss.backward(); TM:950 via optimizer.update) -------
    def cleargrads(self):
        if self._flat_params is None:
            raise RuntimeError('call the model first (parameters are lazily sized)')
        self._ensure_grads().zero_()

    def grad_group_ranges(self):
        """[(start, end)] slices of `_flat_grads`, in the order the backward sweep completes them (include/pivp_hip.h)."""
        if self._offsets is None:
            raise RuntimeError('call the model first (parameters are lazily sized)')
        return list(self._group_ranges)

    def backward(self, on_group=None):
        """Back-propagate the loss of the LAST call through time (needs keep_activations=True).  Gradients
        accumulate into `model._flat_grads` (internal layouts); `grads_reference()` returns them in checkpoint layout.

        on_group(i): optional host callback, invoked from inside the sweep as soon as every kernel that contributes to
        gradient group i (slice grad_group_ranges()[i]) has been enqueued on the model's stream -- the hook the
        data-parallel all-reduce uses to overlap communication with the rest of the sweep."""
        plan = self._active
        if plan is None or self._results is None:
            raise RuntimeError('call the model first')
        if not self.keep_activations:
            raise RuntimeError('backward() needs Model(..., keep_activations=True)')
        self._ensure_grads()
        images, actions, states = self._inputs
        gt_ptr = self._gt_mask.data_ptr() if self._gt_mask is not None else None
        cb = None
        if on_group is not None:
            import ctypes
            errors = []

            def _trampoline(_user, g):
                try:
                    on_group(int(g))
                except BaseException as e:      # never unwind through the C frames
                    errors.append(e)
            cb = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int)(_trampoline)
            _lib.check(plan.lib.pivp_plan_set_grad_callback(plan.h, ctypes.cast(cb, ctypes.c_void_p), None), 'pivp_plan_set_grad_callback')
        try:
            # the model's device must be the CURRENT one for the launches (and for the plan's own side stream, created on first use)
            with torch.cuda.device(self.device):
                _lib.check(plan.lib.pivp_rollout_backward(plan.h, images.data_ptr(), actions.data_ptr(), states.data_ptr(), gt_ptr,
                                                          self._gen.data_ptr(), self._gen_states.data_ptr(), self._stream()),
                           'pivp_rollout_backward')
        finally:
            if cb is not None:
                plan.lib.pivp_plan_set_grad_callback(plan.h, None, None)
        if cb is not None and errors:
            raise errors[0]

    def set_group_join(self, join):
        """join=False: a gradient group's announcement no longer makes the model's stream wait for the group's weight gradients on the
        plan's side stream; the listener must then call wait_group(g, its_stream) (include/pivp_hip.h: pivp_plan_set_group_join)."""
        plan = self._active
        if plan is None:
            raise RuntimeError('call the model first')
        _lib.check(plan.lib.pivp_plan_set_group_join(plan.h, 1 if join else 0), 'pivp_plan_set_group_join')

    def wait_group(self, g, stream):
        """Make `stream` (a torch.cuda.Stream) wait for the side-stream work of gradient group g."""
        plan = self._active
        with torch.cuda.device(self.device):
            _lib.check(plan.lib.pivp_plan_group_wait(plan.h, int(g), stream.cuda_stream), 'pivp_plan_group_wait')

    def grads_reference(self):
        """Gradients in the reference's Chainer-npz layout (same permutations as the parameters)."""
        if self._grads is None:
            raise RuntimeError('no gradients: call cleargrads()/backward() first')
        out = OrderedDict()
        for key, shape in self._shapes().items():
            out[key] = ckpt.from_internal(key, self._grads[key].cpu().numpy(), shape)
        return out

    @property
    def summaries(self):
        """TM:744-759: strings '<prefix>_recon_cost<i>: v', '_psnr<i>', '_state_cost<i>', '_psnr_all', '_loss'."""
        if self._results is None:
            return []
        r = self._results.cpu().numpy()
        nf = self._nf
        p = str(self.prefix)
        out = []
        for i in range(nf):
            out.append(p + '_recon_cost' + str(i) + ': ' + str(r[2 + i]))
            out.append(p + '_psnr' + str(i) + ': ' + str(r[2 + nf + i]))
        for i in range(nf):
            out.append(p + '_state_cost' + str(i) + ': ' + str(r[2 + 2 * nf + i]))
        out.append(p + '_psnr_all: ' + str(r[1]))
        out.append(p + '_loss: ' + str(r[0]))
        return out

    def tap(self, name, step=None):
        """Activation of a timestep, planar NCHW like the reference's encs/hiddens (TM:703-708)."""
        plan = self._active
        if plan is None or self._results is None:
            raise RuntimeError('call the model first')
        cfg = plan.cfg
        T1 = cfg.seq_len - 1
        step = T1 - 1 if step is None else step
        H, W, B = cfg.height, cfg.width, cfg.batch
        lv = {'enc0': (32, 2), 'enc1': (32, 4), 'enc2': (64, 8), 'enc3': (64, 8), 'enc4': (128, 4), 'enc5': (96, 2),
              'enc6': (64, 1), 'hidden1': (32, 2), 'hidden2': (32, 2), 'hidden3': (64, 4), 'hidden4': (64, 4),
              'hidden5': (128, 8), 'hidden6': (64, 4), 'hidden7': (32, 2)}
        for i, (nm, (_, c)) in enumerate(LSTM_SIZES.items()):
            d = [2, 2, 4, 4, 8, 4, 2][i]
            lv[nm + '_h'] = (c, d); lv[nm + '_c'] = (c, d)
        if name in lv:
            C, d = lv[name]
            shape = (B, C, H // d, W // d)
        elif name == 'enc7':
            shape = (B, 25 if self.model_type == 'DNA' else 3, H, W)
        elif name == 'masks':
            shape = (B, self.num_masks + 1, H, W)
        elif name == 'cdna_kerns':
            shape = (B, self.num_masks, 5, 5)
        elif name == 'stp_theta':
            shape = (B, 2, 3)
        else:
            raise KeyError(name)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            n = plan.lib.pivp_get_tap(plan.h, name.encode(), step, out.data_ptr(), self._stream())
        if n < 0:
            _lib.check(int(n), 'pivp_get_tap(%s, %d)' % (name, step))
        assert n == out.numel()
        return out

    @property
    def conv_res(self):
        """TM:715, TM:734: [enc0..enc6, enc7] of the LAST timestep."""
        if self._results is None:
            return []
        return [self.tap(n) for n in ('enc0', 'enc1', 'enc2', 'enc3', 'enc4', 'enc5', 'enc6', 'enc7')]
