"""CPU oracle: NumPy restatement of the reference's ConvLSTM + CDNA/STP/DNA rollout.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product path
(physical-interaction-video-prediction_amd/) never does and fails loudly when the HIP
library is missing.

PARITY UNPINNED: the reference (kristofbc/physical-interaction-video-prediction) ships no
tests, golden vectors, checkpoints or sample data for this path, and its arithmetic lives in
the un-vendored third-party package chainer==2.0.1 (requirements.txt:10), which is not
installed here and cannot be (no network, no Python 2; train_model.py does not parse under
Python 3: TabError at src/models/train_model.py:500).  This file therefore restates, line by
line, what src/models/train_model.py (abbreviated TM below) does with Chainer 2.0.1's
published op semantics; it is pinned by known-answer tests derived from the reference's code
(tests/test_oracle_kat.py) and by agreement with an independently written PyTorch-CPU
restatement (oracle/torch_restatement.py).

All tensors are logical NCHW as in the reference.  `dtype` is float64 for the arbiter and
float32 to mimic the reference's own precision (TM:828-834 load everything as np.float32).

Chainer 2.0.1 semantics relied upon:
  * L.Convolution2D / F.convolution_2d : cross-correlation, zero pad, W (Cout,Cin,kh,kw)
  * L.Deconvolution2D / F.deconvolution_2d: gradient-of-conv, W (Cin,Cout,kh,kw), `outsize`
  * L.LayerNormalization: eps=1e-6 (link default), biased variance over axis 1, gamma/beta per element
  * F.depthwise_convolution_2d(x (N,C,H,W), W (D,C,kh,kw)) -> (N, C*D, H, W), out channel c*D+d
  * F.softmax axis=1, F.mean_squared_error = mean over all elements
  * F.spatial_transformer_grid / _sampler: align-corners bilinear; out-of-range samples follow `stp_border`: 'clamp' (the default here:
    2.0.x clips the sample coordinates to the image, as recollected: SURVEY App. C) or 'zeros' (Chainer >= 3 reads zero outside)
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

RELU_SHIFT = 1e-12   # TM:42
DNA_KERN_SIZE = 5    # TM:45
LN_EPS = 1e-6        # chainer.links.LayerNormalization default eps (2.0.x)

LSTM_SIZES = OrderedDict(  # TM:509-515
    lstm1=32, lstm2=32, lstm3=64, lstm4=64, lstm5=128, lstm6=64, lstm7=32)


# --------------------------------------------------------------------------------------
# Chainer op restatements
# --------------------------------------------------------------------------------------
def relu(x):
    return np.maximum(x, 0)


def sigmoid(x):
    # chainer F.sigmoid: tanh(x*0.5)*0.5+0.5
    return np.tanh(x * 0.5) * 0.5 + 0.5


def conv2d(x, W, b=None, stride=1, pad=0):
    """F.convolution_2d: cross-correlation, zero padding, out = floor((in+2p-k)/s)+1."""
    kh, kw = W.shape[2], W.shape[3]
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    win = sliding_window_view(xp, (kh, kw), axis=(2, 3))[:, :, ::stride, ::stride]
    # win: (B, Cin, OH, OW, kh, kw); contract Cin,kh,kw with W (Cout,Cin,kh,kw)
    y = np.tensordot(win, W, axes=([1, 4, 5], [1, 2, 3]))  # (B, OH, OW, Cout)
    y = np.ascontiguousarray(y.transpose(0, 3, 1, 2))
    if b is not None:
        y = y + b.reshape(1, -1, 1, 1)
    return y


def deconv2d(x, W, b=None, stride=1, pad=0, outsize=None):
    """F.deconvolution_2d: y[b,co,oy,ox] = sum x[b,ci,iy,ix] W[ci,co,ky,kx], oy = iy*s - p + ky.
    W is (Cin, Cout, kh, kw).  With `outsize` the result is cropped/extended to that size
    (TM:505-507 pass outsize = 2*in)."""
    B, Ci, H, Wd = x.shape
    _, Co, kh, kw = W.shape
    if outsize is None:
        OH = stride * (H - 1) + kh - 2 * pad
        OW = stride * (Wd - 1) + kw - 2 * pad
    else:
        OH, OW = outsize
    FH = max(stride * (H - 1) + kh, OH + pad)
    FW = max(stride * (Wd - 1) + kw, OW + pad)
    full = np.zeros((B, Co, FH, FW), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            contrib = np.tensordot(x, W[:, :, ky, kx], axes=([1], [0]))  # (B,H,W,Co)
            full[:, :, ky:ky + stride * H:stride, kx:kx + stride * Wd:stride] += contrib.transpose(0, 3, 1, 2)
    y = full[:, :, pad:pad + OH, pad:pad + OW]
    if b is not None:
        y = y + b.reshape(1, -1, 1, 1)
    return np.ascontiguousarray(y)


def linear(x, W, b=None):
    """L.Linear: y = x W^T + b, W (out, in)."""
    y = x @ W.T
    if b is not None:
        y = y + b
    return y


def layer_norm_flat(x2d, gamma, beta, eps=LN_EPS):
    """L.LayerNormalization on (B, N): biased variance over axis 1, gamma/beta of size N."""
    mu = x2d.mean(axis=1, keepdims=True)
    x_mu = x2d - mu
    var = np.mean(np.square(x_mu), axis=1, keepdims=True)
    inv_std = 1.0 / np.sqrt(var + np.asarray(eps, dtype=x2d.dtype))
    return x_mu * inv_std * gamma[None, :] + beta[None, :]


def layer_norm_conv2d(x, gamma, beta, eps=LN_EPS):
    """LayerNormalizationConv2D.__call__ (TM:203-208): flatten (B, C*H*W), LN, reshape back."""
    B = x.shape[0]
    return layer_norm_flat(x.reshape(B, -1), gamma, beta, eps).reshape(x.shape)


def softmax_axis1(x):
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=1, keepdims=True)


def depthwise_conv2d(x, W, pad):
    """F.depthwise_convolution_2d: x (N,C,H,W), W (D,C,kh,kw) -> (N, C*D, H, W), channel c*D+d."""
    N, C, H, Wd = x.shape
    D, _, kh, kw = W.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    win = sliding_window_view(xp, (kh, kw), axis=(2, 3))  # (N,C,H,W,kh,kw)
    y = np.einsum('nchwij,dcij->ncdhw', win, W)
    return y.reshape(N, C * D, H, Wd)


def mean_squared_error(a, b):
    d = a - b
    return np.mean(d * d)


def peak_signal_to_noise_ratio(true, pred):  # TM:124-134
    return 10.0 * np.log(1.0 / mean_squared_error(true, pred)) / math.log(10.0)


def spatial_transformer_grid(theta, out_hw):
    """F.spatial_transformer_grid: theta (B,2,3) -> grid (B,2,H,W), ch0 = x, ch1 = y, linspace(-1,1)."""
    H, W = out_hw
    ys, xs = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing='ij')
    coords = np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)], axis=0).astype(theta.dtype)  # (3, HW)
    grid = theta @ coords  # (B,2,HW)
    return grid.reshape(theta.shape[0], 2, H, W)


def spatial_transformer_sampler(x, grid, border='clamp'):
    """F.spatial_transformer_sampler: maps [-1,1] -> [0,size-1] (align-corners), bilinear.

    Out-of-range handling is the least pinned piece of Chainer 2.0.1 semantics (SURVEY.md
    App. C): `border='clamp'` clips the sampling coordinates to [-1,1] first (the 2.0.x
    behaviour as recollected: u.clip(-1,1), u0.clip(0,W-2)); `border='zeros'` samples a
    zero-padded image (the behaviour documented from Chainer 3 on).  Both are restated so
    either can be selected; the HIP kernel carries the same switch."""
    B, C, H, W = x.shape
    gu, gv = grid[:, 0], grid[:, 1]
    if border == 'clamp':
        gu = np.clip(gu, -1, 1)
        gv = np.clip(gv, -1, 1)
    u = (gu + 1) * (W - 1) / 2.0
    v = (gv + 1) * (H - 1) / 2.0
    u0 = np.floor(u)
    v0 = np.floor(v)
    if border == 'clamp':
        u0 = np.clip(u0, 0, W - 2)
        v0 = np.clip(v0, 0, H - 2)
    wu1 = u - u0
    wv1 = v - v0
    out = np.zeros((B, C) + u.shape[1:], dtype=x.dtype)
    bidx = np.arange(B)[:, None, None]
    for dv, wv in ((0, 1 - wv1), (1, wv1)):
        for du, wu in ((0, 1 - wu1), (1, wu1)):
            uu = (u0 + du).astype(np.int64)
            vv = (v0 + dv).astype(np.int64)
            ok = (uu >= 0) & (uu < W) & (vv >= 0) & (vv < H)
            uc = np.clip(uu, 0, W - 1)
            vc = np.clip(vv, 0, H - 1)
            val = x[bidx, :, vc, uc]  # (B,H,W,C)
            val = np.moveaxis(val, -1, 1)
            out += val * (wv * wu * ok)[:, None]
    return out


# --------------------------------------------------------------------------------------
# Parameters (Chainer save_npz key layout)
# --------------------------------------------------------------------------------------
def param_shapes(num_masks=10, model_type='CDNA', use_state=True, height=64, width=64):
    """Key -> shape of every parameter, in Chainer's `save_npz` path-key layout."""
    assert height % 8 == 0 and width % 8 == 0
    h2, w2 = height // 2, width // 2
    h4, w4 = height // 4, width // 4
    h8, w8 = height // 8, width // 8
    s = OrderedDict()
    s['enc0/W'] = (32, 3, 5, 5);   s['enc0/b'] = (32,)           # TM:500
    s['enc1/W'] = (32, 32, 3, 3);  s['enc1/b'] = (32,)           # TM:501
    s['enc2/W'] = (64, 64, 3, 3);  s['enc2/b'] = (64,)           # TM:502
    cin3 = 64 + (10 if use_state else 0)
    s['enc3/W'] = (64, cin3, 1, 1); s['enc3/b'] = (64,)          # TM:503
    s['enc4/W'] = (128, 128, 3, 3); s['enc4/b'] = (128,)         # TM:505 deconv (Cin,Cout,kh,kw)
    s['enc5/W'] = (96, 96, 3, 3);   s['enc5/b'] = (96,)          # TM:506
    s['enc6/W'] = (64, 64, 3, 3);   s['enc6/b'] = (64,)          # TM:507
    lstm_in = dict(lstm1=32 + 32, lstm2=32 + 32, lstm3=32 + 64, lstm4=64 + 64,
                   lstm5=64 + 128, lstm6=128 + 64, lstm7=96 + 32)
    for name, c in LSTM_SIZES.items():                            # TM:509-515, TM:224
        s[name + '/conv/W'] = (4 * c, lstm_in[name], 5, 5)
        s[name + '/conv/b'] = (4 * c,)
    ln = OrderedDict(norm_enc0=32 * h2 * w2, norm_enc6=64 * height * width,
                     hidden1=32 * h2 * w2, hidden2=32 * h2 * w2, hidden3=64 * h4 * w4,
                     hidden4=64 * h4 * w4, hidden5=128 * h8 * w8, hidden6=64 * h4 * w4,
                     hidden7=32 * h2 * w2)                        # TM:517-525
    for name, n in ln.items():
        s[name + '/norm/gamma'] = (n,)
        s[name + '/norm/beta'] = (n,)
    s['masks/W'] = (64, num_masks + 1, 1, 1); s['masks/b'] = (num_masks + 1,)   # TM:527
    s['current_state/W'] = (5, 10); s['current_state/b'] = (5,)                 # TM:529
    if model_type == 'CDNA':                                      # TM:288-289
        s['model/enc7/W'] = (64, 3, 1, 1); s['model/enc7/b'] = (3,)
        s['model/cdna_kerns/W'] = (DNA_KERN_SIZE * DNA_KERN_SIZE * num_masks, 128 * h8 * w8)
        s['model/cdna_kerns/b'] = (DNA_KERN_SIZE * DNA_KERN_SIZE * num_masks,)
    elif model_type == 'STP':                                     # TM:429-431
        s['model/enc7/W'] = (64, 3, 1, 1); s['model/enc7/b'] = (3,)
        s['model/stp_input/W'] = (100, 128 * h8 * w8); s['model/stp_input/b'] = (100,)
        s['model/identity_params/W'] = (6, 100); s['model/identity_params/b'] = (6,)
    elif model_type == 'DNA':                                     # TM:364
        s['model/enc7/W'] = (64, DNA_KERN_SIZE ** 2, 1, 1); s['model/enc7/b'] = (DNA_KERN_SIZE ** 2,)
    else:
        raise ValueError("No network specified")                  # TM:540
    return s


def _fan_in(key, shape):
    if key.endswith('/W'):
        if len(shape) == 2:
            return shape[1]
        # chainer get_fans: fan_in = shape[1] * receptive field (also for deconv's (Cin,Cout,kh,kw))
        return shape[1] * shape[2] * shape[3]
    return None


def init_params(seed=1, dtype=np.float32, scale=1.0, **kw):
    """Chainer-2 default initialisation: LeCunNormal (std = sqrt(1/fan_in)) for W, bias 0,
    LN gamma 1 / beta 0.  `scale` perturbs gamma/beta/bias away from the
    trivial values when >0 so parity tests exercise every parameter (scale=0 -> pure defaults)."""
    rs = np.random.RandomState(seed)
    p = OrderedDict()
    for key, shape in param_shapes(**kw).items():
        if key.endswith('/W'):
            p[key] = (rs.standard_normal(shape) * math.sqrt(1.0 / _fan_in(key, shape))).astype(dtype)
        elif key.endswith('/gamma'):
            p[key] = (1.0 + 0.1 * scale * rs.standard_normal(shape)).astype(dtype)
        elif key.endswith('/beta'):
            p[key] = (0.1 * scale * rs.standard_normal(shape)).astype(dtype)
        else:  # bias
            p[key] = (0.1 * scale * rs.standard_normal(shape)).astype(dtype)
    return p


def init_params_widened(seed=1, scale=1.0, **kw):
    """`init_params(dtype=np.float32)` -- the weights every GPU test loads -- widened to float64, so that the float64 oracle and the HIP path run on
    IDENTICAL parameters (the fixtures of tests/golden since round 5; before, the oracle ran on the unrounded float64 draw of the same stream)."""
    p32 = init_params(seed=seed, dtype=np.float32, scale=scale, **kw)
    return OrderedDict((k, v.astype(np.float64)) for k, v in p32.items())


def synthetic_batch(batch, seq_len=10, height=64, width=64, seed=0, dtype=np.float32):
    """Seed-0 synthetic inputs: images U[0,1) (T,B,3,H,W); actions, states 0.1*N(0,1) (T,B,5)."""
    rs = np.random.RandomState(seed)
    images = rs.random_sample((seq_len, batch, 3, height, width)).astype(dtype)
    actions = (0.1 * rs.standard_normal((seq_len, batch, 5))).astype(dtype)
    states = (0.1 * rs.standard_normal((seq_len, batch, 5))).astype(dtype)
    return images, actions, states


def smooth_batch(batch, seq_len=10, height=64, width=64, seed=0, dtype=np.float32, box=11):
    """`synthetic_batch` with every frame box-blurred (box x box, reflect padding): frames with the smoothness of video instead of
    white noise.  The STP warp multiplies a theta error by the image gradient, so white noise is its worst case."""
    images, actions, states = synthetic_batch(batch, seq_len, height, width, seed, dtype)
    r = box // 2
    pad = np.pad(images, ((0, 0), (0, 0), (0, 0), (r, r), (r, r)), mode='reflect')
    images = np.ascontiguousarray(sliding_window_view(pad, (box, box), axis=(3, 4)).mean(axis=(-1, -2))).astype(dtype)
    return images, actions, states


def moving_batch(batch, seq_len=10, height=64, width=64, seed=0, dtype=np.float32, box=9, margin=24):
    """Synthetic VIDEO: per sequence one smooth random scene (box-blurred noise, stretched to [0, 1]) seen through a window that
    drifts with a per-sequence velocity of up to 2 pixels a step (sub-pixel positions, bilinear crop) plus a little per-step jitter.
    `actions[t] = [vx, vy, 0, 0, 0] / 4` is the displacement between frame t and t+1, `states[t] = [px, py, 0, 0, 0] / 16` the window
    position: the roles the gripper command and pose play in the reference's data (make_dataset.py:111-124).  Unlike `smooth_batch`
    the frames of a sequence are predictable from the past, so a model TRAINED on it predicts motion (tests/golden/train_weights.py:
    the trained-weight fixtures)."""
    rs = np.random.RandomState(seed)
    Hs, Ws = height + 2 * margin, width + 2 * margin
    r = box // 2
    noise = rs.random_sample((batch, 3, Hs + 2 * r, Ws + 2 * r))
    c = np.cumsum(np.pad(noise, ((0, 0), (0, 0), (1, 0), (0, 0))), axis=2)          # box sums by running sums (sequential float64 adds)
    rows = c[:, :, box:] - c[:, :, :-box]
    c = np.cumsum(np.pad(rows, ((0, 0), (0, 0), (0, 0), (1, 0))), axis=3)
    scene = (c[:, :, :, box:] - c[:, :, :, :-box]) / float(box * box)
    lo = scene.min(axis=(1, 2, 3), keepdims=True); hi = scene.max(axis=(1, 2, 3), keepdims=True)
    scene = (scene - lo) / (hi - lo)
    vel = rs.uniform(-2.0, 2.0, size=(batch, 2))
    jit = 0.15 * rs.standard_normal((seq_len, batch, 2))
    pos = np.zeros((seq_len, batch, 2)); v = np.zeros((seq_len, batch, 2))
    p = np.full((batch, 2), float(margin))
    for t in range(seq_len):
        pos[t] = p
        v[t] = vel + jit[t]
        p = np.clip(p + v[t], 1.0, 2.0 * margin - 2.0)
    v[:-1] = pos[1:] - pos[:-1]                                    # the displacement actually applied (after clipping)
    # bilinear crop: the (H + 1) x (W + 1) scene window of every (t, b) at its integer position, then one vectorised blend
    x0 = np.floor(pos[..., 0]).astype(np.int64); y0 = np.floor(pos[..., 1]).astype(np.int64)
    fx = (pos[..., 0] - x0)[:, :, None, None, None]; fy = (pos[..., 1] - y0)[:, :, None, None, None]
    g = np.empty((seq_len, batch, 3, height + 1, width + 1))
    for t in range(seq_len):
        for b in range(batch):
            g[t, b] = scene[b, :, y0[t, b]:y0[t, b] + height + 1, x0[t, b]:x0[t, b] + width + 1]
    images = ((1 - fy) * (1 - fx) * g[..., :-1, :-1] + (1 - fy) * fx * g[..., :-1, 1:]
              + fy * (1 - fx) * g[..., 1:, :-1] + fy * fx * g[..., 1:, 1:])
    actions = np.zeros((seq_len, batch, 5)); states = np.zeros((seq_len, batch, 5))
    actions[:, :, :2] = v / 4.0
    states[:, :, :2] = (pos - margin) / 16.0
    return images.astype(dtype), actions.astype(dtype), states.astype(dtype)


def concat_examples(batch):
    """TM:51-71: list of (imgs (T,H,W,3), act (T,5), sta (T,5)) -> time-major (T,B,3,H,W), (T,B,5), (T,B,5)."""
    img = np.array([b[0] for b in batch])
    act = np.array([b[1] for b in batch])
    sta = np.array([b[2] for b in batch])
    act = [np.squeeze(a, axis=1) for a in np.split(act, act.shape[1], axis=1)]
    sta = [np.squeeze(s, axis=1) for s in np.split(sta, sta.shape[1], axis=1)]
    img = [np.rollaxis(np.squeeze(i, axis=1), 3, 1) for i in np.split(img, img.shape[1], axis=1)]
    return np.array(img), np.array(act), np.array(sta)


def scheduled_sample(ground_truth_x, generated_x, batch_size, num_ground_truth, rng=np.random):
    """TM:73-122.  One `shuffle` of arange(B) on the (global) NumPy RNG; the first
    num_ground_truth shuffled indices take the ground-truth frame, the rest the generated one.
    The reference's stitch loop (TM:110-121) reduces to a per-sample select."""
    idx = np.arange(int(batch_size))
    rng.shuffle(idx)
    gt_idx = idx[:num_ground_truth]
    out = np.array(generated_x, copy=True)
    out[gt_idx] = ground_truth_x[gt_idx]
    return out.astype(np.float32)  # TM:120 dtype=np.float32


def num_ground_truth_schedule(batch_size, k, iter_num):
    """TM:654-656."""
    return int(np.int32(np.round(np.float32(batch_size) * (k / (k + np.exp(iter_num / k))))))


# --------------------------------------------------------------------------------------
# Model
# --------------------------------------------------------------------------------------
class Model(object):
    """Restatement of TM:478-764 with the same constructor/call/reset_state surface.

    `train` plays the role of chainer.config.train (TM:649)."""

    def __init__(self, num_masks, is_cdna=True, is_dna=False, is_stp=False, use_state=True,
                 scheduled_sampling_k=-1, num_frame_before_prediction=2, prefix=None,
                 params=None, dtype=np.float64, ln_eps=LN_EPS, stp_border='clamp'):
        if is_cdna:                                              # TM:531-542 precedence cdna > stp > dna
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
        self.num_frame_before_prediction = num_frame_before_prediction
        self.prefix = prefix
        self.dtype = dtype
        self.ln_eps = ln_eps
        self.stp_border = stp_border
        self.train = True
        self.rng = np.random
        self.p = None
        if params is not None:
            self.load_params(params)
        self.taps = None
        self.reset_state()

    def load_params(self, params):
        self.p = OrderedDict((k, np.asarray(v, dtype=self.dtype)) for k, v in params.items())

    def reset_state(self):                                       # TM:604-618
        self.loss = 0.0
        self.psnr_all = 0.0
        self.summaries = []
        self.conv_res = []
        self.lstm_c = {n: None for n in LSTM_SIZES}
        self.lstm_h = {n: None for n in LSTM_SIZES}

    # --- sub-modules -------------------------------------------------------------------
    def _lstm(self, name, inputs, forget_bias=1.0):              # TM:234-276
        C = LSTM_SIZES[name]
        B, _, H, W = inputs.shape
        if self.lstm_c[name] is None:
            self.lstm_c[name] = np.zeros((B, C, H, W), dtype=self.dtype)
        if self.lstm_h[name] is None:
            self.lstm_h[name] = np.zeros((B, C, H, W), dtype=self.dtype)
        inputs_h = np.concatenate((inputs, self.lstm_h[name]), axis=1)           # TM:262
        j_i_f_o = conv2d(inputs_h, self.p[name + '/conv/W'], self.p[name + '/conv/b'], 1, 5 // 2)
        j, i, f, o = np.split(j_i_f_o, 4, axis=1)                               # TM:269
        c = self.lstm_c[name] * sigmoid(f + forget_bias) + sigmoid(i) * np.tanh(j)   # TM:271
        h = np.tanh(c) * sigmoid(o)                                              # TM:272
        self.lstm_c[name], self.lstm_h[name] = c, h
        return h

    def _ln(self, name, x):                                      # TM:203-208
        return layer_norm_conv2d(x, self.p[name + '/norm/gamma'], self.p[name + '/norm/beta'], self.ln_eps)

    def _cdna(self, enc6, hidden5, prev_image):                  # TM:293-351
        B = prev_image.shape[0]
        enc7 = deconv2d(enc6, self.p['model/enc7/W'], self.p['model/enc7/b'])     # TM:315 (1x1)
        enc7 = relu(enc7)                                                         # TM:316
        transformed_list = [sigmoid(enc7)]                                        # TM:317
        cdna_input = hidden5.reshape(B, -1)                                       # TM:321
        k = linear(cdna_input, self.p['model/cdna_kerns/W'], self.p['model/cdna_kerns/b'])  # TM:322
        k = k.reshape(B, self.num_masks, 1, DNA_KERN_SIZE, DNA_KERN_SIZE)         # TM:326
        k = relu(k - RELU_SHIFT) + RELU_SHIFT                                     # TM:327
        norm = k.sum(axis=(2, 3, 4), keepdims=True)                               # TM:328
        k = k / norm                                                              # TM:329
        k = k.reshape(B, self.num_masks, DNA_KERN_SIZE, DNA_KERN_SIZE)            # TM:335
        self.last_cdna_kerns = k
        k = k.transpose(1, 0, 2, 3)                                               # TM:336 (D=masks, C=B)
        pim = prev_image.transpose(1, 0, 2, 3)                                    # TM:338 (N=3, C=B)
        t = depthwise_conv2d(pim, k, DNA_KERN_SIZE // 2)                          # TM:341 -> (3, B*M, H, W)
        H, W = prev_image.shape[2:]
        t = t.reshape(3, B, self.num_masks, H, W)                                 # TM:344
        t = t.transpose(2, 1, 0, 3, 4)                                            # TM:345 -> (M,B,3,H,W)
        transformed_list += [t[m] for m in range(self.num_masks)]                 # TM:346-349
        return transformed_list, enc7

    def _stp(self, enc6, hidden5, prev_image):                   # TM:434-475
        B = prev_image.shape[0]
        enc7 = deconv2d(enc6, self.p['model/enc7/W'], self.p['model/enc7/b'])     # TM:454 (no relu)
        transformed = [sigmoid(enc7)]                                             # TM:455
        s0 = hidden5.reshape(B, -1)
        s1 = relu(linear(s0, self.p['model/stp_input/W'], self.p['model/stp_input/b']))   # TM:458-459
        ident = np.tile(np.array([[1.0, 0.0, 0.0, 0.0, 1.0, 0.0]], dtype=self.dtype), (B, 1))
        for _ in range(self.num_masks - 1):                                       # TM:465 (shared Linear)
            params = linear(s1, self.p['model/identity_params/W'], self.p['model/identity_params/b']) + ident
            params = params.reshape(B, 2, 3)
            grid = spatial_transformer_grid(params, prev_image.shape[2:])
            transformed.append(spatial_transformer_sampler(prev_image, grid, self.stp_border))
        return transformed, enc7

    def _dna(self, enc6, hidden5, prev_image):                   # TM:368-417
        if self.num_masks != 1:
            raise ValueError('Only one mask is supported for DNA model.')        # TM:390
        enc7 = relu(deconv2d(enc6, self.p['model/enc7/W'], self.p['model/enc7/b']))   # TM:387-388
        B, C, H, W = prev_image.shape
        pad = np.pad(prev_image, ((0, 0), (0, 0), (2, 2), (2, 2)))               # TM:395
        inputs = []
        for xk in range(DNA_KERN_SIZE):                                          # TM:397-404
            for yk in range(DNA_KERN_SIZE):
                tmp = pad[:, :, xk:H, yk:W]                                      # TM:400 (the reference's slice quirk)
                tmp = np.pad(tmp, ((0, 0), (0, 0), (0, xk), (0, yk)))            # TM:402
                inputs.append(tmp[:, None])                                      # TM:404 appends tmp.data: DETACHED from the graph
        kin = np.concatenate(inputs, axis=1)                                      # (B,25,C,H,W); constant w.r.t. prev in the backward
        kn = relu(enc7 - RELU_SHIFT) + RELU_SHIFT                                 # TM:408
        kn = kn / kn.sum(axis=1, keepdims=True)                                   # TM:409-410
        out = (kin * kn[:, :, None]).sum(axis=1)                                  # TM:411-414
        return [out], enc7

    # --- one timestep ------------------------------------------------------------------
    def _step(self, prev_image, state_action, taps=None):
        p = self.p
        B = prev_image.shape[0]
        encs = []
        # group 0 TM:595
        x = conv2d(prev_image, p['enc0/W'], p['enc0/b'], 2, 2)
        x = self._ln('norm_enc0', x)
        x = relu(x); encs.append(x)
        # group 1 TM:596
        x = self._lstm('lstm1', x); x = self._ln('hidden1', x); hidden1 = x
        x = self._lstm('lstm2', x); x = self._ln('hidden2', x); hidden2 = x
        x = conv2d(x, p['enc1/W'], p['enc1/b'], 2, 1)
        x = relu(x); encs.append(x)
        # group 2 TM:597
        x = self._lstm('lstm3', x); x = self._ln('hidden3', x); hidden3 = x
        x = self._lstm('lstm4', x); x = self._ln('hidden4', x); hidden4 = x
        x = conv2d(x, p['enc2/W'], p['enc2/b'], 2, 1)
        x = relu(x); encs.append(x)
        # group 3 TM:598 (smear TM:556-567)
        if self.use_state:
            smear = state_action.reshape(B, state_action.shape[1], 1, 1)
            smear = np.tile(smear, (1, 1, x.shape[2], x.shape[3]))
            x = np.concatenate((x, smear), axis=1)
        x = conv2d(x, p['enc3/W'], p['enc3/b'], 1, 0)
        x = relu(x); encs.append(x)
        # group 4 TM:599
        x = self._lstm('lstm5', x); x = self._ln('hidden5', x); hidden5 = x
        x = deconv2d(x, p['enc4/W'], p['enc4/b'], 2, 1, (x.shape[2] * 2, x.shape[3] * 2))
        x = relu(x); encs.append(x)
        # group 5 TM:600
        x = self._lstm('lstm6', x); x = self._ln('hidden6', x); hidden6 = x
        x = np.concatenate((x, encs[1]), axis=1)                                 # TM:574
        x = deconv2d(x, p['enc5/W'], p['enc5/b'], 2, 1, (x.shape[2] * 2, x.shape[3] * 2))
        x = relu(x); encs.append(x)
        # group 6 TM:601
        x = self._lstm('lstm7', x); x = self._ln('hidden7', x); hidden7 = x
        x = np.concatenate((x, encs[0]), axis=1)
        x = deconv2d(x, p['enc6/W'], p['enc6/b'], 2, 1, (x.shape[2] * 2, x.shape[3] * 2))
        x = self._ln('norm_enc6', x)
        x = relu(x); encs.append(x)
        enc6 = x

        head = {'CDNA': self._cdna, 'STP': self._stp, 'DNA': self._dna}[self.model_type]
        transformed, enc7 = head(enc6, hidden5, prev_image)                      # TM:711-714
        encs.append(enc7)

        masks = relu(deconv2d(enc6, p['masks/W'], p['masks/b']))                 # TM:718-719
        H, W = prev_image.shape[2:]
        masks = masks.reshape(-1, self.num_masks + 1)                            # TM:720 flat-11 quirk
        masks = softmax_axis1(masks)                                             # TM:721
        masks = masks.reshape(B, self.num_masks + 1, H, W)                       # TM:722
        mask_list = [masks[:, m:m + 1] for m in range(self.num_masks + 1)]       # TM:723

        output = prev_image * mask_list[0]                                       # TM:725
        for layer, mask in zip(transformed, mask_list[1:]):                      # TM:726 (zip drops the 11th layer)
            output = output + layer * mask
        new_state = linear(state_action, p['current_state/W'], p['current_state/b'])   # TM:730
        if taps is not None:
            taps.update(enc0=encs[0], enc1=encs[1], enc2=encs[2], enc3=encs[3], enc4=encs[4],
                        enc5=encs[5], enc6=encs[6], enc7=enc7, hidden1=hidden1, hidden2=hidden2,
                        hidden3=hidden3, hidden4=hidden4, hidden5=hidden5, hidden6=hidden6,
                        hidden7=hidden7, masks=masks, output=output, state=new_state,
                        transformed=transformed)
        return output, new_state, encs

    # --- rollout -----------------------------------------------------------------------
    def __call__(self, x, iter_num=-1.0, tap_steps=()):         # TM:620-764
        if len(x) > 1:
            images, actions, states = x
        else:
            images, actions, states = x[0], None, None
        images = [np.asarray(i, dtype=self.dtype) for i in images]
        actions = [np.asarray(a, dtype=self.dtype) for a in actions]
        states = [np.asarray(s, dtype=self.dtype) for s in states]
        batch_size = images[0].shape[0]
        gen_states, gen_images = [], []
        current_state = states[0]                                                # TM:646
        if not self.train or self.scheduled_sampling_k == -1:                    # TM:649
            feedself = True
        else:
            num_ground_truth = num_ground_truth_schedule(batch_size, self.scheduled_sampling_k, iter_num)
            feedself = False
        self.taps = {}
        encs = []
        for t, (image, action) in enumerate(zip(images[:-1], actions[:-1])):    # TM:659
            done_warm_start = len(gen_images) > self.num_frame_before_prediction - 1   # TM:663
            if feedself and done_warm_start:
                prev_image = gen_images[-1]
            elif done_warm_start:
                prev_image = scheduled_sample(image, gen_images[-1], batch_size, num_ground_truth,
                                              self.rng).astype(self.dtype)
            else:
                prev_image = image
            state_action = np.concatenate((action, current_state), axis=1)      # TM:676
            taps = {} if t in tap_steps else None
            output, current_state, encs = self._step(prev_image, state_action, taps)
            if taps is not None:
                self.taps[t] = taps
            gen_images.append(output)                                            # TM:728
            gen_states.append(current_state)                                     # TM:731
        self.conv_res = encs                                                     # TM:734

        ctx = self.num_frame_before_prediction
        loss, psnr_all = 0.0, 0.0
        summaries = []
        prefix = str(self.prefix)
        for i, xx, gx in zip(range(len(gen_images)), images[ctx:], gen_images[ctx - 1:]):   # TM:739
            recon = mean_squared_error(xx, gx)
            psnr_i = peak_signal_to_noise_ratio(xx, gx)
            psnr_all += psnr_i
            summaries.append(prefix + '_recon_cost' + str(i) + ': ' + str(recon))
            summaries.append(prefix + '_psnr' + str(i) + ': ' + str(psnr_i))
            loss += recon
        for i, st, gs in zip(range(len(gen_states)), states[ctx:], gen_states[ctx - 1:]):   # TM:749
            state_cost = mean_squared_error(st, gs) * 1e-4
            summaries.append(prefix + '_state_cost' + str(i) + ': ' + str(state_cost))
            loss += state_cost
        summaries.append(prefix + '_psnr_all: ' + str(psnr_all))
        self.psnr_all = psnr_all
        self.loss = loss = loss / np.float32(len(images) - ctx)                  # TM:758
        summaries.append(prefix + '_loss: ' + str(loss))
        self.summaries = summaries
        self.gen_images = gen_images
        self.gen_states = gen_states
        return self.loss


def resize_images(x, out_hw):
    """chainer.functions.resize_images (2.0.x) as used at predict_model.py:120: bilinear on an align-corners grid
    (u = linspace(0, W-1, out_W)), lower neighbour clipped to [0, size-2].  x (B,C,H,W)."""
    B, C, H, W = x.shape
    oh, ow = out_hw
    u = np.linspace(0, W - 1, num=ow); v = np.linspace(0, H - 1, num=oh)
    u0 = np.clip(np.floor(u).astype(np.int64), 0, W - 2); v0 = np.clip(np.floor(v).astype(np.int64), 0, H - 2)
    wu = (u - u0)[None, None, None, :]; wv = (v - v0)[None, None, :, None]
    a = x[:, :, v0][:, :, :, u0]; b = x[:, :, v0][:, :, :, u0 + 1]
    c = x[:, :, v0 + 1][:, :, :, u0]; d = x[:, :, v0 + 1][:, :, :, u0 + 1]
    return (1 - wu) * (1 - wv) * a + wu * (1 - wv) * b + (1 - wu) * wv * c + wu * wv * d


def per_pixel_l2(a, b):
    """Per-pixel L2 over the colour axis: a, b (..., 3, H, W) -> (..., H, W)."""
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return np.sqrt((d * d).sum(axis=-3))
