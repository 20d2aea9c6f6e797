"""Checkpoint layout: the reference's Chainer `save_npz` files <-> the library's internal layouts.

Reference: chainer.serializers.save_npz/load_npz at train_model.py:865, 1035-1037 and
predict_model.py:110; keys are link paths, files carry no extension
(README.md:24, predict_model.py:80).  Shapes in the file: conv W (Cout,Cin,kh,kw); deconv W
(Cin,Cout,kh,kw); Linear W (out,in); LN gamma/beta flat in NCHW order c*H*W + y*W + x.

Internal layouts (csrc/pivp_kernels.h): conv/deconv W of the MFMA kernels [tap][Cin/32][Cout][32]
(K-inner packed); enc0, enc3 and the 1x1 heads [tap][Cin][Cout]; LN gamma/beta NHWC-flat;
cdna_kerns / stp_input W K-major over the NHWC-flat hidden5 index with 256 padded columns."""
import numpy as np

DECONV_KEYS = ('enc4/W', 'enc5/W', 'enc6/W', 'masks/W', 'model/enc7/W')
LN_CHANNELS = {'norm_enc0': 32, 'hidden1': 32, 'hidden2': 32, 'hidden3': 64, 'hidden4': 64,
               'hidden5': 128, 'hidden6': 64, 'hidden7': 32, 'norm_enc6': 64}
SKINNY_KEYS = ('model/cdna_kerns/W', 'model/stp_input/W')
UNPACKED_KEYS = ('enc0/W', 'enc3/W', 'masks/W', 'model/enc7/W')   # Cin not a multiple of 32 or VALU kernels


def _pack_k_inner(t):
    """[tap][Cin][Cout] -> [tap][Cin/32][Cout][32]"""
    taps, cin, cout = t.shape
    return np.ascontiguousarray(t.reshape(taps, cin // 32, 32, cout).transpose(0, 1, 3, 2))


def _unpack_k_inner(f, taps, cin, cout):
    return np.ascontiguousarray(f.reshape(taps, cin // 32, cout, 32).transpose(0, 1, 3, 2)).reshape(taps, cin, cout)


def to_internal(key, arr):
    """Reference-layout array -> flat float32 array in the library's layout."""
    a = np.asarray(arr, dtype=np.float32)
    if key in SKINNY_KEYS:
        nout, K = a.shape
        hw8 = K // 128
        t = a.reshape(nout, 128, hw8).transpose(2, 1, 0).reshape(K, nout)   # k = pix*128 + c
        out = np.zeros((K, 256), np.float32)
        out[:, :nout] = t
        return out.ravel()
    if key.endswith('/W') and a.ndim == 4:
        if key in DECONV_KEYS:
            t = np.ascontiguousarray(a.transpose(2, 3, 0, 1))                # (Cin,Cout,kh,kw) -> [kh][kw][Cin][Cout]
        else:
            t = np.ascontiguousarray(a.transpose(2, 3, 1, 0))                # (Cout,Cin,kh,kw) -> [kh][kw][Cin][Cout]
        t = t.reshape(-1, t.shape[2], t.shape[3])
        if key not in UNPACKED_KEYS:
            t = _pack_k_inner(t)
        return t.ravel()
    if key.endswith('/norm/gamma') or key.endswith('/norm/beta'):
        C = LN_CHANNELS[key.split('/')[0]]
        return np.ascontiguousarray(a.reshape(C, -1).T).ravel()              # c*HW+p -> p*C+c
    return a.ravel()


def from_internal(key, flat, ref_shape):
    """Inverse of to_internal: flat internal array -> array of the reference's shape."""
    f = np.asarray(flat, dtype=np.float32)
    if key in SKINNY_KEYS:
        nout, K = ref_shape
        hw8 = K // 128
        t = f.reshape(K, 256)[:, :nout]
        return np.ascontiguousarray(t.reshape(hw8, 128, nout).transpose(2, 1, 0)).reshape(nout, K)
    if key.endswith('/W') and len(ref_shape) == 4:
        if key in DECONV_KEYS:
            ci, co, kh, kw = ref_shape
        else:
            co, ci, kh, kw = ref_shape
        t = f.reshape(kh * kw, ci, co) if key in UNPACKED_KEYS else _unpack_k_inner(f, kh * kw, ci, co)
        t = t.reshape(kh, kw, ci, co)
        return np.ascontiguousarray(t.transpose(2, 3, 0, 1) if key in DECONV_KEYS else t.transpose(3, 2, 0, 1))
    if key.endswith('/norm/gamma') or key.endswith('/norm/beta'):
        C = LN_CHANNELS[key.split('/')[0]]
        return np.ascontiguousarray(f.reshape(-1, C).T).ravel()
    return f.reshape(ref_shape)


def save_npz(filename, model):
    """chainer.serializers.save_npz(filename, model) equivalent (compressed, no extension added)."""
    params = model.state_dict_reference()
    with open(filename, 'wb') as f:
        np.savez_compressed(f, **params)


def load_npz(filename, model):
    """chainer.serializers.load_npz(filename, model) equivalent; missing keys raise KeyError."""
    with np.load(filename) as npz:
        model.load_state_dict_reference({k: npz[k] for k in npz.files})


def save_optimizer_npz(filename, optimizer, epoch=0):
    """chainer.serializers.save_npz(filename, optimizer) equivalent (train_model.py:1037): `t`, `epoch` and per parameter
    `<path>/t`, `<path>/m`, `<path>/v` in the reference's layouts."""
    state = optimizer.state_dict_reference()
    out = {'t': np.asarray(state.pop('t')), 'epoch': np.asarray(epoch)}
    for k, v in state.items():
        out[k] = v
        if k.endswith('/m'):
            out[k[:-2] + '/t'] = np.asarray(optimizer.t)
    with open(filename, 'wb') as f:
        np.savez_compressed(f, **out)


def load_optimizer_npz(filename, optimizer):
    """Inverse of save_optimizer_npz; the model must already be sized (call it once first)."""
    import torch
    model = optimizer.target
    with np.load(filename) as z:
        optimizer.t = int(z['t'])
        m, v = optimizer._state(model)
        for key, shape in model._shapes().items():
            o, n = model._offsets[key]
            m[o:o + n].copy_(torch.from_numpy(to_internal(key, z[key + '/m'])))
            v[o:o + n].copy_(torch.from_numpy(to_internal(key, z[key + '/v'])))
