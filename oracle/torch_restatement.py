"""CPU oracle, second leg: PyTorch-CPU restatement of the reference rollout.

TEST INFRASTRUCTURE ONLY (see oracle/restatement.py header; PARITY UNPINNED applies here too).

Written independently of oracle/restatement.py with library ops (F.conv2d, F.conv_transpose2d,
F.layer_norm, grouped conv, F.grid_sample) so that the two restatements cross-check each other
(tests/test_oracle_crosscheck.py).  It has three further uses:
  * autograd gives the gradient oracle for the backward kernels and the Adam step,
  * it is the multithreaded "CPU restatement of the reference path" timed by bench.py's
    cpu_baseline leg (same op classes as Chainer's im2col + BLAS GEMM path),
  * fp32 mode mimics the reference's own precision.

Reference citations: TM = src/models/train_model.py of the reference.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

RELU_SHIFT = 1e-12
LSTM_NAMES = ('lstm1', 'lstm2', 'lstm3', 'lstm4', 'lstm5', 'lstm6', 'lstm7')
LSTM_C = dict(lstm1=32, lstm2=32, lstm3=64, lstm4=64, lstm5=128, lstm6=64, lstm7=32)


class TorchModel(object):
    """TM:478-764 with torch ops.  `params`: dict of numpy arrays in Chainer npz key layout."""

    def __init__(self, num_masks, is_cdna=True, is_dna=False, is_stp=False, use_state=True,
                 scheduled_sampling_k=-1, num_frame_before_prediction=2, prefix=None,
                 params=None, dtype=torch.float64, ln_eps=1e-6, stp_border='clamp',
                 requires_grad=False):
        self.model_type = 'CDNA' if is_cdna else 'STP' if is_stp else 'DNA' if is_dna else None
        if self.model_type is None:
            raise ValueError("No network specified")
        self.num_masks = num_masks
        self.use_state = use_state
        self.scheduled_sampling_k = scheduled_sampling_k
        self.ctx = num_frame_before_prediction
        self.prefix = prefix
        self.dtype = dtype
        self.ln_eps = ln_eps
        self.stp_border = stp_border
        self.train = True
        self.rng = np.random
        self.p = OrderedDict()
        if params is not None:
            for k, v in params.items():
                t = torch.tensor(np.asarray(v), dtype=dtype)
                t.requires_grad_(requires_grad)
                self.p[k] = t
        self.reset_state()

    def reset_state(self):                                        # TM:604-618
        self.loss = 0.0
        self.psnr_all = 0.0
        self.c = {n: None for n in LSTM_NAMES}
        self.h = {n: None for n in LSTM_NAMES}

    def _lstm(self, n, x):                                         # TM:234-276
        C = LSTM_C[n]
        if self.h[n] is None:
            self.h[n] = x.new_zeros(x.shape[0], C, x.shape[2], x.shape[3])
            self.c[n] = x.new_zeros(x.shape[0], C, x.shape[2], x.shape[3])
        g = F.conv2d(torch.cat((x, self.h[n]), 1), self.p[n + '/conv/W'], self.p[n + '/conv/b'], padding=2)
        j, i, f, o = torch.split(g, C, dim=1)
        c = self.c[n] * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        h = torch.tanh(c) * torch.sigmoid(o)
        self.c[n], self.h[n] = c, h
        return h

    def _ln(self, n, x):                                           # TM:203-208
        B = x.shape[0]
        g, b = self.p[n + '/norm/gamma'], self.p[n + '/norm/beta']
        return F.layer_norm(x.reshape(B, -1), (g.numel(),), g, b, self.ln_eps).reshape(x.shape)

    def _deconv(self, n, x):                                       # TM:505-507 (outsize = 2*in)
        return F.conv_transpose2d(x, self.p[n + '/W'], self.p[n + '/b'], stride=2, padding=1, output_padding=1)

    def _step(self, prev, sa, prev_head=None):
        """One timestep (TM:676-731).  `prev_head` (tests only) lets a KAT hand the motion head / compositing a separate leaf
        from the one the trunk reads, to isolate d out / d prev through the head; the model always passes one tensor."""
        p = self.p
        B, _, H, W = prev.shape
        enc0 = F.relu(self._ln('norm_enc0', F.conv2d(prev, p['enc0/W'], p['enc0/b'], stride=2, padding=2)))
        x = self._ln('hidden1', self._lstm('lstm1', enc0))
        x = self._ln('hidden2', self._lstm('lstm2', x))
        enc1 = F.relu(F.conv2d(x, p['enc1/W'], p['enc1/b'], stride=2, padding=1))
        x = self._ln('hidden3', self._lstm('lstm3', enc1))
        x = self._ln('hidden4', self._lstm('lstm4', x))
        enc2 = F.relu(F.conv2d(x, p['enc2/W'], p['enc2/b'], stride=2, padding=1))
        x = enc2
        if self.use_state:                                         # TM:556-567
            x = torch.cat((x, sa[:, :, None, None].expand(-1, -1, x.shape[2], x.shape[3])), 1)
        enc3 = F.relu(F.conv2d(x, p['enc3/W'], p['enc3/b']))
        hidden5 = self._ln('hidden5', self._lstm('lstm5', enc3))
        enc4 = F.relu(self._deconv('enc4', hidden5))
        x = self._ln('hidden6', self._lstm('lstm6', enc4))
        enc5 = F.relu(self._deconv('enc5', torch.cat((x, enc1), 1)))
        x = self._ln('hidden7', self._lstm('lstm7', enc5))
        enc6 = F.relu(self._ln('norm_enc6', self._deconv('enc6', torch.cat((x, enc0), 1))))

        if prev_head is not None:
            prev = prev_head

        # 1x1 deconvs == 1x1 convs with W^T (TM:288, TM:527)
        def conv1x1_t(w, b, xin):
            return F.conv2d(xin, w.permute(1, 0, 2, 3), b)

        if self.model_type == 'CDNA':                              # TM:293-351
            enc7 = F.relu(conv1x1_t(p['model/enc7/W'], p['model/enc7/b'], enc6))
            layers = [torch.sigmoid(enc7)]
            k = F.linear(hidden5.reshape(B, -1), p['model/cdna_kerns/W'], p['model/cdna_kerns/b'])
            k = k.reshape(B, self.num_masks, 25)
            k = F.relu(k - RELU_SHIFT) + RELU_SHIFT
            k = k / k.sum(dim=2, keepdim=True)
            self.last_cdna_kerns = k.reshape(B, self.num_masks, 5, 5)
            # one grouped conv: every (b, colour) plane is a group with num_masks output maps
            M = self.num_masks
            w = k.reshape(B, 1, M, 5, 5).expand(B, 3, M, 5, 5).reshape(B * 3 * M, 1, 5, 5)
            t = F.conv2d(prev.reshape(1, B * 3, H, W), w, padding=2, groups=B * 3)
            t = t.reshape(B, 3, M, H, W)
            layers += [t[:, :, m] for m in range(M)]
        elif self.model_type == 'STP':                             # TM:434-475
            enc7 = conv1x1_t(p['model/enc7/W'], p['model/enc7/b'], enc6)
            layers = [torch.sigmoid(enc7)]
            s1 = F.relu(F.linear(hidden5.reshape(B, -1), p['model/stp_input/W'], p['model/stp_input/b']))
            ident = torch.tensor([1.0, 0, 0, 0, 1.0, 0], dtype=self.dtype)
            theta = (F.linear(s1, p['model/identity_params/W'], p['model/identity_params/b']) + ident).reshape(B, 2, 3)
            grid = F.affine_grid(theta, (B, 3, H, W), align_corners=True)
            mode = 'border' if self.stp_border == 'clamp' else 'zeros'
            warped = F.grid_sample(prev, grid, mode='bilinear', padding_mode=mode, align_corners=True)
            layers += [warped for _ in range(self.num_masks - 1)]
        else:                                                      # DNA TM:368-417
            if self.num_masks != 1:
                raise ValueError('Only one mask is supported for DNA model.')
            enc7 = F.relu(conv1x1_t(p['model/enc7/W'], p['model/enc7/b'], enc6))
            padded = F.pad(prev, (2, 2, 2, 2))
            shifted = []
            for xk in range(5):
                for yk in range(5):
                    tmp = padded[:, :, xk:H, yk:W]                 # TM:400 quirk: slice ends at H, W
                    tmp = F.pad(tmp, (0, yk, 0, xk))               # TM:402
                    shifted.append(tmp.detach())                   # TM:404 `kernel_inputs.append(tmp.data)`: graph cut, no gradient into prev
            kin = torch.stack(shifted, 1)                          # (B,25,3,H,W)
            kn = F.relu(enc7 - RELU_SHIFT) + RELU_SHIFT
            kn = kn / kn.sum(dim=1, keepdim=True)
            layers = [(kin * kn[:, :, None]).sum(1)]

        m = F.relu(conv1x1_t(p['masks/W'], p['masks/b'], enc6))    # TM:718-719
        m = torch.softmax(m.reshape(-1, self.num_masks + 1), dim=1).reshape(B, self.num_masks + 1, H, W)  # TM:720-722
        out = prev * m[:, 0:1]
        for q, layer in enumerate(layers[:self.num_masks]):        # TM:726 zip truncation
            out = out + layer * m[:, q + 1:q + 2]
        new_state = F.linear(sa, p['current_state/W'], p['current_state/b'])    # TM:730
        self.last = dict(enc0=enc0, enc1=enc1, enc2=enc2, enc3=enc3, enc4=enc4, enc5=enc5, enc6=enc6,
                         enc7=enc7, hidden5=hidden5, masks=m)
        return out, new_state

    def __call__(self, x, iter_num=-1.0):                          # TM:620-764
        images, actions, states = x
        as_t = lambda a: a if torch.is_tensor(a) else torch.tensor(np.asarray(a), dtype=self.dtype)
        images = [as_t(i).to(self.dtype) for i in images]
        actions = [as_t(a).to(self.dtype) for a in actions]
        states = [as_t(s).to(self.dtype) for s in states]
        B = images[0].shape[0]
        feedself = (not self.train) or self.scheduled_sampling_k == -1
        if not feedself:
            k = self.scheduled_sampling_k
            ngt = int(np.int32(np.round(np.float32(B) * (k / (k + np.exp(iter_num / k))))))
        cur = states[0]
        gen_images, gen_states = [], []
        for image, action in zip(images[:-1], actions[:-1]):
            warm = len(gen_images) > self.ctx - 1
            if feedself and warm:
                prev = gen_images[-1]
            elif warm:                                             # TM:669: detached host round trip
                idx = np.arange(B)
                self.rng.shuffle(idx)
                sel = torch.zeros(B, dtype=torch.bool)
                sel[torch.from_numpy(idx[:ngt])] = True
                prev = torch.where(sel[:, None, None, None], image, gen_images[-1].detach()).float().to(self.dtype)
            else:
                prev = image
            sa = torch.cat((action, cur), 1)
            out, cur = self._step(prev, sa)
            gen_images.append(out)
            gen_states.append(cur)
        loss = 0.0
        psnr_all = 0.0
        for xx, gx in zip(images[self.ctx:], gen_images[self.ctx - 1:]):
            mse = torch.mean((xx - gx) ** 2)
            psnr_all = psnr_all + 10.0 * torch.log(1.0 / mse) / math.log(10.0)
            loss = loss + mse
        for st, gs in zip(states[self.ctx:], gen_states[self.ctx - 1:]):
            loss = loss + torch.mean((st - gs) ** 2) * 1e-4
        loss = loss / float(len(images) - self.ctx)
        self.loss, self.psnr_all = loss, psnr_all
        self.gen_images, self.gen_states = gen_images, gen_states
        return loss


def chainer_adam_step(params, grads, m, v, t, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
    """Chainer 2 AdamRule.update_core: eps added to the UNcorrected sqrt(v).
    All arguments are dicts of numpy arrays updated in place; `t` is the 1-based step count."""
    fix1 = 1.0 - math.pow(beta1, t)
    fix2 = 1.0 - math.pow(beta2, t)
    lr = alpha * math.sqrt(fix2) / fix1
    for k in params:
        g = grads[k]
        m[k] += (1 - beta1) * (g - m[k])
        v[k] += (1 - beta2) * (g * g - v[k])
        params[k] -= lr * m[k] / (np.sqrt(v[k]) + eps)
