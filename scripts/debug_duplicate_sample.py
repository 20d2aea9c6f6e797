"""A sample duplicated inside the batch must give bit-identical frames: locate the first differing tap when it does not."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd
from oracle import restatement as R
P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
a, b = 4, 9
imgs, acts, stas = R.synthetic_batch(32, 10)
imgs[:, b] = imgs[:, a]; acts[:, b] = acts[:, a]; stas[:, b] = stas[:, a]
for train, bwd in ((False, False), (True, False), (True, True)):
    m = pivp_amd.Model(10, prefix='x', keep_activations=True)
    m.load_state_dict_reference(P)
    with pivp_amd.using_config('train', train):
        m([imgs, acts, stas], 0)
        gen0 = torch.stack(m.gen_images).cpu().numpy()
        if bwd:
            m.cleargrads(); m.backward()
    gen = torch.stack(m.gen_images).cpu().numpy()
    print('train', train, 'backward', bwd, 'dup diff before bwd', np.abs(gen0[:, a] - gen0[:, b]).max(axis=(1, 2, 3)), 'after', np.abs(gen[:, a] - gen[:, b]).max(axis=(1, 2, 3)),
          'gen changed by backward', np.abs(gen - gen0).max())
for (a, b) in ((4, 9), (4, 8)):
    imgs, acts, stas = R.synthetic_batch(32, 10)
    imgs[:, b] = imgs[:, a]; acts[:, b] = acts[:, a]; stas[:, b] = stas[:, a]
    m = pivp_amd.Model(10, prefix='x', keep_activations=True, precision='bf16')
    m.load_state_dict_reference(P)
    with pivp_amd.using_config('train', False):
        m([imgs, acts, stas], 0)
    gen = torch.stack(m.gen_images).cpu().numpy()
    first = None
    for n in ['enc0', 'lstm1_h', 'lstm1_c', 'hidden1', 'hidden2', 'enc1', 'hidden3', 'hidden4', 'enc2', 'enc3', 'lstm5_h', 'hidden5', 'enc4', 'hidden6', 'enc5', 'hidden7', 'enc6', 'enc7']:
        t = m.tap(n, 0).cpu().numpy()
        if not np.array_equal(t[a], t[b]):
            first = (n, float(np.abs(t[a] - t[b]).max())); break
    print('bf16', (a, b), 'dup diff per step', np.abs(gen[:, a] - gen[:, b]).max(axis=(1, 2, 3)), 'first differing tap at step 0', first)
# run-to-run determinism of the bf16 mode (a data race in the LDS ring would show here)
outs = []
for rep in range(3):
    m = pivp_amd.Model(10, prefix='x', keep_activations=False, precision='bf16')
    m.load_state_dict_reference(P)
    with pivp_amd.using_config('train', False):
        m([imgs, acts, stas], 0)
    outs.append(torch.stack(m.gen_images).cpu().numpy())
print('bf16 run-to-run max diff', np.abs(outs[0] - outs[1]).max(), np.abs(outs[0] - outs[2]).max())
m.reset_state()
with pivp_amd.using_config('train', False):
    m([imgs, acts, stas], 0)
print('bf16 same model second call max diff', np.abs(torch.stack(m.gen_images).cpu().numpy() - outs[2]).max())
