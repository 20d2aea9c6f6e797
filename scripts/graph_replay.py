"""Does replaying the launch chain from a hipGraph beat enqueueing it?  Captures one rollout (and one train step: forward, BPTT
sweep with its side stream, Adam) of config 2 with torch.cuda.graph and times K replays against K direct calls.  Timing experiment
only: a captured Adam launch carries the step count it was captured with."""
import argparse, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--precision', default='fp32')
ap.add_argument('--leg', default='rollout', choices=['rollout', 'train'])
args = ap.parse_args()
dev = torch.device('cuda:0')
B, T, S, K = args.batch, 10, 64, args.steps
rs = np.random.RandomState(0)
images = torch.from_numpy(rs.random_sample((T, B, 3, S, S)).astype(np.float32)).to(dev)
actions = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).to(dev)
states = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).to(dev)


def timed(fn, k):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = fn()
    torch.cuda.synchronize()
    return g, out


for precision in (args.precision,):
    with pivp_amd.using_config('train', False):
        model = pivp_amd.Model(10, prefix='g', device=dev, keep_activations=False, precision=precision, is_cdna=True)

        def rollout():
            model.reset_state()
            return model([images, actions, states], 0)
        if args.leg != 'rollout':
            break
        direct = timed(rollout, K)
        ref = float(rollout())
        try:
            g, out = capture(rollout)
            replay = timed(g.replay, K)
            g.replay(); torch.cuda.synchronize()
            print('%s rollout: direct %.3f ms, graph replay %.3f ms (loss %.6f vs %.6f)' % (precision, direct, replay, ref, float(out)), flush=True)
        except Exception as e:                                             # noqa
            print('%s rollout: direct %.3f ms, capture failed: %r' % (precision, direct, e), flush=True)
            torch.cuda.synchronize()
for precision in (args.precision,):
    with pivp_amd.using_config('train', False):
        if args.leg != 'train':
            break
        tm = pivp_amd.Model(10, prefix='g', device=dev, keep_activations=True, precision=precision, is_cdna=True)
        op = pivp_amd.Adam(alpha=0.001).setup(tm)
        tm([images, actions, states], 0)

        def train():
            tm.reset_state()
            return op.update(tm, [images, actions, states], 0)
        direct = timed(train, K)
        try:
            g, out = capture(train)
            replay = timed(g.replay, K)
            print('%s train step: direct %.3f ms, graph replay %.3f ms' % (precision, direct, replay), flush=True)
        except Exception as e:                                             # noqa
            print('%s train step: direct %.3f ms, capture failed: %r' % (precision, direct, e), flush=True)
            torch.cuda.synchronize()
