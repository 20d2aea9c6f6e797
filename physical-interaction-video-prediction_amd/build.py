"""Build the gfx950 HIP library in-tree: hipcc -> libpivp_hip.so next to this file.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.  hipcc
cross-compiles for gfx950 without a GPU, so this runs in the build container too."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['igemm_f32.hip', 'igemm_small.hip', 'deconv_tile.hip', 'convlstm_bf16.hip', 'igemm_wgrad.hip', 'wgrad_bf16.hip', 'small_kernels.hip', 'heads.hip', 'backward.hip', 'backward_heads.hip', 'pivp_c_api.hip', 'pivp_plan.hip']
LIB = os.path.join(HERE, 'libpivp_hip.so')
STAMP = os.path.join(HERE, '.libpivp_hip.stamp')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function']
if os.environ.get('PIVP_ABLATE'):   # timing-only diagnostic variants of the igemm kernel (scripts/bench_lstm_layers.py)
    FLAGS.append('-DPIVP_ABLATE')
FLAGS += os.environ.get('PIVP_EXTRA_FLAGS', '').split()   # experiments, e.g. -DPIVP_XCD_MAP=0


def _digest():
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC)) + ['../../include/pivp_hip.h']
    for n in names:
        path = os.path.join(CSRC, n)
        if os.path.isfile(path):
            h.update(n.encode())
            with open(path, 'rb') as f:
                h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into libpivp_hip.so (no-op when up to date)."""
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % src)
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
