"""Build the gfx950 HIP library in-tree: hipcc -> libpivp_hip.so next to this file.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.  hipcc
cross-compiles for gfx950 without a GPU, so this runs in the build container too."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['igemm_f32.hip', 'igemm_small.hip', 'deconv_tile.hip', 'convlstm_bf16.hip', 'conv5x5_bf16.hip', 'igemm_wgrad.hip', 'wgrad3x3s2.hip', 'wgrad5x5p.hip', 'wgrad_bf16.hip', 'small_kernels.hip', 'heads.hip', 'frame_head.hip', 'backward.hip', 'backward_heads.hip', 'pivp_c_api.hip', 'pivp_plan.hip']
LIB = os.path.join(HERE, 'libpivp_hip.so')
STAMP = os.path.join(HERE, '.libpivp_hip.stamp')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function']
EXTRA = os.environ.get('PIVP_EXTRA_FLAGS', '').split()     # instrumented builds, e.g. -DPIVP_F32_STAMPS: recorded in the library (pivp_build_flags)
if os.environ.get('PIVP_ABLATE'):
    EXTRA = ['-DPIVP_ABLATE'] + EXTRA
FLAGS += EXTRA


def _load_digest_module():
    # build.py also runs stand-alone (python build.py, __graft_entry__.build), outside the package: load the sibling file by path
    import importlib.util
    spec = importlib.util.spec_from_file_location('pivp_digest', os.path.join(HERE, '_digest.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _digest():
    """-> (source digest: what the library embeds and `_lib.load()` checks, stamp: source digest + compile flags, the rebuild key)."""
    src = _load_digest_module().source_digest()
    return src, hashlib.sha256((src + ' ' + ' '.join(FLAGS)).encode()).hexdigest()


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into libpivp_hip.so (no-op when up to date)."""
    src_dig, dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    # every header any source may include: a change there recompiles everything, a change in one .hip only that object
    hdr = hashlib.sha256()
    for n in sorted(os.listdir(CSRC)) + ['../../include/pivp_hip.h']:
        if n.endswith('.h'):
            with open(os.path.join(CSRC, n), 'rb') as f:
                hdr.update(n.encode() + f.read())
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        if src == 'pivp_c_api.hip':      # pivp_build_digest(): the digest of the sources this library is built from
            cmd.insert(-4, '-DPIVP_BUILD_DIGEST="%s"' % src_dig)
            cmd.insert(-4, '-DPIVP_BUILD_FLAGS="%s"' % ' '.join(EXTRA).replace('"', "'"))
        objs.append(obj)
        with open(os.path.join(CSRC, src), 'rb') as f:
            key = hashlib.sha256(hdr.digest() + f.read() + ' '.join(cmd).encode()).hexdigest()
        ostamp = obj + '.stamp'
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == key:
            continue
        if verbose:
            print(' '.join(cmd))
        procs.append((src, ostamp, key, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = None
    for src, ostamp, key, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            failed = failed or src
            continue
        with open(ostamp, 'w') as f:
            f.write(key)
        if verbose and out:
            print(out.decode())
    if failed:
        raise RuntimeError('hipcc failed on %s' % failed)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
