"""Digest of the HIP sources a libpivp_hip.so is built from.

build.py embeds it in the library (`pivp_build_digest()`, include/pivp_hip.h) and `_lib.load()` recomputes it from the sources that
travel with the package: a library older than the code it claims to implement refuses to load, so a GPU test can never pass on a stale
build.  Sources only (csrc/*.hip, csrc/*.h, the public header): objects, the .so and compile flags are not part of it -- an
instrumented build (PIVP_EXTRA_FLAGS) of the same sources is still the same code."""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
HEADER = os.path.join(HERE, '..', 'include', 'pivp_hip.h')


def source_files():
    names = sorted(n for n in os.listdir(CSRC) if n.endswith('.hip') or n.endswith('.h'))
    return [os.path.join(CSRC, n) for n in names] + [HEADER]


def source_digest():
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()
