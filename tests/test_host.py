"""CPU-side tests of the host logic and of the C-ABI surface (no compute calls, no GPU)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

import pivp_amd
from pivp_amd import _lib
from oracle import restatement as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    header = open(os.path.join(ROOT, 'include', 'pivp_hip.h')).read()
    declared = set(re.findall(r'\b(pivp_[a-z0-9_]+)\s*\(', header))
    declared -= {'pivp_config', 'pivp_plan'}
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), 'libpivp_hip.so does not export %s' % name
    assert declared == set(_lib.SIGNATURES), 'ctypes table and header disagree: %s' % (declared ^ set(_lib.SIGNATURES))
    assert _lib.load().pivp_abi_version() == 17


def test_stale_library_is_refused(monkeypatch):
    """_lib.load() compares the digest embedded in libpivp_hip.so (pivp_build_digest) with the digest of the sources shipped next to it:
    a library built from other sources must raise, never run (a forgotten rebuild would have tested stale code)."""
    import __graft_entry__ as g
    g.build()
    from pivp_amd import _digest
    lib = _lib.load()
    assert lib.pivp_build_digest().decode() == _digest.source_digest() and len(_digest.source_digest()) == 64
    files = [os.path.basename(f) for f in _digest.source_files()]
    assert 'pivp_hip.h' in files and 'igemm_f32.hip' in files and not [f for f in files if f.endswith('.o') or f.endswith('.so')]
    monkeypatch.setattr(_lib, '_lib', None)                         # force a fresh load ...
    monkeypatch.setattr(_digest, 'source_digest', lambda: 'f' * 64)  # ... against sources that differ from what was compiled
    with pytest.raises(RuntimeError, match='stale'):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load().pivp_abi_version() == 17


def test_product_build_records_no_extra_flags_and_a_missing_source_tree_is_named(monkeypatch):
    """pivp_build_flags(): '' for the product build (an instrumented variant -- PIVP_EXTRA_FLAGS -- has the product's source digest, so it must say
    what it is).  And a package copied without csrc/ or include/ fails the digest check with a sentence, not a bare open() error."""
    import __graft_entry__ as g
    g.build()
    from pivp_amd import _digest
    assert _lib.build_flags() == ''
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_digest, 'HEADER', os.path.join(ROOT, 'include', 'no_such_header.h'))
    with pytest.raises(RuntimeError, match='no_such_header.h is missing'):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load().pivp_abi_version() == 17


def test_plan_param_table_matches_reference_keys():
    lib = _lib.load()
    for mt, nm, code in (('CDNA', 10, 0), ('STP', 10, 1), ('DNA', 1, 2)):
        cfg = _lib.PivpConfig(batch=2, seq_len=10, height=64, width=64, num_masks=nm, model_type=code, use_state=1,
                              context_frames=2, keep_activations=0, ln_eps=1e-6, stp_zero_border=0)
        h = ctypes.c_void_p()
        assert lib.pivp_plan_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
        names = [lib.pivp_param_name(h, i).decode() for i in range(lib.pivp_param_count(h))]
        shapes = pivp_amd.reference_param_shapes(nm, mt, True, 64, 64)
        assert sorted(names) == sorted(shapes)
        assert sorted(names) == sorted(R.param_shapes(num_masks=nm, model_type=mt))
        for i, n in enumerate(names):
            dummy = np.zeros(shapes[n], np.float32)
            assert lib.pivp_param_numel(h, i) == pivp_amd.to_internal(n, dummy).size, n
        assert lib.pivp_plan_workspace_bytes(h) > 0
        lib.pivp_plan_destroy(h)
    # bad configs are rejected, not crashed on
    bad = _lib.PivpConfig(batch=2, seq_len=10, height=60, width=64, num_masks=10, model_type=0, use_state=1,
                          context_frames=2, keep_activations=0, ln_eps=1e-6, stp_zero_border=0)
    h = ctypes.c_void_p()
    assert lib.pivp_plan_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    bad2 = _lib.PivpConfig(batch=2, seq_len=10, height=64, width=64, num_masks=2, model_type=2, use_state=1,
                           context_frames=2, keep_activations=0, ln_eps=1e-6, stp_zero_border=0)
    assert lib.pivp_plan_create(ctypes.byref(bad2), ctypes.byref(h)) == -1   # DNA needs num_masks == 1 (TM:390)


def test_reference_param_shapes_match_oracle_and_count():
    for mt, nm in (('CDNA', 10), ('STP', 10), ('DNA', 1)):
        a = pivp_amd.reference_param_shapes(nm, mt, True, 64, 64)
        b = R.param_shapes(num_masks=nm, model_type=mt)
        assert dict(a) == dict(b)
    assert sum(int(np.prod(s)) for s in pivp_amd.reference_param_shapes(10, 'CDNA', True, 64, 64).values()) == 9212159
    assert sum(int(np.prod(s)) for s in pivp_amd.reference_param_shapes(10, 'CDNA', True, 128, 128).values()) == 18059519


def test_checkpoint_layout_roundtrip_and_permutations():
    shapes = pivp_amd.reference_param_shapes(10, 'CDNA', True, 64, 64)
    rs = np.random.RandomState(0)
    for key in ('enc0/W', 'enc4/W', 'lstm5/conv/W', 'hidden5/norm/gamma', 'masks/W', 'model/cdna_kerns/W', 'enc3/W',
                'current_state/W', 'lstm1/conv/b'):
        a = rs.randn(*shapes[key]).astype(np.float32)
        flat = pivp_amd.to_internal(key, a)
        back = pivp_amd.from_internal(key, flat, shapes[key])
        assert np.array_equal(a, back), key
    # conv: internal [tap][Cin/32][Cout][32] (K-inner packed)
    W = rs.randn(*shapes['lstm5/conv/W']).astype(np.float32)
    f = pivp_amd.to_internal('lstm5/conv/W', W).reshape(25, 6, 512, 32)
    assert f[7, 100 // 32, 300, 100 % 32] == W[300, 100, 1, 2]
    # deconv: reference (Cin,Cout,kh,kw)
    Wd = rs.randn(*shapes['enc5/W']).astype(np.float32)
    fd = pivp_amd.to_internal('enc5/W', Wd).reshape(9, 3, 96, 32)
    assert fd[5, 0, 20, 10] == Wd[10, 20, 1, 2]
    # enc0 (Cin = 3) stays [tap][Cin][Cout]
    W0 = rs.randn(32, 3, 5, 5).astype(np.float32)
    assert pivp_amd.to_internal('enc0/W', W0).reshape(25, 3, 32)[7, 2, 9] == W0[9, 2, 1, 2]
    # LN gamma: NCHW-flat c*HW+p -> NHWC-flat p*C+c
    g = np.arange(8192, dtype=np.float32)
    fg = pivp_amd.to_internal('hidden5/norm/gamma', g)
    assert fg[5 * 128 + 7] == g[7 * 64 + 5]
    # cdna_kerns: in-feature c*64+p -> row p*128+c, 256 padded columns
    Wk = rs.randn(250, 8192).astype(np.float32)
    fk = pivp_amd.to_internal('model/cdna_kerns/W', Wk).reshape(8192, 256)
    assert fk[9 * 128 + 3, 17] == Wk[17, 3 * 64 + 9] and np.all(fk[:, 250:] == 0)


def test_scheduled_sampling_masks_match_reference_rng_use():
    # same global-RNG consumption as the reference: one shuffle(arange(B)) per step t >= ctx (TM:93-94, TM:669)
    B, T, ctx, k, it = 8, 6, 2, 4.0, 3.0
    np.random.seed(11)
    mask = pivp_amd.scheduled_sampling_masks(B, T, ctx, k, it)
    np.random.seed(11)
    ngt = R.num_ground_truth_schedule(B, k, it)
    gt = np.zeros((B, 1, 1, 1), np.float32); gen = np.ones((B, 1, 1, 1), np.float32)
    for t in range(T - 1):
        if t >= ctx:
            out = R.scheduled_sample(gt, gen, B, ngt)
            assert np.array_equal(out[:, 0, 0, 0] == 0, mask[t] == 1)
        else:
            assert mask[t].sum() == 0
    assert mask[ctx:].sum(axis=1).tolist() == [ngt] * (T - 1 - ctx)


def test_rank_offset_sampling_streams():
    # data parallel: every rank draws its scheduled-sampling shuffles from its OWN stream (train.py: RandomState(1 + rank)),
    # leaves the global RNG -- which orders the dataset identically on every rank -- untouched, and is reproducible per rank
    B, T, ctx, k, it = 8, 6, 2, 4.0, 3.0
    np.random.seed(5)
    before = np.random.get_state()[1].copy()
    masks = [pivp_amd.scheduled_sampling_masks(B, T, ctx, k, it, rng=np.random.RandomState(1 + rank)) for rank in range(4)]
    assert np.array_equal(np.random.get_state()[1], before)               # the global stream was not consumed
    assert all(m[ctx:].sum(axis=1).tolist() == masks[0][ctx:].sum(axis=1).tolist() for m in masks)   # same count per step (TM:654-656)
    assert len({m.tobytes() for m in masks}) == 4                         # but different samples on every rank
    again = pivp_amd.scheduled_sampling_masks(B, T, ctx, k, it, rng=np.random.RandomState(1 + 2))
    assert np.array_equal(again, masks[2])


def test_concat_examples_matches_reference_layout():
    rs = np.random.RandomState(0)
    batch = [(rs.rand(4, 8, 8, 3).astype(np.float32), rs.rand(4, 5).astype(np.float32), rs.rand(4, 5).astype(np.float32))
             for _ in range(3)]
    a = pivp_amd.concat_examples(batch)
    b = R.concat_examples(batch)
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(x, y)


def test_constructor_surface_and_errors():
    with pytest.raises(ValueError, match='No network specified'):
        pivp_amd.Model(10, is_cdna=False, is_dna=False, is_stp=False)       # TM:540
    m = pivp_amd.Model(10, is_cdna=True, is_stp=True)
    assert m.model_type == 'CDNA'                                           # precedence cdna > stp > dna (TM:532-537)
    assert pivp_amd.Model(10, is_cdna=False, is_stp=True, is_dna=True).model_type == 'STP'
    assert pivp_amd.config.train is True
    with pivp_amd.using_config('train', False):
        assert pivp_amd.config.train is False
    assert pivp_amd.config.train is True
    with pytest.raises(TypeError):
        m([np.zeros((3, 1, 3, 64, 64), np.float32)])                        # images only -> states[0] fails (TM:646)


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, 'physical-interaction-video-prediction_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'oracle' not in src.replace('no oracle', ''), fn


def test_graft_entry_build_in_a_fresh_interpreter():
    # the driver calls build() from its own process: no module this suite happens to import may be relied on
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.build(); print("ok")'], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]


# the ConvLSTM layers of the model at rank sizes (cx, C, B, H, W), as tests/test_gpu_backward_ops.py runs them on the GPU, plus ragged cuts
_WGRAD_SLOT_SHAPES = [(32, 32, 32, 32, 32), (32, 32, 4, 32, 32), (32, 64, 32, 16, 16), (64, 64, 32, 16, 16), (64, 128, 32, 8, 8), (128, 128, 32, 8, 8),
                      (256, 64, 32, 16, 16), (96, 32, 32, 32, 32), (32, 32, 1, 8, 8), (64, 32, 3, 16, 8), (32, 96, 5, 4, 16), (32, 32, 16, 64, 64)]


@pytest.mark.parametrize('form', [0, 1, 2])
@pytest.mark.parametrize('has_h', [1, 0])
def test_weight_gradient_slot_partition_covers_every_item_once(form, has_h):
    """wgrad5x5p.hip cuts one timestep's (tile, 16-pixel chunk) items into per-block segments, each with a slot of its own, and the reduction adds a tile's
    slots in a fixed order.  Walked on the host through pivp_wgrad5x5_f32_partition (no GPU): every item lies in exactly one segment, no block has more than
    `maxseg` segments, the reduction reads exactly the slots the kernel writes for that tile -- each once, pixel parts then blocks ascending -- and the
    buffer size the sizing entry returns holds them all."""
    lib = _lib.load()
    for cx, C, B, H, W in _WGRAD_SLOT_SHAPES:
        if form == 2 and (4 * C) % 64:
            continue
        geom = np.zeros(8, np.int32)
        ns, nl = ctypes.c_int(0), ctypes.c_int(0)
        args = (cx, C, has_h, B, H, W, form)
        assert lib.pivp_wgrad5x5_f32_partition(*args, geom.ctypes.data, None, 0, ctypes.byref(ns), None, 0, ctypes.byref(nl)) == 0
        J, PP, TP, NTW, T, cpt, maxseg, slot_floats = (int(v) for v in geom)
        cin = cx + (C if has_h else 0)
        assert PP * TP == 8 and NTW in (1, 2) and (form == 0 or NTW == form)
        assert T == 5 * (cin // 32) * (4 * C // (32 * NTW)) and cpt == B * H * W // 16 and slot_floats == 5 * NTW * 1024 + 64
        segs = np.zeros((ns.value, 5), np.int32)
        slots = np.zeros((nl.value, 2), np.int32)
        assert lib.pivp_wgrad5x5_f32_partition(*args, geom.ctypes.data, segs.ctypes.data, len(segs), ctypes.byref(ns), slots.ctypes.data, len(slots),
                                               ctypes.byref(nl)) == 0
        assert ns.value == len(segs) and nl.value == len(slots)
        blk, seg, tile, k0, k1 = segs.T
        assert (k1 > k0).all() and (seg < maxseg).all() and (blk < 8 * J).all() and (tile < T).all() and (k0 >= 0).all() and (k1 <= cpt).all()
        cover = np.zeros((T, cpt + 1), np.int64)                   # difference array over chunks, per tile
        np.add.at(cover, (tile, k0), 1)
        np.add.at(cover, (tile, k1), -1)
        assert (np.cumsum(cover, 1)[:, :cpt] == 1).all(), (cx, C, B, H, W)
        # a block's segments are numbered 0, 1, ... in the order it works through them, and an XCD (block % 8) holds one (pixel part, tile part) pair
        for b in np.unique(blk):
            assert (seg[blk == b] == np.arange((blk == b).sum())).all()
        written = {}
        for b, s, t in zip(blk, seg, tile):
            written.setdefault(int(t), []).append(int(b) * maxseg + int(s))
        read = {}
        for t, s in slots:
            read.setdefault(int(t), []).append(int(s))
        assert set(read) == set(written) == set(range(T))
        for t in range(T):
            assert sorted(read[t]) == sorted(written[t]) and len(set(read[t])) == len(read[t]), (cx, C, B, H, W, t)
            # the fixed order: by pixel part (the XCD's low digit), then by block within the part
            order = [((s // maxseg) % 8 % PP, s // maxseg // 8) for s in read[t]]
            assert order == sorted(order)
        floats = lib.pivp_wgrad5x5_f32_part_floats(cx, C, B, H, W, form)
        assert floats >= 8 * J * maxseg * slot_floats and (max(max(v) for v in written.values()) + 1) * slot_floats <= floats
        # the work is balanced: no block has more than one item above the mean of its part
        items = np.zeros(8 * J, np.int64)
        np.add.at(items, blk, k1 - k0)
        for x in range(8):
            part = items[x::8]
            assert part.max() - part.min() <= 1


def test_weight_gradient_slot_entries_refuse_shapes_the_kernel_does_not_serve():
    lib = _lib.load()
    geom = np.zeros(8, np.int32)
    n = ctypes.c_int(0)
    for cx, C, B, H, W, form in [(32, 32, 2, 24, 24, 0), (32, 32, 2, 16, 4, 0), (16, 32, 2, 16, 16, 0), (32, 24, 2, 16, 16, 0), (32, 32, 0, 16, 16, 0),
                                 (32, 32, 2, 16, 16, 3), (32, 32, 2, 16, 16, -1)]:
        assert lib.pivp_wgrad5x5_f32_partition(cx, C, 1, B, H, W, form, geom.ctypes.data, None, 0, ctypes.byref(n), None, 0, ctypes.byref(n)) != 0
        assert lib.pivp_wgrad5x5_f32_part_floats(cx, C, B, H, W, form) <= 0
    assert lib.pivp_wgrad5x5_f32_partition(32, 32, 1, 2, 16, 16, 0, None, None, 0, ctypes.byref(n), None, 0, ctypes.byref(n)) != 0
