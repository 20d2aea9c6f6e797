"""ctypes binding of libpivp_hip.so (the C ABI declared in include/pivp_hip.h).

There is no CPU fallback: if the library is missing or fails to load, `load()` raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libpivp_hip.so')

PIVP_OK = 0
MODEL_CDNA, MODEL_STP, MODEL_DNA = 0, 1, 2

_c = ctypes
_vp, _i, _f, _ll = _c.c_void_p, _c.c_int, _c.c_float, _c.c_longlong


class PivpConfig(ctypes.Structure):
    _fields_ = [('batch', _i), ('seq_len', _i), ('height', _i), ('width', _i), ('num_masks', _i),
                ('model_type', _i), ('use_state', _i), ('context_frames', _i), ('keep_activations', _i),
                ('ln_eps', _f), ('stp_zero_border', _i)]


class PivpFrameHeadArgs(ctypes.Structure):
    _fields_ = [('e6raw', _vp), ('ln_part', _vp), ('ln_nparts', _i), ('gamma', _vp), ('beta', _vp), ('ln_eps', _f),
                ('masks_w', _vp), ('masks_b', _vp), ('enc7_w', _vp), ('enc7_b', _vp), ('prev', _vp),
                ('partials', _vp), ('kslices', _i), ('head_bias', _vp), ('w2', _vp), ('b2', _vp), ('aux', _vp),
                ('out', _vp), ('masks_out', _vp), ('enc7', _vp),
                ('logits_out', _vp), ('layer0_out', _vp), ('enc6_out', _vp), ('stat_out', _vp), ('kerns_out', _vp), ('vpre_out', _vp),
                ('B', _i), ('H', _i), ('W', _i), ('num_masks', _i), ('model_type', _i), ('stp_zero_border', _i)]


# name -> (restype, argtypes); every symbol include/pivp_hip.h declares
SIGNATURES = {
    'pivp_abi_version': (_i, []),
    'pivp_build_digest': (_c.c_char_p, []),
    'pivp_build_flags': (_c.c_char_p, []),
    'pivp_plan_create': (_i, [_c.POINTER(PivpConfig), _c.POINTER(_vp)]),
    'pivp_plan_destroy': (None, [_vp]),
    'pivp_param_count': (_i, [_vp]),
    'pivp_param_name': (_c.c_char_p, [_vp, _i]),
    'pivp_param_numel': (_ll, [_vp, _i]),
    'pivp_plan_set_param': (_i, [_vp, _i, _vp]),
    'pivp_plan_workspace_bytes': (_ll, [_vp]),
    'pivp_plan_set_workspace': (_i, [_vp, _vp, _ll]),
    'pivp_reset_state': (_i, [_vp, _vp]),
    'pivp_rollout_forward': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pivp_plan_set_grad': (_i, [_vp, _i, _vp]),
    'pivp_rollout_backward': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pivp_plan_set_profiling': (_i, [_vp, _i]),
    'pivp_plan_profile_read': (_i, [_vp, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int), _c.POINTER(_c.c_double)]),
    'pivp_get_tap': (_ll, [_vp, _c.c_char_p, _i, _vp, _vp]),
    'pivp_convlstm': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_convlstm_v': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pivp_param_group': (_i, [_vp, _i]),
    'pivp_param_group_by_name': (_i, [_c.c_char_p]),
    'pivp_plan_set_grad_callback': (_i, [_vp, _vp, _vp]),
    'pivp_plan_set_main_priority': (_i, [_vp, _i]),
    'pivp_plan_set_group_join': (_i, [_vp, _i]),
    'pivp_plan_group_wait': (_i, [_vp, _i, _vp]),
    'pivp_convlstm_ln_scratch_floats': (_ll, [_i, _i, _i, _i]),
    'pivp_convlstm_ln': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _i, _i, _i, _vp, _vp]),
    'pivp_plan_set_precision': (_i, [_vp, _i]),
    'pivp_plan_get_precision': (_i, [_vp]),
    'pivp_plan_set_pack_cache': (_i, [_vp, _i]),
    'pivp_plan_params_changed': (_i, [_vp]),
    'pivp_lstm_bf16_weight_elems': (_ll, [_i, _i]),
    'pivp_pack_lstm_bf16': (_i, [_vp, _vp, _i, _i, _vp]),
    'pivp_convlstm_bf16': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'pivp_conv5x5_bf16_weight_elems': (_ll, [_i, _i]),
    'pivp_conv5x5_bf16': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pivp_conv5x5_bf16x3': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pivp_conv5x5_bf16x6': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pivp_conv5x5_fp16x3': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'pivp_pack_lstm_bf16x3': (_i, [_vp, _vp, _i, _i, _vp]),
    'pivp_convlstm_bf16x3': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'pivp_pack_lstm_bf16x6': (_i, [_vp, _vp, _i, _i, _vp]),
    'pivp_pack_lstm_fp16x3': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'pivp_convlstm_fp16x3': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'pivp_convlstm_bf16x6': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'pivp_deconv3x3s2_bf16x3': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pivp_deconv3x3s2_fp16x3': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'pivp_deconv3x3s2_bf16': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pivp_wgrad5x5_bf16': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_wgrad5x5_bf16_batch': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _vp]),
    'pivp_wgrad5x5_bf16_batch_form': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _i, _vp]),
    'pivp_wgrad5x5_bf16x6_batch': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _vp]),
    'pivp_wgrad5x5_fp16x3_batch': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _vp, _vp]),
    'pivp_convlstm_train': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_convlstm_backward': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                    _i, _i, _i, _vp]),
    'pivp_conv_backward': (_i, [_i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_conv_backward_part_floats': (_ll, [_i, _i, _i, _i, _i, _i]),
    'pivp_conv_wgrad_partial': (_i, [_i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pivp_conv_wgrad_partial_batch': (_i, [_i, _vp, _i, _i, _ll, _vp, _i, _i, _ll, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_conv_wgrad_partial_reduce': (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_convlstm_backward_dx_only': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                            _i, _i, _i, _vp]),
    'pivp_layernorm_train': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    'pivp_layernorm_backward_scratch_floats': (_ll, [_i, _i]),
    'pivp_layernorm_backward': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pivp_gates_backward_ln_scratch_floats': (_ll, [_i, _i]),
    'pivp_gates_backward_ln': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_adam_step': (_i, [_vp, _vp, _vp, _vp, _ll, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _vp]),
    'pivp_grad_pack_bf16': (_i, [_vp, _vp, _ll, _vp]),
    'pivp_grad_unpack_bf16': (_i, [_vp, _vp, _ll, _vp]),
    'pivp_grad_sum_shards': (_i, [_vp, _i, _i, _ll, _vp, _i, _vp]),
    'pivp_conv3x3s2': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pivp_deconv3x3s2': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pivp_deconv3x3s2_ln_fits': (_i, [_i, _i, _i, _i, _i, _i]),
    'pivp_deconv3x3s2_ln': (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'pivp_conv_enc0': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_layernorm_scratch_floats': (_ll, [_i, _i]),
    'pivp_layernorm': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    'pivp_enc3_state': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_heads': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pivp_linear_scratch_floats': (_ll, [_i, _i]),
    'pivp_cdna_kernels': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_stp_params': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'pivp_frame_head_fits': (_i, [_i, _i, _i, _i, _i, _i]),
    'pivp_motion_partials': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'pivp_frame_head': (_i, [_c.POINTER(PivpFrameHeadArgs), _vp]),
    'pivp_composite': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pivp_resize_images': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'pivp_select_frames': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    'pivp_wgrad5x5_f32_part_floats': (_ll, [_i, _i, _i, _i, _i, _i]),
    'pivp_wgrad5x5_f32_batch': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _i, _vp]),
    'pivp_wgrad5x5_f32_reduce': (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pivp_conv5x5_f32': (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pivp_wgrad5x5_f32_partition': (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
}

_lib = None


def load():
    """Load libpivp_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libpivp_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python physical-interaction-video-prediction_amd/build.py`. There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the ABI and the header drift apart
        fn.restype = res
        fn.argtypes = args
    # a library older (or newer) than the sources next to it must never stand in for them: a GPU test would then pass or fail on code
    # other than the code it claims to test
    from . import _digest
    try:
        shipped = _digest.source_digest()
    except OSError as e:      # a copied / installed package without csrc/ or the repo's include/: say what is missing instead of a bare open() error
        raise RuntimeError('cannot check libpivp_hip.so against its sources: %s is missing (the package needs csrc/*.hip, csrc/*.h and '
                           '../include/pivp_hip.h next to it; there is no CPU fallback)' % e.filename) from e
    built = lib.pivp_build_digest().decode()
    if built != shipped:
        raise RuntimeError(
            'libpivp_hip.so is stale: it was built from sources with digest %s..., the sources in %s have %s.... Rebuild it '
            '(`python physical-interaction-video-prediction_amd/build.py`); there is no CPU fallback.' % (built[:12], _digest.CSRC, shipped[:12]))
    _lib = lib
    return lib


def build_flags():
    """Extra compile flags of the loaded library ('' = the product build; anything else is an instrumented or timing-only variant)."""
    return load().pivp_build_flags().decode()


class PivpError(RuntimeError):
    pass


_ERR = {-1: 'PIVP_ERR_BADARG', -2: 'PIVP_ERR_LAUNCH', -3: 'PIVP_ERR_STATE'}


def check(rc, what):
    if rc != PIVP_OK:
        raise PivpError('%s failed: %s (%d)' % (what, _ERR.get(rc, 'unknown'), rc))
