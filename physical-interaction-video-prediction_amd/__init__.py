"""MI355X-native ConvLSTM + CDNA/STP/DNA next-frame prediction hot path.

Drop-in for the `Model` of kristofbc/physical-interaction-video-prediction
(src/models/train_model.py:478-764) on gfx950: Python host code on PyTorch-ROCm for memory and
streams, hand-written HIP kernels behind the C ABI in include/pivp_hip.h for the arithmetic.

The directory name carries a hyphen (it mirrors the reference repository's name), so import it
with importlib or through the `pivp_amd` alias module at the repo root:
    import pivp_amd
    model = pivp_amd.Model(num_masks=10, is_cdna=True, prefix='predict')
"""
from .model import Model, config, using_config, reference_param_shapes, default_init, scheduled_sampling_masks
from .checkpoint import save_npz, load_npz, to_internal, from_internal, save_optimizer_npz, load_optimizer_npz
from . import dataset
from .data import concat_examples
from .optimizer import Adam
from .parallel import GradAllReduce, shard_batch

__all__ = ['Model', 'config', 'using_config', 'reference_param_shapes', 'default_init',
           'scheduled_sampling_masks', 'save_npz', 'load_npz', 'to_internal', 'from_internal', 'concat_examples',
           'Adam', 'GradAllReduce', 'shard_batch', 'save_optimizer_npz', 'load_optimizer_npz', 'dataset']
