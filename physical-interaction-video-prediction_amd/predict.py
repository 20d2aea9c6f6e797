"""Predict entry point with the reference's arguments (predict_model.py:57-76): load a checkpoint, resize the raw frames of
one sequence to the trained size (bilinear, /255: predict_model.py:119-122), run ONE feed-self rollout
(predict_model.py:126-128) and rescale every predicted frame to uint8 by its own min/max (predict_model.py:131-137).
The strip / GIF rendering of predict_model.py:140-246 is out of scope; the frames are written as `.npy`."""
import argparse
import os

import numpy as np
import torch

from . import _lib
from . import dataset as ds
from .checkpoint import load_npz
from .data import concat_examples
from .model import Model, using_config


def resize_images(frames, out_hw, device='cuda:0', scale=1.0):
    """chainer.functions.resize_images on the GPU: frames (B, C, H, W) -> (B, C, out_h, out_w) * scale."""
    lib = _lib.load()
    x = torch.as_tensor(np.ascontiguousarray(frames, dtype=np.float32)).to(device)
    B, C, H, W = x.shape
    out = torch.empty((B, C, out_hw[0], out_hw[1]), dtype=torch.float32, device=device)
    with torch.cuda.device(x.device):
        _lib.check(lib.pivp_resize_images(x.data_ptr(), out.data_ptr(), B * C, H, W, out_hw[0], out_hw[1], float(scale),
                                          torch.cuda.current_stream(x.device).cuda_stream), 'pivp_resize_images')
    return out


def rescale_to_uint8(frame):
    """predict_model.py:133-137: (x - min) / max * 255 per frame, first sample of the batch."""
    r = np.array(frame, dtype=np.float32)
    r -= r.min()
    r /= r.max()
    r *= 255.0
    return r.astype(np.uint8)


def build_parser():
    p = argparse.ArgumentParser(description='Predict the next {time_step} frame based on a trained {model}')
    p.add_argument('model_dir'); p.add_argument('model_name'); p.add_argument('data_index', type=int)
    p.add_argument('--models_dir', default='models')
    p.add_argument('--data_dir', default='data/processed/brain-robotics-data/push/push_testnovel')
    p.add_argument('--time_step', type=int, default=8)
    p.add_argument('--model_type', default='')
    p.add_argument('--schedsamp_k', type=float, default=-1)
    p.add_argument('--context_frames', type=int, default=2)
    p.add_argument('--use_state', type=int, default=1)
    p.add_argument('--num_masks', type=int, default=10)
    p.add_argument('--image_height', type=int, default=64)
    p.add_argument('--image_width', type=int, default=64)
    p.add_argument('--gpu', type=int, default=0)
    p.add_argument('--out', default='')
    return p


def predict(args):
    path = os.path.join(args.models_dir, args.model_dir)
    if not os.path.exists(os.path.join(path, args.model_name)):
        raise ValueError("Directory {} does not exists".format(path))
    if not os.path.exists(args.data_dir):
        raise ValueError("Directory {} does not exists".format(args.data_dir))
    image, image_pred, _, action, state = ds.get_data_info(args.data_dir, args.data_index)
    img_pred, act_pred, sta_pred = concat_examples([[image_pred, action, state]])
    model_type = args.model_type
    if model_type == '':
        parts = args.model_dir.split('-')
        if len(parts) != 4:
            raise ValueError("Model {} is not recognized, use --model_type to describe the type".format(args.model_dir))
        model_type = parts[2]
    device = 'cuda:%d' % args.gpu
    model = Model(num_masks=args.num_masks, is_cdna=model_type == 'CDNA', is_dna=model_type == 'DNA', is_stp=model_type == 'STP',
                  use_state=args.use_state, scheduled_sampling_k=args.schedsamp_k, num_frame_before_prediction=args.context_frames,
                  prefix='predict', device=device)
    load_npz(os.path.join(path, args.model_name), model)
    T = img_pred.shape[0]
    resized = torch.stack([resize_images(img_pred[t], (args.image_height, args.image_width), device, 1.0 / 255.0) for t in range(T)])
    with using_config('train', False):
        loss = model([resized, act_pred, sta_pred], 0)
        predicted = model.gen_images
    frames = np.stack([rescale_to_uint8(p[0].cpu().numpy()) for p in predicted])
    return float(loss), frames


def main(argv=None):
    args = build_parser().parse_args(argv)
    loss, frames = predict(args)
    out = args.out or os.path.join(args.models_dir, args.model_dir, 'prediction-%d.npy' % args.data_index)
    np.save(out, frames)
    print('loss %.6f; %d predicted frames -> %s' % (loss, len(frames), out))


if __name__ == '__main__':
    main()
