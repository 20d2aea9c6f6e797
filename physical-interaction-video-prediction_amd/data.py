"""Batch layout helper of the reference's host loop (train_model.py:51-71)."""
import numpy as np


def concat_examples(batch):
    """list of (images (T,H,W,3), actions (T,5), states (T,5)) -> time-major float32 arrays
    (T,B,3,H,W), (T,B,5), (T,B,5): the reference splits per timestep and rolls NHWC to NCHW
    (np.rollaxis(img, 3, 1), train_model.py:69)."""
    img = np.asarray([b[0] for b in batch])
    act = np.asarray([b[1] for b in batch])
    sta = np.asarray([b[2] for b in batch])
    img = np.ascontiguousarray(img.transpose(1, 0, 4, 2, 3))
    return img, np.ascontiguousarray(act.transpose(1, 0, 2)), np.ascontiguousarray(sta.transpose(1, 0, 2))
