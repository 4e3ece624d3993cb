"""Sample-time data augmentation shared by the game packages: plain torch, runs on whatever device the
replay tensors live on."""
import torch


def mirror_augment(batch, axes):
    """`batch` is a tuple of tensors with the sample index first; `axes[i]` names the axis of field i that runs over
    board columns (None for per-sample scalars).  Returns the same tuple with every field doubled: the originals
    followed by their mirror images."""
    if len(batch) != len(axes):
        raise ValueError(f"batch has {len(batch)} fields, expected {len(axes)}")
    out = []
    for field, axis in zip(batch, axes):
        twin = field if axis is None else field.flip(axis)
        out.append(torch.cat((field, twin), dim=0))
    return tuple(out)
