"""Boundary types of the hot path: NestedTensor, interpolate, targets_to.

Same contract as /root/reference/util/misc.py:171-250 (padded image batch + bool padding mask).
"""
from typing import Any, Dict, List

import torch


class NestedTensor:
    def __init__(self, tensors, mask):
        self.tensors = tensors
        self.mask = mask

    def to(self, *args, **kwargs):
        t = self.tensors.to(*args, **kwargs)
        m = self.mask.to(*args, **kwargs) if self.mask is not None else None
        return type(self)(t, m)

    def decompose(self):
        return self.tensors, self.mask

    @classmethod
    def from_tensor_list(cls, tensor_list, do_round=False):
        """Pad a list of [C,h,w] images to the batch maximum (optionally to multiples of 128);
        mask is True on padding (reference misc.py:185-209)."""
        if tensor_list[0].ndim != 3:
            raise ValueError("not supported")
        c = tensor_list[0].shape[0]
        h = max(int(t.shape[1]) for t in tensor_list)
        w = max(int(t.shape[2]) for t in tensor_list)
        if do_round:
            h = (h + 127) // 128 * 128
            w = (w + 127) // 128 * 128
        b = len(tensor_list)
        dtype, device = tensor_list[0].dtype, tensor_list[0].device
        tensor = torch.zeros((b, c, h, w), dtype=dtype, device=device)
        mask = torch.ones((b, h, w), dtype=torch.bool, device=device)
        for i, img in enumerate(tensor_list):
            tensor[i, : img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
            mask[i, : img.shape[1], : img.shape[2]] = False
        return cls(tensor, mask)

    def __repr__(self):
        return repr(self.tensors)


def interpolate(input, size=None, scale_factor=None, mode="nearest", align_corners=None):
    """torch.nn.functional.interpolate that also accepts an empty channel axis (misc.py:215-230)."""
    if input.numel() == 0 and input.shape[1] == 0 and input.shape[0] != 0:
        out = torch.nn.functional.interpolate(input.transpose(0, 1), size, scale_factor, mode, align_corners)
        return out.transpose(0, 1)
    return torch.nn.functional.interpolate(input, size, scale_factor, mode, align_corners)


_HOST_KEYS = {"questionId", "tokens_positive", "noun_tokens_positive", "tokens", "dataset_name", "sentence_id",
              "original_img_id", "nb_eval", "task_id", "original_id", "idx", "cat_name"}


def targets_to(targets: List[Dict[str, Any]], device):
    """Move the tensor entries of each target dict to `device` (misc.py:234-250); `caption` is dropped."""
    return [{k: (v if k in _HOST_KEYS else v.to(device)) for k, v in t.items() if k != "caption"} for t in targets]
