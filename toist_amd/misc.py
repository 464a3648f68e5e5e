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


# ---- data-side collate (reference util/misc.py:40-127) and host -> HBM staging ---------------------------------------
def _batched_positive_map(targets):
    """[sum_i T_i, max_i L_i] fp32: the per-image positive maps stacked and right-padded with zeros (misc.py:64-73, 112-124)."""
    width = max(int(t["positive_map"].shape[1]) for t in targets)
    rows = sum(int(t["positive_map"].shape[0]) for t in targets)
    out = torch.zeros((rows, width), dtype=torch.bool)
    at = 0
    for t in targets:
        pm = t["positive_map"]
        out[at:at + pm.shape[0], :pm.shape[1]] = pm
        at += pm.shape[0]
    return out.float()


def _example_rel(items):
    return [i for i, group in enumerate(items) for _ in range(len(group))]


def collate_fn_plain(do_round, batch):
    """Plain (non-distillation) collate: every dataset item is (list of images, list of targets); the lists are
    flattened, images padded into one NestedTensor, positive maps stacked (misc.py:93-127)."""
    images, targets = zip(*batch)
    flat_targets = [t for group in targets for t in group]
    out = {"example_rel": _example_rel(images),
           "samples": NestedTensor.from_tensor_list([im for group in images for im in group], do_round),
           "targets": flat_targets}
    if flat_targets and "positive_map" in flat_targets[0]:
        out["positive_map"] = _batched_positive_map(flat_targets)
    return out


def collate_fn(do_round, batch):
    """Distillation collate: every item carries a (noun, pronoun) pair of images and of targets; each side is batched
    on its own -> two-element lists under "samples" / "targets" / "positive_map" (misc.py:40-91)."""
    images, targets = zip(*batch)
    sides_t = [[pair[s] for pair in targets] for s in (0, 1)]
    out = {"samples": [NestedTensor.from_tensor_list([pair[s] for pair in images], do_round) for s in (0, 1)],
           "example_rel": _example_rel(images), "targets": sides_t}
    if "positive_map" in sides_t[0][0]:
        out["positive_map"] = [_batched_positive_map(side) for side in sides_t]
    return out


class DeviceStager:
    """Pinned-memory, asynchronous host -> HBM staging of collated batches on its own HIP stream (replaces the
    `samples.to(device)` / `targets_to` / `positive_map.to(device)` sequence of engine.py:54-60 with copies that
    overlap the previous step).  stage(batch) starts the copies; the returned batch's tensors may be used on the
    current stream after wait()."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceStager stages into GPU memory: there is no CPU path")
        self.stream = torch.cuda.Stream(device=self.device)
        self._keep = []
        self._staged = []      # device tensors of the batch in flight (allocated on the copy stream)

    def _move(self, t):
        if not torch.is_tensor(t):
            return t
        src = t if t.is_pinned() else t.pin_memory()
        self._keep.append(src)                      # pinned source must outlive the asynchronous copy
        dst = src.to(self.device, non_blocking=True)
        self._staged.append(dst)
        return dst

    def _move_targets(self, targets):
        return [{k: (v if k in _HOST_KEYS else self._move(v)) for k, v in t.items() if k != "caption"} for t in targets]

    def stage(self, batch):
        self._keep = []
        self._staged = []
        out = dict(batch)
        with torch.cuda.stream(self.stream):
            s = batch["samples"]
            one = lambda nt: NestedTensor(self._move(nt.tensors), self._move(nt.mask))
            out["samples"] = [one(x) for x in s] if isinstance(s, (list, tuple)) else one(s)
            t = batch["targets"]
            out["targets"] = [self._move_targets(x) for x in t] if t and isinstance(t[0], list) else self._move_targets(t)
            if "positive_map" in batch:
                pm = batch["positive_map"]
                out["positive_map"] = [self._move(x) for x in pm] if isinstance(pm, (list, tuple)) else self._move(pm)
        return out

    def wait(self):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        # The blocks were allocated while the copy stream was current: tell the caching allocator that the consumer stream uses them
        # too, otherwise a dropped batch returns them to the copy stream's pool and a later stage() may overwrite them while kernels
        # of an earlier step (the host runs ahead of the GPU) are still reading.
        for t in self._staged:
            t.record_stream(cur)
