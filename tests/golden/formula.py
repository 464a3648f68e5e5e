"""Closed-form tensors shared by the golden-vector generator (which runs the real reference) and the
tests (which run the oracle / the HIP path): every value is a function of (tensor name, flat index), so
no large weights are stored and nothing depends on torch's RNG streams."""
import zlib

import numpy as np
import torch


def unit(name, n):
    """n pseudo-uniform values in [-0.5, 0.5), deterministic in (name, index)."""
    salt = np.uint64(zlib.crc32(name.encode()))
    i = np.arange(n, dtype=np.uint64)
    x = (i * np.uint64(2654435761) + salt * np.uint64(40503)) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(2246822519) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(13))) & np.uint64(0xFFFFFFFF)
    return (x.astype(np.float64) / 4294967296.0 - 0.5).astype(np.float32)


def tensor(name, shape, scale=1.0, shift=0.0):
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy(unit(name, n) * np.float32(scale) + np.float32(shift)).reshape(shape)


def fill_state_dict(sd):
    """Deterministic values for every entry of a state_dict (returns a new dict)."""
    out = {}
    for name, t in sd.items():
        shape = tuple(t.shape)
        if not t.is_floating_point():
            out[name] = t.clone()
            continue
        leaf = name.split(".")[-1]
        if "running_var" in name:
            v = tensor(name, shape, 0.5, 1.0)
        elif "running_mean" in name:
            v = tensor(name, shape, 0.2)
        elif ("norm" in name.lower() or ".bn" in name or "downsample.1" in name) and leaf == "weight":
            v = tensor(name, shape, 0.2, 1.0)
        elif leaf == "bias" or len(shape) == 1:
            v = tensor(name, shape, 0.2)
        elif "embeddings" in name or "query_embed" in name:
            v = tensor(name, shape, 1.0)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = tensor(name, shape, 3.0 / np.sqrt(fan_in))
        out[name] = v.to(t.dtype)
    return out


def sample_indices(n, count=4096):
    """Fixed subset of flat positions used to store large outputs compactly."""
    if n <= count:
        return np.arange(n)
    return (np.arange(count, dtype=np.int64) * 7919 + 13) % n


class FakeTokenized(dict):
    """Stand-in for a HF BatchEncoding in fixtures: deterministic char -> token map (4 characters per token, every 4th
    character is a separator that maps to no token; like BatchEncoding, a single argument means batch 0)."""

    def __init__(self, length):
        super().__init__(length=length)

    def to(self, device):
        return self

    def char_to_token(self, batch_or_char, char=None):
        c = batch_or_char if char is None else char
        if c < 0 or c % 4 == 3:
            return None
        t = c // 4 + 1
        return t if t < self["length"] - 1 else None
