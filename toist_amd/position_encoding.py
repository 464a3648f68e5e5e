"""Positional encodings (reference: /root/reference/models/position_encoding.py)."""
import torch
from torch import nn

from . import kernels as k
from .misc import NestedTensor


class PositionEmbeddingSine(nn.Module):
    """Sine/cosine image position encoding (position_encoding.py:13-49), computed by the HIP kernel
    toist_sine_position.  Only the normalize=True / scale=2*pi configuration built by
    build_position_encoding (:89-93) is on the hot path."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=True, scale=None):
        super().__init__()
        if not normalize:
            raise NotImplementedError("PositionEmbeddingSine: only normalize=True (the reference's configuration)")
        self.num_pos_feats = num_pos_feats
        self.temperature = float(temperature)

    def forward(self, tensor_list: NestedTensor):
        mask = tensor_list.mask
        B, H, W = mask.shape
        out = torch.empty(B, 2 * self.num_pos_feats, H, W, dtype=torch.float32, device=mask.device)
        m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()
        k.sine_position(m8, self.num_pos_feats, self.temperature, out_nchw=out)
        return out

    def tokens(self, mask, tail=0):
        """bf16 [B, H*W + tail, 2F] token-major encoding for the native encoder path; the `tail` rows behind every image's tokens (the
        caption tokens of the cross-modal sequence) are zero."""
        B, H, W = mask.shape
        out = torch.empty(B, H * W + tail, 2 * self.num_pos_feats, dtype=torch.bfloat16, device=mask.device)
        m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()
        if tail:
            k.sine_position_seq(m8, self.num_pos_feats, self.temperature, out)
        else:
            k.sine_position(m8, self.num_pos_feats, self.temperature, out_tok=out)
        return out


class PositionEmbeddingLearned(nn.Module):
    """Learned absolute position embedding (position_encoding.py:52-86); plain table lookups."""

    def __init__(self, num_pos_feats=256):
        super().__init__()
        self.row_embed = nn.Embedding(50, num_pos_feats)
        self.col_embed = nn.Embedding(50, num_pos_feats)
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, tensor_list: NestedTensor):
        x = tensor_list.tensors
        h, w = x.shape[-2:]
        xe = self.col_embed(torch.arange(w, device=x.device))
        ye = self.row_embed(torch.arange(h, device=x.device))
        pos = torch.cat([xe.unsqueeze(0).repeat(h, 1, 1), ye.unsqueeze(1).repeat(1, w, 1)], dim=-1)
        return pos.permute(2, 0, 1).unsqueeze(0).repeat(x.shape[0], 1, 1, 1)


def build_position_encoding(args):
    n_steps = args.hidden_dim // 2
    if args.position_embedding in ("v2", "sine"):
        return PositionEmbeddingSine(n_steps, normalize=True)
    if args.position_embedding in ("v3", "learned"):
        return PositionEmbeddingLearned(n_steps)
    raise ValueError(f"not supported {args.position_embedding}")
