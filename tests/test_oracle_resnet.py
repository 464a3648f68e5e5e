"""oracle/model_ref.resnet_body -- the torchvision ResNet-101 (v1.5) body the reference wraps at /root/reference/models/backbone.py:64-91 -- against an INDEPENDENT
third-party implementation of the same architecture.  torchvision is not installed in the image (the restatement was so far checked by key names / shapes /
parameter count and by reading); Hugging Face `transformers` IS, and its ResNetModel with the microsoft/resnet-101 configuration (bottleneck layers, depths
3-4-23-3, widths 256-512-1024-2048, stride on the 3x3 convolution = "v1.5", 7x7/2 stem + 3x3/2 max-pool pad 1, projection shortcuts in the first block of every
stage) is the network torchvision's resnet101 checkpoints are converted to.  Random weights in torchvision's naming are copied into both; BatchNorm runs on its
running statistics (= the reference's FrozenBatchNorm2d, backbone.py:21-60, eps 1e-5).  CPU only."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_ref  # noqa: E402


def _torchvision_style_weights(seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5

    def bn(name, c, gain=1.0):
        sd[name + ".weight"] = (0.5 + torch.rand(c, generator=g)) * gain
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)
    conv("body.conv1", 64, 3, 7)
    bn("body.bn1", 64)
    cin = 64
    for li, (nb, width) in enumerate(zip(model_ref.RESNET101_BLOCKS, (64, 128, 256, 512)), start=1):
        for bi in range(nb):
            p = f"body.layer{li}.{bi}."
            conv(p + "conv1", width, cin, 1)
            bn(p + "bn1", width)
            conv(p + "conv2", width, width, 3)
            bn(p + "bn2", width)
            conv(p + "conv3", 4 * width, width, 1)
            bn(p + "bn3", 4 * width, gain=0.3)           # keeps 33 residual blocks of random weights from blowing the activations up
            if bi == 0:
                conv(p + "downsample.0", 4 * width, cin, 1)
                bn(p + "downsample.1", 4 * width)
            cin = 4 * width
    return sd


def _to_hf(sd):
    out = {}

    def put(dst, src):
        out[dst + ".convolution.weight"] = sd[src[0] + ".weight"]
        for k_ in ("weight", "bias", "running_mean", "running_var"):
            out[dst + ".normalization." + k_] = sd[src[1] + "." + k_]
    put("embedder.embedder", ("body.conv1", "body.bn1"))
    for li, nb in enumerate(model_ref.RESNET101_BLOCKS, start=1):
        for bi in range(nb):
            p, q = f"body.layer{li}.{bi}.", f"encoder.stages.{li - 1}.layers.{bi}."
            for j in (1, 2, 3):
                put(q + f"layer.{j - 1}", (p + f"conv{j}", p + f"bn{j}"))
            if bi == 0:
                put(q + "shortcut", (p + "downsample.0", p + "downsample.1"))
    return out


@pytest.mark.parametrize("hw", [(64, 96), (75, 101)])
def test_resnet101_body_against_the_hf_transformers_implementation(hw):
    transformers = pytest.importorskip("transformers")
    cfg = transformers.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=list(model_ref.RESNET101_BLOCKS), layer_type="bottleneck",
                                    hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    hf = transformers.ResNetModel(cfg).eval()
    sd = _torchvision_style_weights()
    missing, unexpected = hf.load_state_dict(_to_hf(sd), strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (missing[:5], unexpected[:5])
    assert sum(v.numel() for k_, v in sd.items() if k_.endswith("weight") and v.dim() == 4) == 42_394_816          # the convolution parameters of ResNet-101
    x = torch.randn(2, 3, *hw, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = hf(x, output_hidden_states=True).hidden_states[1:]                 # the four stage outputs
        got = model_ref.resnet_body(x, sd, "body.")
    assert len(got) == 4
    for i, (a, b) in enumerate(zip(got, ref)):
        assert a.shape == b.shape == (2, 256 * 2 ** i, -(-hw[0] // (4 * 2 ** i)), -(-hw[1] // (4 * 2 ** i))), (i, a.shape, b.shape)
        err = float((a - b).abs().max() / b.abs().max())
        assert err < 1e-4, (i, err)
