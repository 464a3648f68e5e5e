"""ResNet-101 backbone with frozen BatchNorm, NHWC bf16 implicit-GEMM convolutions on the MI355X.

Mirrors the interface of /root/reference/models/backbone.py (FrozenBatchNorm2d :21-58, BackboneBase
:61-80, Backbone :83-91, Joiner :165-178, build_backbone :181-198) with the same state_dict keys
(`0.body.layer3.5.conv2.weight`, `0.body.bn1.running_var`, ...), so published checkpoints load.  The
architecture is torchvision's resnet v1.5 bottleneck net (stride on the 3x3), which the reference
instantiates through `torchvision.models.resnet101` (backbone.py:87-89); torchvision itself is not
used.  FrozenBN is folded: the bf16 compute copy of each conv weight is pre-multiplied by the BN
scale and the shift rides in the GEMM epilogue.
"""
import os
from collections import OrderedDict

import torch
from torch import nn

from .knobs import knob
from . import engine, functions
from . import kernels as k
from . import ops
from .misc import NestedTensor
from .position_encoding import build_position_encoding

BF16 = torch.bfloat16
FUSED_STEM = knob("TOIST_FUSED_STEM", True)     # conv1 + bn1 + relu + maxpool as one launch (csrc/stem.hip)


_BN_EPOCH = [0]   # bumped whenever any FrozenBatchNorm2d invalidates its folded scale/shift


class FrozenBatchNorm2d(nn.Module):
    """Buffers only (weight, bias, running_mean, running_var); reference backbone.py:21-58."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._ss = None
        _BN_EPOCH[0] += 1

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        self._ss = None
        _BN_EPOCH[0] += 1
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

        self._ss = None
        _BN_EPOCH[0] += 1

    def _apply(self, fn, *a, **kw):
        self._ss = None
        _BN_EPOCH[0] += 1
        return super()._apply(fn, *a, **kw)

    def scale_shift(self):
        """(scale, shift) fp32 of x*scale+shift; cached (the buffers are frozen), reset on load / move."""
        if self._ss is None or self._ss[0].device != self.weight.device:
            eps = 1e-5
            scale = self.weight * (self.running_var + eps).rsqrt()
            shift = self.bias - self.running_mean * scale
            self._ss = (scale.float().contiguous(), shift.float().contiguous())
        return self._ss


class ConvWeight(nn.Module):
    """Bias-free convolution parameter holder; `weight` is [Co,Cin,R,S] stored channels_last."""

    def __init__(self, cin, cout, ksize, stride=1, padding=0):
        super().__init__()
        w = torch.empty(cout, cin, ksize, ksize)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.stride, self.padding, self.ksize = stride, padding, ksize

    def _load_from_state_dict(self, state_dict, prefix, *args):
        key = prefix + "weight"
        if key in state_dict:
            state_dict[key] = state_dict[key].contiguous(memory_format=torch.channels_last)
        super()._load_from_state_dict(state_dict, prefix, *args)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = ConvWeight(inplanes, planes, 1)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = ConvWeight(planes, planes, 3, stride=stride, padding=1)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = ConvWeight(planes, planes * 4, 1)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.stride = stride
        if downsample:
            self.downsample = nn.Sequential(ConvWeight(inplanes, planes * 4, 1, stride=stride), FrozenBatchNorm2d(planes * 4))
        else:
            self.downsample = None


RESNET_BLOCKS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


class ResNetBody(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.conv1 = ConvWeight(3, 64, 7, stride=2, padding=3)
        self.bn1 = FrozenBatchNorm2d(64)
        inplanes = 64
        for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), blocks), start=1):
            layer = []
            for bi in range(nb):
                stride = 2 if (bi == 0 and li > 1) else 1
                layer.append(Bottleneck(inplanes, planes, stride, downsample=(bi == 0)))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*layer))
        self.blocks = blocks


class BackboneBase(nn.Module):
    """Reference BackboneBase (backbone.py:61-80): freezes everything outside layer2-4 (all of it when
    train_backbone is False) and returns `layer4` only, or layer1..4 when return_interm_layers."""

    def __init__(self, body, train_backbone, num_channels, return_interm_layers):
        super().__init__()
        for name, parameter in body.named_parameters():
            if not train_backbone or ("layer2" not in name and "layer3" not in name and "layer4" not in name):
                parameter.requires_grad_(False)
        self.body = body
        self.num_channels = num_channels
        self.return_interm_layers = return_interm_layers
        self._cache = {}

    # ---- native path ----------------------------------------------------------------------------
    def _transforms(self):
        """bf16 compute weights: KRSC, BN scale folded in; the stem additionally padded to 8 channels.  Rebuilt only when a
        FrozenBatchNorm2d dropped its cached scale/shift (load / device move)."""
        cached = self.__dict__.get("_tr_cache")
        if cached is not None and cached[0] == (_BN_EPOCH[0], id(self)):   # id: a deepcopy must fold its own BN buffers
            return cached[1]
        tr = {}
        for mod_name, m in self.body.named_modules():
            if not isinstance(m, ConvWeight):
                continue
            bn = self._bn_of(mod_name)
            scale = bn.scale_shift()[0]

            def make(w, scale=scale, stem=(mod_name == "conv1")):
                wf = w.detach() * scale.view(-1, 1, 1, 1)
                if stem:
                    wf = torch.nn.functional.pad(wf, (0, 0, 0, 0, 0, 8 - wf.shape[1]))
                return wf.to(BF16).contiguous(memory_format=torch.channels_last)

            if mod_name != "conv1":  # same physical layout as the master: the optimizer tail may rewrite it
                make.elementwise, make.row_scale = True, scale
            tr[mod_name + ".weight"] = make
        self.__dict__["_tr_cache"] = ((_BN_EPOCH[0], id(self)), tr)
        return tr

    def _bn_of(self, conv_name):
        parts = conv_name.split(".")
        if parts[-1].startswith("conv"):
            parts[-1] = "bn" + parts[-1][4:]
        else:  # downsample.0 -> downsample.1
            parts[-1] = "1"
        return self.body.get_submodule(".".join(parts))

    def _program(self, levels, first=1, last=4, stem=True):
        """Forward program of residual stages first .. last (with the stem in front when `stem`): prog(tape, ps, x) with x = the image
        batch (stem) or the previous stage's NHWC bf16 output; ps holds the parameters under their names inside `body`."""
        body = self.body
        bn_cache = {}

        def bn(name):
            if name not in bn_cache:
                bn_cache[name] = body.get_submodule(name).scale_shift()
            return bn_cache[name]

        def prog(tape, ps, images):
            x = images.data
            if stem:
                N, C, H, W = x.shape
                s0, t0 = bn("bn1")
                w1 = engine.krsc(ps["conv1.weight"].w)
                if FUSED_STEM and tuple(w1.shape) == (64, 7, 7, 8) and x.dtype == torch.float32:
                    # conv1 + bn1 + relu + maxpool in one launch, straight from the fp32 NCHW batch (csrc/stem.hip): the stem is frozen
                    # (backbone.py:64-66 of the reference), so its 105 MB convolution output is not needed by anything
                    CH, CW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                    pooled = torch.empty(N, (CH - 1) // 2 + 1, (CW - 1) // 2 + 1, 64, dtype=BF16, device=x.device)
                    k.stem_fwd(x.contiguous(), w1, t0, pooled)
                else:
                    xin = torch.empty(N, H, W, 8, dtype=BF16, device=x.device)
                    k.pack_image(x.contiguous(), xin)
                    y = ops.conv2d(xin, w1, stride=2, pad=3, shift=t0, act=k.ACT_RELU, cin_real=C)
                    OH, OW = (y.shape[1] + 2 - 3) // 2 + 1, (y.shape[2] + 2 - 3) // 2 + 1
                    pooled = torch.empty(N, OH, OW, 64, dtype=BF16, device=x.device)
                    k.maxpool3x3s2(y, pooled)
                    del y
                cur = engine.Var(pooled, needs_grad=False)
            else:
                cur = images          # the previous stage's output: its gradient (w.r.t. the pre-ReLU sum, masked by the first block here) flows back
            outs = []
            for li in range(first, last + 1):
                layer = getattr(body, f"layer{li}")
                for bi, blk in enumerate(layer):
                    pre = f"layer{li}.{bi}."
                    Wd = {"conv1": ps[pre + "conv1.weight"], "conv2": ps[pre + "conv2.weight"], "conv3": ps[pre + "conv3.weight"]}
                    bnd = {"bn1": bn(pre + "bn1"), "bn2": bn(pre + "bn2"), "bn3": bn(pre + "bn3")}
                    has_down = blk.downsample is not None
                    if has_down:
                        Wd["down"] = ps[pre + "downsample.0.weight"]
                        bnd["down"] = bn(pre + "downsample.1")
                    train = Wd["conv1"].g is not None
                    nxt = engine.bottleneck(tape, cur, Wd, bnd, blk.stride, has_down, train)
                    nxt.needs_grad = train
                    cur = nxt
                if li in levels:
                    outs.append(cur)      # needs_grad = this stage trains: a consumer (the mask head's adapter on the frozen layer1 output) does not compute a gradient nobody takes
            return outs, None

        return prog

    @staticmethod
    def _with_output_masks(prog, premasked):
        """Wrap a stage program: a gradient fed back into output i is taken w.r.t. the post-ReLU values and masked by (out > 0) here,
        unless i is listed in `premasked` (the consumer's data-gradient epilogue already did it)."""
        def wrapped(tape, ps, img):
            outs, extra = prog(tape, ps, img)
            finals = []
            for oi, o in enumerate(outs):
                f = engine.Var(o.data, needs_grad=o.needs_grad)

                def bwd(o=o, f=f, oi=oi):
                    g = f.take_grad()
                    if g is None or not o.needs_grad:
                        return
                    # blocks expect gradients w.r.t. their pre-ReLU sum: mask by (out > 0)
                    gm = g if oi in premasked else torch.where(o.data > 0, g, torch.zeros_like(g))
                    engine.accumulate(o, gm)

                tape.record(bwd)
                finals.append(f)
            return finals, extra
        return wrapped

    STAGES = ((1, 2, True), (3, 3, False), (4, 4, False))     # stem + layer1 + layer2 | layer3 | layer4: one backward program (and one flat gradient buffer) each

    def forward_native(self, images, levels=(4,), premasked=(), stage_cuts=None):
        """images: fp32 NCHW on the device -> tuple of NHWC bf16 feature maps (post-ReLU) for `levels`.
        A gradient fed back into an output is taken w.r.t. the post-ReLU values and masked by (out > 0)
        here, unless its index is listed in `premasked` (the consumer's dgrad epilogue already did it:
        input_proj, toist_amd/mdetr.py).

        stage_cuts (a list, data-parallel jobs): the body runs as THREE programs (STAGES) instead of one and the autograd graph is
        cut between them -- (stage output, detached leaf fed to the next stage) pairs are appended to the list -- so that the gradient
        all-reduce of layer4 / layer3 can travel underneath the backward pass of the stages below (toist_amd.parallel.backward_cut);
        only for levels == (4,) (the detection model)."""
        if stage_cuts is not None and tuple(levels) == (4,):
            return self._forward_staged(images, premasked, stage_cuts)
        named = engine.named_cache(self, "body", lambda: OrderedDict(self.body.named_parameters()))
        wrapped = self._with_output_masks(self._program(levels), premasked)
        return functions.run_program(wrapped, named, [images], cache=self._cache, training=self.training, transforms=self._transforms(),
                                     group_wgrads=True, store_once=lambda n, t: t.dim() == 4)     # every conv weight: one weight-gradient GEMM each

    def _forward_staged(self, images, premasked, stage_cuts):
        tr_all = self._transforms()
        caches = self.__dict__.setdefault("_stage_caches", [{} for _ in self.STAGES])
        x = images
        for si, (first, last, stem) in enumerate(self.STAGES):
            def build(first=first, last=last, stem=stem):
                keep = tuple(f"layer{li}." for li in range(first, last + 1)) + (("conv1.",) if stem else ())
                return OrderedDict((n, p) for n, p in self.body.named_parameters() if n.startswith(keep))
            named = engine.named_cache(self, f"body.stage{si}", build)
            tr = {n: f for n, f in tr_all.items() if n in named}
            is_last = si == len(self.STAGES) - 1
            # an inner stage hands its last block's output on as it is: the next stage's first block returns the gradient already masked
            prog = self._program((last,), first, last, stem)
            wrapped = self._with_output_masks(prog, premasked if is_last else (0,))
            (y,) = functions.run_program(wrapped, named, [x], cache=caches[si], training=self.training, transforms=tr, group_wgrads=True,
                                         store_once=lambda n, t: t.dim() == 4)
            if not is_last and y.requires_grad:
                leaf = y.detach().requires_grad_(True)
                stage_cuts.append((f"backbone.layer{last}", y, leaf))      # named by the stage that produced it (the cut sits at layer `last`'s output)
                y = leaf
            x = y
        return (x,)

    # ---- reference-compatible API ------------------------------------------------------------------
    def forward(self, tensor_list: NestedTensor):
        levels = (1, 2, 3, 4) if self.return_interm_layers else (4,)
        feats = self.forward_native(tensor_list.tensors, levels)
        out = OrderedDict()
        names = [str(i) for i in range(len(levels))] if self.return_interm_layers else ["0"]
        for name, f in zip(names, feats):
            x = f.permute(0, 3, 1, 2).float()
            mask = nearest_mask(tensor_list.mask, x.shape[-2:])
            out[name] = NestedTensor(x, mask)
        return out


_NEAREST_IDX = {}


def nearest_mask(mask, hw):
    """F.interpolate(mask[None].float(), size=hw).bool()[0] (backbone.py:78) as ONE index gather; the source rows / columns of a
    (size, device) pair are computed once on the host."""
    H, W = mask.shape[-2:]
    h, w = hw
    key = (H, W, h, w, str(mask.device))
    idx = _NEAREST_IDX.get(key)
    if idx is None:
        iy = torch.div(torch.arange(h) * H, h, rounding_mode="floor")
        ix = torch.div(torch.arange(w) * W, w, rounding_mode="floor")
        idx = _NEAREST_IDX[key] = ((iy[:, None] * W + ix[None, :]).reshape(-1).to(mask.device),)
    return mask.reshape(mask.shape[0], H * W).index_select(1, idx[0]).view(mask.shape[0], h, w)


class Backbone(BackboneBase):
    """ResNet backbone with frozen BatchNorm (reference backbone.py:83-91).  No pretrained download."""

    def __init__(self, name, train_backbone, return_interm_layers, dilation):
        if name not in RESNET_BLOCKS:
            raise ValueError(f"unsupported backbone {name} (resnet50 / resnet101)")
        if dilation:
            raise NotImplementedError("dilation (DC5) is not part of the hot path")
        super().__init__(ResNetBody(RESNET_BLOCKS[name]), train_backbone, 2048, return_interm_layers)


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, tensor_list):
        xs = self[0](tensor_list)
        out, pos = [], []
        for _, x in xs.items():
            out.append(x)
            pos.append(self[1](x).to(x.tensors.dtype))
        return out, pos


def build_backbone(args):
    position_embedding = build_position_encoding(args)
    train_backbone = args.lr_backbone > 0
    backbone = Backbone(args.backbone, train_backbone, args.masks, getattr(args, "dilation", False))
    model = Joiner(backbone, position_embedding)
    model.num_channels = backbone.num_channels
    return model
