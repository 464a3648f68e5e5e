"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol of
include/toist_hip.h, struct layouts agree with the header, the module tree is state_dict compatible
with the reference, and the product path refuses to run without the HIP device (no CPU fallback)."""
import ctypes
import json
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from toist_amd import _lib
    header = open(os.path.join(ROOT, "include", "toist_hip.h")).read()
    declared = set(re.findall(r"\b(toist_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    handle = _lib.lib()
    missing = [n for n in sorted(declared) if not hasattr(handle, n)]
    assert not missing, f"symbols declared in toist_hip.h but not exported: {missing}"
    assert set(_lib.exported_symbols()) == declared, (sorted(set(_lib.exported_symbols()) ^ declared))
    assert handle.toist_version() == 1


def test_struct_layouts_match_header(tmp_path):
    from toist_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include "toist_hip.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(toist_operand), '
                   'sizeof(toist_epilogue), sizeof(toist_gemm), offsetof(toist_gemm, epi), offsetof(toist_gemm, a_colsum));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(_lib.Operand), ctypes.sizeof(_lib.Epilogue), ctypes.sizeof(_lib.Gemm), _lib.Gemm.epi.offset,
                   _lib.Gemm.a_colsum.offset]
    # the optimizer-tail table rows (built with numpy in toist_amd/optim.py) and the split-K reduction descriptor
    src.write_text('#include "toist_hip.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(toist_opt_tensor), '
                   'offsetof(toist_opt_tensor, numel), offsetof(toist_opt_tensor, group), sizeof(toist_opt_state), offsetof(toist_opt_state, step), '
                   'sizeof(toist_reduce_desc));return 0;}\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    from toist_amd import optim
    dt = optim._TENSOR_DT
    assert got == [dt.itemsize, dt.fields["numel"][1], dt.fields["group"][1], 32, 16, ctypes.sizeof(_lib.ReduceDesc)]
    # the grouped-launch table rows (six int64 per problem, toist_amd/kernels.py: group_table) and the descriptor's group pointer
    src.write_text('#include "toist_hip.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(toist_group), '
                   'offsetof(toist_group, c_off), offsetof(toist_group, colsum_off), offsetof(toist_gemm, group));return 0;}\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [48, 16, 32, _lib.Gemm.group.offset]


def test_bad_arguments_return_error_codes_without_a_gpu():
    from toist_amd import _lib
    handle = _lib.lib()
    d = _lib.Gemm()  # all zero: bad shape
    rc = handle.toist_gemm_bf16(ctypes.byref(d), None)
    assert rc == -1 and "bad shape" in _lib.last_error()


def test_no_cpu_fallback():
    from toist_amd import kernels, ops
    x = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(x, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        kernels.layernorm_fwd(x, torch.ones(8), torch.zeros(8), 1e-5, x)


@pytest.fixture(scope="module")
def model():
    import toist_amd
    from toist_amd import harness
    torch.manual_seed(0)
    return toist_amd.build_model(harness.default_args(device="cpu"))


def test_state_dict_is_reference_compatible(model):
    m, criterion, cluster, weight_dict = model
    assert cluster is None
    with open(os.path.join(ROOT, "tests", "golden", "reference_state_dict_shapes.json")) as f:
        ref = json.load(f)
    sd = m.state_dict()
    for k, shape in ref.items():
        if k.endswith("position_ids") or k.startswith("contrastive_align"):
            continue  # buffer of newer HF versions / built only with --contrastive_align_loss
        assert k in sd, f"missing reference key {k}"
        assert list(sd[k].shape) == shape, (k, list(sd[k].shape), shape)
    # torchvision resnet101 naming under backbone.0.body (public architecture; reference backbone.py:87-89)
    for k, shape in {"backbone.0.body.conv1.weight": [64, 3, 7, 7], "backbone.0.body.bn1.running_var": [64],
                     "backbone.0.body.layer1.0.downsample.0.weight": [256, 64, 1, 1], "backbone.0.body.layer3.22.conv2.weight": [256, 256, 3, 3],
                     "backbone.0.body.layer4.2.bn3.bias": [2048]}.items():
        assert list(sd[k].shape) == shape
    conv_params = sum(v.numel() for k, v in sd.items() if k.startswith("backbone.0.body") and ("conv" in k or "downsample.0" in k))
    assert conv_params == 42_394_816  # torchvision resnet101: 44,549,160 total - 2,049,000 (fc) - 105,344 (BN affine)
    assert not m.backbone[0].body.layer1[0].conv1.weight.requires_grad and m.backbone[0].body.layer2[0].conv1.weight.requires_grad
    assert sorted(weight_dict)[:3] == ["loss_bbox", "loss_bbox_0", "loss_bbox_1"] and len(weight_dict) == 18
    # main.py:351-366 builds LR groups from these substrings
    names = [n for n, _ in m.named_parameters()]
    assert any("backbone" in n for n in names) and any("text_encoder" in n for n in names)
    # checkpoints with BatchNorm bookkeeping load (backbone.py:40-42 drops num_batches_tracked)
    extra = dict(sd)
    extra["backbone.0.body.bn1.num_batches_tracked"] = torch.tensor(5)
    missing, unexpected = m.load_state_dict(extra, strict=False)
    assert not missing and not unexpected
    import copy
    copy.deepcopy(m)  # EMA copy in main.py:333


def test_nested_tensor_and_targets():
    from toist_amd.misc import NestedTensor, targets_to
    a, b = torch.ones(3, 5, 7), torch.ones(3, 6, 4)
    nt = NestedTensor.from_tensor_list([a, b])
    assert nt.tensors.shape == (2, 3, 6, 7) and nt.mask.dtype == torch.bool
    assert not nt.mask[0, :5, :7].any() and nt.mask[0, 5].all() and nt.mask[1, :, 4:].all()
    nt = NestedTensor.from_tensor_list([a, b], do_round=True)
    assert nt.tensors.shape == (2, 3, 128, 128)
    t = targets_to([{"boxes": torch.zeros(1, 4), "caption": "x", "tokens_positive": [[(0, 1)]]}], "cpu")
    assert "caption" not in t[0] and t[0]["tokens_positive"] == [[(0, 1)]]


def test_nearest_mask_matches_interpolate():
    from toist_amd.backbone import nearest_mask
    g = torch.Generator().manual_seed(0)
    m = torch.rand(3, 97, 131, generator=g) > 0.5
    for hw in [(4, 5), (13, 17), (25, 33), (97, 131)]:
        ref = torch.nn.functional.interpolate(m[None].float(), size=hw).bool()[0]
        assert torch.equal(nearest_mask(m, hw), ref)


def test_matcher_module_contract_on_cpu_raises():
    from toist_amd.matcher import HungarianMatcher
    m = HungarianMatcher(1.0, 5.0, 2.0)
    out = {"pred_logits": torch.zeros(1, 4, 8), "pred_boxes": torch.rand(1, 4, 4)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(out, [{"boxes": torch.rand(2, 4)}], torch.rand(2, 8))
    with pytest.raises(AssertionError):
        HungarianMatcher(0, 0, 0)


def test_segmentation_wrapper_state_dict():
    """DETRsegm keeps the reference names (segmentation.py:17-38; main.py:481,499 rely on `.detr` and
    'mask_head.adapter3.bias') and freezes DETR before creating the mask branch when asked to."""
    import toist_amd
    from toist_amd import harness
    torch.manual_seed(0)
    args = harness.default_args(device="cpu", masks=True, mask_model="smallconv", frozen_weights="some.pth")
    model, criterion, _, wd = toist_amd.build_model(args)
    with open(os.path.join(ROOT, "tests", "golden", "reference_segm_state_dict_shapes.json")) as f:
        ref = json.load(f)
    sd = model.state_dict()
    for k, shape in ref.items():
        assert k in sd and list(sd[k].shape) == shape, k
    assert "detr.class_embed.weight" in sd and hasattr(model, "detr")
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert trainable == sum(p.numel() for n, p in model.named_parameters() if not n.startswith("detr."))  # only the mask branch
    assert 1_300_000 < trainable < 1_400_000  # SURVEY: 1.33 M parameters
    assert "masks" in criterion.losses and wd["loss_mask"] == 1.0 and wd["loss_dice"] == 1.0


def test_compute_copy_registry_is_keyed_by_tensor_identity():
    """The caching allocator reuses a freed parameter's address for the next model: a registry hit must be the same tensor
    object, not merely the same address (a stale row in the optimizer's pointer table writes past the smaller copy)."""
    import gc
    import torch
    from toist_amd import engine
    cache = {}
    t = torch.nn.Parameter(torch.randn(8, 4))
    w = engine.compute_copy(t, engine._cast_bf16, cache, "w")
    assert engine.copy_of(t) is not None and engine.copy_of(t).w is w
    alias = t.detach()[:2]                                    # same address, another tensor (and another size)
    assert alias.data_ptr() == t.data_ptr() and engine.copy_of(alias) is None
    ptr = t.data_ptr()
    del t, alias
    gc.collect()
    engine.prune_copies()
    assert ptr not in engine.COPIES


def test_store_once_row_ranges_are_tracked_on_the_root_view():
    """ADVICE r2 (engine.py ParamView): the store-once state of a packed parameter is a set of row ranges on the ROOT view -- a second slice
    of the same rows sees the first launch (it must accumulate, not store), marking one slice does not mark its siblings, and the rows
    nothing wrote are reported for zeroing."""
    import torch
    from toist_amd.engine import ParamView
    g = torch.zeros(12, 4)
    root = ParamView(None, g, fresh=True)
    q1, v1 = root.rows(0, 8), root.rows(8, 12)
    assert not root.written and not q1.written and not v1.written
    q1.mark_written()
    assert q1.written and not v1.written and not root.written           # a sibling slice and the whole slot stay unwritten
    assert root.rows(0, 8).written and root.rows(2, 6).written          # a NEW view of the stored rows accumulates
    assert not root.rows(4, 10).written
    assert root.unwritten_rows() == [(8, 12)]
    v1.mark_written()
    assert root.written and root.unwritten_rows() == []
    root.written = False
    assert not q1.written and root.unwritten_rows() == [(0, 12)]
    root.mark_written()
    assert q1.written and root.rows(3, 4).rows(0, 1).written and root.unwritten_rows() == []
    nested = ParamView(None, torch.zeros(12, 4), fresh=True)
    nested.rows(4, 12).rows(2, 4).mark_written()                         # rows 6..8 of the root
    assert nested.unwritten_rows() == [(0, 6), (8, 12)]


def test_memory_cache_materialises_lazy_entries_on_read():
    """transformer.MemoryCache: the fp32 API copies of memory_cache are thunks until somebody reads them; reads, membership, iteration and
    assignment behave like the plain dict of the reference (transformer.py:146-157)."""
    from toist_amd.transformer import MemoryCache
    calls = []

    def thunk(name, value):
        def f():
            calls.append(name)
            return value
        return f

    mc = MemoryCache({"mask": 1}, lazy={"img_memory": thunk("img", torch.ones(2)), "pos_embed": thunk("pos", torch.zeros(2))})
    mc._lazy["text_memory"] = lambda: mc["img_memory"][-1:]
    assert "img_memory" in mc and mc.is_lazy("img_memory") and len(mc) == 4 and calls == []
    assert mc["mask"] == 1 and calls == []
    assert torch.equal(mc["text_memory"], torch.ones(1)) and calls == ["img"] and not mc.is_lazy("img_memory")
    assert mc["img_memory"] is mc["img_memory"] and calls == ["img"]              # materialised once
    mc["pos_embed"] = "replaced"                                                  # assignment drops the thunk
    assert mc.get("pos_embed") == "replaced" and "pos" not in calls
    mc2 = MemoryCache({}, lazy={"a": thunk("a", 5)})
    assert dict(mc2.items()) == {"a": 5} and calls[-1] == "a" and list(mc2) == ["a"]
    assert mc2.get("missing", 7) == 7 and mc2.pop("a") == 5 and "a" not in mc2


def test_distill_tables_follow_in_place_refills():
    """distill._cached_table keys a batch's caption-driven tables on the identity AND the in-place version of its token tensor: a loop that refills a
    static input_ids buffer in place (a captured step's feed) must not be served the previous batch's tables."""
    import torch
    from toist_amd import distill
    ids = torch.zeros(2, 4, dtype=torch.long)
    calls = []
    make = lambda: calls.append(1) or len(calls)
    assert distill._cached_table("t_test", (ids,), (), make) == 1
    assert distill._cached_table("t_test", (ids,), (), make) == 1          # same contents: cached
    ids.copy_(torch.ones(2, 4, dtype=torch.long))                          # refilled in place
    assert distill._cached_table("t_test", (ids,), (), make) == 2


@pytest.mark.parametrize("order", ["alone", "extra_before", "extra_after"])
def test_matched_rows_sentinel_survives_autograd_accumulation(order):
    """The hand-off of toist_amd/segmentation.py (`_MaskLossFn.backward` -> `matched_rows_of`), driven on the CPU with the same objects: the
    producer returns the zero sentinel (one element, expanded, kept alive by the sink); the consumer must see THAT tensor iff nothing else
    consumed pred_masks, and a correct dense sum otherwise -- in particular when the other consumer was created first, the order in which
    autograd's input buffer accumulates onto the first-arrived gradient (in place, if it owns it alone: VERDICT r5 weak #1)."""
    from toist_amd import segmentation as sg
    sink = sg._MatchedRows()
    seen = {}

    class Producer(torch.autograd.Function):          # stands for the mask program: receives the gradient of pred_masks
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            zero = sink.sentinel
            seen["is_sentinel"] = bool(zero is not None and g.data_ptr() == zero.data_ptr() and not any(g.stride()) and zero._version == sink.version)
            seen["g"] = g.clone()
            return g.contiguous()

    class Loss(torch.autograd.Function):              # stands for _MaskLossFn
        @staticmethod
        def forward(ctx, pred):
            ctx.shape = pred.shape
            return pred.sum() * 0.0

        @staticmethod
        def backward(ctx, g):
            zero = sg._zero_sentinel(torch.device("cpu"))
            sink.grad, sink.sentinel, sink.version = torch.ones(1), zero, zero._version
            return zero.expand(ctx.shape)

    x = torch.randn(3, 4, 5, requires_grad=True)
    masks = Producer.apply(x)
    extra = (masks * masks).mean() if order == "extra_before" else None
    loss = Loss.apply(masks.view(12, 5))
    if order == "extra_after":
        extra = (masks * masks).mean()
    (loss if extra is None else loss + extra).backward()
    assert seen["is_sentinel"] == (order == "alone")
    want = torch.zeros_like(x) if order == "alone" else 2 * x.detach() / x.numel()
    assert torch.allclose(seen["g"], want, atol=1e-7)
    assert float(sg._zero_sentinel(torch.device("cpu"))) == 0.0 and sg._zero_sentinel(torch.device("cpu"))._version == sink.version


def test_box_ops_against_the_reference_fixture():
    """toist_amd/box_ops.py (API-edge helpers, rewritten in round 6 as one broadcast over the four corners) against values the REAL reference's
    util/box_ops.py produced (tests/golden/box_ops.npz, make_golden.py)."""
    import numpy as np
    from toist_amd import box_ops
    d = np.load(os.path.join(ROOT, "tests", "golden", "box_ops.npz"))
    a, b = torch.from_numpy(d["a"]), torch.from_numpy(d["b"])
    axy, bxy = box_ops.box_cxcywh_to_xyxy(a), box_ops.box_cxcywh_to_xyxy(b)
    assert np.allclose(axy.numpy(), d["axy"], atol=1e-7)
    assert np.allclose(box_ops.box_xyxy_to_cxcywh(axy).numpy(), d["back"], atol=1e-6)
    iou, union = box_ops.box_iou(axy, bxy)
    assert np.allclose(iou.numpy(), d["iou"], atol=1e-6) and np.allclose(union.numpy(), d["union"], atol=1e-6)
    assert np.allclose(box_ops.generalized_box_iou(axy, bxy).numpy(), d["giou"], atol=1e-6)
    bad = axy.clone()
    bad[0, 2] = bad[0, 0] - 1.0
    with pytest.raises(AssertionError):
        box_ops.generalized_box_iou(bad, bxy)


def test_postprocess_forwards_pred_isfinal():
    """PostProcess keeps the reference's `pred_isfinal` -> `scores_refexp` branch (/root/reference/models/postprocessors.py:49-54), and
    its scores / boxes equal the reference fixture's (postprocess.npz) when the key is absent."""
    import numpy as np
    from toist_amd.postprocessors import PostProcess
    d = np.load(os.path.join(ROOT, "tests", "golden", "postprocess.npz"))
    out = {"pred_logits": torch.from_numpy(d["logits"]), "pred_boxes": torch.from_numpy(d["boxes"])}
    sizes = torch.from_numpy(d["sizes"])
    res = PostProcess()(out, sizes)
    for i, r in enumerate(res):
        assert set(r) == {"scores", "labels", "boxes"}
        assert np.allclose(r["scores"].numpy(), d["scores"][i], atol=1e-6) and np.allclose(r["boxes"].numpy(), d["out_boxes"][i], atol=1e-4)
    fin = torch.linspace(-3, 3, d["logits"].shape[0] * d["logits"].shape[1]).view(d["logits"].shape[0], -1, 1)
    res2 = PostProcess()(dict(out, pred_isfinal=fin), sizes)
    for i, r in enumerate(res2):
        assert torch.allclose(r["scores_refexp"], res[i]["scores"] * torch.sigmoid(fin[i, :, 0]), atol=1e-7)
