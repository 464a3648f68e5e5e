"""toist_amd -- MI355X-native (gfx950) implementation of the TOIST/MDETR forward/backward hot path.

Host code is Python on PyTorch-ROCm (memory, streams, torch.distributed only); all compute runs in
hand-written HIP kernels behind the C ABI of include/toist_hip.h.  The public surface mirrors
/root/reference/models/__init__.py and friends: build_model(args), SetCriterion, HungarianMatcher,
PostProcess / PostProcessSegm, NestedTensor.
"""


def build_model(args):
    """models.build_model of the reference (models/__init__.py:7-8): -> (model, criterion,
    cluster_criterion, weight_dict)."""
    from .mdetr import build
    return build(args)


def __getattr__(name):
    import importlib
    table = {
        "SetCriterion": ("mdetr", "SetCriterion"), "weighted_total": ("mdetr", "weighted_total"), "MDETR": ("mdetr", "MDETR"), "HungarianMatcher": ("matcher", "HungarianMatcher"),
        "build_matcher": ("matcher", "build_matcher"), "PostProcess": ("postprocessors", "PostProcess"),
        "PostProcessSegm": ("postprocessors", "PostProcessSegm"), "build_postprocessors": ("postprocessors", "build_postprocessors"),
        "NestedTensor": ("misc", "NestedTensor"), "targets_to": ("misc", "targets_to"),
        "TDODCocoEvaluator": ("coco_eval", "TDODCocoEvaluator"), "CocoGroundTruth": ("coco_eval", "CocoGroundTruth"),
    }
    if name in table:
        mod, attr = table[name]
        return getattr(importlib.import_module("." + mod, __name__), attr)
    raise AttributeError(name)
