"""Golden vectors for PostProcessSegm from the REAL reference (models/postprocessors.py:59-109): both of its branches
(all sizes equal -> one batched resize; ragged -> per-image crop + resize).  Masks are stored bit-packed.
Run in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_postsegm.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import formula  # noqa: E402
import make_golden as mg  # noqa: E402  (creates the import stubs, puts /root/reference on sys.path)

from models.postprocessors import PostProcessSegm  # noqa: E402


def main():
    B, Q, h, w = 2, 5, 24, 32
    pred = formula.tensor("ps.masks", (B, Q, 1, h, w), 8.0)
    rec = {"pred_masks": pred}
    cases = {"equal": (torch.tensor([[96, 128], [96, 128]]), torch.tensor([[75, 100], [75, 100]])),
             "ragged": (torch.tensor([[96, 128], [80, 112]]), torch.tensor([[120, 161], [60, 84]]))}
    for name, (max_sizes, orig_sizes) in cases.items():
        res = PostProcessSegm()([{} for _ in range(B)], {"pred_masks": pred}, orig_sizes, max_sizes)
        rec[name + "_max"], rec[name + "_orig"] = max_sizes, orig_sizes
        for i, r in enumerate(res):
            m = r["masks"].numpy()
            assert m.dtype == np.bool_ and m.shape == (Q, 1, int(orig_sizes[i, 0]), int(orig_sizes[i, 1]))
            rec[f"{name}_bits{i}"] = np.packbits(m.reshape(-1))
    mg.save("postprocess_segm.npz", **rec)


if __name__ == "__main__":
    main()
