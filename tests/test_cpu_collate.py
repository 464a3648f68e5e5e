"""Data-side collate (toist_amd/misc.py) against vectors produced by the reference's collate_fn / collate_fn_plain
(tests/golden/make_golden_collate.py).  Exact: padding, masks and stacked positive maps are copies."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import formula  # noqa: E402

from toist_amd.misc import collate_fn, collate_fn_plain  # noqa: E402

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "collate.npz"))
SIZES, BOXES, WIDTHS = [(33, 40), (48, 21), (17, 64)], [2, 0, 3], [9, 5, 12]


def item(i, tag):
    h, w = SIZES[i]
    tgt = {"boxes": formula.tensor(f"collate.{tag}.box{i}", (BOXES[i], 4), 0.5, 0.5),
           "positive_map": formula.tensor(f"collate.{tag}.pm{i}", (BOXES[i], WIDTHS[i])) > 0, "dataset_name": f"task_{i + 1}_train.json"}
    return formula.tensor(f"collate.{tag}.img{i}", (3, h, w)), tgt


def test_collate_plain_matches_reference():
    batch = [([item(i, "p")[0]], [item(i, "p")[1]]) for i in range(3)]
    for do_round in (False, True):
        b = collate_fn_plain(do_round, batch)
        k = f"plain{int(do_round)}."
        assert np.array_equal(b["samples"].tensors.numpy(), Z[k + "tensors"]) and np.array_equal(b["samples"].mask.numpy(), Z[k + "mask"])
        assert np.array_equal(b["positive_map"].numpy(), Z[k + "positive_map"]) and b["positive_map"].dtype == torch.float32
        assert b["example_rel"] == Z[k + "example_rel"].tolist() and len(b["targets"]) == 3
    assert collate_fn_plain(True, batch)["samples"].tensors.shape[-2:] == (128, 128)


def test_collate_pairs_matches_reference():
    pairs = [((item(i, "n")[0], item((i + 1) % 3, "s")[0]), (item(i, "n")[1], item((i + 1) % 3, "s")[1])) for i in range(3)]
    b = collate_fn(False, pairs)
    for s, name in enumerate(("noun", "sth")):
        assert np.array_equal(b["samples"][s].tensors.numpy(), Z[f"pair.{name}.tensors"])
        assert np.array_equal(b["samples"][s].mask.numpy(), Z[f"pair.{name}.mask"])
        assert np.array_equal(b["positive_map"][s].numpy(), Z[f"pair.{name}.positive_map"])
        assert np.array_equal(b["targets"][s][-1]["boxes"].numpy(), Z[f"pair.{name}.boxes_last"])
    assert b["example_rel"] == Z["pair.example_rel"].tolist()
