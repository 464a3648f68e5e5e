"""Golden vectors of the data-side collate: the REAL reference's collate_fn / collate_fn_plain
(/root/reference/util/misc.py:40-127) on closed-form ragged inputs.  Runs only in the build container.
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_collate.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import formula  # noqa: E402

STUBS = "/tmp/toist_ref_stubs"
for rel, text in {"IPython/__init__.py": "def embed(*a, **k):\n    pass\n", "torchvision/__init__.py": ""}.items():
    path = os.path.join(STUBS + "_collate", rel)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write(text)
sys.dont_write_bytecode = True
sys.path[:0] = [STUBS + "_collate", "/root/reference"]
from util.misc import collate_fn, collate_fn_plain  # noqa: E402

SIZES = [(33, 40), (48, 21), (17, 64)]
BOXES = [2, 0, 3]
WIDTHS = [9, 5, 12]


def item(i, tag):
    h, w = SIZES[i]
    img = formula.tensor(f"collate.{tag}.img{i}", (3, h, w))
    pm = formula.tensor(f"collate.{tag}.pm{i}", (BOXES[i], WIDTHS[i])) > 0
    tgt = {"boxes": formula.tensor(f"collate.{tag}.box{i}", (BOXES[i], 4), 0.5, 0.5), "positive_map": pm, "dataset_name": f"task_{i + 1}_train.json"}
    return img, tgt


def main():
    out = {}
    plain = [([item(i, "p")[0]], [item(i, "p")[1]]) for i in range(3)]
    for do_round in (False, True):
        b = collate_fn_plain(do_round, plain)
        k = f"plain{int(do_round)}."
        out[k + "tensors"], out[k + "mask"], out[k + "positive_map"] = b["samples"].tensors, b["samples"].mask, b["positive_map"]
        out[k + "example_rel"] = np.asarray(b["example_rel"])
    pairs = [((item(i, "n")[0], item((i + 1) % 3, "s")[0]), (item(i, "n")[1], item((i + 1) % 3, "s")[1])) for i in range(3)]
    b = collate_fn(False, pairs)
    for s, name in enumerate(("noun", "sth")):
        out[f"pair.{name}.tensors"], out[f"pair.{name}.mask"] = b["samples"][s].tensors, b["samples"][s].mask
        out[f"pair.{name}.positive_map"] = b["positive_map"][s]
        out[f"pair.{name}.boxes_last"] = b["targets"][s][-1]["boxes"]
    out["pair.example_rel"] = np.asarray(b["example_rel"])
    np.savez_compressed(os.path.join(HERE, "collate.npz"), **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    print("wrote collate.npz", sorted(out))


if __name__ == "__main__":
    main()
