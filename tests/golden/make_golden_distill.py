"""Golden vectors of the distillation path from the REAL reference (/root/reference/models/mdetr.py): the list branch
of SetCriterion (noun_/sth_ losses, loss_nsthl2, loss_softkd per layer) and ClusterCriterion.update_memory / forward.
Runs only in the build container (CPU): `.cuda()` and the process-group queries of ClusterCriterion.__init__ are
neutralised here, the arithmetic is the reference's.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_distill.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import formula  # noqa: E402
from transformers import RobertaModel  # noqa: E402,F401  (before the stubs)

STUBS = "/tmp/toist_ref_stubs"
for rel, text in {"IPython/__init__.py": "def embed(*a, **k):\n    pass\n", "torchvision/__init__.py": "from . import ops, models\n",
                  "torchvision/ops/__init__.py": "from . import boxes\n",
                  "torchvision/ops/boxes.py": "def box_area(b):\n    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])\n",
                  "torchvision/models/__init__.py": "from . import _utils\n", "torchvision/models/_utils.py": "class IntermediateLayerGetter:\n    pass\n",
                  "timm/__init__.py": "from . import models\n", "timm/models/__init__.py": "def create_model(*a, **k):\n    raise RuntimeError('stub')\n"}.items():
    path = os.path.join(STUBS, rel)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write(text)
sys.dont_write_bytecode = True
sys.path[:0] = [STUBS, "/root/reference"]
from models.matcher import HungarianMatcher  # noqa: E402
from models.mdetr import ClusterCriterion, SetCriterion  # noqa: E402

B, Q, K, LT, D, LAYERS = 2, 12, 256, 10, 16, 3
SPANS = {"noun": [[[(0, 7)], [(8, 11), (16, 19)]], [[(3, 10)]]], "sth": [[[(4, 7)], [(12, 19)]], [[(0, 3)]]]}   # per image, per box: char spans
T = [2, 1]


def side(tag):
    def layer(l):
        return {"pred_logits": formula.tensor(f"dst.{tag}.logits{l}", (B, Q, K), 4.0), "pred_boxes": formula.tensor(f"dst.{tag}.boxes{l}", (B, Q, 4), 0.3, 0.5),
                "proj_queries": torch.zeros(B, Q, 4), "tokenized": formula.FakeTokenized(LT)}
    out = layer(LAYERS - 1)
    out["aux_outputs"] = [layer(l) for l in range(LAYERS - 1)]
    targets, pms = [], []
    for i in range(B):
        pm = torch.zeros(T[i], K)
        pm[:, 1 + i:4 + i] = 1.0 / 3
        targets.append({"boxes": formula.tensor(f"dst.{tag}.tbox{i}", (T[i], 4), 0.25, 0.5), "labels": torch.ones(T[i], dtype=torch.int64),
                        "noun_tokens_positive": SPANS[tag][i], "dataset_name": f"task_{3 + 2 * i}_train.json"})
        pms.append(pm)
    mc = {"text_memory": formula.tensor(f"dst.{tag}.text", (LT, B, D), 2.0), "tokenized": formula.FakeTokenized(LT)}
    return out, targets, torch.cat(pms), mc


def criterion_pair(out):
    args = types.SimpleNamespace(num_queries=Q, nsthl2_loss=True, softkd_loss=True)
    crit = SetCriterion(args, 255, matcher=HungarianMatcher(1, 5, 2), eos_coef=0.1, losses=["labels", "boxes", "cardinality", "nsthl2", "softkd"], temperature=0.07, contrastive_hdim=64)
    (on, tn, pn, mn), (os_, ts, ps, ms) = side("noun"), side("sth")
    losses = crit([mn, ms], [on, os_], [tn, ts], [pn, ps], None)
    for k, v in losses.items():
        out["pair." + k] = v.detach()
    print("pair keys", len(losses))


def cluster(out):
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.distributed.get_world_size = lambda *a, **k: 1
    torch.distributed.all_gather = lambda lst, t, *a, **k: lst[0].copy_(t)
    MEM, HW = 24, 6
    args = types.SimpleNamespace(train_batch_size=B, fifo_memory=False)
    cc = ClusterCriterion(feature_dim=D, memory_size=MEM, cluster_num=3, task_count=14, args=args)
    cc.feature_bank.copy_(formula.tensor("dst.bank", (14, MEM, D), 2.0))
    cc.cluster_centers.copy_(formula.tensor("dst.centers", (14, 3, D), 2.0))
    cc.full_label.fill_(1)
    cc.update_count.fill_(100)
    _, tn, _, mn = side("noun")
    mn["img_memory"] = formula.tensor("dst.noun.img", (HW + LT, B, D), 1.5)
    mn["text_memory"] = mn["img_memory"][-LT:]
    mc = cc.update_memory(mn, tn, ["a noun caption"] * B)
    out["cl.noun.img_memory_mod"], out["cl.bank_after_update"], out["cl.centers_after_update"] = mc["img_memory_mod"].clone(), cc.feature_bank.clone(), cc.cluster_centers.clone()
    _, ts, _, ms = side("sth")
    ms["img_memory"] = formula.tensor("dst.sth.img", (HW + LT, B, D), 1.5)
    ms["text_memory"] = ms["img_memory"][-LT:]
    captions = ["put something on it", "use something"]
    mc2, loss = cc(ms, ts, captions)
    out["cl.sth.img_memory_mod"], out["cl.centers_after_forward"] = mc2["img_memory_mod"].clone(), cc.cluster_centers.clone()
    out["cl.loss_cluster_feature"], out["cl.loss_cluster_choice"] = loss["loss_cluster_feature"].detach(), loss["loss_cluster_choice"].detach()


def main():
    out = {}
    criterion_pair(out)
    cluster(out)
    np.savez_compressed(os.path.join(HERE, "distill.npz"), **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    print("wrote distill.npz", len(out), {k: float(np.asarray(v)) for k, v in out.items() if np.asarray(v).ndim == 0 and ("softkd" in k or "nsthl2" in k or "cluster" in k)})


if __name__ == "__main__":
    main()
