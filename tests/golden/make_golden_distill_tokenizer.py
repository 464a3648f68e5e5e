"""Golden vectors of the distillation path's CAPTION handling from the REAL reference with a REAL Hugging Face fast tokenizer (tests/golden/tiny_tokenizer.py:
byte-level BPE, RoBERTa post-processing): the char-span -> token lookups of ClusterCriterion.update_memory / forward and SetCriterion.loss_nsthl2
(/root/reference/models/mdetr.py:112-141, 240-260, 684-711) acting on genuine BatchEncodings -- multi-token words, spans that start / end on a space (the
reference's fallbacks, which drop the batch index), captions of different lengths with padding.  make_golden_distill.py does the same with a synthetic
4-characters-per-token encoding.  Runs only in the build container (CPU).  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_distill_tokenizer.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import formula  # noqa: E402
import make_golden_distill as base  # noqa: E402  (import stubs + the reference's classes)
import tiny_tokenizer  # noqa: E402

B, Q, K, D, LAYERS = 3, 12, 256, 16, 2
CAPTIONS = {"noun": ["use the screwdriver to cut the paper up", "sit comfortably on the armchair", "dig a hole with the umbrella handle"],
            "sth": ["use something to cut the paper up", "sit comfortably on something", "dig a hole with something"]}
# per image, per box: character spans (token numbers from the BatchEncodings of tiny_tokenizer).
#  image 0  "use the screwdriver to cut the paper up": (8, 19) = "screwdriver" (tokens 3 .. 12); (7, 19) starts ON the space before it (first lookup None -> the
#           reference retries beg + 1 WITHOUT the batch index, i.e. in caption 0); (31, 37) = "paper " ends on a space (end - 1 None -> retries end - 2)
#  image 1  "sit comfortably on the armchair": (23, 31) = "armchair"; (3, 15) = " comfortably" starts on a space: the retry looks at character 4 of CAPTION 0
#           ("the", token 2) -- the reference's quirk; it happens to be the token of "comfortably" in caption 1 as well
#  image 2  "dig a hole with the umbrella handle": (20, 35) = two words, tokens 7 .. 18; (19, 28) starts on a space: the retry finds "to" of caption 0 (token 13),
#           so the box's span is the single token 13 instead of "umbrella" = 7 .. 13 (the quirk changes the feature)
#  pronoun side: "something" = (4, 13) / (19, 28) / (16, 25); (15, 25) starts on a space: character 16 of caption 0 is a space too -> retries beg + 2 = "cut"
SPANS = {"noun": [[[(8, 19)], [(7, 19), (31, 37)]], [[(3, 15), (23, 31)]], [[(20, 35)], [(19, 28)]]],
         "sth": [[[(4, 13)], [(4, 13)]], [[(19, 28)]], [[(16, 25)], [(15, 25)]]]}
T = [2, 1, 2]


def main():
    tok = tiny_tokenizer.build()
    enc = {tag: tok(CAPTIONS[tag], padding="longest", return_tensors="pt") for tag in ("noun", "sth")}
    LT = {tag: int(enc[tag]["input_ids"].shape[1]) for tag in enc}
    out = {"noun.input_ids": enc["noun"]["input_ids"], "sth.input_ids": enc["sth"]["input_ids"]}

    def side(tag):
        def layer(l):
            return {"pred_logits": formula.tensor(f"dtk.{tag}.logits{l}", (B, Q, K), 4.0), "pred_boxes": formula.tensor(f"dtk.{tag}.boxes{l}", (B, Q, 4), 0.3, 0.5),
                    "proj_queries": torch.zeros(B, Q, 4), "tokenized": enc[tag]}
        o = layer(LAYERS - 1)
        o["aux_outputs"] = [layer(l) for l in range(LAYERS - 1)]
        targets, pms = [], []
        for i in range(B):
            pm = torch.zeros(T[i], K)
            pm[:, 1 + i:4 + i] = 1.0 / 3
            targets.append({"boxes": formula.tensor(f"dtk.{tag}.tbox{i}", (T[i], 4), 0.25, 0.5), "labels": torch.ones(T[i], dtype=torch.int64),
                            "noun_tokens_positive": SPANS[tag][i], "dataset_name": f"task_{2 + 3 * i}_train.json"})
            pms.append(pm)
        mc = {"text_memory": formula.tensor(f"dtk.{tag}.text", (LT[tag], B, D), 2.0), "tokenized": enc[tag]}
        return o, targets, torch.cat(pms), mc

    args = types.SimpleNamespace(num_queries=Q, nsthl2_loss=True, softkd_loss=True)
    crit = base.SetCriterion(args, 255, matcher=base.HungarianMatcher(1, 5, 2), eos_coef=0.1, losses=["labels", "boxes", "cardinality", "nsthl2", "softkd"], temperature=0.07,
                             contrastive_hdim=64)
    (on, tn, pn, mn), (os_, ts, ps, ms) = side("noun"), side("sth")
    losses = crit([mn, ms], [on, os_], [tn, ts], [pn, ps], None)
    for k, v in losses.items():
        out["pair." + k] = v.detach()

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.distributed.get_world_size = lambda *a, **k: 1
    torch.distributed.all_gather = lambda lst, t, *a, **k: lst[0].copy_(t)
    MEM, HW = 24, 6
    cc = base.ClusterCriterion(feature_dim=D, memory_size=MEM, cluster_num=3, task_count=14, args=types.SimpleNamespace(train_batch_size=B, fifo_memory=False))
    cc.feature_bank.copy_(formula.tensor("dtk.bank", (14, MEM, D), 2.0))
    cc.cluster_centers.copy_(formula.tensor("dtk.centers", (14, 3, D), 2.0))
    cc.full_label.fill_(1)
    cc.update_count.fill_(100)
    _, tn, _, mn = side("noun")
    mn["img_memory"] = formula.tensor("dtk.noun.img", (HW + LT["noun"], B, D), 1.5)
    mn["text_memory"] = mn["img_memory"][-LT["noun"]:]
    mc = cc.update_memory(mn, tn, CAPTIONS["noun"])
    out["cl.noun.img_memory_mod"], out["cl.bank_after_update"], out["cl.centers_after_update"] = mc["img_memory_mod"].clone(), cc.feature_bank.clone(), cc.cluster_centers.clone()
    _, ts, _, ms = side("sth")
    ms["img_memory"] = formula.tensor("dtk.sth.img", (HW + LT["sth"], B, D), 1.5)
    ms["text_memory"] = ms["img_memory"][-LT["sth"]:]
    mc2, loss = cc(ms, ts, CAPTIONS["sth"])
    out["cl.sth.img_memory_mod"], out["cl.centers_after_forward"] = mc2["img_memory_mod"].clone(), cc.cluster_centers.clone()
    out["cl.loss_cluster_feature"] = loss["loss_cluster_feature"].detach()
    np.savez_compressed(os.path.join(HERE, "distill_tokenizer.npz"), **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    print("wrote distill_tokenizer.npz", len(out), "tokens", LT, {k: float(np.asarray(v)) for k, v in out.items() if np.asarray(v).ndim == 0 and ("nsthl2" in k or "cluster" in k)})
    print(enc["noun"]["input_ids"].tolist())


if __name__ == "__main__":
    main()
