"""Distillation oracle (oracle/distill_ref.py) against the vectors the REAL reference produced
(tests/golden/distill.npz): loss_softkd per layer, loss_nsthl2, the memory-bank update and the prototype
substitution.  CPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import formula  # noqa: E402

from oracle import distill_ref, matcher_ref  # noqa: E402

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "distill.npz"))
B, Q, K, LT, D, LAYERS = 2, 12, 256, 10, 16, 3
SPANS = {"noun": [[[(0, 7)], [(8, 11), (16, 19)]], [[(3, 10)]]], "sth": [[[(4, 7)], [(12, 19)]], [[(0, 3)]]]}
T = [2, 1]


def side(tag):
    layers = [(formula.tensor(f"dst.{tag}.logits{l}", (B, Q, K), 4.0), formula.tensor(f"dst.{tag}.boxes{l}", (B, Q, 4), 0.3, 0.5)) for l in range(LAYERS)]
    targets, pms = [], []
    for i in range(B):
        pm = torch.zeros(T[i], K)
        pm[:, 1 + i:4 + i] = 1.0 / 3
        targets.append({"boxes": formula.tensor(f"dst.{tag}.tbox{i}", (T[i], 4), 0.25, 0.5), "noun_tokens_positive": SPANS[tag][i],
                        "dataset_name": f"task_{3 + 2 * i}_train.json"})
        pms.append(pm)
    return layers, targets, torch.cat(pms), formula.tensor(f"dst.{tag}.text", (LT, B, D), 2.0)


def test_softkd_and_nsthl2_match_reference():
    (ln, tn, pn, xn), (ls, ts, ps, xs) = side("noun"), side("sth")
    tok = formula.FakeTokenized(LT)
    for l in range(LAYERS):
        idx_n = matcher_ref.hungarian_match(ln[l][0], ln[l][1], [t["boxes"] for t in tn], pn)
        idx_s = matcher_ref.hungarian_match(ls[l][0], ls[l][1], [t["boxes"] for t in ts], ps)
        got = distill_ref.loss_softkd(ln[l][0], ls[l][0], ln[l][1], ls[l][1], idx_n, idx_s)
        want = float(Z["pair.loss_softkd" + ("" if l == LAYERS - 1 else f"_{l}")])
        assert abs(float(got) - want) <= 1e-5 * abs(want) + 1e-8, (l, float(got), want)
    idx_s = matcher_ref.hungarian_match(ls[-1][0], ls[-1][1], [t["boxes"] for t in ts], ps)
    got = distill_ref.loss_nsthl2(xn.permute(1, 0, 2), xs.permute(1, 0, 2), tok, tok, tn, ts, [len(s) for s, _ in idx_s])
    assert abs(float(got) - float(Z["pair.loss_nsthl2"])) <= 1e-5 * float(Z["pair.loss_nsthl2"])


def test_cluster_update_and_substitution_match_reference():
    MEM, HW = 24, 6
    bank = formula.tensor("dst.bank", (14, MEM, D), 2.0)
    centers = formula.tensor("dst.centers", (14, 3, D), 2.0)
    tok = formula.FakeTokenized(LT)
    _, tn, _, _ = side("noun")
    img = formula.tensor("dst.noun.img", (HW + LT, B, D), 1.5)
    text = img[-LT:].permute(1, 0, 2)
    feats = distill_ref.noun_features(text, tok, tn)
    for i, tgt in enumerate(tn):                       # update_memory_queue: one new feature per task, nearest replaced
        t = int(tgt["dataset_name"].split("_")[1]) - 1
        bank[t] = distill_ref.replace_nearest(bank[t], feats[i:i + 1])
    np.testing.assert_allclose(bank.numpy(), Z["cl.bank_after_update"], rtol=1e-6, atol=1e-7)
    mod = img.clone()
    for i, tgt in enumerate(tn):
        t = int(tgt["dataset_name"].split("_")[1]) - 1
        pos = distill_ref.positions(tok, i, [s for box in tgt["noun_tokens_positive"] for s in box], LT)
        mod, centers[t], _ = distill_ref.cluster_substitute(mod, LT, i, pos, bank[t], centers[t], feats[i], 3)
    np.testing.assert_allclose(centers.numpy(), Z["cl.centers_after_update"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mod.numpy(), Z["cl.noun.img_memory_mod"], rtol=1e-5, atol=1e-6)
    # student side: 'something' -> prototype, loss_cluster_feature
    _, ts, _, _ = side("sth")
    img_s = formula.tensor("dst.sth.img", (HW + LT, B, D), 1.5)
    mod_s, loss = img_s.clone(), 0.0
    for i, cap in enumerate(["put something on it", "use something"]):
        beg = cap.find("something")
        pos = torch.arange(tok.char_to_token(i, beg), tok.char_to_token(i, beg + 8) + 1)
        feature = img_s[-LT:].permute(1, 0, 2)[i][pos].mean(0)
        t = int(ts[i]["dataset_name"].split("_")[1]) - 1
        mod_s, centers[t], centre = distill_ref.cluster_substitute(mod_s, LT, i, pos, bank[t], centers[t], feature, 3)
        loss += float(torch.nn.functional.mse_loss(feature, centre))
    np.testing.assert_allclose(mod_s.numpy(), Z["cl.sth.img_memory_mod"], rtol=1e-5, atol=1e-6)
    assert abs(loss / B - float(Z["cl.loss_cluster_feature"])) <= 1e-5 * float(Z["cl.loss_cluster_feature"])
