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


# ---- a REAL Hugging Face fast tokenizer (tests/golden/tiny_tokenizer.py) driving the span lookups: fixture from the real reference -------------------------
import tiny_tokenizer  # noqa: E402

ZT = np.load(os.path.join(os.path.dirname(__file__), "golden", "distill_tokenizer.npz"))
TK_CAPTIONS = {"noun": ["use the screwdriver to cut the paper up", "sit comfortably on the armchair", "dig a hole with the umbrella handle"],
               "sth": ["use something to cut the paper up", "sit comfortably on something", "dig a hole with something"]}
TK_SPANS = {"noun": [[[(8, 19)], [(7, 19), (31, 37)]], [[(3, 15), (23, 31)]], [[(20, 35)], [(19, 28)]]],
            "sth": [[[(4, 13)], [(4, 13)]], [[(19, 28)]], [[(16, 25)], [(15, 25)]]]}
TK_T = [2, 1, 2]


def tk_side(tag, enc, layers=2, B=3, Q=12, K=256, D=16):
    LT_ = int(enc["input_ids"].shape[1])
    lay = [(formula.tensor(f"dtk.{tag}.logits{l}", (B, Q, K), 4.0), formula.tensor(f"dtk.{tag}.boxes{l}", (B, Q, 4), 0.3, 0.5)) for l in range(layers)]
    targets, pms = [], []
    for i in range(B):
        pm = torch.zeros(TK_T[i], K)
        pm[:, 1 + i:4 + i] = 1.0 / 3
        targets.append({"boxes": formula.tensor(f"dtk.{tag}.tbox{i}", (TK_T[i], 4), 0.25, 0.5), "labels": torch.ones(TK_T[i], dtype=torch.int64),
                        "noun_tokens_positive": TK_SPANS[tag][i], "dataset_name": f"task_{2 + 3 * i}_train.json"})
        pms.append(pm)
    return lay, targets, torch.cat(pms), formula.tensor(f"dtk.{tag}.text", (LT_, B, D), 2.0)


def test_real_tokenizer_spans_match_reference():
    """The char-span -> token lookups (mdetr.py:112-141, 240-260, 684-711) on genuine BatchEncodings: captions of different lengths (padding), a word of ten BPE
    tokens, spans that start / end on a space -- where `char_to_token` returns None and the reference retries WITHOUT the batch index, i.e. in caption 0 (one span
    of image 2 thereby selects one token instead of a word: the fixture holds what the reference really computes).  oracle/distill_ref.py must reproduce
    the real reference's loss_nsthl2, memory bank, prototypes, substituted memories and loss_cluster_feature (tests/golden/make_golden_distill_tokenizer.py)."""
    tok = tiny_tokenizer.build()
    enc = {tag: tok(TK_CAPTIONS[tag], padding="longest", return_tensors="pt") for tag in ("noun", "sth")}
    assert np.array_equal(enc["noun"]["input_ids"].numpy(), ZT["noun.input_ids"]) and np.array_equal(enc["sth"]["input_ids"].numpy(), ZT["sth.input_ids"])
    assert enc["noun"].char_to_token(0, 7) is None and enc["noun"].char_to_token(8) == 3        # a space; the batch-index-free form reads caption 0
    (ln, tn, pn, xn), (ls, ts, ps, xs) = tk_side("noun", enc["noun"]), tk_side("sth", enc["sth"])
    idx_s = matcher_ref.hungarian_match(ls[-1][0], ls[-1][1], [t["boxes"] for t in ts], ps)
    got = distill_ref.loss_nsthl2(xn.permute(1, 0, 2), xs.permute(1, 0, 2), enc["noun"], enc["sth"], tn, ts, [len(s) for s, _ in idx_s])
    assert abs(float(got) - float(ZT["pair.loss_nsthl2"])) <= 1e-5 * float(ZT["pair.loss_nsthl2"]), (float(got), float(ZT["pair.loss_nsthl2"]))
    # the quirk is visible in the token positions themselves
    assert distill_ref.positions(enc["noun"], 2, [(19, 28)], 20).tolist() == [13] and distill_ref.positions(enc["noun"], 2, [(20, 28)], 20).tolist() == list(range(7, 14))
    MEM, HW, D, B = 24, 6, 16, 3
    LTn, LTs = int(enc["noun"]["input_ids"].shape[1]), int(enc["sth"]["input_ids"].shape[1])
    bank = formula.tensor("dtk.bank", (14, MEM, D), 2.0)
    centers = formula.tensor("dtk.centers", (14, 3, D), 2.0)
    img = formula.tensor("dtk.noun.img", (HW + LTn, B, D), 1.5)
    feats = distill_ref.noun_features(img[-LTn:].permute(1, 0, 2), enc["noun"], tn)
    task = lambda t: int(t["dataset_name"].split("_")[1]) - 1
    for i, tgt in enumerate(tn):
        bank[task(tgt)] = distill_ref.replace_nearest(bank[task(tgt)], feats[i:i + 1])
    np.testing.assert_allclose(bank.numpy(), ZT["cl.bank_after_update"], rtol=1e-6, atol=1e-7)
    mod = img.clone()
    for i, tgt in enumerate(tn):
        pos = distill_ref.positions(enc["noun"], i, [s for box in tgt["noun_tokens_positive"] for s in box], LTn)
        mod, centers[task(tgt)], _ = distill_ref.cluster_substitute(mod, LTn, i, pos, bank[task(tgt)], centers[task(tgt)], feats[i], 3)
    np.testing.assert_allclose(centers.numpy(), ZT["cl.centers_after_update"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mod.numpy(), ZT["cl.noun.img_memory_mod"], rtol=1e-5, atol=1e-6)
    img_s = formula.tensor("dtk.sth.img", (HW + LTs, B, D), 1.5)
    mod_s, loss = img_s.clone(), 0.0
    for i, cap in enumerate(TK_CAPTIONS["sth"]):
        beg = cap.find("something")
        pos = torch.arange(enc["sth"].char_to_token(i, beg), enc["sth"].char_to_token(i, beg + 8) + 1)
        feature = img_s[-LTs:].permute(1, 0, 2)[i][pos].mean(0)
        mod_s, centers[task(ts[i])], centre = distill_ref.cluster_substitute(mod_s, LTs, i, pos, bank[task(ts[i])], centers[task(ts[i])], feature, 3)
        loss += float(torch.nn.functional.mse_loss(feature, centre))
    np.testing.assert_allclose(mod_s.numpy(), ZT["cl.sth.img_memory_mod"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(centers.numpy(), ZT["cl.centers_after_forward"], rtol=1e-5, atol=1e-6)
    assert abs(loss / B - float(ZT["cl.loss_cluster_feature"])) <= 1e-5 * float(ZT["cl.loss_cluster_feature"])


def test_distill_tables_pack_equals_the_per_batch_tables_on_a_real_tokenizer():
    """distill.DistillTables.pack (the fixed-address image of a captured distillation step) holds exactly what the list path builds per batch
    (noun_token_weights, span positions, task grouping) -- on real BatchEncodings with padding, None lookups and the reference's fallbacks."""
    from toist_amd import distill
    tok = tiny_tokenizer.build()
    for tag, pronoun in (("noun", False), ("sth", True)):
        enc = tok(TK_CAPTIONS[tag], padding="longest", return_tensors="pt")
        L = int(enc["input_ids"].shape[1])
        _, targets, _, _ = tk_side(tag, enc)
        targets[1]["dataset_name"] = targets[2]["dataset_name"]             # two images in one task
        tb = distill.DistillTables(3, L, "cpu", pronoun_side=pronoun)
        W_span, W_sth, sub_span, sub_sth, task, group_task, group_off, members = tb._views(tb.pack(enc, targets, TK_CAPTIONS[tag]))
        want = distill.noun_token_weights(enc, targets, L, "cpu")
        assert torch.equal(W_span, want)
        for i, t in enumerate(targets):
            assert sub_span[:, i].nonzero().reshape(-1).tolist() == distill.span_positions(enc, i, [s for b_ in t["noun_tokens_positive"] for s in b_], L).tolist()
        assert task.tolist() == [1, 7, 7] and group_task.tolist()[:2] == [1, 7] and group_off.tolist() == [0, 1, 3, 3] and members.tolist() == [0, 1, 2]
        if pronoun:
            for i, cap in enumerate(TK_CAPTIONS[tag]):
                beg = cap.find("something")
                pos = list(range(enc.char_to_token(i, beg), enc.char_to_token(i, beg + 8) + 1))
                assert sub_sth[:, i].nonzero().reshape(-1).tolist() == pos and abs(float(W_sth[i].sum()) - 1.0) < 1e-6
        else:
            assert float(W_sth.abs().sum()) == 0.0 and int(sub_sth.sum()) == 0
