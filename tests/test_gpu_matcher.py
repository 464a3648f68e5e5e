"""GPU parity of the one-wavefront Hungarian matcher against the oracle (oracle/matcher_ref.py +
oracle/lsap.c, themselves pinned to the reference / SciPy on the CPU).  Indices must be bit-identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(seed, B, Q, K, sizes, L=1, dup=False, same_pm=True):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(L, B, Q, K, generator=g) * 2.0
    cxcy = torch.rand(L, B, Q, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(L, B, Q, 2, generator=g) * 0.35 + 0.05
    boxes = torch.cat([cxcy, wh], -1)
    tgt = []
    for t in sizes:
        c = torch.rand(t, 2, generator=g) * 0.6 + 0.2
        s = torch.rand(t, 2, generator=g) * 0.35 + 0.05
        bx = torch.cat([c, s], -1)
        if dup and t >= 2:
            bx[1] = bx[0]
        tgt.append(bx)
    T = sum(sizes)
    if same_pm:
        pm = torch.zeros(T, K)
        pm[:, 1:15] = 1.0 / 14.0
    else:
        pm = torch.rand(T, K, generator=g)
        pm = pm / pm.sum(-1, keepdim=True)
    return logits, boxes, tgt, pm


def _run_gpu(dev, logits, boxes, tgt, pm, w=(1.0, 5.0, 2.0), want_cost=False):
    from toist_amd import kernels as k
    L, B, Q, K = logits.shape
    sizes = [int(t.shape[0]) for t in tgt]
    off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32) if sizes else torch.zeros(1, dtype=torch.int32)
    moff = torch.tensor([0] + list(torch.tensor([min(Q, s) for s in sizes]).cumsum(0)), dtype=torch.int32)
    Mtot, Ttot = int(moff[-1]), int(off[-1])
    tb = torch.cat(tgt) if Ttot else torch.zeros(1, 4)
    src = torch.full((L, max(Mtot, 1)), -1, dtype=torch.int64, device=dev)
    dst = torch.full((L, max(Mtot, 1)), -1, dtype=torch.int64, device=dev)
    status = torch.full((L * B,), -7, dtype=torch.int32, device=dev)
    cost = torch.zeros(L, B * Q, max(Ttot, 1), device=dev) if want_cost else None
    pmd = pm.to(dev) if Ttot else torch.zeros(1, K, device=dev)
    k.matcher(logits.to(dev), boxes.to(dev), tb.to(dev), pmd, off.to(dev), moff.to(dev), max(sizes) if sizes else 0, w[0], w[1], w[2],
              src, dst, status, cost)
    torch.cuda.synchronize()
    return src.cpu(), dst.cpu(), status.cpu(), moff, (cost.cpu() if want_cost else None)


@pytest.mark.parametrize("seed,B,Q,K,sizes,dup,same_pm", [
    (0, 8, 100, 256, [0, 1, 4, 10, 30, 2, 7, 3], False, True),
    (1, 8, 100, 256, [4, 4, 4, 4, 4, 4, 4, 4], True, True),
    (2, 4, 100, 256, [10, 0, 0, 9], True, False),
    (3, 2, 20, 64, [30, 20], False, False),      # T > Q and T == Q
    (4, 3, 97, 256, [97, 96, 98], False, False),  # near-square (softkd-sized)
    (5, 1, 100, 256, [0], False, True),
])
def test_indices_bit_exact(dev, seed, B, Q, K, sizes, dup, same_pm):
    from oracle import matcher_ref
    L = 3
    logits, boxes, tgt, pm = _case(seed, B, Q, K, sizes, L=L, dup=dup, same_pm=same_pm)
    src, dst, status, moff, cost = _run_gpu(dev, logits, boxes, tgt, pm, want_cost=True)
    assert int(status.abs().sum()) == 0, status
    mism = 0
    for l in range(L):
        ref = matcher_ref.hungarian_match(logits[l], boxes[l], tgt, pm)
        if sum(sizes):
            cref = matcher_ref.cost_matrix(logits[l], boxes[l], torch.cat(tgt), pm)
        for b, (ri, rj) in enumerate(ref):
            lo, hi = int(moff[b]), int(moff[b + 1])
            if not (torch.equal(src[l, lo:hi], ri) and torch.equal(dst[l, lo:hi], rj)):
                mism += 1
            if sizes[b]:
                t0 = sum(sizes[:b])
                got = cost[l, b * Q:(b + 1) * Q, t0:t0 + sizes[b]]
                want = cref[b, :, t0:t0 + sizes[b]]
                assert torch.allclose(got, want, rtol=1e-5, atol=2e-6), float((got - want).abs().max())
    assert mism == 0, f"{mism} images with different assignment"


@pytest.mark.parametrize("name", ["rand", "dup", "big", "wide", "samepm", "neartie"])
def test_reference_fixtures_direct(dev, name):
    """toist_matcher on the inputs the REAL reference's HungarianMatcher was run on (tests/golden/make_golden.py): the index
    arrays must equal the reference's bit for bit -- including `neartie`, whose assignment hinges on 1-ulp cost differences."""
    import os
    import numpy as np
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"matcher_{name}.npz"))
    sizes = d["sizes"].tolist()
    tgt = [torch.from_numpy(d[f"tgt{i}"]) for i in range(len(sizes))]
    src, dst, status, moff, _ = _run_gpu(dev, torch.from_numpy(d["logits"])[None], torch.from_numpy(d["boxes"])[None], tgt, torch.from_numpy(d["pm"]))
    assert int(status.abs().sum()) == 0
    bad = [i for i in range(len(sizes))
           if not (np.array_equal(src[0, int(moff[i]):int(moff[i + 1])].numpy(), d[f"src{i}"]) and
                   np.array_equal(dst[0, int(moff[i]):int(moff[i + 1])].numpy(), d[f"dst{i}"]))]
    assert not bad, f"{name}: images {bad} differ from the reference's assignment"


def test_invalid_cost_flagged(dev):
    logits, boxes, tgt, pm = _case(9, 2, 100, 256, [3, 2])
    logits[0, 1, 5, 7] = float("nan")
    _, _, status, _, _ = _run_gpu(dev, logits, boxes, tgt, pm)
    assert status.tolist() == [0, 1]


def test_generic_lsap_matches_oracle(dev):
    """toist_lsap (arbitrary cost matrices: memory-bank replacement mdetr.py:100, softkd matcher mdetr.py:539) against
    oracle/lsap.c, which is pinned to SciPy: tall, wide, square, tied and 1024-column problems in one launch."""
    import numpy as np
    from oracle import lsap as oracle_lsap
    from toist_amd.matcher import linear_sum_assignment_batch
    g = torch.Generator().manual_seed(7)
    costs = [torch.rand(97, 97, generator=g), torch.rand(5, 1024, generator=g) * 3, torch.rand(40, 7, generator=g),
             torch.randint(0, 4, (30, 30), generator=g).float(), torch.rand(1, 1, generator=g), torch.rand(16, 100, generator=g) - 0.5]
    got = linear_sum_assignment_batch([c.to(dev) for c in costs])
    for c, (r, col) in zip(costs, got):
        rr, cc = oracle_lsap.linear_sum_assignment(c.double().numpy())
        assert np.array_equal(r.cpu().numpy(), rr) and np.array_equal(col.cpu().numpy(), cc), c.shape
    bad = torch.rand(4, 4)
    bad[1, 2] = float("nan")
    with pytest.raises(ValueError):
        linear_sum_assignment_batch([bad.to(dev)])


def test_generic_lsap_committed_known_answers(dev):
    """toist_lsap against SciPy's own answers (tests/golden/lsap_kat.json: exact ties, duplicated rows / columns, tall, wide,
    97 x 97, empty) -- one launch for all 40 problems."""
    import json
    import os
    from toist_amd.matcher import linear_sum_assignment_batch
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lsap_kat.json")) as f:
        kat = json.load(f)
    cases = [c for c in kat["cases"] if c["rows"] and c["cols"]]
    costs = [torch.tensor(c["cost"], dtype=torch.float32).reshape(c["rows"], c["cols"]).to(dev) for c in cases]
    got = linear_sum_assignment_batch(costs)
    for i, (c, (r, k_)) in enumerate(zip(cases, got)):
        assert r.cpu().tolist() == c["row_ind"] and k_.cpu().tolist() == c["col_ind"], (i, c["rows"], c["cols"])


def test_register_resident_lsap_on_ties_shapes_and_infinities(dev):
    """Round 6: the solver keeps a column's path cost / dual / owner / POSITION in SciPy's `remaining` list in registers (csrc/matcher.hip
    lsap_solve_wave_reg, 2 .. 16 columns per lane) and resolves ties at the minimum by position.  300 problems chosen to stress exactly that --
    integer costs in {0..3} and {0, 1} (ties at every step), duplicated rows and columns, every lane-count bracket (<= 128, 256, 512, 1024
    columns), tall ones (solved on the transpose), +inf entries (feasible and infeasible) -- against oracle/lsap.c (pinned to SciPy): the
    assignments must be bit-identical, infeasible problems must be reported as such."""
    import numpy as np
    from oracle import lsap as oracle_lsap
    from toist_amd import matcher as tm
    g = torch.Generator().manual_seed(2026)
    costs = []
    for k_ in range(300):
        kind = k_ % 6
        r = int(torch.randint(1, 130, (1,), generator=g))
        c = int(torch.randint(1, 130, (1,), generator=g))
        if k_ % 25 == 0:
            c = [200, 256, 300, 512, 700, 1024][(k_ // 25) % 6]
            r = int(torch.randint(1, 9, (1,), generator=g))
        if kind == 0:
            m = torch.randint(0, 4, (r, c), generator=g).float()
        elif kind == 1:
            m = torch.randint(0, 2, (r, c), generator=g).float()
        elif kind == 2:
            m = torch.rand(r, c, generator=g)
            m[:, c // 2:] = m[:, :c - c // 2].clone()               # duplicated columns
        elif kind == 3:
            m = torch.rand(r, c, generator=g).round(decimals=1)
            m[r // 2:] = m[:r - r // 2].clone()                     # duplicated rows
        elif kind == 4:
            m = torch.rand(r, c, generator=g) - 0.5
            m[torch.rand(r, c, generator=g) < 0.3] = float("inf")
            if (k_ // 6) % 3 == 0:
                m[0, :] = float("inf")                              # a row that cannot be assigned: infeasible whenever rows <= columns
        else:
            m = torch.rand(r, c, generator=g) * 100 - 50
        costs.append(m)
    shapes = [(int(m.shape[0]), int(m.shape[1])) for m in costs]
    sizes = [a * b for a, b in shapes]
    flat = torch.cat([m.reshape(-1) for m in costs]).to(dev)
    ri, ci, out_off, pairs, status = tm.lsap_blocks(flat, shapes, [sum(sizes[:i]) for i in range(len(sizes))], 0)
    ri, ci, status = ri.cpu().numpy(), ci.cpu().numpy(), status.cpu().tolist()
    n_inf = 0
    for m, o, n, st in zip(costs, out_off, pairs, status):
        try:
            rr, cc = oracle_lsap.linear_sum_assignment(m.double().numpy())
        except ValueError:
            assert st == 2, (m.shape, st)                           # infeasible
            n_inf += 1
            continue
        assert st == 0, (m.shape, st)
        assert np.array_equal(ri[o:o + n], rr) and np.array_equal(ci[o:o + n], cc), m.shape
    assert n_inf > 0
