"""Pins oracle/lsap.c (the CPU restatement of scipy.optimize.linear_sum_assignment) against SciPy
itself and the known-answer tests of SURVEY.md section 4.  CPU only."""
import numpy as np
import pytest

from oracle import lsap


def test_known_answers():
    r, c = lsap.linear_sum_assignment(np.zeros((4, 2)))
    assert r.tolist() == [0, 1] and c.tolist() == [0, 1]
    r, c = lsap.linear_sum_assignment(np.zeros((2, 4)))
    assert r.tolist() == [0, 1] and c.tolist() == [0, 1]
    r, c = lsap.linear_sum_assignment(np.zeros((100, 0)))
    assert r.size == 0 and c.size == 0
    r, c = lsap.linear_sum_assignment(np.array([[1, 2, 3], [2, 4, 6], [3, 6, 9], [0, 0, 0], [5, 1, 1.0]]))
    assert r.tolist() == [0, 3, 4] and c.tolist() == [0, 1, 2]
    r, c = lsap.linear_sum_assignment(np.array([[np.inf, 1], [1, np.inf]]))
    assert r.tolist() == [0, 1] and c.tolist() == [1, 0]
    with pytest.raises(ValueError, match="invalid numeric entries"):
        lsap.linear_sum_assignment(np.array([[np.nan, 1.0], [1.0, 2.0]]))
    with pytest.raises(ValueError, match="invalid numeric entries"):
        lsap.linear_sum_assignment(np.array([[-np.inf, 1.0], [1.0, 2.0]]))


def test_against_scipy_random_and_ties():
    scipy_opt = pytest.importorskip("scipy.optimize")
    rng = np.random.default_rng(0)
    for it in range(3000):
        nr, nc = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        mode = it % 6
        if mode == 0:
            c = rng.random((nr, nc))
        elif mode == 1:
            c = rng.integers(0, 3, (nr, nc)).astype(float)
        elif mode == 2:
            c = np.zeros((nr, nc))
        elif mode == 3:
            c = rng.random((nr, nc))
            if nc > 1:
                c[:, 1] = c[:, 0]
        elif mode == 4:
            c = rng.integers(0, 5, (nr, nc)).astype(float)
            if nr > 2:
                c[2] = c[0]
        else:
            c = rng.random((nr, nc)).astype(np.float32).astype(float)
            c[rng.random((nr, nc)) < 0.1] = np.inf
        try:
            a, b = scipy_opt.linear_sum_assignment(c)
        except ValueError:
            with pytest.raises(ValueError):
                lsap.linear_sum_assignment(c)
            continue
        r, k = lsap.linear_sum_assignment(c)
        assert np.array_equal(a, r) and np.array_equal(b, k), (it, mode, nr, nc)


def test_matcher_shaped_blocks():
    scipy_opt = pytest.importorskip("scipy.optimize")
    rng = np.random.default_rng(1)
    for T in (1, 4, 10, 30, 97, 100, 130):
        c = rng.random((100, T)).astype(np.float32).astype(np.float64)
        if T >= 2:
            c[:, 1] = c[:, 0]  # duplicated ground-truth box
        a, b = scipy_opt.linear_sum_assignment(c)
        r, k = lsap.linear_sum_assignment(c)
        assert np.array_equal(a, r) and np.array_equal(b, k)
        assert np.all(np.diff(r) > 0)


def test_committed_known_answers():
    """tests/golden/lsap_kat.json (SciPy's answers, committed: the GPU box need not have SciPy)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lsap_kat.json")) as f:
        kat = json.load(f)
    assert len(kat["cases"]) >= 40
    for i, c in enumerate(kat["cases"]):
        cost = np.array(c["cost"], dtype=np.float64).reshape(c["rows"], c["cols"])
        r, k = lsap.linear_sum_assignment(cost)
        assert r.tolist() == c["row_ind"] and k.tolist() == c["col_ind"], i
