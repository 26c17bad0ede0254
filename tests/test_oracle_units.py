"""Host-logic units of the oracle: forest parser and the sort-match tail quirks."""
import numpy as np

from oracle.pyoracle import Settings


def naive_find(ss, sk, ts, tk):
    """Literal model of the reference loop (inference.hpp:231-252) on stably sorted sets."""
    so = np.argsort(ss, kind="stable")
    to = np.argsort(ts, kind="stable")
    S, SK, T, TK = ss[so], sk[so], ts[to], tk[to]
    out, j, i = [], 0, 0
    while i < len(S):
        unique = True
        while i + 1 < len(S) and S[i] == S[i + 1]:
            i += 1
            unique = False
        if unique:
            while j < len(T) - 1 and T[j] < S[i]:
                j += 1
            if j != len(T) - 1 and T[j] == S[i] and (j + 1 == len(T) - 1 or T[j] != T[j + 1]):
                out.append((SK[i], TK[j]))
        i += 1
    return out


def test_forest_parser(oracle, forest_paths):
    rc, f = oracle.read_forest(forest_paths["zero"], 1024, 436)
    assert (rc, f.num_tests, f.type, f.discarded) == (0, 30, 0, 0)
    assert f.offs[0] == -3 + -3 * 1024 and f.offs[1] == 2 + 3 * 1024
    rc, f = oracle.read_forest(forest_paths["tau"], 96, 64)
    assert (rc, f.num_tests, f.type) == (0, 30, 1)
    assert list(f.tau[:5]) == [1, 1, -2, -5, -10]
    rc, f = oracle.read_forest("/nonexistent/forest.txt", 96, 64)
    assert rc == -1 and f.num_tests == 0 and f.type == 0


def test_forest_truncation(oracle):
    lines = ["16"]
    for fern in range(16):
        lines.append("%d m 20" % fern)
        for t in range(20):
            # only a DISCARDED test has tau != 0: type must still become 1 (inference.hpp:433)
            tau = 3 if (fern == 15 and t == 19) else 0
            lines.append("%d %d %d %d %d %d" % (t, t % 7 - 3, fern % 5 - 2, 3 - t % 6, 2 - fern % 4, tau))
    rc, f = oracle.parse_forest_text("\n".join(lines), 64, 48)
    assert (rc, f.num_tests, f.discarded, f.type) == (0, 32, 288, 1)


def test_find_correspondences_quirks(oracle):
    W = 1000
    k = lambda *v: np.array(v, np.int32)
    s = lambda *v: np.array(v, np.uint64)
    # Q1: the last sorted target never matches
    assert len(oracle.find_correspondences(s(5), k(1), s(3, 5), k(10, 11), W)) == 0
    assert len(oracle.find_correspondences(s(3), k(1), s(3, 5), k(10, 11), W)) == 1
    # Q2: a hit at n_t-2 skips the target-uniqueness test
    c = oracle.find_correspondences(s(5), k(1), s(3, 5, 5), k(10, 11, 12), W)
    assert len(c) == 1 and c[0]["tx"] == 11
    assert len(oracle.find_correspondences(s(5), k(1), s(5, 5, 5), k(10, 11, 12), W)) == 0
    assert len(oracle.find_correspondences(s(5), k(1), s(5, 5, 7), k(10, 11, 12), W)) == 0
    # degenerate sets
    assert len(oracle.find_correspondences(s(5), k(1), s(5), k(10), W)) == 0
    assert len(oracle.find_correspondences(s(5), k(1), s(), k(), W)) == 0
    assert len(oracle.find_correspondences(s(), k(), s(5), k(1), W)) == 0
    # duplicates in the source are skipped
    assert len(oracle.find_correspondences(s(5, 5), k(1, 2), s(5, 9), k(10, 11), W)) == 0


def test_find_correspondences_random(oracle):
    rng = np.random.default_rng(11)
    for trial in range(500):
        ns, nt = rng.integers(0, 12, 2)
        ss = rng.integers(0, 8, ns).astype(np.uint64)
        ts = rng.integers(0, 8, nt).astype(np.uint64)
        sk = rng.permutation(100)[:ns].astype(np.int32)
        tk = rng.permutation(100)[:nt].astype(np.int32) + 200
        got = oracle.find_correspondences(ss, sk, ts, tk, 1000)
        want = naive_find(ss, sk, ts, tk) if nt > 0 else []
        assert [(int(c["sx"]) + 1000 * int(c["sy"]), int(c["tx"]) + 1000 * int(c["ty"])) for c in got] == \
               [(int(a), int(b)) for a, b in want]


def test_rectified_filter(oracle):
    from oracle.pyoracle import CORR_DTYPE
    corr = np.array([(50, 20, 40, 20), (50, 21, 40, 22), (300, 30, 100, 30), (10, 5, 30, 5)], CORR_DTYPE)
    out = oracle.rectified_filter(corr, Settings(5, 128, 0, 1))
    assert [(int(s["x"]), int(s["y"]), float(s["d"])) for s in out] == [(50, 20, 10.0), (10, 5, -20.0)]
    out = oracle.rectified_filter(corr, Settings(5, 1000, 1, 1))
    assert len(out) == 4
