"""GPU tests of scoring against a prepared item table (csrc/als_scores_prepared.inc): the item factor converted once
into two fp16 pieces per power-of-two-scaled value, in the score kernel's operand order.

What is scored is SimpleMFRecAlgo.recommend's X[u] . Y[i] (reclibwh/core/RecAlgos.py:20-32); the reference has no
reduced-precision path, so the bar is the float64 product: every score within 16 x 2^-24 x sum |x_k y_k| (the
float32 dot product's own bound is a few 2^-24 of that sum; item values more than 2^17 below the table's largest add an
absolute floor, see _check_bound), rankings equal to the float64 ranking wherever scores
differ by more than that, and the two prepared calls (materialised scores, fused top-k) equal to each other bit for bit.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd
    return recommender_amd


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _check_bound(S, X, Y, factor=16.0):
    """|error| <= factor x 2^-24 x sum_k |x_k| (|y_k| + 2^-17 max|Y|): 22-bit operands, and for item values more than
    2^17 below the table's largest an absolute floor (their second fp16 piece falls below fp16's normal range)."""
    exact = X.astype(np.float64) @ Y.astype(np.float64).T
    Yabs = np.abs(Y).astype(np.float64) + 2.0 ** -17 * float(np.abs(Y[np.isfinite(Y)]).max())
    bound = factor * 2.0 ** -24 * (np.abs(X).astype(np.float64) @ Yabs.T) + 1e-30
    err = np.abs(S.astype(np.float64) - exact)
    assert np.all(err <= bound), float(np.max(err / bound))


def test_prepared_scores_edge_shapes(ra):
    """Every slice count (f = 4 .. 128), ragged user / item / factor tails, one user, one item."""
    rng = np.random.default_rng(2)
    for (nu, M, f) in [(1, 1, 4), (65, 257, 20), (3, 1000, 64), (70, 300, 52), (129, 33, 36), (200, 4000, 100),
                       (33, 70, 128), (300, 20000, 96), (257, 16385, 128), (5, 31, 32), (64, 32, 68)]:
        X = rng.normal(size=(nu, f)).astype(np.float32); Y = rng.normal(size=(M, f)).astype(np.float32)
        P = ra.prepare_items(_dev(Y))
        assert P.shape == (M, f)
        S = ra.scores(_dev(X), P).cpu().numpy()
        _check_bound(S, X, Y)


def test_prepared_scores_over_wide_magnitudes(ra):
    """User rows are scaled per row and the table by its largest value: rows of very different magnitude - users from
    1e-6 to 1e4, items spanning three decades, a few entries a thousand times the rest - keep the relative bound."""
    rng = np.random.default_rng(5)
    nu, M, f = 96, 3000, 128
    X = rng.normal(size=(nu, f)) * 10.0 ** rng.uniform(-6, 4, size=(nu, 1))
    Y = rng.normal(size=(M, f)) * 10.0 ** rng.uniform(-2, 1, size=(M, 1))
    Y[rng.integers(0, M, 20), rng.integers(0, f, 20)] *= 1e3
    X[3] = 0.0; Y[7] = 0.0
    X = X.astype(np.float32); Y = Y.astype(np.float32)
    S = ra.scores(_dev(X), ra.prepare_items(_dev(Y))).cpu().numpy()
    _check_bound(S, X, Y)
    assert not S[3].any() and not S[:, 7].any()
    # without outliers in the table the floor plays no part: the plain relative bound holds
    Y2 = (rng.normal(size=(M, f)) * 10.0 ** rng.uniform(-2, 1, size=(M, 1))).astype(np.float32)
    S2 = ra.scores(_dev(X), ra.prepare_items(_dev(Y2))).cpu().numpy().astype(np.float64)
    exact = X.astype(np.float64) @ Y2.astype(np.float64).T
    assert np.all(np.abs(S2 - exact) <= 16 * 2.0 ** -24 * (np.abs(X).astype(np.float64) @ np.abs(Y2).astype(np.float64).T) + 1e-30)


def test_a_score_depends_on_its_two_rows_and_the_table_scale_only(ra):
    """Same bits whichever other users share the call, whatever prefix of the items is scored, with user ids and with
    padded leading dimensions; and a table rebuilt from the same factor is the same bytes."""
    rng = np.random.default_rng(6)
    f, ld = 64, 80
    Xs = _dev(rng.normal(size=(500, ld))); Ys = _dev(rng.normal(size=(3000, ld)))
    X, Y = Xs[:, :f], Ys[:, :f]
    P = ra.prepare_items(Y)
    P2 = ra.prepare_items(Y.contiguous())
    assert torch.equal(P.table, P2.table)
    users = torch.from_numpy(rng.permutation(500)[:201]).cuda()
    full = ra.scores(X.contiguous(), P)
    assert torch.equal(ra.scores(X, P, users), full[users])
    assert torch.equal(ra.scores(X, P, users[:7], M=1000), full[users[:7], :1000])
    assert torch.equal(ra.scores(X, P, users[:1], M=33), full[users[:1], :33])


def test_a_non_finite_item_row_spoils_its_own_scores_only(ra):
    rng = np.random.default_rng(8)
    nu, M, f = 70, 900, 96
    X = rng.normal(size=(nu, f)).astype(np.float32); Y = rng.normal(size=(M, f)).astype(np.float32)
    good = ra.scores(_dev(X), ra.prepare_items(_dev(Y)))
    Y[13, 5] = np.nan; Y[400, 0] = np.inf
    bad = ra.scores(_dev(X), ra.prepare_items(_dev(Y)))
    keep = [i for i in range(M) if i not in (13, 400)]
    assert torch.equal(bad[:, keep], good[:, keep])          # the scale ignores non-finite values
    assert not torch.isfinite(bad[:, [13, 400]]).any()


def _exclusion_rows(rng, nu, M, best):
    rows = [np.unique(np.concatenate([best[u], rng.integers(0, M, size=50)])).astype(np.int32) for u in range(nu)]
    ptr = np.zeros(nu + 1, dtype=np.int64); np.cumsum([len(r) for r in rows], out=ptr[1:])
    return rows, (ptr, np.concatenate(rows))


@pytest.mark.parametrize("f,k", [(32, 100), (64, 1024), (100, 10), (128, 100)])
def test_prepared_fused_topk_equals_two_call_path(ra, f, k):
    """Quantised factors, so that equal scores abound (ties rank by ascending item id): the fused top-k on the table is
    scores + top-k on the materialised matrix of the same table, bit for bit, with and without exclusions - and, the
    products of quarter-integers being exact in either arithmetic, also what the float32-operand path returns."""
    rng = np.random.default_rng(f + k)
    nu, M = 300, 200000
    X = _dev(rng.integers(-2, 3, size=(nu, f)) / 4); Y = _dev(rng.integers(-2, 3, size=(M, f)) / 4)
    P = ra.prepare_items(Y)
    users = torch.arange(nu, device="cuda")
    i0, v0 = ra.ops.recommend_topk(X, P, users, k, fused=False)
    i1, v1 = ra.ops.recommend_topk(X, P, users, k)
    i2, v2 = ra.ops.recommend_topk(X, Y, users, k)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)
    assert torch.equal(i0, i2) and torch.equal(v0, v2)
    rows, ex = _exclusion_rows(rng, nu, M, i0[:, :30].cpu().numpy())
    j0, w0 = ra.ops.recommend_topk(X, P, users, k, exclude_csr=ex, fused=False)
    j1, w1 = ra.ops.recommend_topk(X, P, users, k, exclude_csr=ex, block_users=128)
    assert torch.equal(j0, j1) and torch.equal(w0, w1)
    for u in range(0, nu, 37):
        assert not np.intersect1d(j1[u].cpu().numpy(), rows[u]).size


@pytest.mark.parametrize("nu,M,f,k", [(1, 16385, 4, 1), (3, 16384, 8, 16), (5, 16416, 64, 100), (2, 65537, 128, 1024),
                                      (129, 20000, 36, 1024), (7, 300000, 100, 200), (4, 1024, 16, 1024),
                                      (260, 262145, 32, 7)])
def test_prepared_fused_topk_edge_shapes(ra, nu, M, f, k):
    """Range boundaries (one item past the first range, exactly the first range, fewer items than a range), k = 1024, a
    single user, more users than one workgroup takes - on real-valued factors against the two-call path, bit for bit,
    and against the float64 ranking wherever neighbouring scores differ by more than the error bound."""
    rng = np.random.default_rng(nu + M + f + k)
    Xh = rng.normal(size=(nu + 3, f)).astype(np.float32); Yh = rng.normal(size=(M, f)).astype(np.float32)
    X, Y = _dev(Xh), _dev(Yh)
    P = ra.prepare_items(Y)
    users = torch.arange(nu, device="cuda") + 2
    i0, v0 = ra.ops.recommend_topk(X, P, users, k, fused=False)
    i1, v1 = ra.ops.recommend_topk(X, P, users, k)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)
    for u in range(0, nu, max(1, nu // 3)):
        exact = Yh.astype(np.float64) @ Xh[u + 2].astype(np.float64)
        order = np.argsort(-exact, kind="stable")[:k]
        got = i1[u].cpu().numpy()
        tol = 16 * 2.0 ** -24 * float((np.abs(Yh).astype(np.float64) @ np.abs(Xh[u + 2]).astype(np.float64)).max())
        assert np.all(np.abs(exact[got] - exact[order]) <= 2 * tol)          # rank by rank the same score, to the bound
        # where a rank's neighbours on both sides are clearly apart, the item itself must agree
        apart = np.abs(np.diff(exact[order])) > 4 * tol
        if len(order) > 2:
            assert np.all((got == order)[1:-1][apart[:-1] & apart[1:]])


def test_prepared_heavy_users_overflow_alone(ra):
    """As for the float32-operand path: users whose 20 000 best items are all excluded overflow their candidate rows,
    are marked, and are redone alone - by scores + top-k on the same table."""
    rng = np.random.default_rng(12)
    nu, M, f, k = 300, 100000, 64, 100
    Xh = rng.normal(size=(nu, f)).astype(np.float32); Yh = rng.normal(size=(M, f)).astype(np.float32)
    X, Y = _dev(Xh), _dev(Yh)
    P = ra.prepare_items(Y)
    heavy = [7, 150, 299]
    rows = [np.sort(rng.choice(M, size=int(c), replace=False)).astype(np.int32) for c in rng.integers(0, 200, size=nu)]
    for u in heavy:
        sc = Yh.astype(np.float64) @ Xh[u].astype(np.float64)
        rows[u] = np.sort(np.argsort(-sc)[:20000]).astype(np.int32)
    ptr = np.zeros(nu + 1, dtype=np.int64); np.cumsum([len(r) for r in rows], out=ptr[1:])
    ex = (ptr, np.concatenate(rows))
    users = torch.arange(nu, device="cuda")
    j0, w0 = ra.ops.recommend_topk(X, P, users, k, exclude_csr=ex, fused=False)
    j1, w1 = ra.ops.recommend_topk(X, P, users, k, exclude_csr=ex)
    assert torch.equal(j0, j1) and torch.equal(w0, w1)
    for u in heavy:
        assert not np.intersect1d(j1[u].cpu().numpy(), rows[u]).size


def test_prepared_arguments(ra):
    rng = np.random.default_rng(1)
    Y = _dev(rng.normal(size=(100, 64))); X = _dev(rng.normal(size=(10, 64)))
    with pytest.raises(ValueError):
        ra.prepare_items(_dev(rng.normal(size=(100, 132))))          # f > 128
    with pytest.raises(ValueError):
        ra.prepare_items(_dev(rng.normal(size=(100, 30))))           # f % 4 != 0
    with pytest.raises(ValueError):
        ra.prepare_items(Y, M=101)
    P = ra.prepare_items(Y, M=60)
    assert P.shape == (60, 64)
    with pytest.raises(ValueError):
        ra.scores(_dev(rng.normal(size=(10, 32))), P)                # factor counts differ
    with pytest.raises(ValueError):
        ra.scores(X, P, M=61)                                        # more items than the table holds
    with pytest.raises(ValueError):
        ra.ops.recommend_topk(X, P, torch.arange(10, device="cuda"), 5, M=50)   # a table is ranked whole
    S = ra.scores(X, P)
    assert S.shape == (10, 60)
    assert torch.equal(S, ra.scores(X, ra.prepare_items(Y))[:, :60])
    i, v = ra.ops.recommend_topk(X, P, torch.arange(10, device="cuda"), 5)
    assert int(i.max()) < 60
