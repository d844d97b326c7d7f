"""The device form of the reference's evaluation (metrics.mean_average_precision_device: torch operations, no
per-user loop) against the host loop over the reference's own compute_ap (reclibwh/utils/eval_utils.py:23-37,
AUCEval.evaluate EvalProto.py:28-62), on CPU tensors: the same number to rounding on random rankings, including
users without test items, users whose list holds none of theirs, lists that are all training items and lim larger
than the list."""
import numpy as np
import pytest
import scipy.sparse as sps

torch = pytest.importorskip("torch")

from recommender_amd import metrics  # noqa: E402


def _random_case(seed, n_users=60, n_items=90, lim=40, width=50):
    rng = np.random.default_rng(seed)
    train = sps.random(n_users, n_items, density=0.15, random_state=seed, format="csr", data_rvs=lambda n: rng.integers(1, 11, n) * 0.5)
    test = sps.random(n_users, n_items, density=0.08, random_state=seed + 1, format="csr", data_rvs=lambda n: rng.integers(1, 11, n) * 0.5)
    train.sort_indices(); test.sort_indices()
    test = test.tolil(); test[:5] = 0; test = test.tocsr(); test.eliminate_zeros(); test.sort_indices()   # users without test items
    users = rng.choice(n_users, size=35, replace=False)
    recs = np.stack([rng.permutation(n_items)[:width] for _ in users])
    # one user whose whole list is training items
    tr0 = train.indices[train.indptr[users[0]]:train.indptr[users[0] + 1]]
    if len(tr0):
        recs[0, :] = np.resize(tr0, width)
    return train, test, users, recs, n_items, lim


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
@pytest.mark.parametrize("lim", [40, 200])
def test_device_map_equals_host_loop(seed, lim):
    train, test, users, recs, M, _ = _random_case(seed)
    want = metrics.mean_average_precision(recs, users, train, test, 3.0, lim)
    tk = metrics.pair_keys(train, M, "cpu")
    rk, rc = metrics.relevant_test_keys(test, M, "cpu", 3.0)
    got = metrics.mean_average_precision_device(torch.from_numpy(recs), torch.from_numpy(users), tk, rk, rc, M, lim)
    assert got == pytest.approx(want, rel=1e-12, abs=1e-15)


def test_device_map_no_counted_user_is_nan():
    train = sps.csr_matrix((4, 6)); test = sps.csr_matrix((4, 6))
    tk = metrics.pair_keys(train, 6, "cpu")
    rk, rc = metrics.relevant_test_keys(test, 6, "cpu")
    got = metrics.mean_average_precision_device(torch.arange(6).repeat(2, 1), torch.tensor([0, 1]), tk, rk, rc, 6, 5)
    assert np.isnan(got) and np.isnan(metrics.mean_average_precision(np.arange(6)[None].repeat(2, 0), [0, 1], train, test))


def test_device_rmse():
    a = torch.tensor([1.0, 2.0, 4.0]); b = torch.tensor([1.5, 2.0, 2.0])
    assert metrics.rmse_device(a, b) == pytest.approx(metrics.rmse(a.numpy(), b.numpy()), rel=1e-12)
