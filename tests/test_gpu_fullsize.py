"""Full-size checks on the shapes of the benchmark configurations (MovieLens-25M-shaped matrix,
f=64; top-100 against 1 M items, f=128), through properties that do not need the CPU oracle to
walk the whole input: sampled rows (the longest ones included, which are the rows cut into
chunks) against the float64 oracle, bit-for-bit repeatability, row shards = whole, the sum of the
per-row changes, and the fused top-k against a selection done by torch on the same scores.

The same on one GPU's eighth of the 1 B-entry matrix (1.25 M users x 1 M items, 125 M entries, f = 128: exact and
conjugate-gradient solves) and on a slice of the 4 B-entry one (all 5 M items, f = 256: conjugate gradients, the
formed-matrix form over 4 wavefronts and the lock-step form swept in row slabs).

Bounds: factors 1e-4 relative (max-abs error over the sample / max-abs value), the north-star
bar; 2e-4 for three conjugate-gradient steps (a truncated iteration carries the float32 input
rounding through every step; the same bound as the small CG tests)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import als_ref, c_oracle  # noqa: E402

pytestmark = pytest.mark.gpu

LAMB, ALPHA, F = 0.01, 40.0, 64


@pytest.fixture(scope="module")
def c2():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd as ra
    from recommender_amd import data
    from recommender_amd.als import init_factors
    wl = data.WORKLOADS["c2_ml25m_syn"]
    N, M = wl["n_users"], wl["n_items"]
    U, Ut = data.synth_matrix(N, M, wl["nnz"], seed=1234, device="cuda")
    X0, Y0 = init_factors(N, M, F, "tf")
    return dict(ra=ra, N=N, M=M, U=U, Ut=Ut, X0=X0.astype(np.float32), Y0=Y0.astype(np.float32))


def _sample_rows(ptr, rng, n_random, n_longest):
    lens = np.diff(ptr)
    longest = np.argsort(-lens, kind="stable")[:n_longest]
    shortest = np.argsort(lens, kind="stable")[:5]
    rows = np.unique(np.concatenate([longest, shortest, rng.choice(len(lens), size=n_random, replace=False)]))
    return rows, lens


def _sub_csr(ptr, col, val, rows):
    lens = ptr[rows + 1] - ptr[rows]
    sp = np.zeros(len(rows) + 1, dtype=np.int64); np.cumsum(lens, out=sp[1:])
    idx = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in rows])
    return sp, col[idx], val[idx].astype(np.float64)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _check_side(c2, csr_triplet, shape, fixed, own, seed, min_longest=2 * 2048):
    ra = c2["ra"]
    ptr_d, col_d, val_d = csr_triplet
    csr = ra.DeviceCSR(ptr_d, col_d, val_d, shape, "cuda")
    Yd = torch.from_numpy(fixed).cuda()
    gc = ra.gram(Yd, LAMB)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    X = torch.from_numpy(own).cuda()
    d = ra.half_sweep(csr, Yd, gc, X, ALPHA, info=info)
    assert int(info.item()) == 0
    # the reported change is the sum of the per-row changes
    ref_d = (torch.from_numpy(own).cuda().double() - X.double()).abs().sum()
    assert float(d.item()) == pytest.approx(float(ref_d.item()), rel=1e-6)
    # sampled rows against the float64 oracle: the longest rows (cut into chunks) are in the sample
    ptr, col, val = (t.cpu().numpy() for t in csr_triplet)
    rows, lens = _sample_rows(ptr, np.random.default_rng(seed), 300, 25)
    assert lens[rows].max() > min_longest                   # rows of several chunks
    sp, sc, sv = _sub_csr(ptr, col, val, rows)
    ref = c_oracle.half_sweep(sp, sc, sv, fixed.astype(np.float64), own[rows].astype(np.float64), LAMB, ALPHA,
                              dtype=np.float64, nthreads=8)
    got = X[torch.from_numpy(rows).cuda()].cpu().numpy().astype(np.float64)
    assert _rel(got, ref) < 1e-4, _rel(got, ref)
    # bit-for-bit repeatable
    X2 = torch.from_numpy(own).cuda()
    d2 = ra.half_sweep(csr, Yd, gc, X2, ALPHA)
    assert torch.equal(X, X2) and float(d.item()) == float(d2.item())
    # two row shards (balanced by entries, as two devices would hold them) give the same bits
    from recommender_amd.als import balanced_ranges
    b = [int(x) for x in balanced_ranges(ptr, 2)]
    X3 = torch.from_numpy(own).cuda()
    for r0, r1 in zip(b[:-1], b[1:]):
        ra.half_sweep(csr.row_range(r0, r1), Yd, gc, X3[r0:r1], ALPHA)
    assert torch.equal(X, X3)
    # three conjugate-gradient steps, formed-matrix form, same sample
    X4 = torch.from_numpy(own).cuda()
    ra.half_sweep_cg(csr, Yd, gc, X4, ALPHA, 3)
    rows_cg = rows[:: max(1, len(rows) // 60)]
    spc, scc, svc = _sub_csr(ptr, col, val, rows_cg)
    ref_cg = als_ref.half_sweep_cg(spc, scc, svc, fixed.astype(np.float64), own[rows_cg].astype(np.float64), LAMB, ALPHA, 3)
    got_cg = X4[torch.from_numpy(rows_cg).cuda()].cpu().numpy().astype(np.float64)
    assert _rel(got_cg, ref_cg) < 2e-4, _rel(got_cg, ref_cg)
    return X.cpu().numpy()


def test_c2_user_and_item_half_sweeps_at_full_size(c2):
    N, M = c2["N"], c2["M"]
    X1 = _check_side(c2, c2["U"], (N, M), c2["Y0"], c2["X0"], seed=1)
    _check_side(c2, c2["Ut"], (M, N), X1, c2["Y0"], seed=2)


def _device_factors(n, f, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    return (torch.randn((n, f), generator=g, device="cuda") / f ** 0.5).cpu().numpy()


def test_c3_shard_user_and_item_half_sweeps_at_full_size():
    """One GPU's eighth of the 10 M x 1 M matrix at f = 128: both factor tables exceed the caches, rows are short
    (100 entries a user), and with 1.25 M rows every resident wavefront of the stream kernel has its own."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd as ra
    from recommender_amd import data
    wl = data.WORKLOADS["c3_shard_syn"]
    N, M, f = wl["n_users"], wl["n_items"], wl["f"]
    U, Ut = data.synth_matrix(N, M, wl["nnz"], seed=4321, device="cuda")
    ctx = dict(ra=ra)
    X0, Y0 = _device_factors(N, f, 11), _device_factors(M, f, 12)
    X1 = _check_side(ctx, U, (N, M), Y0, X0, seed=3, min_longest=64)
    del U
    _check_side(ctx, Ut, (M, N), X1, Y0, seed=4, min_longest=2048)


def test_c3_rank_item_half_sweep_against_the_whole_user_table():
    """The item half-sweep as ONE rank of the eight-GPU run of the 10 M x 1 M matrix executes it (f = 128): its
    125 000 items against all 10 M users - rows eight times longer than in the one-block matrix (the longest are cut into
    hundreds of chunks), gathered from the whole 5 GB user table.  Sampled rows (up to 300 000 entries long) against the
    float64 oracle, which gets the rows of the user table that the sample touches and the Gram of the whole table;
    the reported change, repeatability, two row shards = whole."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd as ra
    from recommender_amd import data
    from recommender_amd.als import balanced_ranges
    wl = data.WORKLOADS["c3_rank_syn"]
    W, f = wl["of_world"], wl["f"]
    N, mi = wl["n_users"] * W, wl["n_items"] // W
    keys, r = data.synth_pairs(N, mi, wl["nnz"], seed=4321, device="cuda")
    tk, tr = data.transpose_keys(keys, r, N, mi)
    del keys, r
    Ut = data.csr_from_sorted_keys(tk, tr, mi, N)
    del tk, tr
    g = torch.Generator(device="cuda"); g.manual_seed(31)
    X = torch.randn((N, f), generator=g, device="cuda") / f ** 0.5
    Y0 = torch.randn((mi, f), generator=g, device="cuda") / f ** 0.5
    csr = ra.DeviceCSR(Ut[0], Ut[1], Ut[2], (mi, N), "cuda", chunk_nnz=8192)
    gc = ra.gram(X, LAMB)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    Y = Y0.clone()
    d = ra.half_sweep(csr, X, gc, Y, ALPHA, info=info)
    assert int(info.item()) == 0
    assert float(d.item()) == pytest.approx(float((Y0.double() - Y.double()).abs().sum().item()), rel=1e-6)
    ptr, col, val = (t.cpu().numpy() for t in Ut)
    lens = np.diff(ptr)
    assert lens.max() > 100 * 8192                       # rows of hundreds of chunks exist
    order = np.argsort(-lens, kind="stable")
    capped = order[lens[order] <= 300_000]
    rng = np.random.default_rng(9)
    rows = np.unique(np.concatenate([capped[:3], order[-5:], rng.choice(mi, size=40, replace=False)]))
    assert lens[rows].max() > 4 * 8192                   # the sample holds rows of several chunks
    sp, sc, sv = _sub_csr(ptr, col, val, rows)
    used, sc_local = np.unique(sc, return_inverse=True)
    Xsub = X[torch.from_numpy(used.astype(np.int64)).cuda()].double().cpu().numpy()
    G = torch.zeros((f, f), dtype=torch.float64, device="cuda")
    for r0 in range(0, N, 1 << 20):                      # float64 Gram of the whole table, a slab at a time
        blk = X[r0:r0 + (1 << 20)].double()
        G += blk.T @ blk
    sel = torch.from_numpy(rows).cuda()
    ref = c_oracle.half_sweep(sp, sc_local.astype(np.int32), sv, Xsub, Y0[sel].double().cpu().numpy(), LAMB, ALPHA,
                              dtype=np.float64, nthreads=8, Y2=G.cpu().numpy())
    got = Y[sel].cpu().numpy().astype(np.float64)
    assert _rel(got, ref) < 1e-4, _rel(got, ref)
    Y2 = Y0.clone()
    d2 = ra.half_sweep(csr, X, gc, Y2, ALPHA)
    assert torch.equal(Y, Y2) and float(d.item()) == float(d2.item())
    b = [int(x) for x in balanced_ranges(ptr, 2)]
    Y3 = Y0.clone()
    for r0, r1 in zip(b[:-1], b[1:]):
        ra.half_sweep(csr.row_range(r0, r1), X, gc, Y3[r0:r1], ALPHA)
    assert torch.equal(Y, Y3)


def test_c4_slice_cg_forms_at_f256():
    """A slice of one GPU's eighth of the 50 M x 5 M matrix (every item, 1/20 of the shard's users) at f = 256: three
    conjugate-gradient steps in the formed-matrix form (a row's tiles over 4 wavefronts) and in the lock-step form
    swept in row slabs that share one bounded workspace; sampled rows (the longest included) against the float64
    restatement, which gets the rows of the item table that the sample touches and the Gram of the whole table."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd as ra
    from recommender_amd import data, ops
    from recommender_amd.als import balanced_ranges
    N, M, nnz, f = 312_500, 5_000_000, 25_000_000, 256
    keys, r = data.synth_pairs(N, M, nnz, 99, "cuda")
    U = data.csr_from_sorted_keys(keys, r, N, M)
    del keys, r
    g = torch.Generator(device="cuda"); g.manual_seed(21)
    Y = torch.randn((M, f), generator=g, device="cuda") / f ** 0.5
    X0 = torch.randn((N, f), generator=g, device="cuda") / f ** 0.5
    csr = ra.DeviceCSR(U[0], U[1], U[2], (N, M), "cuda")
    gc = ra.gram(Y, LAMB)
    ptr, col, val = (t.cpu().numpy() for t in U)
    rows, lens = _sample_rows(ptr, np.random.default_rng(8), 60, 10)
    sp, sc, sv = _sub_csr(ptr, col, val, rows)
    used, sc_local = np.unique(sc, return_inverse=True)
    Ysub = Y[torch.from_numpy(used.astype(np.int64)).cuda()].double().cpu().numpy()
    Y2 = (Y.double().T @ Y.double()).cpu().numpy()
    ref = als_ref.half_sweep_cg(sp, sc_local, sv, Ysub, X0[torch.from_numpy(rows).cuda()].double().cpu().numpy(),
                                LAMB, ALPHA, 3, Y2=Y2)
    sel = torch.from_numpy(rows).cuda()
    out = {}
    for form in ("matrix", "lock_step"):
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        X = X0.clone()
        old = ops.LOCKSTEP_WS_LIMIT
        ops.LOCKSTEP_WS_LIMIT = 96 << 20          # 4 f-vectors a row: 4 KB a row at f = 256 -> slabs of ~24 000 rows
        try:
            d = ra.half_sweep_cg(csr, Y, gc, X, ALPHA, 3, info=info, form=form)
            X2 = X0.clone()
            d2 = ra.half_sweep_cg(csr, Y, gc, X2, ALPHA, 3, form=form)
        finally:
            ops.LOCKSTEP_WS_LIMIT = old
        assert int(info.item()) == 0
        assert torch.equal(X, X2) and float(d.item()) == float(d2.item())          # bit-for-bit repeatable
        ref_d = (X0.double() - X.double()).abs().sum()
        assert float(d.item()) == pytest.approx(float(ref_d.item()), rel=1e-6)
        got = X[sel].cpu().numpy().astype(np.float64)
        assert _rel(got, ref) < 2e-4, (form, _rel(got, ref))
        out[form] = X
    # the slabs' factors are those of one call with the whole workspace
    X3 = X0.clone()
    ra.half_sweep_cg(csr, Y, gc, X3, ALPHA, 3, form="lock_step")
    assert torch.equal(out["lock_step"], X3)
    # two row shards (balanced by entries) give the same bits as the whole, in either form
    b = [int(x) for x in balanced_ranges(ptr, 2)]
    for form in ("matrix", "lock_step"):
        X4 = X0.clone()
        for r0, r1 in zip(b[:-1], b[1:]):
            ra.half_sweep_cg(csr.row_range(r0, r1), Y, gc, X4[r0:r1], ALPHA, 3, form=form)
        assert torch.equal(out[form], X4)


@pytest.mark.parametrize("prepared", [False, True])
def test_c5_fused_topk_at_one_million_items(prepared):
    """prepared: against the item table in the score kernel's operand form (two fp16 pieces per value), else against
    the float32 factor (three bf16 pieces)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd as ra
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    f, M, nu, k = 128, 1_000_000, 300, 100
    X = torch.randn((4000, f), generator=g, device="cuda") / f ** 0.5
    Y = torch.randn((M, f), generator=g, device="cuda") / f ** 0.5
    users = torch.randperm(4000, generator=g, device="cuda")[:nu]
    rng = np.random.default_rng(0)
    ex_rows = [np.sort(rng.choice(M, size=int(c), replace=False)).astype(np.int32) for c in rng.integers(0, 400, size=nu)]
    ptr = np.zeros(nu + 1, dtype=np.int64); np.cumsum([len(r) for r in ex_rows], out=ptr[1:])
    # exclusions are given per user id over ALL users: scatter the block's rows into a full CSR
    full_rows = [np.zeros(0, np.int32)] * 4000
    for u, r in zip(users.cpu().numpy(), ex_rows):
        full_rows[u] = r
    fptr = np.zeros(4001, dtype=np.int64); np.cumsum([len(r) for r in full_rows], out=fptr[1:])
    ex = (fptr, np.concatenate(full_rows))
    T = ra.prepare_items(Y) if prepared else Y
    idx, val = ra.ops.recommend_topk(X, T, users, k, exclude_csr=ex)
    # the selection done by torch on the library's own scores (same arithmetic, so the same bits)
    S = ra.scores(X, T, users)
    for j, r in enumerate(ex_rows):
        S[j, torch.from_numpy(r.astype(np.int64)).cuda()] = -float("inf")
    tv, ti = torch.topk(S, k, dim=1, sorted=True)
    assert torch.equal(val, tv)
    assert (idx.long() == ti).float().mean().item() > 0.9999        # equal scores may swap in torch's order, not in ours
    assert bool((val[:, :-1] >= val[:, 1:]).all())
    for j in range(0, nu, 29):
        got = idx[j].cpu().numpy()
        assert len(np.unique(got)) == k and not np.intersect1d(got, ex_rows[j]).size
    # independent of the library's arithmetic: every user's float64 scores of all items, the exclusions applied, ranked
    # by torch - the k best scores agree rank by rank to float32 rounding, and so do the items but for swaps of scores
    # that close (the score GEMM at M = 1 M, all 300 users, not a sample)
    del S, tv, ti
    S64 = X[users].double() @ Y.double().T
    for j, r in enumerate(ex_rows):
        S64[j, torch.from_numpy(r.astype(np.int64)).cuda()] = -float("inf")
    tv64, ti64 = torch.topk(S64, k, dim=1, sorted=True)
    assert float((val.double() - tv64).abs().max()) < 2e-6
    assert float((S64.gather(1, idx.long()) - val.double()).abs().max()) < 2e-6
    assert (idx.long() == ti64).float().mean().item() > 0.999


def test_entry_offsets_beyond_two_to_the_31():
    """A CSR with more than 2^31 stored entries (2 M rows x 1 100 entries = 2.2 B; the 4 B-entry configuration needs
    64-bit entry offsets everywhere - the reference casts its indices to int32, reclibwh/core/ALS.py:220-222, and could
    not hold it).  Built on the device without a sort: row r holds the arithmetic progression
    base_r + j * stride_r of columns.  One exact half-sweep (f = 64) and one lock-step conjugate-gradient half-sweep:
    sampled rows from the LAST 1 % of the entry range (offsets above 2.178e9) against the float64 oracle, a repeat
    run bit-equal, and the rows whose entries straddle 2^31 included in the sample."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import recommender_amd as ra
    dev = "cuda"
    n_rows, L, M, f = 2_000_000, 1100, 200_000, 64
    assert n_rows * L > 2 ** 31
    g = torch.Generator(device=dev); g.manual_seed(5)
    stride = torch.randint(1, 151, (n_rows,), generator=g, device=dev, dtype=torch.int32)
    base = torch.randint(0, M - L * 150, (n_rows,), generator=g, device=dev, dtype=torch.int32)
    col = torch.empty(n_rows * L, dtype=torch.int32, device=dev)
    val = torch.empty(n_rows * L, dtype=torch.float32, device=dev)
    j = torch.arange(L, device=dev, dtype=torch.int32)
    step = 100_000
    for r0 in range(0, n_rows, step):                 # in row slabs: no temporary larger than the slab
        r1 = min(n_rows, r0 + step)
        c = base[r0:r1, None] + j[None, :] * stride[r0:r1, None]
        col[r0 * L: r1 * L] = c.reshape(-1)
        rr = torch.arange(r0, r1, device=dev, dtype=torch.int32)
        val[r0 * L: r1 * L] = (((rr[:, None] + j[None, :]) % 10 + 1).to(torch.float32) * 0.5).reshape(-1)
        del c
    ptr = torch.arange(n_rows + 1, device=dev, dtype=torch.int64) * L
    assert int(ptr[-1]) > 2 ** 31
    csr = ra.DeviceCSR(ptr, col, val, (n_rows, M), dev)
    gy = torch.Generator(device=dev); gy.manual_seed(6)
    Y = torch.randn((M, f), generator=gy, device=dev) / np.sqrt(f)
    X0 = torch.randn((n_rows, f), generator=gy, device=dev) / np.sqrt(f)
    cache = ra.gram(Y, LAMB)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    # rows of the last 1 % of the entry range, and the rows around entry 2^31
    rng = np.random.default_rng(9)
    first_last = int(0.99 * n_rows)
    cross = (2 ** 31) // L
    rows = np.unique(np.concatenate([rng.choice(np.arange(first_last, n_rows), size=120, replace=False),
                                     np.arange(cross - 2, cross + 3), [0, n_rows - 1]]))
    assert (rows.astype(np.int64) * L).max() > 0.99 * n_rows * L
    Yh = Y.cpu().numpy().astype(np.float64)
    sub_ptr = np.arange(len(rows) + 1, dtype=np.int64) * L
    idx = (torch.from_numpy(rows.astype(np.int64)).to(dev)[:, None] * L + torch.arange(L, device=dev)[None, :]).reshape(-1)
    sub_col = col[idx].cpu().numpy(); sub_val = val[idx].cpu().numpy().astype(np.float64)
    X0s = X0[torch.from_numpy(rows).to(dev)].cpu().numpy().astype(np.float64)
    want = c_oracle.half_sweep(sub_ptr, sub_col, sub_val, Yh, X0s, LAMB, ALPHA, nthreads=c_oracle.max_threads())

    def exact():
        X = X0.clone()
        d = ra.half_sweep(csr, Y, cache, X, ALPHA, info=info)
        torch.cuda.synchronize()
        return X, float(d.item())
    Xa, da = exact()
    got = Xa[torch.from_numpy(rows).to(dev)].cpu().numpy().astype(np.float64)
    assert int(info.item()) == 0
    assert _rel(got, want) < 1e-4, _rel(got, want)
    Xb, db = exact()
    assert torch.equal(Xa, Xb) and da == db
    del Xb
    # lock-step conjugate gradients, 2 steps, against the same iteration in float64
    steps = 2
    want_cg = als_ref.half_sweep_cg(sub_ptr, sub_col, sub_val, Yh, X0s, LAMB, ALPHA, steps, Y2=cache.gram.cpu().numpy().astype(np.float64))

    def cg():
        X = X0.clone()
        ra.half_sweep_cg(csr, Y, cache, X, ALPHA, steps, info=info, form="lock_step")
        torch.cuda.synchronize()
        return X
    Xc = cg()
    got_cg = Xc[torch.from_numpy(rows).to(dev)].cpu().numpy().astype(np.float64)
    assert _rel(got_cg, want_cg) < 2e-4, _rel(got_cg, want_cg)
    assert torch.equal(Xc, cg())
