"""CPU-only tests: host logic, the C-ABI library's exported symbols, metrics and init against
the golden vectors.  No compute call is made here (there is no GPU in this environment)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "als_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(als_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 14
    lib = ctypes.CDLL(os.path.join(ROOT, "recommender_amd", "libals_hip.so"))
    for n in sorted(names):
        assert hasattr(lib, n), "libals_hip.so does not export %s" % n
    from recommender_amd import _lib
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)
    assert _lib.lib.als_abi_version() >= 1


def test_option_struct_is_the_same_in_header_binding_and_guide():
    """als_sweep_opts_t: the header's fields, the ctypes mirror of recommender_amd/_lib.py and the one printed in the
    integration guide agree in names, order and C types, and both bindings state the library's ABI version - a field
    added to one of them only would be read as garbage by the library."""
    from recommender_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "als_hip.h")).read()
    body = re.search(r"typedef struct \{(.*?)\} als_sweep_opts_t;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if decl:
            name = re.search(r"(\w+)$", decl).group(1)
            ctype = decl[: -len(name)].strip()
            fields.append((name, ctypes.c_void_p if "*" in ctype else {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64,
                                                                      "float": ctypes.c_float}[ctype]))
    assert fields == list(_lib.SweepOpts._fields_)
    assert _lib.lib.als_abi_version() == _lib.ABI_VERSION
    guide = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = next(b for b in re.findall(r"```python\n(.*?)```", guide, flags=re.S) if "class SweepOpts" in b)
    names = re.findall(r'\("(\w+)", (\w+)\)', re.search(r"_fields_ = \[(.*?)\]", stub, flags=re.S).group(1))
    short = {"i32": ctypes.c_int32, "i64": ctypes.c_int64, "f32": ctypes.c_float, "vp": ctypes.c_void_p}
    assert [(n, short[t]) for n, t in names] == fields
    assert "als_abi_version() == %d" % _lib.ABI_VERSION in stub


def test_error_strings_and_argument_checks_without_gpu():
    from recommender_amd._lib import lib
    n = ctypes.c_int64()
    ptr = np.array([0, 3, 1], dtype=np.int64)       # decreasing row pointer
    rc = lib.als_plan_count(ptr.ctypes.data, 2, 2048, ctypes.byref(n), ctypes.byref(n), ctypes.byref(n))
    assert rc == 1 and b"non-decreasing" in lib.als_last_error_string()
    rc = lib.als_plan_count(ptr.ctypes.data, 2, 100, ctypes.byref(n), ctypes.byref(n), ctypes.byref(n))
    assert rc == 1 and b"multiple of 64" in lib.als_last_error_string()
    assert lib.als_gram_tiles_bytes(0) == 0 and lib.als_gram_tiles_bytes(64) == 40 * 64 * 4 + 256 and lib.als_gram_partial_bytes(64) == 40 * 64 * 4
    assert lib.als_half_sweep_workspace_bytes(10, 0, 1000) == 0


@pytest.mark.parametrize("chunk", [64, 2048])
def test_plan_covers_every_row_once(chunk):
    from recommender_amd import build_plan
    rng = np.random.default_rng(3)
    lens = np.concatenate([rng.integers(0, 300, size=500), [0, 0, 5000, 64, 65, chunk, chunk + 1, 3 * chunk]])
    ptr = np.zeros(len(lens) + 1, dtype=np.int64); np.cumsum(lens, out=ptr[1:])
    items, long_rows, n_slots = build_plan(ptr, chunk)
    direct = items[items[:, 1] < 0]
    chunks = items[items[:, 1] >= 0]
    assert sorted(direct[:, 0].tolist() + long_rows[:, 0].tolist()) == list(range(len(lens)))
    assert np.all(lens[direct[:, 0]] <= chunk) and np.all(lens[long_rows[:, 0]] > chunk)
    assert np.all(np.diff(lens[direct[:, 0]]) <= 0)                      # longest first
    assert n_slots == len(chunks) and sorted(chunks[:, 2].tolist()) == list(range(n_slots))
    for r, first, n, _ in long_rows:
        assert n == -(-lens[r] // chunk)
        mine = chunks[chunks[:, 0] == r]
        assert mine[:, 1].tolist() == list(range(n)) and mine[:, 2].tolist() == list(range(first, first + n))
    assert len(items) == len(direct) + n_slots
    # every item carries its own entry range: begin (int64 in words 4-5) and count (word 3)
    begin = items[:, 4:6].copy().view(np.int64)[:, 0]
    assert np.array_equal(begin[items[:, 1] < 0], ptr[direct[:, 0]]) and np.array_equal(direct[:, 3], lens[direct[:, 0]])
    assert np.array_equal(begin[items[:, 1] >= 0], ptr[chunks[:, 0]] + chunks[:, 1].astype(np.int64) * chunk)
    assert np.array_equal(chunks[:, 3], np.minimum(chunk, lens[chunks[:, 0]] - chunks[:, 1] * chunk))


def test_init_factors_match_reference_streams(golden):
    from recommender_amd.als import init_factors
    g = golden("g2_powerlaw_train.npz")
    N, M = (int(x) for x in g["U_shape"])
    for K in (8, 20, 64):
        for conv in ("ref", "tf"):
            X0, Y0 = init_factors(N, M, K, conv)
            assert np.array_equal(X0, g["K%d_%s_X0" % (K, conv)]) and np.array_equal(Y0, g["K%d_%s_Y0" % (K, conv)])


def test_metrics_match_reference(golden):
    from recommender_amd import metrics
    g = golden("g4_metrics.npz")
    for k in range(g["rec"].shape[0]):
        rec = g["rec"][k][g["rec"][k] >= 0]; te = g["test"][k][g["test"][k] >= 0]; tr = g["train"][k][g["train"][k] >= 0]
        assert metrics.compute_ap(rec, te, tr) == pytest.approx(g["ap"][k], abs=1e-15)
        assert metrics.compute_auc(rec, te, tr) == pytest.approx(g["auc"][k], abs=1e-15)
    assert metrics.rmse(g["rmse_x"], g["rmse_y"]) == pytest.approx(float(g["rmse"]), abs=1e-15)


def test_balanced_ranges():
    from recommender_amd.als import balanced_ranges
    rng = np.random.default_rng(0)
    lens = rng.zipf(1.5, size=10000).clip(max=5000)
    ptr = np.zeros(len(lens) + 1, dtype=np.int64); np.cumsum(lens, out=ptr[1:])
    for parts in (1, 2, 3, 8):
        b = balanced_ranges(ptr, parts)
        assert b[0] == 0 and b[-1] == len(lens) and np.all(np.diff(b) >= 0) and len(b) == parts + 1
        share = np.diff(ptr[b])
        assert share.max() <= ptr[-1] / parts + lens.max()
    assert balanced_ranges(np.zeros(11, dtype=np.int64), 2).tolist() == [0, 5, 10]


def test_synthetic_generator_is_canonical_and_seeded():
    import torch
    from recommender_amd import data
    U, Ut = data.synth_matrix(300, 200, 5000, seed=11, device="cpu")
    U2, _ = data.synth_matrix(300, 200, 5000, seed=11, device="cpu")
    assert all(torch.equal(a, b) for a, b in zip(U, U2))
    ptr, col, val = (t.numpy() for t in U)
    assert ptr[-1] == 5000 == len(col) and val.min() >= 0.5 and val.max() <= 5.0
    for r in range(300):
        c = col[ptr[r]:ptr[r + 1]]
        assert np.all(np.diff(c) > 0)                                      # sorted, duplicate free
    A = data.to_scipy(U, (300, 200)); At = data.to_scipy(Ut, (200, 300))
    assert (A.T != At).nnz == 0


def test_blocked_solve_index_logic_in_emulation():
    """The lane/register index maps of the in-register block Cholesky (MFMA accumulator layout, DPP
    row broadcasts, XtZ tile products), emulated in numpy lane by lane, solve SPD systems for every
    tile-grid size the kernels instantiate."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("emul", os.path.join(ROOT, "tools", "emul_blocked_solve.py"))
    emul = importlib.util.module_from_spec(spec); spec.loader.exec_module(emul)
    for VW, f in [(1, 5), (2, 20), (4, 64), (8, 100)]:
        assert emul.run(VW, f, seed=VW) < 2e-4


def test_csv_ingest_matches_reference_normalisation(tmp_path):
    """Raw CSV -> dense ids in sorted-key order (what pandas category codes give, the reference's
    normalizeDf, utils.py:19-33), 80/20 split, CSR [N, M] float64 with duplicates summed
    (ItemMetadata.py:402-405); the saved folder reads back to the same matrices."""
    import pandas as pd
    import scipy.sparse as sps
    from recommender_amd.ingest import ExplicitDataFromCSV
    rng = np.random.default_rng(4)
    n = 500
    df = pd.DataFrame({"userId": rng.choice([3, 17, 4, 99, 1000, 42], size=n),
                       "movieId": rng.choice([10, 7, 260, 1196, 5, 8, 31, 2], size=n),
                       "rating": rng.integers(1, 11, size=n) * 0.5, "timestamp": np.arange(n)})
    csv = tmp_path / "ratings.csv"; df.to_csv(csv, index=False)
    d = ExplicitDataFromCSV(False, ratings_csv=str(csv), seed=1)
    assert (d.N, d.M, d.Nranked) == (6, 8, n)
    assert np.array_equal(d.user, df["userId"].astype("category").cat.codes.to_numpy())
    assert np.array_equal(d.item, df["movieId"].astype("category").cat.codes.to_numpy())
    tr, te = d.make_training_datasets(dtype="sparse")
    assert tr.shape == te.shape == (6, 8) and tr.dtype == np.float64
    assert abs(int((d.split == 0).sum()) - 0.8 * n) <= 1
    whole = sps.csr_matrix((d.rating, (d.user, d.item)), shape=(6, 8))
    assert np.allclose((tr + te).toarray(), whole.toarray())
    d.save(str(tmp_path / "saved"))
    d2 = ExplicitDataFromCSV(True, data_folder=str(tmp_path / "saved"))
    tr2, te2 = d2.make_training_datasets(dtype="sparse")
    assert (d2.N, d2.M) == (6, 8) and (tr2 != tr).nnz == 0 and (te2 != te).nnz == 0
    assert np.array_equal(d2.user_keys, [3, 4, 17, 42, 99, 1000])


def test_every_stream_lookup_sits_inside_a_device_guard():
    """The library is handed torch's CURRENT stream of the CURRENT device; an operator whose tensors live on another
    device must therefore look the stream up inside `with torch.cuda.device(<that tensor's device>)`, or its kernels
    run on device 0's stream with another device's pointers.  Checked on the source: every `_stream()` call in
    recommender_amd/ops.py has such a `with` among its ancestors."""
    import ast
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "recommender_amd", "ops.py")
    tree = ast.parse(open(path).read())
    parents = {}
    for node in ast.walk(tree):
        for child in ast.iter_child_nodes(node):
            parents[child] = node

    def is_guard(w):
        for item in w.items:
            c = item.context_expr
            if isinstance(c, ast.Call) and ast.unparse(c.func) == "torch.cuda.device":
                return True
        return False

    calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "_stream"]
    assert len(calls) >= 8          # gram x3, the sweeps, scoring, top-k ...
    for call in calls:
        node, guarded = call, False
        while node in parents:
            node = parents[node]
            if isinstance(node, ast.With) and is_guard(node):
                guarded = True
                break
        assert guarded, "ops.py:%d: _stream() outside `with torch.cuda.device(...)`" % call.lineno


def test_strong_scaling_shards_tile_the_one_matrix():
    """bench.py's fixed-size workloads (wl["blocks"]) are the same matrix for any number of GPUs: the user and item
    shards of 2 and 3 ranks, cut by stored entries, concatenate to the single-rank CSR bit for bit, and no shard
    holds much more than its share of the entries."""
    import importlib.util
    import torch
    from recommender_amd import data
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    wl = dict(data.WORKLOADS["c3_full_small_syn"], n_users=500, n_items=300, nnz=6000, blocks=4)
    U1, Ut1, ub1, ib1, N, M, nnz = bench.build_shards(wl, 1, 0, torch.device("cpu"))
    assert (N, M) == (2000, 300) and ub1 == [0, N] and ib1 == [0, M] and int(U1[0][-1]) == nnz == int(Ut1[0][-1])
    for world in (2, 3):
        parts = [bench.build_shards(wl, world, r, torch.device("cpu")) for r in range(world)]
        ub, ib = parts[0][2], parts[0][3]
        assert all(p[2] == ub and p[3] == ib and p[4:] == (N, M, nnz) for p in parts)
        assert ub[0] == 0 and ub[-1] == N and ib[0] == 0 and ib[-1] == M
        for side, whole, bounds in ((0, U1, ub), (1, Ut1, ib)):
            ptr = torch.cat([torch.zeros(1, dtype=torch.int64)] +
                            [p[side][0][1:] + int(whole[0][bounds[r]]) for r, p in enumerate(parts)])
            assert torch.equal(ptr, whole[0])
            assert torch.equal(torch.cat([p[side][1] for p in parts]), whole[1])
            assert torch.equal(torch.cat([p[side][2] for p in parts]), whole[2])
            share = [int(whole[0][bounds[r + 1]] - whole[0][bounds[r]]) for r in range(world)]
            assert max(share) < 1.25 * nnz / world, share
