"""Implicit-ALS training engine (Hu-Koren-Volinsky) on MI355X.

Host-side mirror of the reference's algorithm layer, reclibwh/core/ALS.py:

  init_factors        ALSRef.__init__ (ALS.py:56-65) / als_initizalize (ALS.py:248-260)
  als_update_cache    ALS.py:235-246
  run_als_epoch       ALS.py:201-233   (one half-sweep; returns mean |delta| per element)
  ALSEngine.fit       ALSTrainer.fit (ALS.py:320-357) / ALSRef.train (ALS.py:122-141)

One process drives one GPU.  With torch.distributed initialised (backend "nccl" = RCCL on
ROCm, "gloo" in CPU tests) every rank keeps full replicas of X and Y, sweeps its own
contiguous shard of user rows (then item rows) and the shards are all-gathered (the user shard
chunk by chunk, beside the sweep of the next chunk); the Gram is summed from fixed 65 536-row partials in row order on every rank, so
factors are bit-identical to a single-GPU run for any world size.
"""
import numpy as np
import torch

from . import _lib, ops


# ----------------------------------------------------------------------------- init
def init_factors(n_users, n_items, f, convention="ref", seed=42):
    """Initial factors ~ N(0, 1/sqrt(f)) from numpy's legacy MT19937 global stream, float64.

    "ref": re-seed before X and before Y, as ALSRef.__init__ does (ALS.py:59-65).
    "tf":  seed once, draw X then Y, as als_initizalize does (ALS.py:255-257); pass the
           layer row counts (N+1, M+1 in the training script, ALS.py:456) to reproduce it.
    """
    if convention not in ("ref", "tf"):
        raise ValueError("convention must be 'ref' or 'tf'")
    np.random.seed(seed)
    X = np.random.normal(0, 1 / np.sqrt(f), size=(n_users, f))
    if convention == "ref":
        np.random.seed(seed)
    Y = np.random.normal(0, 1 / np.sqrt(f), size=(n_items, f))
    return X, Y


def balanced_ranges(rowptr, parts):
    """Split rows into `parts` contiguous ranges with about equal stored entries (prefix sum
    on the row pointers).  rowptr: host int64 array.  Returns parts+1 boundaries."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n = rowptr.shape[0] - 1
    total = int(rowptr[-1])
    if total == 0:
        return np.linspace(0, n, parts + 1).astype(np.int64)
    targets = (np.arange(1, parts, dtype=np.float64) * total / parts)
    cuts = np.searchsorted(rowptr, targets, side="left").astype(np.int64)
    b = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(np.clip(b, 0, n))


# ----------------------------------------------------------------------------- operators
def als_update_cache(Y, lamb, cache=None, group=None, backend=ops):
    """Y2 <- Y^T Y over all rows of Y (pad/phantom rows included), LambdaI <- lamb I
    (ALS.py:235-246).  With a process group, each rank reduces its share of the fixed
    65 536-row partials and the partials are all-gathered and summed in row order.
    `backend` supplies gram / gram_partials / gram_finalize (default: the HIP operators)."""
    world = torch.distributed.get_world_size(group) if group is not None else 1
    n, f = Y.shape
    n_super = -(-n // _lib.GRAM_SUPER_ROWS)
    if world == 1 or n_super < 2 * world:
        return backend.gram(Y, lamb, out=cache)
    rank = torch.distributed.get_rank(group)
    cuts = [(n_super * r) // world for r in range(world + 1)]
    s0, s1 = cuts[rank], cuts[rank + 1]
    mine = backend.gram_partials(Y, s0 * _lib.GRAM_SUPER_ROWS, min(n, s1 * _lib.GRAM_SUPER_ROWS))
    cnt_max = max(cuts[r + 1] - cuts[r] for r in range(world))
    elems = mine.shape[1]
    xdev = "cpu" if _host_staged(Y, group) else Y.device
    send = torch.zeros((cnt_max, elems), dtype=mine.dtype, device=xdev)
    send[: mine.shape[0]] = mine
    recv = [torch.empty((cnt_max, elems), dtype=mine.dtype, device=xdev) for _ in range(world)]
    torch.distributed.all_gather(recv, send, group=group)       # a few KB per rank
    parts = [recv[r][: cuts[r + 1] - cuts[r]] for r in range(world)]
    return backend.gram_finalize(torch.cat(parts, 0).to(Y.device).contiguous(), f, lamb, out=cache)


def run_als_epoch(csr, X, Y, cache, alpha, absdiff=None, info=None, backend=ops, cg_steps=0, cg_form="auto",
                  opts=None, short_rows=False):
    """One half-sweep over the rows of `csr`: re-solves those rows of X in place against the
    fixed Y and its cache (ALS.py:201-233).  Returns the device scalar sum|X_old - X_new|
    (float64); the reference's return value is that divided by n_rows * f.
    cg_steps > 0: that many conjugate-gradient steps per row instead of the exact solve.
    opts: ops.sweep_opts(...) - per-call kernel options (accumulate mode, timing taps).
    short_rows: rows with fewer entries than factors are solved in their entry-space form from cache.whitened
    (ops.whiten; exact solve only)."""
    if cg_steps:
        return backend.half_sweep_cg(csr, Y, cache, X, alpha, cg_steps, absdiff=absdiff, info=info, form=cg_form,
                                     opts=opts)
    if short_rows:
        return backend.half_sweep(csr, Y, cache, X, alpha, absdiff=absdiff, info=info, opts=opts, short_rows=True)
    return backend.half_sweep(csr, Y, cache, X, alpha, absdiff=absdiff, info=info, opts=opts)


def default_cg_steps(f):
    """Conjugate-gradient steps per row and half-sweep at which 15 iterations give the exact solve's MAP@200,
    precision@10 and held-out RMSE to 3 decimals: max(10, ceil(f / 6)).  Pinned at f = 20 (10 steps; 8 miss
    precision@10), f = 128 (22; 16 miss precision@10 by 1.6e-3, 18 hold) and f = 256 (43; 36 miss it by 6e-4, 42 hold)
    by the metric-level tests of tests/test_gpu_recommender.py; the truncated iteration needs about f / 6 steps
    whatever the preconditioning (with the Gram's Cholesky factor as preconditioner 8 steps reach the accuracy of 10
    plain ones at f = 128: not worth a second factor table)."""
    return max(10, -(-int(f) // 6))


def short_rows_pay(n_short, n_rows, n_opp, f):
    """Whether a half-sweep should solve its short rows (at most ops.short_rows_max_entries(f) entries) in their
    entry-space form.  The form needs the whitened table of the fixed factor - a factorisation of the Gram (0.2 ms)
    and one GEMM over its n_opp rows (0.5 ms per million at f = 128) per half-sweep - and saves about 12 ns per
    short row at f = 128: with a margin of two it pays from 32 768 + n_opp / 12 short rows on.  At f <= 64 the f x f
    solve is cheap enough that the two come out even (measured on the 162 541 x 59 047 matrix: sweep -7 %, whitening
    +7 %), so there the f x f form stays.  n_opp >= 16 f keeps the Gram that is factored well away from rank
    deficiency.  All arguments are totals over the whole matrix, never a rank's share, so every rank of a sharded
    run decides alike and the factors do not depend on the sharding."""
    return bool(ops.short_rows_max_entries(f) > 0 and f > 64 and n_opp >= 16 * f and n_short >= 32768 + n_opp // 12)


def _host_staged(T, group):
    """RCCL ("nccl") moves device tensors directly and in stream order.  Any other backend (gloo in
    tests and one-GPU rehearsals) gets host tensors: gloo's device path never returns from a collective
    issued under a side stream that waits for a kernel on the default stream - with one rank as with
    two, and with nothing of this package involved (tools/gloo_device_probe.py, variants C and D)."""
    return T.is_cuda and torch.distributed.get_backend(group) != "nccl"


_PADDED_RECV = {}      # (device, dtype, world, trailing shape) -> (receive, send) buffers of the padded form, grown on demand
UNEVEN_FORMS = ("padded", "list")


def _all_gather_row_blocks(T, starts, sizes, rank, group, uneven="padded", before=None):
    """All-gather of one row block per rank of the replicated T, in place: rank r owns
    T[starts[r] : starts[r] + sizes[r]] and every rank ends up with all of them:

      * RCCL, equal blocks that tile T: all_gather_into_tensor straight into T;
      * RCCL, equal blocks elsewhere: all_gather into the list of T's own row-block views;
      * RCCL, blocks of different sizes, uneven="padded": padded to the longest through ONE receive buffer per
        (device, dtype, world, row shape) that is kept and only ever grows (one all_gather_into_tensor, then each
        block is copied to its place: a second copy of the gathered rows and one more pass over them);
        uneven="list": all_gather into the views, which torch issues as one broadcast per rank inside one RCCL
        group - every receive lands in place, no staging buffer, no copy pass;
      * other backends (gloo in tests): one broadcast per rank of host copies - they get host tensors (_host_staged).
    before: called immediately in front of the collective call, behind any packing of this rank's
    block - the overlapped sweep records there the event its main stream waits for (ALSEngine._sweep_users)."""
    if uneven not in UNEVEN_FORMS:
        raise ValueError("uneven must be one of %s" % (UNEVEN_FORMS,))
    world = len(sizes)
    if max(sizes) == 0:
        return
    mine = T[starts[rank]: starts[rank] + sizes[rank]]
    equal = len(set(sizes)) == 1
    if _host_staged(T, group) or not T.is_cuda:
        host = T.is_cuda
        if before is not None:      # (no collective kernel to make room for; called so that rehearsals walk the same code)
            before()
        for r in range(world):
            if not sizes[r]:
                continue
            blk = T[starts[r]: starts[r] + sizes[r]]
            if host:
                buf = blk.cpu() if r == rank else torch.empty(blk.shape, dtype=T.dtype, device="cpu")
            else:
                buf = blk
            torch.distributed.broadcast(buf, src=torch.distributed.get_global_rank(group, r), group=group)
            if host and r != rank:
                blk.copy_(buf)
        return
    if equal and T.is_contiguous() and all(starts[r] == r * sizes[0] for r in range(world)) and sizes[0] * world == T.shape[0]:
        if before is not None:
            before()
        torch.distributed.all_gather_into_tensor(T, mine, group=group)      # blocks tile T: no staging at all
        return
    if equal or uneven == "list":
        if before is not None:
            before()
        torch.distributed.all_gather([T[starts[r]: starts[r] + sizes[r]] for r in range(world)], mine, group=group)
        return
    _all_gather_padded(T, starts, sizes, rank, group, before)


def _all_gather_padded(T, starts, sizes, rank, group, before=None):
    """Row blocks of different sizes, padded to the longest: ONE all_gather_into_tensor through a receive and a send
    buffer kept per (device, dtype, world, row shape) - sized for the longest block seen so far, so the chunks of an
    overlapped sweep share one pair -, then every other rank's block is copied to its place in T."""
    world = len(sizes)
    cap = max(sizes)
    tail = tuple(T.shape[1:])
    key = (T.device, T.dtype, world, tail)
    bufs = _PADDED_RECV.get(key)
    if bufs is None or bufs[1].shape[0] < cap:        # a send buffer of its own: no reliance on in-place collectives
        bufs = _PADDED_RECV[key] = (torch.empty((world * cap,) + tail, dtype=T.dtype, device=T.device),
                                    torch.zeros((cap,) + tail, dtype=T.dtype, device=T.device))
    recv = bufs[0][: world * cap].view((world, cap) + tail)
    send = bufs[1][:cap]
    send[: sizes[rank]] = T[starts[rank]: starts[rank] + sizes[rank]]
    if before is not None:
        before()
    torch.distributed.all_gather_into_tensor(recv.view((world * cap,) + tail), send, group=group)
    for r in range(world):
        if r != rank and sizes[r]:
            T[starts[r]: starts[r] + sizes[r]].copy_(recv[r, : sizes[r]])


def _all_gather_rows(T, bounds, rank, group, uneven="padded"):
    """In-place all-gather of the row shards [bounds[r], bounds[r+1]) of the replicated T."""
    world = len(bounds) - 1
    _all_gather_row_blocks(T, [int(bounds[r]) for r in range(world)],
                           [int(bounds[r + 1] - bounds[r]) for r in range(world)], rank, group, uneven)


def _chunk_cuts(n, parts):
    """Boundaries of `parts` row chunks of a shard of n rows, sizes in the ratio parts : parts-1 : ... : 1
    (the exchange of the last chunk is the one no sweep hides, so it is the smallest).  Every rank can
    compute every other rank's cuts from the shard boundaries alone."""
    total = parts * (parts + 1) // 2
    cuts, acc = [0], 0
    for c in range(parts):
        acc += parts - c
        cuts.append((n * acc) // total)
    return cuts


# ----------------------------------------------------------------------------- engine
class ALSEngine:
    """Alternating least squares over a user x item CSR and its transpose.

    csr_users / csr_items are this rank's row shards (ops.DeviceCSR): rows
    [user_bounds[rank], user_bounds[rank+1]) of U and [item_bounds[rank], ...) of U^T, with
    GLOBAL column ids.  X [n_users(+phantom), f] and Y [n_items(+phantom), f] are full
    replicas.  Single GPU: bounds = [0, n].
    """

    def __init__(self, csr_users, csr_items, X, Y, lamb, alpha, user_bounds=None, item_bounds=None, group=None,
                 backend=ops, overlap_chunks=None, item_overlap_chunks=None, solver="auto", cg_steps=None, cg_form="auto", short_rows="auto",
                 uneven_exchange="padded", stream_reserve_cus=None, exchange_gap_ticks=20000):
        self.group = group
        if uneven_exchange not in UNEVEN_FORMS:
            raise ValueError("uneven_exchange must be one of %s" % (UNEVEN_FORMS,))
        # how RCCL exchanges shards of different sizes: "padded" (one all_gather_into_tensor through a kept
        # buffer + a copy pass) or "list" (receives land in place; see _all_gather_row_blocks)
        self.uneven_exchange = uneven_exchange
        self.backend = backend      # the HIP operators; tests inject a CPU stand-in to exercise the sharding logic
        self.world = torch.distributed.get_world_size(group) if group is not None else 1
        self.rank = torch.distributed.get_rank(group) if group is not None else 0
        self.csr_users, self.csr_items = csr_users, csr_items
        self.X, self.Y = X, Y
        self.f = X.shape[1]
        self.lamb, self.alpha = float(lamb), float(alpha)
        self.user_bounds = [0, csr_users.shape[0]] if user_bounds is None else [int(b) for b in user_bounds]
        self.item_bounds = [0, csr_items.shape[0]] if item_bounds is None else [int(b) for b in item_bounds]
        if len(self.user_bounds) != self.world + 1 or len(self.item_bounds) != self.world + 1:
            raise ValueError("need world+1 shard boundaries")
        u0, u1 = self.user_bounds[self.rank], self.user_bounds[self.rank + 1]
        i0, i1 = self.item_bounds[self.rank], self.item_bounds[self.rank + 1]
        if csr_users.shape[0] != u1 - u0 or csr_items.shape[0] != i1 - i0:
            raise ValueError("CSR shard row counts do not match the shard boundaries")
        self.n_users, self.n_items = self.user_bounds[-1], self.item_bounds[-1]
        if X.shape[0] < self.n_users or Y.shape[0] < self.n_items:
            raise ValueError("factor matrices are smaller than the rating matrix")
        # "cholesky": exact solve (f <= 128).  "cg": cg_steps conjugate-gradient steps per row and
        # half-sweep, warm-started (any f <= 256).  "auto": exact where it exists.
        if solver not in ("auto", "cholesky", "cg"):
            raise ValueError("solver must be 'auto', 'cholesky' or 'cg'")
        if solver == "auto":
            solver = "cholesky" if self.f <= _lib.MAX_FACTORS_EXACT else "cg"
        if solver == "cholesky" and self.f > _lib.MAX_FACTORS_EXACT:
            raise ValueError("the exact solve supports f <= %d; use solver='cg'" % _lib.MAX_FACTORS_EXACT)
        self.solver = solver
        if cg_steps is None:
            cg_steps = default_cg_steps(self.f)
        self.cg_steps = int(cg_steps) if solver == "cg" else 0
        if solver == "cg" and self.cg_steps < 1:
            raise ValueError("cg_steps must be >= 1")
        self.cg_form = cg_form
        # short_rows: "auto" (short_rows_pay on the matrix's totals), True or False - whether the user / item
        # half-sweep solves rows with fewer entries than factors in their entry-space form (exact solver only)
        self.short = self._decide_short_rows(short_rows)
        self._share_rating_bound()
        self.sweep_opts = None         # callable("users" | "items") -> ops.sweep_opts(...) for that launch, or None
        self.cache_Y = backend.GramCache(self.f, lamb, X.device)
        self.cache_X = backend.GramCache(self.f, lamb, X.device)
        self._dX = torch.zeros(1, dtype=torch.float64, device=X.device)
        self._dY = torch.zeros(1, dtype=torch.float64, device=X.device)
        self.info = torch.zeros(1, dtype=torch.int32, device=X.device)
        self.trace = []
        # Multi-GPU: the user sweep runs as `overlap_chunks` row chunks; the all-gather of chunk c
        # runs on a side stream beside the sweep of chunk c+1, so only the last chunk's exchange is
        # exposed.  A row's result does not depend on which launch computes it, so factors are
        # unchanged bit for bit.  (X is the large factor; Y is exchanged in one piece.)
        if overlap_chunks is None:
            overlap_chunks = 4 if self.world > 1 else 1
        self.overlap_chunks = max(1, int(overlap_chunks))      # the same on every rank
        # the item sweep likewise, where the item factor is large enough for its exchange to matter (16 MB a rank)
        if item_overlap_chunks is None:
            big = self.n_items * self.f * 4 >= self.world * (16 << 20)
            item_overlap_chunks = 4 if (self.world > 1 and big) else 1
        self.item_overlap_chunks = max(1, int(item_overlap_chunks))
        self._parts_cache = {}
        self._side = None
        # Overlap of the exchange of row chunk c with the sweep of chunk c + 1, as measured on one MI355X with a stand-in
        # of RCCL's kernel (tools/overlap_probe.py; its workgroups of 256 threads and 261-280 registers need a wholly
        # free compute unit each):
        #  * the collective and the next sweep become runnable at the same instant, when the chip drains - and the sweep's
        #    tens of thousands of small workgroups take every unit first, after which the collective finds no free
        #    unit until that sweep has ENDED (f = 64: done 10.8 ms after the event instead of 4.5).  Hence the gap: the
        #    main stream waits for an event the side stream records immediately in front of the collective call, plus a
        #    few microseconds of a one-wave spin kernel - the collective's workgroups are resident before the sweep
        #    starts, and the sweep runs on the units that are left;
        #  * the persistent row kernel of 64 < f <= 128 hands its rows out while it runs, so the units the collective
        #    holds cost their share of the chip and no more;
        #  * stream_reserve_cus > 0 makes that kernel leave units free in the chunks that have an exchange beside them
        #    (als_sweep_opts_t.reserve_cus) - the remedy for a collective that becomes runnable while the kernel is
        #    already resident.  With the gap it is not needed (measured: no gain, -12 % on the kernel), so 0.
        if stream_reserve_cus is None:
            stream_reserve_cus = 0
        self.stream_reserve_cus = int(stream_reserve_cus)
        self.exchange_gap_ticks = int(exchange_gap_ticks)      # spin kernel of the gap, in clock ticks; 0: event only; < 0: no gap
        # benchmarks set this to {"x_wait": [], "y_gather": []}: per step, event pairs around the main stream's wait for
        # the user exchange still in flight when the user sweep has been queued, and around the item exchange
        self.exchange_taps = None

    def _decide_short_rows(self, mode):
        off = {"users": False, "items": False}
        if mode not in ("auto", True, False):
            raise ValueError("short_rows must be 'auto', True or False")
        lim = ops.short_rows_max_entries(self.f) if self.backend is ops else 0
        if mode is False or self.solver != "cholesky" or lim == 0 or self.f % 4 or not self.X.is_cuda:
            if mode is True:
                raise ValueError("short_rows=True needs the exact solver, 32 < f <= 128 and f a multiple of 4")
            return off
        if mode is True:
            return {"users": True, "items": True}
        nb = lim // 16
        mine = [int(c.short_bounds[-1] - c.short_bounds[nb]) for c in (self.csr_users, self.csr_items)]
        if self.world > 1:          # totals over the ranks: every rank takes the same decision
            t = torch.tensor(mine, dtype=torch.int64, device="cpu" if _host_staged(self.X, self.group) else self.X.device)
            torch.distributed.all_reduce(t, group=self.group)
            mine = [int(v) for v in t.cpu().tolist()]
        return {"users": short_rows_pay(mine[0], self.n_users, self.Y.shape[0], self.f),
                "items": short_rows_pay(mine[1], self.n_items, self.X.shape[0], self.f)}

    def _share_rating_bound(self):
        """The fp16 x 2 accumulate scales its operands by a power of two taken from an upper bound of the ratings.  Both
        CSRs of every rank get the largest rating of the WHOLE matrix, so that the scale - and with it every bit of
        the factors - does not depend on how the rows were cut into shards."""
        csrs = [c for c in (self.csr_users, self.csr_items) if hasattr(c, "max_rating")]
        if not csrs:
            return
        m = max(c.max_rating for c in csrs)
        if self.world > 1:
            t = torch.tensor([m], dtype=torch.float64, device="cpu" if _host_staged(self.X, self.group) else self.X.device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.group)
            m = float(t.cpu()[0])
        for c in csrs:
            c.max_rating = m

    def _side_of(self, which):
        """(csr shard, own factor, fixed factor, its cache, shard bounds, chunks, |delta| buffer, rows of the whole factor)"""
        if which == "users":
            return (self.csr_users, self.X, self.Y, self.cache_Y, self.user_bounds, self.overlap_chunks, self._dX, self.n_users)
        return (self.csr_items, self.Y, self.X, self.cache_X, self.item_bounds, self.item_overlap_chunks, self._dY, self.n_items)

    def _parts(self, which="users"):
        """[(row0, row1, csr chunk or None if empty, |delta| buffer)], chunk c of this rank's user (item) shard."""
        if which not in self._parts_cache:
            csr, own, _, _, _, chunks, dbuf, _ = self._side_of(which)
            n = csr.shape[0]
            if chunks == 1:
                self._parts_cache[which] = [(0, n, csr if n else None, dbuf)]
            else:
                cuts = _chunk_cuts(n, chunks)
                self._parts_cache[which] = [
                    (cuts[c], cuts[c + 1], csr.row_range(cuts[c], cuts[c + 1]) if cuts[c + 1] > cuts[c] else None,
                     torch.zeros(1, dtype=torch.float64, device=own.device))
                    for c in range(chunks)]
        return self._parts_cache[which]

    def launches_per_step(self):
        """Half-sweep calls of one step (benchmarks size their event pools by it)."""
        return len(self._parts("users")) + len(self._parts("items"))

    def _sweep_side(self, which):
        """The half-sweep over this rank's user (item) shard and the exchange of the solved rows.  With several ranks
        and `chunks` > 1 the shard is swept as row chunks and the all-gather of chunk c runs on a side stream beside
        the sweep of chunk c + 1 (the gap: see __init__); the main stream waits for the exchanges at the end."""
        csr_all, own, fixed, cache, bounds, chunks, dsum, n_all = self._side_of(which)
        r0_shard = bounds[self.rank]
        parts = self._parts(which)
        chunked = self.world > 1 and chunks > 1
        on_gpu = own.is_cuda
        if chunked and on_gpu and self._side is None:
            self._side = torch.cuda.Stream(device=own.device)
        done = []
        for c, (r0, r1, csr, dbuf) in enumerate(parts):
            if csr is not None:
                opts = self.sweep_opts(which) if self.sweep_opts else None
                if chunked and c > 0 and self.stream_reserve_cus and self.backend is ops:
                    if opts is None:
                        opts = ops.sweep_opts()
                    opts.reserve_cus = self.stream_reserve_cus
                run_als_epoch(csr, own[r0_shard + r0:r0_shard + r1], fixed, cache, self.alpha, dbuf, self.info,
                              self.backend, self.cg_steps, self.cg_form, opts, self.short[which])
            else:
                dbuf.zero_()
            if not chunked:
                continue
            # chunk c of every rank: rows [bounds[r] + cuts_r[c], bounds[r] + cuts_r[c+1])
            starts, sizes = [], []
            for r in range(self.world):
                cuts = _chunk_cuts(bounds[r + 1] - bounds[r], chunks)
                starts.append(bounds[r] + cuts[c]); sizes.append(cuts[c + 1] - cuts[c])
            if on_gpu:
                main = torch.cuda.current_stream(own.device)
                self._side.wait_stream(main)
                released = []
                with torch.cuda.stream(self._side):
                    _all_gather_row_blocks(own, starts, sizes, self.rank, self.group, self.uneven_exchange,
                                           before=lambda: released.append(self._side.record_event()))
                    done.append(self._side.record_event())
                if released and c + 1 < len(parts) and self.exchange_gap_ticks >= 0:
                    # the gap: the next chunk's sweep starts once the collective has been released, and a moment later
                    main.wait_event(released[0])
                    if self.exchange_gap_ticks > 0 and hasattr(torch.cuda, "_sleep"):
                        torch.cuda._sleep(self.exchange_gap_ticks)
            else:
                _all_gather_row_blocks(own, starts, sizes, self.rank, self.group, self.uneven_exchange)
        tap = self._tap_begin("x_wait" if which == "users" else "y_gather") if self.world > 1 else None
        for ev in done:
            torch.cuda.current_stream(own.device).wait_event(ev)
        if len(parts) > 1:
            torch.sum(torch.cat([p[3] for p in parts]), 0, keepdim=True, out=dsum)
        if self.world > 1 and not chunked:
            _all_gather_rows(own[:n_all], bounds, self.rank, self.group, self.uneven_exchange)
        self._tap_end(tap)

    def _sweep_users(self):
        self._sweep_side("users")

    def _tap_begin(self, which):
        if self.exchange_taps is None or not self.X.is_cuda:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.exchange_taps[which].append((e0, e1))
        return e1

    def _tap_end(self, e1):
        if e1 is not None:
            e1.record()

    # one ALS iteration = X half-sweep + Y half-sweep, Gram refreshes included
    def step(self):
        als_update_cache(self.Y, self.lamb, self.cache_Y, self.group, self.backend)
        if self.short["users"]:
            self.backend.whiten(self.Y, self.cache_Y, info=self.info)
        self._sweep_users()
        als_update_cache(self.X, self.lamb, self.cache_X, self.group, self.backend)
        if self.short["items"]:
            self.backend.whiten(self.X, self.cache_X, info=self.info)
        self._sweep_side("items")

    def last_delta(self):
        """mean|dX| + mean|dY| of the last step, as ALSRef.train's trace (ALS.py:136-139).
        Synchronises with the device."""
        d = torch.stack([self._dX[0] / (self.n_users * self.f), self._dY[0] / (self.n_items * self.f)])
        if self.world > 1:
            if _host_staged(d, self.group):
                d = d.cpu()
            torch.distributed.all_reduce(d, group=self.group)
        return float(d.sum().item())

    def fit(self, steps, callback=None):
        """`steps` iterations; records the reference's convergence trace; `callback(step)` runs
        after each iteration (ALS.py:352)."""
        for s in range(steps):
            self.step()
            self.trace.append(self.last_delta())
            if callback is not None:
                callback(s)
        if int(self.info.item()) != 0:
            raise FloatingPointError("%d row systems were not positive definite in float32 (lamb too small?)"
                                     % int(self.info.item()))
        return self.trace
