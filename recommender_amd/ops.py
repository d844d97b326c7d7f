"""Device-side operators of the ALS hot path: thin torch wrappers over the C ABI.

  DeviceCSR      the CSR the kernels walk (replaces the padded batches of
                 reclibwh/data/iterators.py:160-193 + reclibwh/utils/utils.py:142-173)
  gram           als_update_cache        (reclibwh/core/ALS.py:235-246)
  half_sweep     run_als_epoch / run_als_step / als_run_regression
                                          (reclibwh/core/ALS.py:160-233)
  half_sweep_cg  the same half-sweep with a fixed number of conjugate-gradient steps per row
                 instead of the exact solve (no reference counterpart; the path for f > 128)

torch is used for device memory and streams only; every computation is a HIP kernel
in libals_hip.so.  Arguments are validated here (dtype, contiguity, device, index
range) so that the kernels can assume well-formed input.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t, dtype, name, dim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError("%s must be a CUDA/HIP tensor" % name)
    if t.dtype != dtype:
        raise ValueError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    if dim is not None and t.dim() != dim:
        raise ValueError("%s must be %d-dimensional" % (name, dim))


def _index(t, dtype, name):
    """1-D index tensor the kernels read as a dense array: right dtype AND contiguous."""
    _need(t, dtype, name, 1)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous (a strided view would be read as a dense array)" % name)


def _same_device(first, *others):
    """Every tensor of one call must live on one device: the call is launched on that device's stream."""
    for t in others:
        if t is not None and t.device != first.device:
            raise ValueError("tensors of one call live on different devices (%s, %s)" % (first.device, t.device))


def _factor_ld(t, name):
    """Row-major [n, f] with unit inner stride; returns the leading dimension in elements."""
    _need(t, torch.float32, name, 2)
    if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        raise ValueError("%s must be row-major with unit inner stride" % name)
    return t.stride(0)


def sweep_opts(accumulate="default", row_kernel="auto", tap=None, reserve_cus=0):
    """Per-call options of the half-sweeps (als_sweep_opts_t).
    accumulate: "fp16x2" (= "default" for a DeviceCSR, which knows its largest rating: operands scaled by a power of
    two and cut into two fp16 pieces, three piece products on the fp16 matrix cores), "bf16x3" (three bf16 pieces,
    six products) or "fp32" (float32 matrix-core fma chain).
    row_kernel: "auto", "wave" or "stream" - which per-row kernel handles 64 < f <= 128.
    tap: (begin, end) torch.cuda.Event pair recorded around the call's per-row kernels.
    reserve_cus: compute units the persistent stream kernel leaves free for kernels of other streams (the RCCL
    exchange of a multi-GPU run); the results do not depend on it."""
    if accumulate not in _lib.ACCUMULATE_MODES:
        raise ValueError("accumulate must be one of %s" % sorted(_lib.ACCUMULATE_MODES))
    if row_kernel not in _lib.ROW_KERNELS:
        raise ValueError("row_kernel must be one of %s" % sorted(_lib.ROW_KERNELS))
    if int(reserve_cus) < 0:
        raise ValueError("reserve_cus must be >= 0")
    o = _lib.SweepOpts(accumulate=_lib.ACCUMULATE_MODES[accumulate], row_kernel=_lib.ROW_KERNELS[row_kernel],
                       reserve_cus=int(reserve_cus))
    if tap is not None:
        o.tap_begin, o.tap_end = tap[0].cuda_event, tap[1].cuda_event
        o._tap = tap            # keeps the events alive, and lets a call made of several library calls record them itself
    return o


def _opts_ref(opts):
    if opts is None:
        return None
    if not isinstance(opts, _lib.SweepOpts):
        raise ValueError("opts must come from ops.sweep_opts()")
    return ctypes.byref(opts)


# ----------------------------------------------------------------------------- plan (host only)
def default_chunk_nnz(f):
    """Entries per chunk item of a long row for a training run at factor count f.  A row with more entries is cut into
    chunk items whose partial sums a second kernel combines; the longer the chunks, the fewer partial sums - but a whole
    row is one wavefront's work, and one that outlasts the rest of the launch holds it up.  Measured on the 25 M-entry
    matrix (162 541 x 59 047; per iteration, chunk 2048 / 8192 / 16384 / 32768): f = 64 2.08 / 1.99 / 1.96 / 2.24 ms,
    f = 128 6.86 / 6.6 / 7.05 / 9.7 ms; the lock-step conjugate-gradient form of f > 128 gains 1 % from 8192.  The value depends on f alone - never on the matrix or a rank's shard of it - so
    that every shard of a distributed run is cut like the whole (a row's bits depend on its chunking)."""
    f = int(f)
    return 16384 if f <= 64 else 8192


def build_plan(indptr, chunk_nnz=_lib.DEFAULT_CHUNK_NNZ):
    """Row schedule for a CSR row-pointer array (host numpy int64).  Returns
    (items [n_items, 8] int32 - 32-byte als_item_t records -, long_rows [n_long, 4] int32, n_slots)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    n_rows = indptr.shape[0] - 1
    if n_rows < 0:
        raise ValueError("indptr must have at least one entry")
    ni, nl, ns = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    check(lib.als_plan_count(indptr.ctypes.data, n_rows, chunk_nnz, ctypes.byref(ni), ctypes.byref(nl),
                             ctypes.byref(ns)), "als_plan_count")
    items = np.empty((ni.value, 8), dtype=np.int32)
    long_rows = np.empty((max(nl.value, 1), 4), dtype=np.int32)
    check(lib.als_plan_fill(indptr.ctypes.data, n_rows, chunk_nnz, items.ctypes.data, long_rows.ctypes.data),
          "als_plan_fill")
    return items, long_rows[: nl.value], int(ns.value)


class DeviceCSR:
    """CSR matrix resident in HBM (int64 row pointers, int32 columns, float32 values) plus its
    row schedule.  `row_ids`, if given, maps CSR row u to the factor row it updates."""

    def __init__(self, indptr, indices, data, shape, device, chunk_nnz=_lib.DEFAULT_CHUNK_NNZ, row_ids=None):
        """indptr / indices / data: numpy arrays or torch tensors (host or device)."""
        dev = torch.empty(0, device=device).device           # normalised ("cuda" -> "cuda:0")
        n_rows, n_cols = int(shape[0]), int(shape[1])

        def as_dev(a, dtype):
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
            return t.to(device=dev, dtype=dtype).contiguous()

        rowptr = as_dev(indptr, torch.int64)
        col = as_dev(indices, torch.int32)
        val = as_dev(data, torch.float32)
        host_ptr = rowptr.cpu().numpy()
        if n_rows < 1:
            raise ValueError("CSR needs at least one row")
        if host_ptr.shape[0] != n_rows + 1 or host_ptr[0] != 0 or host_ptr[-1] != col.shape[0] \
                or col.shape[0] != val.shape[0] or col.dim() != 1:
            raise ValueError("inconsistent CSR arrays")
        if np.any(np.diff(host_ptr) < 0):
            raise ValueError("indptr must be non-decreasing")
        if col.numel() and (int(col.min()) < 0 or int(col.max()) >= n_cols):
            raise ValueError("column index out of range [0, %d)" % n_cols)
        if val.numel() and not bool(torch.all(val >= 0)):      # also rejects NaN
            raise ValueError("ratings must be >= 0 (confidence 1 + alpha r; the kernels take sqrt(alpha r))")
        self.max_rating = float(val.max()) if val.numel() else 0.0      # the operand bound of the fp16 x 2 accumulate
        self.shape = (n_rows, n_cols)
        self.nnz = int(col.shape[0])
        self.chunk_nnz = int(chunk_nnz)
        self.device = dev
        items, long_rows, self.n_slots = build_plan(host_ptr, chunk_nnz)
        self.n_items, self.n_long = items.shape[0], long_rows.shape[0]
        # whole rows are listed by entry count: short_bounds[j] = first item with at most 16 j entries (host array,
        # handed to the half-sweep when it solves short rows in their entry-space form)
        self.short_bounds = np.zeros(_lib.SHORT_MAX_GROUPS + 2, dtype=np.int64)
        check(lib.als_plan_short_bounds(host_ptr.ctypes.data, n_rows, self.chunk_nnz, self.short_bounds.ctypes.data),
              "als_plan_short_bounds")
        self.rowptr = rowptr
        if self.nnz == 0:   # keep the pointers non-null for a matrix without entries
            col = torch.zeros(1, dtype=torch.int32, device=dev)
            val = torch.zeros(1, dtype=torch.float32, device=dev)
        self.col, self.val = col, val
        self.items = torch.from_numpy(items).to(dev)
        self.long_rows = torch.from_numpy(long_rows).to(dev) if self.n_long else None
        self.row_ids = None
        if row_ids is not None:
            row_ids = np.ascontiguousarray(row_ids, dtype=np.int64)
            if row_ids.shape[0] != n_rows or (row_ids.size and row_ids.min() < 0):
                raise ValueError("row_ids must hold one non-negative id per CSR row")
            self.row_ids_max = int(row_ids.max())
            self.row_ids = torch.from_numpy(row_ids).to(dev)
        self._ws = {}

    @classmethod
    def from_scipy(cls, U, device, chunk_nnz=_lib.DEFAULT_CHUNK_NNZ, row_ids=None):
        import scipy.sparse as sps
        U = sps.csr_matrix(U)
        U.sort_indices()
        return cls(U.indptr, U.indices, U.data, U.shape, device, chunk_nnz, row_ids)

    def row_range(self, r0, r1):
        """Rows [r0, r1) as a DeviceCSR of their own (own schedule and workspace; the column and
        value arrays are views of this one's)."""
        r0, r1 = int(r0), int(r1)
        if not (0 <= r0 < r1 <= self.shape[0]):
            raise ValueError("bad row range")
        ptr = self.rowptr[r0:r1 + 1]
        lo, hi = int(ptr[0]), int(ptr[-1])
        ids = None if self.row_ids is None else self.row_ids[r0:r1].cpu().numpy()
        sub = DeviceCSR(ptr - lo, self.col[lo:hi], self.val[lo:hi], (r1 - r0, self.shape[1]), self.device,
                        self.chunk_nnz, ids)
        sub.max_rating = self.max_rating        # the whole matrix's bound: any cut of the rows accumulates at one scale
        return sub

    def transpose(self):
        """The CSR of the transposed matrix (canonical: sorted columns), built on the device with
        one key sort - what sps.csr_matrix(U.T) does on the host in the reference's training script
        (reclibwh/core/ALS.py:432)."""
        if self.row_ids is not None:
            raise ValueError("transpose of a row-mapped CSR is not defined")
        n_rows, n_cols = self.shape
        if self.nnz == 0:
            return DeviceCSR(np.zeros(n_cols + 1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32),
                             (n_cols, n_rows), self.device, self.chunk_nnz)
        rows = torch.repeat_interleave(torch.arange(n_rows, device=self.device, dtype=torch.int64),
                                       self.rowptr[1:] - self.rowptr[:-1])
        keys, order = torch.sort(self.col.to(torch.int64) * n_rows + rows)
        trow = torch.div(keys, n_rows, rounding_mode="floor")
        ptr = torch.zeros(n_cols + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(torch.bincount(trow, minlength=n_cols), 0, out=ptr[1:])
        return DeviceCSR(ptr, (keys - trow * n_rows).to(torch.int32), self.val[order], (n_cols, n_rows), self.device,
                         self.chunk_nnz)

    def lockstep_slabs(self, f, limit):
        """[(row0, row1, DeviceCSR of those rows)] and ONE workspace all of them fit: contiguous row slabs whose
        lock-step CG workspace stays within `limit` bytes (cached)."""
        key = ("slabs", f, limit)
        hit = self._ws.get(key)
        if hit is None:
            n = self.shape[0]
            per_row = max(1, lib.als_half_sweep_cg_workspace_bytes(n, self.n_slots, f) // max(n, 1))
            rows = max(1, min(n, int(limit // per_row)))
            while True:
                slabs = [(r0, min(n, r0 + rows), self.row_range(r0, min(n, r0 + rows))) for r0 in range(0, n, rows)]
                need = max(lib.als_half_sweep_cg_workspace_bytes(s.shape[0], s.n_slots, f) for _, _, s in slabs)
                if need <= limit or rows == 1:
                    break
                rows = max(1, rows * 3 // 4)          # slots of long rows pushed a slab over the limit
            hit = self._ws[key] = (slabs, torch.empty(need, dtype=torch.uint8, device=self.device))
        return hit

    def workspace_cg(self, f):
        ws = self._ws.get(("cg", f))
        if ws is None:
            nbytes = lib.als_half_sweep_cg_workspace_bytes(self.shape[0], self.n_slots, f)
            if nbytes == 0:
                raise ValueError("unsupported factor count f=%d" % f)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[("cg", f)] = ws
        return ws

    def workspace(self, f):
        ws = self._ws.get(f)
        if ws is None:
            nbytes = lib.als_half_sweep_workspace_bytes(self.shape[0], self.n_slots, f)
            if nbytes == 0:
                raise ValueError("unsupported factor count f=%d" % f)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[f] = ws
        return ws


# ----------------------------------------------------------------------------- gram
class GramCache:
    """Y^T Y (`gram`, [f, f]) and its kernel-order copy with lamb*I added (`tiles`):
    the Y2 / LambdaI pair of reclibwh/core/ALS.py:235-246."""

    def __init__(self, f, lamb, device):
        nbytes = lib.als_gram_tiles_bytes(f)
        if nbytes == 0:
            raise ValueError("unsupported factor count f=%d (1..%d)" % (f, _lib.MAX_FACTORS))
        self.f, self.lamb = int(f), float(lamb)
        self.gram = torch.zeros((f, f), dtype=torch.float32, device=device)
        self.tiles = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        self._ws = None
        self.whitened = None        # [n, 2 f] table of whiten() for the factor this Gram belongs to, or None
        self._wws = None


def gram(Y, lamb, out=None, n=None):
    """als_update_cache: Gram of the first n rows of Y (default all) and lamb*I."""
    ldy = _factor_ld(Y, "Y")
    n = Y.shape[0] if n is None else int(n)
    f = Y.shape[1]
    if n < 1 or n > Y.shape[0]:
        raise ValueError("n out of range")
    if out is None:
        out = GramCache(f, lamb, Y.device)
    out.lamb = float(lamb)
    need = lib.als_gram_workspace_bytes(n, f)
    if out._ws is None or out._ws.numel() < need:
        out._ws = torch.empty(need, dtype=torch.uint8, device=Y.device)
    _same_device(Y, out.gram, out.tiles, out._ws)
    with torch.cuda.device(Y.device):
        check(lib.als_gram_f32(_ptr(Y), n, f, ldy, float(lamb), _ptr(out.gram), _ptr(out.tiles), _ptr(out._ws),
                               out._ws.numel(), _stream()), "als_gram_f32")
    return out


def short_rows_max_entries(f):
    """Rows with at most this many stored entries can be solved in their entry-space form (0: not for this f)."""
    return 16 * int(lib.als_short_rows_max_groups(int(f)))


def whiten(Y, cache, n=None, info=None):
    """The table the short-row form of the half-sweep gathers from: cache.whitened[i] = [ y_i R^-1 | y_i G^-1 ] with
    G = cache.gram + lamb I = R^T R (als_whiten_f32).  cache must hold the Gram of these rows of Y (ops.gram just
    before).  The table is kept in the cache and reused as a buffer by the next call."""
    ldy = _factor_ld(Y, "Y")
    n = Y.shape[0] if n is None else int(n)
    f = Y.shape[1]
    if cache.f != f or not (1 <= n <= Y.shape[0]):
        raise ValueError("cache and Y disagree (f=%d, cache f=%d, n=%d)" % (f, cache.f, n))
    need = lib.als_whiten_workspace_bytes(f)
    if need == 0:
        raise ValueError("the short-row form exists for 32 < f <= 128 with f a multiple of 4 (f=%d)" % f)
    if cache._wws is None:
        cache._wws = torch.empty(need, dtype=torch.uint8, device=Y.device)
    wz = cache.whitened
    if wz is None or wz.shape[0] < n or wz.device != Y.device:
        wz = cache.whitened = torch.empty((n, 2 * f), dtype=torch.float32, device=Y.device)
    _same_device(Y, cache.gram, wz, cache._wws, info)
    with torch.cuda.device(Y.device):
        check(lib.als_whiten_f32(_ptr(cache.gram), float(cache.lamb), f, _ptr(Y), n, ldy, _ptr(wz), wz.stride(0), _ptr(info),
                                 _ptr(cache._wws), cache._wws.numel(), _stream()), "als_whiten_f32")
    return wz


def gram_partials(Y, row_begin, row_end, super_out=None, ws=None):
    """Second-level Gram partials of rows [row_begin, row_end) (row_begin a multiple of
    GRAM_SUPER_ROWS); returns a [n_super, tile_elems] float32 tensor."""
    ldy = _factor_ld(Y, "Y")
    f = Y.shape[1]
    if not (0 <= row_begin < row_end <= Y.shape[0]):
        raise ValueError("bad row range")
    n_super = -(-(row_end - row_begin) // _lib.GRAM_SUPER_ROWS)
    elems = lib.als_gram_partial_bytes(f) // 4
    if elems == 0:
        raise ValueError("unsupported factor count f=%d" % f)
    if super_out is None:
        super_out = torch.empty((n_super, elems), dtype=torch.float32, device=Y.device)
    need = lib.als_gram_workspace_bytes(row_end - row_begin, f)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=Y.device)
    _same_device(Y, super_out, ws)
    with torch.cuda.device(Y.device):
        check(lib.als_gram_partials_f32(_ptr(Y), row_begin, row_end, f, ldy, _ptr(super_out), _ptr(ws), ws.numel(),
                                        _stream()), "als_gram_partials_f32")
    return super_out


def gram_finalize(super_partials, f, lamb, out=None):
    _need(super_partials, torch.float32, "super_partials", 2)
    if not super_partials.is_contiguous():
        raise ValueError("super_partials must be contiguous")
    if out is None:
        out = GramCache(f, lamb, super_partials.device)
    out.lamb = float(lamb)
    _same_device(super_partials, out.gram, out.tiles)
    with torch.cuda.device(super_partials.device):
        check(lib.als_gram_finalize_f32(_ptr(super_partials), super_partials.shape[0], f, float(lamb), _ptr(out.gram),
                                        _ptr(out.tiles), _stream()), "als_gram_finalize_f32")
    return out


# ----------------------------------------------------------------------------- half sweep
def half_sweep(csr, Yopp, gram_cache, Xown, alpha, absdiff=None, info=None, opts=None, static_scratch=False,
               short_rows=False):
    """One half-sweep: every row of `csr` re-solves its row of Xown (in place) against the
    fixed factor Yopp.  Returns the device scalar sum|X_old - X_new| (float64, 1 element);
    divide by n*k for run_als_epoch's return value (reclibwh/core/ALS.py:233).  opts: sweep_opts(...).
    static_scratch=True: als_half_sweep_csr_static_f32 - no workspace is allocated or passed, the library's built-in
    per-GPU scratch buffer is used (small inputs only; absdiff must then be given to keep the call allocation free).
    short_rows=True: rows with at most short_rows_max_entries(f) stored entries are solved in their entry-space form
    from gram_cache.whitened (ops.whiten(Yopp, gram_cache) after ops.gram; the same solution to float32 rounding)."""
    if not isinstance(csr, DeviceCSR):
        raise ValueError("csr must be a DeviceCSR")
    ldy = _factor_ld(Yopp, "Yopp")
    ldx = _factor_ld(Xown, "Xown")
    f = Yopp.shape[1]
    if Xown.shape[1] != f or gram_cache.f != f:
        raise ValueError("factor counts differ: Yopp %d, Xown %d, gram %d" % (f, Xown.shape[1], gram_cache.f))
    if Yopp.shape[0] < csr.shape[1]:
        raise ValueError("Yopp has %d rows but the CSR has %d columns" % (Yopp.shape[0], csr.shape[1]))
    need_rows = (csr.row_ids_max + 1) if csr.row_ids is not None else csr.shape[0]
    if Xown.shape[0] < need_rows:
        raise ValueError("Xown has %d rows, the CSR updates row %d" % (Xown.shape[0], need_rows - 1))
    if Yopp.device != csr.device or Xown.device != csr.device:
        raise ValueError("tensors live on different devices")
    if Xown.data_ptr() == Yopp.data_ptr():
        raise ValueError("Xown and Yopp must not alias")
    if absdiff is None:
        absdiff = torch.zeros(1, dtype=torch.float64, device=csr.device)
    o = _lib.SweepOpts()
    if opts is not None:
        o.accumulate, o.row_kernel, o.tap_begin, o.tap_end = opts.accumulate, opts.row_kernel, opts.tap_begin, opts.tap_end
        o.reserve_cus = opts.reserve_cus
    o.max_rating = csr.max_rating          # the bound the fp16 x 2 accumulate (the default) takes its scale from
    o._keep = (opts,)
    opts = o
    if short_rows:
        wz = gram_cache.whitened
        if wz is None or wz.shape[0] < csr.shape[1] or wz.shape[1] != 2 * f or wz.device != csr.device:
            raise ValueError("short_rows=True needs gram_cache.whitened for this factor (ops.whiten)")
        o.whitened, o.ld_whitened, o.short_bounds = wz.data_ptr(), wz.stride(0), csr.short_bounds.ctypes.data
        o._keep = (o._keep, wz, csr.short_bounds)
    if static_scratch:
        with torch.cuda.device(csr.device):
            check(lib.als_half_sweep_csr_static_f32(
                _ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), csr.shape[0], _ptr(csr.row_ids), _ptr(csr.items),
                csr.n_items, _ptr(csr.long_rows), csr.n_long, csr.n_slots, csr.chunk_nnz, _ptr(Yopp), csr.shape[1], ldy,
                _ptr(gram_cache.tiles), f, float(alpha), _ptr(Xown), ldx, _ptr(absdiff), _ptr(info), _opts_ref(opts),
                _stream()), "als_half_sweep_csr_static_f32")
        return absdiff
    ws = csr.workspace(f)
    with torch.cuda.device(csr.device):
        check(lib.als_half_sweep_csr_f32(
            _ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), csr.shape[0], _ptr(csr.row_ids), _ptr(csr.items),
            csr.n_items, _ptr(csr.long_rows), csr.n_long, csr.n_slots, csr.chunk_nnz, _ptr(Yopp), csr.shape[1], ldy,
            _ptr(gram_cache.tiles), f, float(alpha), _ptr(Xown), ldx, _ptr(absdiff), _ptr(info), _opts_ref(opts),
            _ptr(ws), ws.numel(), _stream()), "als_half_sweep_csr_f32")
    return absdiff


LOCKSTEP_WS_LIMIT = 8 << 30     # bytes of lock-step CG workspace beyond which the rows are swept slab by slab


def half_sweep_cg(csr, Yopp, gram_cache, Xown, alpha, cg_steps, absdiff=None, info=None, form="auto", opts=None,
                  _ws=None):
    """One half-sweep by `cg_steps` conjugate-gradient steps per row, warm-started at the current
    rows of Xown (updated in place).  Same contract as half_sweep otherwise; 1 <= f <= 256.
    form: "matrix" - one gather pass, the steps run on the row's formed matrix in registers;
    "lock_step" - cg_steps + 1 gather passes with all rows in lock step; "auto" - matrix for
    f <= 128, lock_step above (see include/als_hip.h).
    The lock-step form keeps four f-vectors per row in its workspace (16 f bytes per row: 205 GB for 50 M rows at
    f = 256).  When that exceeds LOCKSTEP_WS_LIMIT the rows are swept in contiguous slabs that share one workspace of
    that size; rows are independent, so the factors are the same bits as from one call."""
    if form not in _lib.CG_FORMS:
        raise ValueError("form must be one of %s" % sorted(_lib.CG_FORMS))
    if not isinstance(csr, DeviceCSR):
        raise ValueError("csr must be a DeviceCSR")
    ldy = _factor_ld(Yopp, "Yopp")
    ldx = _factor_ld(Xown, "Xown")
    f = Yopp.shape[1]
    if Xown.shape[1] != f or gram_cache.f != f:
        raise ValueError("factor counts differ: Yopp %d, Xown %d, gram %d" % (f, Xown.shape[1], gram_cache.f))
    if Yopp.shape[0] < csr.shape[1]:
        raise ValueError("Yopp has %d rows but the CSR has %d columns" % (Yopp.shape[0], csr.shape[1]))
    need_rows = (csr.row_ids_max + 1) if csr.row_ids is not None else csr.shape[0]
    if Xown.shape[0] < need_rows:
        raise ValueError("Xown has %d rows, the CSR updates row %d" % (Xown.shape[0], need_rows - 1))
    if Yopp.device != csr.device or Xown.device != csr.device:
        raise ValueError("tensors live on different devices")
    if Xown.data_ptr() == Yopp.data_ptr():
        raise ValueError("Xown and Yopp must not alias")
    cg_steps = int(cg_steps)
    if not (1 <= cg_steps <= 1024):
        raise ValueError("cg_steps must be in [1, 1024]")
    if absdiff is None:
        absdiff = torch.zeros(1, dtype=torch.float64, device=csr.device)
    lock_step = form == "lock_step" or (form == "auto" and f > _lib.MAX_FACTORS_EXACT)
    if _ws is None and lock_step and csr.shape[0] > 1 \
            and lib.als_half_sweep_cg_workspace_bytes(csr.shape[0], csr.n_slots, f) > LOCKSTEP_WS_LIMIT:
        slabs, ws = csr.lockstep_slabs(f, LOCKSTEP_WS_LIMIT)
        parts = torch.zeros(len(slabs), dtype=torch.float64, device=csr.device)
        tap = getattr(opts, "_tap", None) if opts is not None else None
        if tap is not None:         # the tap brackets all slabs, not the last one
            opts = _lib.SweepOpts(accumulate=opts.accumulate, row_kernel=opts.row_kernel, max_rating=opts.max_rating)
            with torch.cuda.device(csr.device):
                tap[0].record(torch.cuda.current_stream())
        for k, (r0, r1, sub) in enumerate(slabs):
            Xs = Xown if sub.row_ids is not None else Xown[r0:r1]
            half_sweep_cg(sub, Yopp, gram_cache, Xs, alpha, cg_steps, absdiff=parts[k:k + 1], info=info, form=form,
                          opts=opts, _ws=ws)
        if tap is not None:
            with torch.cuda.device(csr.device):
                tap[1].record(torch.cuda.current_stream())
        torch.sum(parts, 0, keepdim=True, out=absdiff)           # slab order: a fixed order
        return absdiff
    o = _lib.SweepOpts()
    if opts is not None:
        o.accumulate, o.row_kernel, o.tap_begin, o.tap_end = opts.accumulate, opts.row_kernel, opts.tap_begin, opts.tap_end
        o.reserve_cus = opts.reserve_cus
    o.max_rating = csr.max_rating          # the formed-matrix form accumulates as the exact half-sweep does
    o._keep = (opts,)
    opts = o
    ws = csr.workspace_cg(f) if _ws is None else _ws
    if ws.numel() < lib.als_half_sweep_cg_workspace_bytes(csr.shape[0], csr.n_slots, f):
        raise ValueError("shared workspace too small for this slab")
    with torch.cuda.device(csr.device):
        check(lib.als_half_sweep_cg_f32(
            _ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), csr.shape[0], _ptr(csr.row_ids), _ptr(csr.items),
            csr.n_items, _ptr(csr.long_rows), csr.n_long, csr.n_slots, csr.chunk_nnz, _ptr(Yopp), csr.shape[1], ldy,
            _ptr(gram_cache.gram), _ptr(gram_cache.tiles), float(gram_cache.lamb), f, float(alpha), cg_steps,
            _lib.CG_FORMS[form], _ptr(Xown), ldx, _ptr(absdiff),
            _ptr(info), _opts_ref(opts), _ptr(ws), ws.numel(), _stream()), "als_half_sweep_cg_f32")
    return absdiff


# ----------------------------------------------------------------------------- scoring
class PreparedItems:
    """An item factor in the operand form of the fp16 matrix cores (als_items_prepare_f32): every value scaled by one
    power of two and cut into two fp16 pieces, laid out as the score kernel's lanes read it.  Built once per model;
    scores() and recommend_topk() take it in place of Y and then run three piece products per block instead of six,
    with no conversion work in the loop (scores to 22 operand bits instead of 24).  Same size as the float32 factor."""

    def __init__(self, table, n_items, f):
        self.table, self.shape, self.device = table, (int(n_items), int(f)), table.device


def prepared_items_supported(f):
    return lib.als_items_prepared_bytes(1, int(f)) > 0


def prepare_items(Y, M=None):
    """The first M rows (default: all) of the item factor Y as PreparedItems.  f <= 128, f % 4 == 0, 16-byte aligned
    rows; the factors must be finite."""
    ldy = _factor_ld(Y, "Y")
    M = Y.shape[0] if M is None else int(M)
    f = Y.shape[1]
    if not (1 <= M <= Y.shape[0]):
        raise ValueError("M out of range")
    nbytes = lib.als_items_prepared_bytes(M, f)
    if nbytes == 0:
        raise ValueError("prepared items need f <= 128 and f % 4 == 0 (f=%d)" % f)
    table = torch.empty(nbytes, dtype=torch.uint8, device=Y.device)
    with torch.cuda.device(Y.device):
        check(lib.als_items_prepare_f32(_ptr(Y), ldy, M, f, _ptr(table), nbytes, _stream()), "als_items_prepare_f32")
    return PreparedItems(table, M, f)


def scores(X, Y, user_ids=None, M=None, out=None):
    """scores[u, i] = <X[user_ids[u]], Y[i]> for i < M: the [nu, M] matrix that
    SimpleMFRecAlgo.recommend builds pair by pair (reclibwh/core/RecAlgos.py:20-30).  Y: the item factor, or its
    PreparedItems."""
    ldx = _factor_ld(X, "X")
    prepared = Y if isinstance(Y, PreparedItems) else None
    ldy = 0 if prepared else _factor_ld(Y, "Y")
    f = X.shape[1]
    if Y.shape[1] != f:
        raise ValueError("factor counts differ")
    M = Y.shape[0] if M is None else int(M)
    if not (1 <= M <= Y.shape[0]):
        raise ValueError("M out of range")
    if user_ids is not None:
        _index(user_ids, torch.int64, "user_ids")
        nu = user_ids.shape[0]
        if nu and (int(user_ids.min()) < 0 or int(user_ids.max()) >= X.shape[0]):
            raise ValueError("user id out of range")
    else:
        nu = X.shape[0]
    if nu < 1:
        raise ValueError("no users")
    if out is None:
        out = torch.empty((nu, M), dtype=torch.float32, device=X.device)
    _need(out, torch.float32, "out", 2)
    if out.shape[0] < nu or out.shape[1] < M or out.stride(1) != 1:
        raise ValueError("out too small or not row-major")
    _same_device(X, prepared.table if prepared else Y, user_ids, out)
    with torch.cuda.device(X.device):
        if prepared:
            check(lib.als_scores_prepared_f32(_ptr(X), ldx, _ptr(user_ids), nu, _ptr(prepared.table), prepared.shape[0], M,
                                              f, _ptr(out), out.stride(0), _stream()), "als_scores_prepared_f32")
        else:
            check(lib.als_scores_f32(_ptr(X), ldx, _ptr(user_ids), nu, _ptr(Y), ldy, M, f, _ptr(out), out.stride(0),
                                     _stream()), "als_scores_f32")
    return out[:nu, :M]


def topk_rows(score_rows, k, exclude=None, exclude_rows=None):
    """Per row: the k largest scores, descending, ties by ascending column.  `exclude` =
    (rowptr int64, col int32) device tensors: columns forced to -inf first (the score rows are
    modified in place).  Score row r takes CSR row r, or CSR row exclude_rows[r] (int64 device
    tensor) when given - a CSR over all users then serves any block of them.
    Returns (idx int32 [n, k], values float32 [n, k])."""
    _need(score_rows, torch.float32, "score_rows", 2)
    if score_rows.stride(1) != 1:
        raise ValueError("score rows must have unit inner stride")
    n, M = score_rows.shape
    k = int(k)
    if not (1 <= k <= min(1024, M)):
        raise ValueError("k must be in [1, min(1024, M)]")
    ex_ptr = ex_col = None
    if exclude is not None:
        ex_ptr, ex_col = exclude
        _index(ex_ptr, torch.int64, "exclude rowptr")
        _index(ex_col, torch.int32, "exclude col")
        _same_device(score_rows, ex_ptr, ex_col, exclude_rows)
        if exclude_rows is not None:
            _index(exclude_rows, torch.int64, "exclude_rows")
            if exclude_rows.shape[0] != n:
                raise ValueError("exclude_rows must have one entry per score row")
            if n and (int(exclude_rows.min()) < 0 or int(exclude_rows.max()) + 1 >= ex_ptr.shape[0]):
                raise ValueError("exclude_rows points outside the exclusion CSR")
        elif ex_ptr.shape[0] != n + 1:
            raise ValueError("exclude rowptr must have n_rows + 1 entries")
    elif exclude_rows is not None:
        raise ValueError("exclude_rows without exclude")
    idx = torch.empty((n, k), dtype=torch.int32, device=score_rows.device)
    val = torch.empty((n, k), dtype=torch.float32, device=score_rows.device)
    with torch.cuda.device(score_rows.device):
        check(lib.als_topk_rows_f32(_ptr(score_rows), score_rows.stride(0), n, M, _ptr(ex_ptr), _ptr(ex_col),
                                    _ptr(exclude_rows), k, _ptr(idx), _ptr(val), _stream()), "als_topk_rows_f32")
    return idx, val


def predict_pairs(X, Y, u, i):
    """out[p] = <X[u[p]], Y[i[p]]>: the ALS model's output for explicit (user, item) pairs
    (reclibwh/core/ALS.py:359-362)."""
    ldx = _factor_ld(X, "X"); ldy = _factor_ld(Y, "Y")
    _index(u, torch.int64, "u"); _index(i, torch.int64, "i")
    if u.shape != i.shape or u.numel() < 1:
        raise ValueError("u and i must be equally long, non-empty")
    if int(u.min()) < 0 or int(u.max()) >= X.shape[0] or int(i.min()) < 0 or int(i.max()) >= Y.shape[0]:
        raise ValueError("index out of range")
    out = torch.empty(u.shape[0], dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        check(lib.als_predict_pairs_f32(_ptr(X), ldx, _ptr(Y), ldy, _ptr(u), _ptr(i), u.shape[0], X.shape[1],
                                        _ptr(out), _stream()), "als_predict_pairs_f32")
    return out


def normalize_rows(E, out=None):
    """Rows scaled to unit 2-norm, zero rows stay zero (reclibwh/core/RecAlgos.py:45-50)."""
    lde = _factor_ld(E, "E")
    if out is None:
        out = torch.empty_like(E)
    ldo = _factor_ld(out, "out")
    with torch.cuda.device(E.device):
        check(lib.als_normalize_rows_f32(_ptr(E), lde, E.shape[0], E.shape[1], _ptr(out), ldo, _stream()),
              "als_normalize_rows_f32")
    return out


def device_exclusions(exclude_csr, device):
    """(rowptr int64, col int32) device tensors of an exclusion CSR over ALL users, from host numpy arrays, device
    tensors or a DeviceCSR (its own arrays, nothing copied)."""
    if isinstance(exclude_csr, DeviceCSR):
        return exclude_csr.rowptr, exclude_csr.col
    ptr, col = exclude_csr
    if not torch.is_tensor(ptr):
        ptr = torch.from_numpy(np.ascontiguousarray(ptr, dtype=np.int64))
    if not torch.is_tensor(col):
        col = torch.from_numpy(np.ascontiguousarray(col, dtype=np.int32))
    ptr = ptr.to(device=device, dtype=torch.int64).contiguous()
    col = col.to(device=device, dtype=torch.int32).contiguous()
    if col.numel() == 0:
        col = torch.zeros(1, dtype=torch.int32, device=device)       # a pointer the library accepts; never read
    return ptr, col


def recommend_topk(X, Y, users, k, M=None, exclude_csr=None, block_users=4096, fused=True):
    """Top-k items for `users` (int64 device tensor), block_users users at a time.
    exclude_csr: optional CSR over ALL users (sorted columns) whose columns are filtered per user: (rowptr, col) as
    host numpy arrays or device tensors, or a DeviceCSR (the training matrix as it lies on the device).  The kernels
    index it by user id; nothing is built per block.  Returns (idx int32 [nu, k], scores [nu, k]).

    Y: the item factor or its PreparedItems (prepare_items: built once per model, about three times the throughput).

    fused=True: als_recommend_topk_f32 - only the first 16 384+ items of a block are materialised,
    the rest is filtered in the score kernel's registers.  Users whose candidate list overflows (the library marks
    their rows: e.g. thousands of excluded items that all score near the top) are redone by the two-call path - they
    alone, not their block -, as is everyone when fused=False."""
    prepared = Y if isinstance(Y, PreparedItems) else None
    ldx = _factor_ld(X, "X"); ldy = 0 if prepared else _factor_ld(Y, "Y")
    M = Y.shape[0] if M is None else int(M)
    f = X.shape[1]
    if Y.shape[1] != f or not (1 <= M <= Y.shape[0]):
        raise ValueError("factor counts differ or M out of range")
    if prepared and M != Y.shape[0]:
        raise ValueError("a prepared item table is ranked whole (prepare_items(Y, M) the rows wanted)")
    _index(users, torch.int64, "users")
    _same_device(X, prepared.table if prepared else Y, users)
    nu = users.shape[0]
    k = int(k)
    if not (1 <= k <= min(1024, M)):
        raise ValueError("k must be in [1, min(1024, M)]")
    if nu and (int(users.min()) < 0 or int(users.max()) >= X.shape[0]):
        raise ValueError("user id out of range")
    idx = torch.empty((nu, k), dtype=torch.int32, device=X.device)
    val = torch.empty((nu, k), dtype=torch.float32, device=X.device)
    ex = None
    if exclude_csr is not None:
        ex = device_exclusions(exclude_csr, X.device)
        if nu and int(users.max()) + 1 >= ex[0].shape[0]:
            raise ValueError("a user id lies outside the exclusion CSR")
    bu = min(block_users, max(nu, 1))
    buf = None
    blocks = [(s, min(nu, s + block_users)) for s in range(0, nu, block_users)]
    redo = None                     # positions (rows of the output) to be done by the two-call path; None = all
    if fused and blocks:
        ws = torch.empty(lib.als_recommend_topk_workspace_bytes(bu, M, k), dtype=torch.uint8, device=X.device)
        overflow = torch.zeros(len(blocks), dtype=torch.int32, device=X.device)
        for b, (s, e) in enumerate(blocks):
            with torch.cuda.device(X.device):
                tail = (_ptr(ex[0]) if ex else None, _ptr(ex[1]) if ex else None, 1, _ptr(idx[s:e]), _ptr(val[s:e]),
                        _ptr(overflow[b:b + 1]), _ptr(ws), ws.numel(), _stream())
                if prepared:
                    check(lib.als_recommend_topk_prepared_f32(_ptr(X), ldx, _ptr(users[s:e]), e - s, _ptr(prepared.table),
                                                              M, f, k, *tail), "als_recommend_topk_prepared_f32")
                else:
                    check(lib.als_recommend_topk_f32(_ptr(X), ldx, _ptr(users[s:e]), e - s, _ptr(Y), ldy, M, f, k, *tail),
                          "als_recommend_topk_f32")
        # users whose candidate list overflowed (marked with item -1) are redone, the others stand; one
        # synchronisation for all blocks
        redo = torch.nonzero(idx[:, 0] < 0).flatten() if bool(overflow.any()) else idx.new_zeros(0, dtype=torch.int64)
    # the materialised path works on row blocks whose score matrix stays below 4 GiB
    rows_per = max(1, min(bu, (1 << 30) // M))
    n_redo = nu if redo is None else int(redo.numel())
    for s in range(0, n_redo, rows_per):
        e = min(n_redo, s + rows_per)
        pos = None if redo is None else redo[s:e]
        us = users[s:e] if pos is None else users[pos].contiguous()
        if buf is None:
            buf = torch.empty((min(rows_per, n_redo), M), dtype=torch.float32, device=X.device)
        sc = scores(X, Y, us, M, out=buf)
        i_, v_ = topk_rows(sc, k, ex, us if ex else None)
        if pos is None:
            idx[s:e] = i_; val[s:e] = v_
        else:
            idx[pos] = i_; val[pos] = v_
    return idx, val
