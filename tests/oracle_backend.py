"""TEST INFRASTRUCTURE: a CPU stand-in for recommender_amd.ops built on the oracle, so that the
engine's multi-process logic (row sharding, in-place all-gather, rank-ordered Gram assembly) can
be exercised with the gloo backend on machines without a GPU.  Never imported by the product."""
import numpy as np
import torch

from oracle import c_oracle
from recommender_amd import _lib


class HostCSR:
    def __init__(self, indptr, indices, data, shape):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        self.shape = (int(shape[0]), int(shape[1]))
        self.nnz = int(self.indices.shape[0])

    def row_range(self, r0, r1):
        p = self.indptr
        return HostCSR(p[r0:r1 + 1] - p[r0], self.indices[p[r0]:p[r1]], self.data[p[r0]:p[r1]], (r1 - r0, self.shape[1]))


class GramCache:
    def __init__(self, f, lamb, device):
        self.f, self.lamb = int(f), float(lamb)
        self.gram = torch.zeros((f, f), dtype=torch.float64)


def gram_partials(Y, row_begin, row_end, super_out=None, ws=None):
    S = _lib.GRAM_SUPER_ROWS
    assert row_begin % S == 0
    f = Y.shape[1]
    blocks = []
    for b in range(row_begin, row_end, S):
        blk = Y[b:min(row_end, b + S)].to(torch.float64)
        blocks.append((blk.T @ blk).reshape(1, f * f))
    return torch.cat(blocks, 0)


def gram_finalize(super_partials, f, lamb, out=None):
    out = GramCache(f, lamb, None) if out is None else out
    acc = super_partials[0].clone()
    for i in range(1, super_partials.shape[0]):       # fixed order, as the HIP finalize kernel
        acc += super_partials[i]
    out.gram = acc.reshape(f, f)
    out.lamb = float(lamb)
    return out


def gram(Y, lamb, out=None, n=None):
    n = Y.shape[0] if n is None else n
    return gram_finalize(gram_partials(Y, 0, n), Y.shape[1], lamb, out)


def half_sweep(csr, Yopp, cache, Xown, alpha, absdiff=None, info=None, opts=None):
    old = Xown.numpy().astype(np.float64)
    new = c_oracle.half_sweep(csr.indptr, csr.indices, csr.data, Yopp.numpy().astype(np.float64), old, cache.lamb,
                              alpha, Y2=cache.gram.numpy())
    Xown.copy_(torch.from_numpy(new).to(Xown.dtype))
    if absdiff is None:
        absdiff = torch.zeros(1, dtype=torch.float64)
    absdiff[0] = float(np.abs(old - new).sum())
    return absdiff


def half_sweep_cg(csr, Yopp, cache, Xown, alpha, cg_steps, absdiff=None, info=None, form="auto", opts=None):
    from oracle import als_ref
    old = Xown.numpy().astype(np.float64)
    new = als_ref.half_sweep_cg(csr.indptr, csr.indices, csr.data, Yopp.numpy().astype(np.float64), old, cache.lamb,
                                alpha, cg_steps)
    Xown.copy_(torch.from_numpy(new).to(Xown.dtype))
    if absdiff is None:
        absdiff = torch.zeros(1, dtype=torch.float64)
    absdiff[0] = float(np.abs(old - new).sum())
    return absdiff
