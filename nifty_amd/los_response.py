"""Line-of-sight response and mask (BASELINE config 4).

``LOSResponse`` (reference library/los_response.py:100-253): every line of sight is walked through the pixel
grid; the path length inside each pixel -- optionally weighted by the survival function of a Gaussian
uncertainty of the inverse distance -- becomes one entry of a sparse ``n_los x n_pix`` matrix (float32
weights, int32 indices as in the reference, :194-197).  TIMES = SpMV, ADJOINT_TIMES = SpMV^T.  The set-up runs
on the host in numpy (as in the reference); on a GPU both products run in libniftyk as row sums of a CSR matrix
(``nk_csr_rowsum``: the matrix for TIMES, its transpose -- built once on the host -- for ADJOINT_TIMES, so every
output is summed in a fixed order without atomics), on the host through scipy.sparse.
"""
import numpy as np
import torch
from scipy.special import erfc

from . import backend as B
from .domains import DomainTuple, RGSpace, UnstructuredDomain
from .field import Field
from .operators import LinearOperator


def _gaussian_sf(x):
    """Survival function of the standard normal (los_response.py:29-30)."""
    return 0.5 * erfc(x / np.sqrt(2.0))


def _weight_by_distance(wgt, dist, lo, mid, hi, sig):
    """Parallax weighting of the segments (los_response.py:91-97): nothing beyond `hi`, survival probability of the
    line still being inside the volume between `lo` and `hi`."""
    wgt = wgt.copy()
    wgt[dist > hi] = 0.0
    sel = (dist > lo) & (dist <= hi)
    wgt[sel] *= _gaussian_sf((-1.0 / dist[sel] + 1.0 / mid) / sig)
    return wgt


def _walk_line(start, end, shape, strides, dist, lo, mid, hi, sig):
    """Pixels crossed by the segment start -> end (pixel units, box [0, shape]) and the physical path length inside
    each of them (los_response.py:33-88).  Returns (flat pixel indices int64, weights float64)."""
    shape = np.asarray(shape, dtype=np.float64)
    direction = end - start
    moving = direction != 0.0
    # parameter interval of the line inside the box, per axis then intersected
    safe = np.where(moving, direction, 1e-12)
    t_in = np.where(moving, -start / safe, ((start > 0) - 0.5) * 1e12)
    t_out = np.where(moving, (shape - start) / safe, ((start < shape) - 0.5) * -1e12)
    tmin = max(0.0, float(np.minimum(t_in, t_out).max()))
    tmax = min(1.0, float(np.maximum(t_in, t_out).min()))
    tmax = max(tmin, tmax)
    tmin += 1e-7  # stay clear of grid crossings at the ends (los_response.py:57-59)
    tmax -= 1e-7
    if tmin >= tmax:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float64)
    # first grid plane crossed on every axis, then one crossing per 1/|direction|
    first = np.ceil(start + direction * tmin)
    first = np.where(direction > 0.0, first, first - 1.0)
    first = (first - start) / safe
    pix0 = int(np.sum((start + tmin * direction).astype(np.int64) * strides))
    ts, steps = [], []
    for j in np.nonzero(moving)[0]:
        tj = np.arange(first[j], tmax, abs(1.0 / direction[j]))
        ts.append(tj)
        steps.append(np.full(len(tj), strides[j] if direction[j] > 0 else -strides[j], dtype=np.int64))
    ts = np.concatenate(ts) if ts else np.zeros(0)
    steps = np.concatenate(steps) if steps else np.zeros(0, dtype=np.int64)
    order = np.argsort(ts)
    knots = np.concatenate([[tmin], ts[order], [tmax]]) * np.linalg.norm(direction * dist)
    wgt = _weight_by_distance(np.diff(knots), 0.5 * (knots[:-1] + knots[1:]), lo, mid, hi, sig)
    return np.cumsum(np.concatenate([[pix0], steps[order]])), wgt


def los_matrix(shape, distances, starts, ends, sigmas=None, truncation=3.0):
    """CSR arrays (rowptr int64, col int32, wgt float32) of the response (los_response.py:144-221)."""
    ndim = len(shape)
    starts, ends = np.array(starts, dtype=np.float64), np.array(ends, dtype=np.float64)
    if starts.ndim != 2 or starts.shape[0] != ndim:
        raise TypeError("dimension mismatch")
    nlos = starts.shape[1]
    sigmas = np.zeros(nlos, dtype=np.float32) if sigmas is None else np.array(sigmas)
    if nlos != sigmas.shape[0] or starts.shape != ends.shape:
        raise TypeError("dimension mismatch")
    diffs = ends - starts
    length = np.linalg.norm(diffs, axis=0)
    diffs = diffs / length
    far = 1.0 / (1.0 / length - truncation * sigmas)
    if np.any(far < 0):
        raise ValueError("parallax error truncation to high: getting negative distances")
    near = 1.0 / (1.0 / length + truncation * sigmas)
    dist = np.array(distances, dtype=np.float64)
    pstart = starts / dist[:, None] + 0.5
    pend = (starts + diffs * far) / dist[:, None] + 0.5
    strides = np.ones(ndim, dtype=np.int64)
    for j in range(ndim - 2, -1, -1):
        strides[j] = strides[j + 1] * shape[j + 1]
    cols, wgts, rowptr = [], [], np.zeros(nlos + 1, dtype=np.int64)
    for i in range(nlos):
        c, w = _walk_line(pstart[:, i], pend[:, i], shape, strides, dist, near[i], length[i], far[i], sigmas[i])
        cols.append(c)
        wgts.append(w)
        rowptr[i + 1] = rowptr[i] + len(c)
    col = np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, dtype=np.int32)
    wgt = np.concatenate(wgts).astype(np.float32) if wgts else np.zeros(0, dtype=np.float32)
    return rowptr, col, wgt


class LOSResponse(LinearOperator):
    """Line-of-sight response: RGSpace -> UnstructuredDomain(n_los) (reference los_response.py:100-253; same
    arguments).  `starts`, `ends`: (ndim, n_los) arrays in physical units; `sigmas`: optional standard deviations
    of the inverse line lengths (parallax model) with `truncation` in sigmas."""

    def __init__(self, domain, starts, ends, sigmas=None, truncation=3.0):
        self._domain = DomainTuple.make(domain)
        self._capability = self.TIMES | self.ADJOINT_TIMES
        if len(self._domain) != 1 or not isinstance(self._domain[0], RGSpace):
            raise TypeError("The domain must be exactly one RGSpace instance.")
        sp = self._domain[0]
        self._rowptr, self._col, self._wgt = los_matrix(sp.shape, sp.distances, starts, ends, sigmas, truncation)
        self._nlos = len(self._rowptr) - 1
        self._target = DomainTuple.make(UnstructuredDomain(self._nlos))
        self._host = None
        self._host_t = None
        self._dev = {}

    def _host_matrix(self):
        if self._host is None:
            from scipy.sparse import csr_matrix

            self._host = csr_matrix((self._wgt, self._col, self._rowptr), shape=(self._nlos, self._domain.size))
        return self._host

    def _transposed(self):
        """CSR arrays of R^T (rows = pixels; the entries of a pixel in ascending line order: scipy's csr -> csc pass walks
        the lines in order), made once."""
        if self._host_t is None:
            mt = self._host_matrix().tocsc()
            self._host_t = (mt.indptr.astype(np.int64), mt.indices.astype(np.int32), mt.data.astype(np.float32))
        return self._host_t

    def _device_arrays(self, device, transposed=False):
        key = (str(device), transposed)
        if key not in self._dev:
            arrs = self._transposed() if transposed else (self._rowptr, self._col, self._wgt)
            self._dev[key] = tuple(torch.from_numpy(a).to(device) for a in arrs)
        return self._dev[key]

    def apply(self, x, mode):
        self._check_input(x, mode)
        v = x.val
        if v.is_cuda:
            nnz = len(self._col)
            if mode == self.TIMES:
                rowptr, col, wgt = self._device_arrays(v.device)
                return Field(self._target, B.spmv(rowptr, col, wgt, v.contiguous().reshape(-1), self._nlos,
                                                  B.lanes_for(nnz, self._nlos)))
            rowptr, col, wgt = self._device_arrays(v.device, transposed=True)
            res = B.spmv(rowptr, col, wgt, v.contiguous(), self._domain.size, B.lanes_for(nnz, self._domain.size))
            return Field(self._domain, res.reshape(self._domain.shape))
        m = self._host_matrix()
        a = v.numpy()
        if mode == self.TIMES:
            return Field(self._target, torch.from_numpy(np.asarray(m @ a.reshape(-1))))
        return Field(self._domain, torch.from_numpy(np.asarray(m.T @ a).reshape(self._domain.shape)))


class SparseResponse:
    """A linear response signal space -> data space as ONE sparse matrix for the fused engine (FusedModel(response=...)):
    the rows of a LOSResponse that a following MaskOperator keeps (reference demos/cl/getting_started_3.py:98-100:
    ``Mask @ LOSResponse``), held on the device as CSR arrays of the matrix and of its transpose.  Both products are
    nk_csr_rowsum launches (fixed summation order); float32 weights, fp64 accumulation, like LOSResponse itself."""

    def __init__(self, rowptr, col, wgt, n_pix):
        from scipy.sparse import csr_matrix

        self.n_data, self.n_pix = len(rowptr) - 1, int(n_pix)
        self._m = csr_matrix((np.asarray(wgt, dtype=np.float32), np.asarray(col, dtype=np.int32),
                              np.asarray(rowptr, dtype=np.int64)), shape=(self.n_data, self.n_pix))
        self._dev = {}

    @staticmethod
    def from_operators(los, mask=None):
        """LOSResponse, optionally followed by a MaskOperator on its target."""
        m = los._host_matrix()
        if mask is not None:
            if mask.domain is not los.target and mask.domain != los.target:
                raise ValueError("mask does not act on the response's target")
            m = m[mask._keep.numpy()]
        m = m.tocsr()
        return SparseResponse(m.indptr, m.indices, m.data, los.domain.size)

    @property
    def host_matrix(self):
        return self._m

    def _arrays(self, device):
        key = str(device)
        if key not in self._dev:
            mt = self._m.tocsc()
            host = (self._m.indptr.astype(np.int64), self._m.indices.astype(np.int32), self._m.data.astype(np.float32),
                    mt.indptr.astype(np.int64), mt.indices.astype(np.int32), mt.data.astype(np.float32))
            self._dev[key] = tuple(torch.from_numpy(a).to(device) for a in host)
        return self._dev[key]

    def times(self, x):
        """R x: x any tensor with n_pix entries -> [n_data]."""
        a = self._arrays(x.device)
        return B.spmv(a[0], a[1], a[2], x.contiguous().reshape(-1), self.n_data, B.lanes_for(self._m.nnz, self.n_data))

    def adjoint(self, y, shape=None):
        """R^T y: [n_data] -> n_pix entries (reshaped to `shape`)."""
        a = self._arrays(y.device)
        out = B.spmv(a[3], a[4], a[5], y.contiguous(), self.n_pix, B.lanes_for(self._m.nnz, self.n_pix))
        return out if shape is None else out.reshape(shape)
