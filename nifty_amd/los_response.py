"""Line-of-sight response and mask (BASELINE config 4).

``LOSResponse`` (reference library/los_response.py:100-253): every line of sight is walked through the pixel
grid; the path length inside each pixel -- optionally weighted by the survival function of a Gaussian
uncertainty of the inverse distance -- becomes one entry of a sparse ``n_los x n_pix`` matrix (float32
weights, int32 indices as in the reference, :194-197).  TIMES = SpMV, ADJOINT_TIMES = SpMV^T.  The set-up runs
on the host in numpy (as in the reference); on a GPU both products run in libniftyk as row sums of a CSR matrix
(``nk_csr_rowsum``: the matrix for TIMES, its transpose -- built once on the host -- for ADJOINT_TIMES, so every
output is summed in a fixed order without atomics), on the host through scipy.sparse.
"""
import os

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


# ---- the response re-ordered by tiles of the grid (nk_tiled_rowsum, include/niftyk.h) --------------------------------
TILED_BLOCK, TILED_STEP, TILED_SEGMAX = 8, 64, 128  # entries per block, blocks per wavefront step, longest segment (nk_vec.hip)


def tiled_plan(rowptr, col, wgt, grid_shape, th=None, tw=None):
    """Host arrays of ``nk_tiled_csr`` for the CSR matrix (rowptr, col, wgt) whose columns are the points of a C-ordered
    grid.  Entries are sorted by (tile of th x tw points of the last two axes, row, original position) and cut into SEGMENTS
    (one row inside one tile, at most TILED_SEGMAX entries), every segment padded with zero weights to whole BLOCKS of
    TILED_BLOCK entries -- one lane of a wavefront takes one block.  A tile's blocks are walked in STEPS of TILED_STEP blocks
    (one wavefront instruction stream each); the blocks of one segment inside one step form a PIECE and share a partial-sum
    slot, the slots of a row being consecutive.  Index bookkeeping at set-up, like the reference's own matrix construction
    (library/los_response.py:194-221); returns a dict of numpy arrays + the grid / tile sizes."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    wgt = np.asarray(wgt, dtype=np.float32)
    n_rows, nnz = len(rowptr) - 1, len(col)
    shape = tuple(int(n) for n in grid_shape)
    ny, nx = (1, shape[0]) if len(shape) == 1 else shape[-2:]
    if th is None or tw is None:
        th, tw = (1, 4096) if ny == 1 else (32, 64)
    th, tw = min(int(th), ny), min(int(tw), nx)
    if th * tw > 32768:
        raise ValueError("a tile holds at most 32768 grid points")
    nty, ntx = -(-ny // th), -(-nx // tw)
    o, rem = np.divmod(col, ny * nx)
    gy, gx = np.divmod(rem, nx)
    tile = (o * nty + gy // th) * ntx + gx // tw
    loc = ((gy % th) * tw + gx % tw).astype(np.uint16)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr))
    order = np.argsort(tile * n_rows + rows, kind="stable")
    tile, rows, loc, wgt = tile[order], rows[order], loc[order], wgt[order]
    # segments: runs of equal (tile, row), split after every TILED_SEGMAX entries
    pos = np.arange(nnz)
    new_run = np.ones(nnz, dtype=bool)
    new_run[1:] = (tile[1:] != tile[:-1]) | (rows[1:] != rows[:-1])
    run_start = np.maximum.accumulate(np.where(new_run, pos, 0))
    seg_first = np.nonzero(new_run | ((pos - run_start) % TILED_SEGMAX == 0))[0]
    n_segs = len(seg_first)
    seg_len = np.diff(np.concatenate([seg_first, [nnz]]))
    seg_tile, seg_row = tile[seg_first], rows[seg_first]
    seg_blocks = -(-seg_len // TILED_BLOCK)
    seg_blk0 = np.concatenate([[0], np.cumsum(seg_blocks)])  # first block of every segment; [-1] = number of blocks
    n_blocks = int(seg_blk0[-1])
    # items = non-empty tiles with their block ranges
    new_item = np.ones(n_segs, dtype=bool)
    new_item[1:] = seg_tile[1:] != seg_tile[:-1]
    item_first_seg = np.nonzero(new_item)[0]
    item_tile = seg_tile[item_first_seg].astype(np.int32)
    item_blk = np.concatenate([seg_blk0[item_first_seg], [n_blocks]]).astype(np.int64)
    # blocks -> pieces (same segment, same step of the tile) -> slots grouped by row
    blk_seg = np.repeat(np.arange(n_segs), seg_blocks)
    item_of_seg = np.cumsum(new_item) - 1
    blk_step = (np.arange(n_blocks) - item_blk[:-1][item_of_seg[blk_seg]]) // TILED_STEP if n_blocks else np.zeros(0, dtype=np.int64)
    new_piece = np.ones(n_blocks, dtype=bool)
    new_piece[1:] = (blk_seg[1:] != blk_seg[:-1]) | (blk_step[1:] != blk_step[:-1])
    piece_of_blk = np.cumsum(new_piece) - 1
    piece_row = seg_row[blk_seg[new_piece]]
    n_pieces = len(piece_row)
    by_row = np.argsort(piece_row, kind="stable")
    piece_slot = np.empty(n_pieces, dtype=np.int32)
    piece_slot[by_row] = np.arange(n_pieces, dtype=np.int32)
    row_slot = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(piece_row, minlength=n_rows), out=row_slot[1:])
    # padded entries: segment s at [8 blk0, 8 blk0 + len); pads repeat the segment's first position with weight zero
    seg_of_entry = np.repeat(np.arange(n_segs), seg_len)
    dst = TILED_BLOCK * seg_blk0[:-1][seg_of_entry] + (pos - seg_first[seg_of_entry])
    loc_p = np.repeat(loc[seg_first], TILED_BLOCK * seg_blocks).astype(np.uint16)
    wgt_p = np.zeros(TILED_BLOCK * n_blocks, dtype=np.float32)
    loc_p[dst], wgt_p[dst] = loc, wgt
    return dict(n_rows=n_rows, n_slots=n_pieces, n_items=len(item_tile), ny=int(ny), nx=int(nx), th=th, tw=tw, nnz=nnz,
                item_tile=item_tile, item_blk=item_blk, blk_slot=piece_slot[piece_of_blk].astype(np.int32), row_slot=row_slot,
                loc=loc_p, wgt=wgt_p)


def tiled_rowsum_host(plan, x):
    """What the two launches of nk_tiled_rowsum compute, addition by addition (numpy; the emulation the CPU tests hold the
    plan builder to -- and the device result bit for bit): per block eight products added in order, then the lanes of a
    piece joined by the shuffle-down steps 1, 2, 4, 8 among equal slots; per row its slots, 16 lanes + tree."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, plan["ny"], plan["nx"])
    th, tw = plan["th"], plan["tw"]
    nty, ntx = -(-plan["ny"] // th), -(-plan["nx"] // tw)
    partial = np.zeros(plan["n_slots"])
    for it in range(plan["n_items"]):
        t = int(plan["item_tile"][it])
        o, ty, tx = t // (nty * ntx), (t // ntx) % nty, t % ntx
        tile = np.zeros((th, tw))
        blk = x[o, ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
        tile[:blk.shape[0], :blk.shape[1]] = blk
        tile = tile.reshape(-1)
        b0, b1 = int(plan["item_blk"][it]), int(plan["item_blk"][it + 1])
        for s0 in range(b0, b1, TILED_STEP):
            s1 = min(s0 + TILED_STEP, b1)
            acc = np.zeros(TILED_STEP)
            slot = np.full(TILED_STEP + 8, -2, dtype=np.int64)
            slot[:s1 - s0] = plan["blk_slot"][s0:s1]
            e = slice(TILED_BLOCK * s0, TILED_BLOCK * s1)
            prod = (plan["wgt"][e].astype(np.float64) * tile[plan["loc"][e]]).reshape(-1, TILED_BLOCK)
            for i in range(TILED_BLOCK):
                acc[:s1 - s0] += prod[:, i]
            for off in (1, 2, 4, 8):
                other = np.concatenate([acc[off:], np.zeros(off)])
                acc = acc + np.where(slot[off:off + TILED_STEP] == slot[:TILED_STEP], other, 0.0)
            head = np.ones(s1 - s0, dtype=bool)
            head[1:] = slot[1:s1 - s0] != slot[:s1 - s0 - 1]
            partial[slot[:s1 - s0][head]] = acc[:s1 - s0][head]
    y = np.zeros(plan["n_rows"])
    for r in range(plan["n_rows"]):
        y[r] = _lane_tree_sum(partial[plan["row_slot"][r]:plan["row_slot"][r + 1]])
    return y


def _lane_tree_sum(v):
    """16 lanes add v[q], v[q + 16], ... in order; shuffle-down tree 8, 4, 2, 1 (lane 0 holds the result)."""
    lanes = np.zeros(16)
    for q in range(16):
        for t in v[q::16]:
            lanes[q] += t
    for off in (8, 4, 2, 1):
        lanes[:off] = lanes[:off] + lanes[off:2 * off]
    return lanes[0]


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

    def _tiled_matrix(self, device):
        """TIMES through the grid-tiled order of the matrix (nk_tiled_rowsum) when the lines are long enough to pay for it."""
        key = ("tiled", str(device))
        if key not in self._dev:
            use = os.environ.get("NK_TILED_RESPONSE", "1") != "0" and len(self._col) >= 64 * max(1, self._nlos)
            self._dev[key] = B.TiledMatrix(tiled_plan(self._rowptr, self._col, self._wgt, self._domain[0].shape), device) if use else None
        return self._dev[key]

    def apply(self, x, mode):
        self._check_input(x, mode)
        v = x.val
        if v.is_cuda:
            nnz = len(self._col)
            if mode == self.TIMES:
                tm = self._tiled_matrix(v.device)
                if tm is not None:
                    y = torch.empty(self._nlos, dtype=v.dtype, device=v.device)
                    return Field(self._target, tm.rowsum([v.contiguous().reshape(-1)], [y])[0])
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

    def __init__(self, rowptr, col, wgt, n_pix, grid_shape=None):
        from scipy.sparse import csr_matrix

        self.n_data, self.n_pix = len(rowptr) - 1, int(n_pix)
        self.grid_shape = None if grid_shape is None else tuple(int(n) for n in grid_shape)
        self._tiled = {}
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
        return SparseResponse(m.indptr, m.indices, m.data, los.domain.size, los.domain.shape)

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

    def tiled(self, device):
        """The matrix re-ordered by grid tiles for TIMES (``tiled_plan`` / nk_tiled_rowsum) on `device`, or None: rows too
        short to pay for it, no grid geometry, or NK_TILED_RESPONSE=0 (the wavefront-per-row kernel then)."""
        key = str(device)
        if key not in self._tiled:
            use = (self.grid_shape is not None and os.environ.get("NK_TILED_RESPONSE", "1") != "0"
                   and self._m.nnz >= 64 * max(1, self.n_data))
            self._tiled[key] = None
            if use:
                th_tw = [int(v) for v in os.environ["NK_TILED_SHAPE"].split(",")] if "NK_TILED_SHAPE" in os.environ else [None, None]
                self._tiled[key] = B.TiledMatrix(tiled_plan(self._m.indptr, self._m.indices, self._m.data, self.grid_shape,
                                                            *th_tw), device)
        return self._tiled[key]

    def times(self, x):
        """R x: x any tensor with n_pix entries -> [n_data]."""
        tm = self.tiled(x.device)
        if tm is not None:
            y = torch.empty(self.n_data, dtype=x.dtype, device=x.device)
            return tm.rowsum([x.contiguous().reshape(-1)], [y])[0]
        a = self._arrays(x.device)
        return B.spmv(a[0], a[1], a[2], x.contiguous().reshape(-1), self.n_data, B.lanes_for(self._m.nnz, self.n_data))

    def adjoint(self, y, shape=None):
        """R^T y: [n_data] -> n_pix entries (reshaped to `shape`)."""
        a = self._arrays(y.device)
        out = B.spmv(a[3], a[4], a[5], y.contiguous(), self.n_pix, B.lanes_for(self._m.nnz, self.n_pix))
        return out if shape is None else out.reshape(shape)
