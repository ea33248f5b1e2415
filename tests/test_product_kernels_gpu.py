"""nk_product_field / nk_product_marginal / nk_mirror_combine (include/niftyk.h; reference library/correlated_fields.py:713-764)
through the C ABI against numpy: the amplitude field of a product spectrum and its tangent, the weighted marginal sums of
its adjoint, and the per-sub-space Hartley transform rebuilt from the genuine N-D one."""
import ctypes

import numpy as np
import pytest
import scipy.fft
import torch

pytestmark = pytest.mark.gpu


def _dev(a, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(a if dtype is None else a.astype(dtype))).cuda()


@pytest.mark.parametrize("sizes,nbs", [((16, 48), (5, 9)), ((6, 10, 8), (3, 4, 5)), ((128,), (7,)), ((40, 1, 33), (4, 1, 6))])
def test_product_field_tangent_and_marginals(sizes, nbs):
    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    lib = L.load()
    rng = np.random.default_rng(2)
    nsub = len(sizes)
    pidx = [rng.integers(0, nb, size=s).astype(np.int32) for s, nb in zip(sizes, nbs)]
    tab = [rng.uniform(0.5, 2.0, size=nb) for nb in nbs]
    dtab = [rng.normal(size=nb) for nb in nbs]
    scale, dscale = 1.7, -0.3
    keep = [_dev(p) for p in pidx] + [_dev(t) for t in tab] + [_dev(t) for t in dtab] + [_dev(np.array([scale])), _dev(np.array([dscale]))]
    q = L.Product()
    q.nsub = nsub
    for i in range(nsub):
        q.size[i], q.pidx[i], q.tab[i], q.dtab[i] = sizes[i], keep[i].data_ptr(), keep[nsub + i].data_ptr(), keep[2 * nsub + i].data_ptr()
    q.scale, q.dscale = keep[-2].data_ptr(), keep[-1].data_ptr()
    # numpy: outer products of the gathered tables
    fac = [t[p] for t, p in zip(tab, pidx)]
    dfac = [t[p] for t, p in zip(dtab, pidx)]

    def outer(parts):
        out = parts[0]
        for p in parts[1:]:
            out = np.multiply.outer(out, p)
        return out

    field = scale * outer(fac)
    tangent = dscale * outer(fac) + scale * sum(outer([dfac[j] if j == i else fac[j] for j in range(nsub)]) for i in range(nsub))
    n = int(np.prod(sizes))
    for dtype, tol in ((torch.float64, 1e-14), (torch.float32, 1e-6)):
        out = torch.empty(n, dtype=dtype, device="cuda")
        L.check(lib.nk_product_field(ctypes.byref(q), 0, out.data_ptr(), B.dtype_code(out), B._stream()))
        assert np.max(np.abs(out.cpu().numpy().reshape(field.shape) - field)) < tol * np.max(np.abs(field))
        L.check(lib.nk_product_field(ctypes.byref(q), 1, out.data_ptr(), B.dtype_code(out), B._stream()))
        assert np.max(np.abs(out.cpu().numpy().reshape(field.shape) - tangent)) < 10 * tol * np.max(np.abs(tangent))
    # adjoint: weighted marginals, twice the same bits
    w = rng.normal(size=field.shape)
    wd = _dev(w.reshape(-1))
    for which in range(nsub):
        others = scale * outer([np.ones_like(fac[j]) if j == which else fac[j] for j in range(nsub)])
        axes = tuple(j for j in range(nsub) if j != which)
        want = (w * others).sum(axis=axes) if axes else w * others
        scratch = torch.empty(max(1, lib.nk_product_marginal_scratch(ctypes.byref(q), which) // 8), dtype=torch.float64, device="cuda")
        got = [torch.empty(sizes[which], dtype=torch.float64, device="cuda") for _ in range(2)]
        for g in got:
            L.check(lib.nk_product_marginal(ctypes.byref(q), which, wd.data_ptr(), scratch.data_ptr(), g.data_ptr(), B._stream()))
        assert torch.equal(got[0], got[1])
        assert np.max(np.abs(got[0].cpu().numpy() - want)) < 1e-12 * max(1.0, np.max(np.abs(want)))


@pytest.mark.parametrize("shape,group", [((16, 12), (0, 1)), ((8, 6, 10), (0, 1, 1)), ((8, 6, 10), (0, 0, 1)), ((6, 4, 10), (0, 1, 2))])
def test_mirror_combine_turns_the_genuine_hartley_into_the_separable_one(shape, group):
    from nifty_amd import _lib as L
    from nifty_amd import backend as B
    from nifty_amd.correlated_fields import _ProductFieldNode

    lib = L.load()
    rng = np.random.default_rng(5)
    x = rng.normal(size=shape)

    def hartley(a, axes):
        f = scipy.fft.fftn(a, axes=axes)
        return f.real + f.imag

    genuine = hartley(x, tuple(range(len(shape))))
    separable = x
    for g in sorted(set(group)):
        separable = hartley(separable, tuple(ax for ax, gg in enumerate(group) if gg == g))
    nsub = len(set(group))
    coef = (ctypes.c_double * (1 << nsub))(*_ProductFieldNode.MIRROR[nsub])
    shp = (ctypes.c_int64 * len(shape))(*shape)
    grp = (ctypes.c_int * len(shape))(*group)
    src = _dev(genuine.reshape(-1))
    dst = torch.empty_like(src)
    L.check(lib.nk_mirror_combine(len(shape), shp, grp, nsub, coef, src.data_ptr(), dst.data_ptr(), 1.0, 0.25, B.dtype_code(src),
                                  B._stream()))
    got = dst.cpu().numpy().reshape(shape)
    assert np.max(np.abs(got - (separable + 0.25))) < 1e-12 * np.max(np.abs(separable))
