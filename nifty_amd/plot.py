"""A small `ift.Plot` for the fields of the path, so that reference scripts that end in a plot run unchanged (reference
nifty/cl/plot.py:532-720 is the interface; this is not a port of its renderer): line plots of fields on a one-dimensional
RGSpace or a PowerSpace (log-log), images of fields on a two-dimensional RGSpace -- one panel per index of any further
one-dimensional sub-domains --, histograms of fields on an UnstructuredDomain, energy histories.  Spherical maps and the RGB
rendering of multi-frequency cubes are out of scope (NotImplementedError).  matplotlib is imported on first use; without it
`output()` warns and draws nothing, like the reference."""
from itertools import product
from warnings import warn

import numpy as np

from .domains import DomainTuple, PowerSpace, RGSpace, UnstructuredDomain
from .field import Field, MultiField
from .minimization import EnergyHistory

_PER_CURVE = ("label", "alpha", "linewidth", "color", "linestyle")


def _per_curve(kwargs, n):
    """the i-th entry of list-valued curve options, the value itself for scalars"""
    return [{k: (v[i] if isinstance(v, (list, tuple)) else v) for k, v in kwargs.items() if k in _PER_CURVE and v is not None}
            for i in range(n)]


def _limits(ax, kw):
    ax.set_xlim(kw.get("xmin"), kw.get("xmax"))
    ax.set_ylim(kw.get("ymin"), kw.get("ymax"))


def _lines(fields, ax, kw):
    space = fields[0].domain[0]
    if isinstance(space, PowerSpace):
        x = np.asarray(space.k_lengths, dtype=float).copy()
        x[0] = 0.5 * x[1] if len(x) > 1 else 1.0  # the k = 0 bin on a logarithmic axis
        ax.set_xscale("log")
        ax.set_yscale("log")
    else:
        x = np.arange(space.shape[0]) * space.distances[0]
    for f, opts in zip(fields, _per_curve(kw, len(fields))):
        ax.plot(x, f.asnumpy(), **opts)
    if kw.get("label") is not None:
        ax.legend()
    _limits(ax, kw)


def _image(field, ax, kw):
    space = field.domain[0]
    nx, ny = space.shape
    dx, dy = space.distances
    im = ax.imshow(field.asnumpy().T, origin="lower", extent=[0, nx * dx, 0, ny * dy], cmap=kw.get("cmap"),
                   vmin=kw.get("vmin", kw.get("zmin")), vmax=kw.get("vmax", kw.get("zmax")), norm=kw.get("norm"),
                   aspect=kw.get("aspect", "equal"))
    ax.figure.colorbar(im, ax=ax)
    _limits(ax, kw)


def _histogram(fields, ax, kw):
    for f, opts in zip(fields, _per_curve(kw, len(fields))):
        ax.hist(f.asnumpy().ravel(), bins=kw.get("bins", 50), density=True, histtype="step", **opts)
    _limits(ax, kw)


def _history(histories, ax, kw):
    for h, opts in zip(histories, _per_curve(kw, len(histories))):
        ax.plot(np.asarray(h.time_stamps, dtype=float) - (0.0 if kw.get("skip_timestamp_conversion") else h.time_stamps[0]),
                h.energy_values, **opts)
    if kw.get("yscale"):
        ax.set_yscale(kw["yscale"])
    _limits(ax, kw)


def _draw(content, ax, kw):
    for key, setter in (("title", ax.set_title), ("xlabel", ax.set_xlabel), ("ylabel", ax.set_ylabel)):
        if kw.get(key) is not None:
            setter(kw[key])
    first = content[0]
    if isinstance(first, EnergyHistory):
        return _history(content, ax, kw)
    dom = first.domain
    if len(dom) != 1:
        raise NotImplementedError("Plot: fields over several sub-domains are split into panels by add()")
    space = dom[0]
    if isinstance(space, UnstructuredDomain):
        return _histogram(content, ax, kw)
    if isinstance(space, PowerSpace) or (isinstance(space, RGSpace) and len(space.shape) == 1):
        return _lines(content, ax, kw)
    if isinstance(space, RGSpace) and len(space.shape) == 2:
        if len(content) != 1:
            raise ValueError("Plot: one field per image panel")
        return _image(first, ax, kw)
    raise NotImplementedError(f"Plot: no renderer for fields on {space!r} (spherical maps, grids beyond two dimensions)")


class Plot:
    """`add()` panels, then `output()` them in a grid -- to `name` (a file) or on screen."""

    def __init__(self):
        self._panels = []

    def add(self, f, **kwargs):
        if f is None:
            self._panels.append((None, {}))
            return
        items = [f] if isinstance(f, (Field, MultiField, EnergyHistory)) else list(f)
        if not items or not isinstance(items[0], (Field, MultiField, EnergyHistory)):
            raise TypeError("Incorrect data type. You can only add Fields or EnergyHistories, or None")
        if all(isinstance(it, MultiField) for it in items):  # one panel per key
            for key in items[0].domain.keys():
                title = f"{key} {kwargs['title']}" if "title" in kwargs else str(key)
                self.add([it[key] for it in items], **dict(kwargs, title=title))
            return
        dom = None if isinstance(items[0], EnergyHistory) else items[0].domain
        if dom is not None and len(dom) > 1:
            if "freq_space_idx" in kwargs:
                raise NotImplementedError("Plot: the RGB rendering of multi-frequency cubes is out of scope")
            ranks = [len(sub.shape) for sub in dom]
            if ranks.count(2) != 1 or ranks.count(1) != len(ranks) - 1:
                raise NotImplementedError("Plot: need one two-dimensional sub-domain and one-dimensional ones beside it")
            image = ranks.index(2)
            counts = [sub.size for i, sub in enumerate(dom) if i != image]
            for index in product(*[range(n) for n in counts]):
                where = list(index)
                where[image:image] = [slice(None), slice(None)]
                for it in items:
                    self._panels.append(([Field.from_raw(DomainTuple.make(dom[image]), it.asnumpy()[tuple(where)])], kwargs))
            return
        self._panels.append((items, kwargs))

    def prepare(self, **kwargs):
        """The matplotlib figure of the panels added so far (the caller closes it); None without matplotlib."""
        try:
            import matplotlib.pyplot as plt
        except ImportError:
            warn("Since matplotlib is not installed, no plots are generated.")
            return None
        count = len(self._panels)
        if count == 0:
            raise ValueError("Use .add to add plots to your plotting routine.")
        nx, ny = kwargs.get("nx", 0), kwargs.get("ny", 0)
        if nx == ny == 0:
            ny = int(np.ceil(np.sqrt(count)))
        if nx == 0:
            nx = int(np.ceil(count / ny))
        if ny == 0:
            ny = int(np.ceil(count / nx))
        if nx * ny < count:
            raise ValueError(f"Figure dimensions not sufficient for number of plots. Available plot slots: {nx * ny}, "
                             f"number of plots: {count}")
        fig = plt.figure(figsize=(kwargs.get("xsize", 6 * nx), kwargs.get("ysize", 6 * ny)))
        if kwargs.get("title") is not None:
            fig.suptitle(kwargs["title"])
        for i, (content, kw) in enumerate(self._panels):
            if content is not None:
                _draw(content, fig.add_subplot(ny, nx, i + 1), kw)
        fig.tight_layout()
        return fig

    def output(self, **kwargs):
        """Draw; `name`: the file to write (else the figure is shown), `dpi`, `block` as in matplotlib, the rest: prepare()."""
        fig = self.prepare(**kwargs)
        if fig is None:
            return
        import matplotlib.pyplot as plt

        if kwargs.get("name") is None:
            plt.show(block=kwargs.get("block", True))
        else:
            fig.savefig(kwargs["name"], dpi=kwargs.get("dpi"))
        plt.close(fig)


def single_plot(field, **kwargs):
    """One field, one figure: the options of Plot.add and Plot.output in one call (sugar.py:521-529)."""
    p = Plot()
    p.add(field, **kwargs)
    kwargs.pop("title", None)
    p.output(**kwargs)


def plot_priorsamples(op, n_samples=5, common_colorbar=True, **kwargs):
    """`n_samples` prior draws pushed through `op`, images side by side or curves in one panel (sugar.py:532-561)."""
    from .field import from_random

    draws = [op(from_random(op.domain)) for _ in range(n_samples)]
    if not isinstance(draws[0], Field):
        raise TypeError("the operator must map to a Field")
    space = draws[0].domain[0] if len(draws[0].domain) == 1 else None
    p = Plot()
    if isinstance(space, RGSpace) and len(space.shape) == 2:
        lo = min(float(d.asnumpy().min()) for d in draws) if common_colorbar else None
        hi = max(float(d.asnumpy().max()) for d in draws) if common_colorbar else None
        for d in draws:
            p.add(d, vmin=lo, vmax=hi, **kwargs)
            kwargs.pop("title", None)
    else:
        p.add(draws, **kwargs)
    p.output(**kwargs)
