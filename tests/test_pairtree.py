"""Host logic of the pairwise sum over samples (reference utilities.py:349-414): the binary-counter bookkeeping that the fused
engine uses to build the sum INSIDE the VJP epilogues (engine._PairTree) brackets its terms exactly like parallel.pair_tree,
for every number of samples and every number of partial sums the epilogue can carry."""
import pytest

from nifty_amd import parallel
from nifty_amd.engine import _PairTree


class _Vec:
    def __init__(self, name):
        self.name, self.expr = name, None


@pytest.mark.parametrize("max_carries", [0, 1, 2])
def test_pair_tree_bookkeeping_brackets_like_the_reference(max_carries, monkeypatch):
    monkeypatch.setattr(_PairTree, "MAX_CARRIES", max_carries)
    for n in range(1, 34):
        made = []

        def scratch():
            made.append(_Vec(f"S{len(made)}"))
            return made[-1]

        out = _Vec("out")
        tree = _PairTree(out, scratch)
        for i in range(n):
            dest = tree.place(i == n - 1)
            assert len(dest.carries) <= max(max_carries, 0)
            g = f"c{i}"
            for c in dest.carries:  # innermost first, as the epilogue adds them
                g = f"({c.expr}+{g})"
            if dest.accumulate:
                g = f"({dest.xi.expr}+{g})"
            dest.xi.expr = g
            for target, source in dest.after:  # explicit additions beyond what the epilogue carries
                target.expr = f"({target.expr}+{source.expr})"
        assert tree.total() is out
        want = parallel.tree_fold([f"c{i}" for i in range(n)], lambda a, b: f"({a}+{b})")
        assert out.expr == want, (n, max_carries)
        # the output and at most floor(log2(n - 1)) scratch vectors, whatever the epilogue can carry
        assert len(made) <= max(0, (n - 1).bit_length() - 1)


def test_sample_plan_run_together_equals_run():
    """parallel.SamplePlan.run_together (draw everything a pair needs, solve the pairs together, finish sample by sample)
    visits the samples, their random contexts and their mirrored flags exactly like run()."""
    import numpy as np

    from nifty_amd import random

    def recipe(use_together):
        random.push_sseq_from_seed(77)
        try:
            plan = parallel.SamplePlan(3, True)
            seen = []

            def prepare(seed):
                return float(random.Random.normal(np.float64, (1,))[0])

            def solve_one(x):
                return (x, 2.0 * x)

            def finish(pair, mirrored):
                seen.append(float(random.Random.normal(np.float64, (1,))[0]))  # finish() runs inside the sample's context too
                return (-pair[1] if mirrored else pair[1], mirrored)

            if use_together:
                out = plan.run_together(prepare, lambda jobs: [solve_one(j) for j in jobs], finish)
            else:
                out = plan.run(lambda seed: solve_one(prepare(seed)), finish)
            return out, seen, plan.n_total
        finally:
            random.pop_sseq()

    a, b = recipe(False), recipe(True)
    assert a[0] == b[0] and a[2] == b[2] == 6
    assert len(a[0]) == 6 and [m for _, m in a[0]] == [False, True] * 3
