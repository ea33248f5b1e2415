"""The host part of a sample's prior draws made one sample ahead on a worker thread (engine._HostDrawAhead,
FusedModel.host_draws_before_xi; reference draw order: multi_field.py:143-153, kl_energies.py:132-146): the same numbers and
the same generator state as drawing them in place inside the sample's random context."""
import numpy as np

from nifty_amd import engine as E


class _Model:
    nb = 1234
    host_draws_before_xi = E.FusedModel.host_draws_before_xi


def test_draws_ahead_are_the_in_place_draws():
    model = _Model()
    seeds = np.random.SeedSequence(5).spawn(4)
    ahead = E._HostDrawAhead(model, seeds)
    before_xi = list(E.LATENT_KEYS[:E.LATENT_KEYS.index("xi")])
    assert before_xi == ["asperity", "flexibility", "fluctuations", "loglogavgslope", "spectrum"]
    for sq in seeds:
        vals, state, fresh = ahead.take(sq)
        rng = np.random.default_rng(sq)  # what random.Context(sq) puts on the stack
        assert fresh == rng.bit_generator.state  # (draw_prior adopts `state` only from a context generator still in this state)
        for k in before_xi:
            ref = rng.normal(0.0, 1.0, (2, model.nb - 2) if k == "spectrum" else ())
            assert np.array_equal(vals[k], ref)
        assert state == rng.bit_generator.state
        # the generator that continues from the state makes the draws that follow xi in the reference's order
        cont = np.random.default_rng(0)
        cont.bit_generator.state = state
        assert cont.normal() == rng.normal()
    # a seed that was not announced gets nothing (the caller then draws in place)
    assert ahead.take(np.random.SeedSequence(99)) is None


def test_close_drops_the_pending_draw():
    """An exception between two samples must not leave the worker behind: close() cancels what was not taken."""
    ahead = E._HostDrawAhead(_Model(), np.random.SeedSequence(6).spawn(3))
    ahead.close()
    assert not ahead._pending and ahead._pool._shutdown
