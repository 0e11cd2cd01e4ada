"""Last file of the GPU suite (files run in name order): nothing in this pytest process had to repeat an analysis
after a launch time-out (PW_E_TIMEOUT), and the shared per-device context saw no pacing gate expire.  A green suite
with silent repeats would otherwise look like a healthy one (round-4 review, reliability item)."""
import pytest

pytestmark = pytest.mark.gpu


def test_no_analysis_was_repeated_after_a_timeout():
    from pywindow_amd import _lib, engine

    assert _lib.retries_total() == 0, "an analysis was repeated after PW_E_TIMEOUT during this test session"
    ctx = engine.context(0)
    assert ctx.retries == 0
    gates = ctx.gate_timeouts
    assert gates["residency"] == 0, gates


def test_timeout_is_a_code_of_its_own_and_is_counted():
    """PW_E_TIMEOUT maps to PwTimeoutError; a binding that repeats an analysis says so and the counters see it
    (on a context of its own, so that the session's own count above is not disturbed -- this test runs after it)."""
    from pywindow_amd import _lib

    ctx = _lib.Context(0)
    try:
        before = _lib.retries_total()
        assert ctx.retries == 0
        with pytest.raises(_lib.PwTimeoutError):
            _lib._check(_lib.E_TIMEOUT, "a call")
        _lib.load().pw_context_count_retry(ctx._h)
        assert ctx.retries == 1 and _lib.retries_total() == before + 1
        q = ctx.queue_state()
        assert len(q) == 4 and all(s["error"] == 0 for s in q)
    finally:
        ctx.close()
