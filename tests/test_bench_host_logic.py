"""Host-side logic of bench.py that the contract depends on (no GPU): a timed region holds EXACTLY K steps also when one
call of the trainer stands for several of them, and Python's cyclic collector stays out of the region."""
import gc
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench  # noqa: E402


class _Ctx:
    def __init__(self):
        self.syncs = 0

    def sync(self):
        self.syncs += 1

    def max_over_ranks(self, seconds):
        return seconds


def test_every_nth_any_window_of_k_calls_holds_k_steps():
    for n in (1, 2, 4):
        done = []
        step = bench.every_nth(n, lambda: done.append(n))
        for phase in range(7):                 # warm-up / settle phases of any length before the window
            for _ in range(phase):
                step()
            before = sum(done)
            for _ in range(20):                # K = 20: a multiple of 1, 2 and 4
                step()
            assert sum(done) - before == 20, (n, phase)


def test_timed_runs_k_steps_between_two_syncs_with_the_collector_off():
    ctx, seen = _Ctx(), []
    assert gc.isenabled()
    el = bench.timed(ctx, lambda: seen.append((gc.isenabled(), ctx.syncs)), 9)
    assert len(seen) == 9 and el >= 0.0
    assert all(not on and syncs == 1 for on, syncs in seen)      # one sync before the first step, collector disabled inside
    assert ctx.syncs == 2 and gc.isenabled()                     # one sync after the last step, collector back on
    gc.disable()
    try:
        bench.timed(ctx, lambda: None, 1)
        assert not gc.isenabled()                                # ... and left as it was found
    finally:
        gc.enable()
