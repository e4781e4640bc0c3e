"""fold.get_context: one context per (device, host thread); contexts of threads that ended without close_contexts() are closed
by the next call instead of piling up (no GPU: the Context class is replaced by a counter)."""
import importlib
import threading

F = importlib.import_module("trrosettax2-dynamics_amd.fold")


class FakeCtx:
    made, closed = 0, 0

    def __init__(self, device):
        self.device, self.lanes = device, 1
        FakeCtx.made += 1

    def set_lanes(self, n):
        self.lanes = n

    def close(self):
        FakeCtx.closed += 1


def test_dead_threads_contexts_are_closed(monkeypatch):
    monkeypatch.setattr(F, "Context", FakeCtx)
    F.close_contexts(all_threads=True)
    FakeCtx.made = FakeCtx.closed = 0
    seen = []

    def worker():
        c = F.get_context(0)
        assert F.get_context(0) is c          # cached for the thread
        seen.append(c)

    for _ in range(5):                        # thread churn without close_contexts()
        t = threading.Thread(target=worker)
        t.start(); t.join()
    mine = F.get_context(0, lanes=2)          # sweeps the five dead threads' contexts
    assert mine.lanes == 2 and mine not in seen
    assert FakeCtx.closed == FakeCtx.made - 1 and len(F._CTX) == 1
    F.close_contexts()
    assert FakeCtx.closed == FakeCtx.made and not F._CTX and not F._CTX_OWNER


def test_live_threads_keep_their_contexts(monkeypatch):
    monkeypatch.setattr(F, "Context", FakeCtx)
    F.close_contexts(all_threads=True)
    FakeCtx.made = FakeCtx.closed = 0
    go, done = threading.Event(), threading.Event()
    got = {}

    def worker():
        got["a"] = F.get_context(0)
        done.set(); go.wait(10)
        got["b"] = F.get_context(0)
        F.close_contexts()

    t = threading.Thread(target=worker)
    t.start(); done.wait(10)
    F.get_context(0)                          # another thread's call must not close the live worker's context
    assert FakeCtx.closed == 0
    go.set(); t.join()
    assert got["a"] is got["b"]
    F.close_contexts(all_threads=True)
    assert FakeCtx.closed == FakeCtx.made == 2
