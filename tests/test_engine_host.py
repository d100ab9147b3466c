"""Host-side logic of the engine that needs no GPU: channel chunking, sort size classes, the bounded size-guess cache."""
import pytest
import torch

from deblur4dgs_amd import engine


def test_channel_chunks_cover_every_count_with_instantiated_widths():
    for D in range(0, 70):
        ch = engine.channel_chunks(D)
        assert ch[0][0] == 0 and ch[-1][1] == D
        for (a, b, w), nxt in zip(ch, ch[1:] + [None]):
            assert w in engine.SUPPORTED_D and b - a <= w and (b - a == 16 or nxt is None)
            if nxt is not None:
                assert nxt[0] == b
        if D in engine.SUPPORTED_D:
            assert ch == [(0, D, D)]  # the common case: one pass, nothing padded
    # the reference's layouts: 3 + mask + 3B track channels (+ depth rides on the last chunk)
    assert engine.channel_chunks(16) == [(0, 16, 16)]
    assert engine.channel_chunks(3 + 1 + 18) == [(0, 16, 16), (16, 22, 8)]


def test_sort_classes_leave_headroom_and_fall_back_to_unknown():
    assert engine._sort_class(0) == 512 and engine._sort_class(341) == 512  # one wave, registers only (k_tile_sort_w)
    assert engine._sort_class(342) == 2048 and engine._sort_class(1365) == 2048
    assert engine._sort_class(1366) == 4096 and engine._sort_class(5000) == 8192 and engine._sort_class(10922) == 16384
    assert engine._sort_class(10923) == 0  # longer than the largest LDS class with 50 % headroom: every class is launched


def test_size_guess_cache_is_bounded_and_buckets_n():
    dev = torch.device("cuda", 0)  # only .index is used
    k1, k2 = engine._size_key(dev, 8, 300_000, 512, 288), engine._size_key(dev, 8, 310_000, 512, 288)
    assert k1 == k2  # densification moves N a little: same entry (two leading bits)
    assert engine._size_key(dev, 8, 600_000, 512, 288) != k1
    engine._SIZE_GUESS.clear()
    for i in range(3 * engine._SIZE_GUESS_MAX):
        engine._guess_put(("k", i), (i, 2048))
    assert len(engine._SIZE_GUESS) == engine._SIZE_GUESS_MAX
    assert engine._guess_get(("k", 0)) is None and engine._guess_get(("k", 3 * engine._SIZE_GUESS_MAX - 1)) is not None
    engine._SIZE_GUESS.clear()


def test_control_stats_sink_is_validated():
    import pytest

    good = {"xys_grad_norm_acc": torch.zeros(5), "vis_count": torch.zeros(5, dtype=torch.int64), "max_radii": torch.zeros(5),
            "batch_size": 2}
    assert engine._check_stats(good, 5) is good
    with pytest.raises(ValueError):
        engine._check_stats(good, 6)
    with pytest.raises(ValueError):
        engine._check_stats({**good, "vis_count": torch.zeros(5)}, 5)
    with pytest.raises(ValueError):
        engine._check_stats({**good, "batch_size": 0}, 5)


def test_deferred_records_queue_up_and_are_all_verified():
    """Host logic of `deferred_size_check` without a GPU: several pending records per shape, every one is read, an
    overflow in ANY of them raises (only the newest used to survive)."""
    from deblur4dgs_amd import engine

    class Ev:
        def __init__(self, done):
            self.done = done

        def query(self):
            return self.done

        def synchronize(self):
            self.done = True

    key = ("unit", 1, 2, 3, 4)
    mk = lambda n, mt, live=0: torch.tensor([n, mt, 50, live], dtype=torch.int64)  # D4gsProjOut.n_isect: 50 sampled entries
    engine._DEFERRED[key] = [(mk(100, 10, 45), Ev(True), 200, 2048), (mk(900, 10, 50), Ev(True), 200, 2048),
                             (mk(120, 10, 10), Ev(False), 200, 2048)]
    with pytest.raises(RuntimeError, match="needed 900 intersections"):
        engine._deferred_poll(key)           # reads the two that have landed; the second one overflowed
    assert len(engine._DEFERRED[key]) == 1   # the pending one is still queued
    assert engine._guess_get(key)[0] >= 900  # the guess follows the largest count seen
    assert engine._LIVE_FRAC[key] == 45 / 50  # live fraction of the render that fitted (the overflowed one composited nothing)
    assert engine._deferred_poll(key) is None and len(engine._DEFERRED[key]) == 1  # not landed yet, non-blocking
    engine.check_deferred()                  # blocking drain: fits, no error
    assert key not in engine._DEFERRED
    assert engine._LIVE_FRAC[key] == 10 / 50
    engine._SIZE_GUESS.pop(key, None)
    engine._LIVE_FRAC.pop(key, None)


def test_row_mode_follows_the_measured_live_fraction(monkeypatch):
    """auto: the library's capacity heuristic until a render of that shape has been measured, then sparse below 50 % live."""
    from deblur4dgs_amd import _lib as L, engine

    dev = torch.device("cuda", 0)
    cfg = engine.RenderCfg(N=1000, G=0, K=0, T=0, S=2, D=3, width=64, height=48)
    key = engine._size_key(dev, 2, 1000, 64, 48)
    engine._LIVE_FRAC.pop(key, None)
    assert engine.row_mode_for(cfg, dev) == L.ROWS_AUTO
    engine._live_put(key, 150, 1000)
    assert engine.row_mode_for(cfg, dev) == L.ROWS_SPARSE
    engine._live_put(key, 930, 1000)
    assert engine.row_mode_for(cfg, dev) == L.ROWS_DENSE
    monkeypatch.setattr(engine, "BWD_ROWS", "sparse")
    assert engine.row_mode_for(cfg, dev) == L.ROWS_SPARSE
    engine._LIVE_FRAC.pop(key, None)


def test_exact_tiles_auto_keeps_one_size_guess_per_flag_and_counts_again_after_on_to_off(monkeypatch):
    """with D4GS_EXACT_TILES=auto the lists binned under the flag are 15-40 % shorter; the capacity measured with the
    flag on must never size a render that runs with it off.  The guesses are keyed on the resolved flag, an on -> off flip
    leaves NO guess for the off key (the next render counts first, also under deferred_size_check), and the live-fraction
    condition has hysteresis."""
    dev = torch.device("cuda", 0)
    monkeypatch.setattr(engine, "EXACT_TILES", "auto")
    monkeypatch.setattr(engine, "seg_state_elems", lambda cfg: 0)
    S, N, W, H = 2, 1000, 640, 480
    base = engine._size_key(dev, S, N, W, H)
    mk = lambda: engine.RenderCfg(N=N, G=0, K=0, T=0, S=S, D=3, width=W, height=H)
    cap = lambda per_inst: int(per_inst * S * N * 1.25) + 4096
    for k in (base + (False,), base + (True,)):
        engine._guess_drop(k)
    engine._XT_ON.pop(base, None)
    engine._LIVE_FRAC.pop(base, None)
    try:
        cfg = engine.resolve_lazy(mk(), dev)
        assert cfg.exact_tiles is False and engine._list_key(dev, cfg) == base + (False,)  # nothing measured yet
        engine._guess_put(base + (False,), (cap(4.0), 2048))  # rectangles: 4 intersections per instance -> on
        cfg = engine.resolve_lazy(mk(), dev)
        assert cfg.exact_tiles is True and engine._list_key(dev, cfg) == base + (True,)
        assert engine._guess_get(engine._list_key(dev, cfg)) is None  # the render with the flag on does not inherit the off capacity
        engine._guess_put(base + (True,), (cap(2.8), 512))  # what that render measured: shorter lists, a smaller sort class
        engine._live_put(base, 45, 100)                      # 45 % live: below the 0.5 that turns it on, above the 0.4 that turns it off
        assert engine.resolve_lazy(mk(), dev).exact_tiles is True
        engine._live_put(base, 30, 100)                      # now it goes off ...
        cfg = engine.resolve_lazy(mk(), dev)
        assert cfg.exact_tiles is False
        assert engine._guess_get(engine._list_key(dev, cfg)) is None  # ... and the stale off-key capacity is gone: count first
        assert engine._guess_get(base + (True,)) == (cap(2.8), 512)    # a late deferred record of the on-render lands under ITS key
        engine._live_put(base, 45, 100)
        engine._guess_put(base + (False,), (cap(4.0), 2048))
        assert engine.resolve_lazy(mk(), dev).exact_tiles is False     # 45 % does not turn it back on
    finally:
        for k in (base + (False,), base + (True,)):
            engine._guess_drop(k)
        engine._XT_ON.pop(base, None)
        engine._LIVE_FRAC.pop(base, None)
