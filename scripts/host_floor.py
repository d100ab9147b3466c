"""Host-side floor of the example training step: tiny GPU work, cProfile by cumulative time."""
import cProfile, pstats, os, sys, io, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("ex", os.path.join(os.path.dirname(__file__), "..", "examples", "train_dynamic_step.py"))
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
kw = dict(W=64, H=48, n_fg=1500, n_bg=1500, K=20, verbose=False)
ex.train(steps=5, **kw)
_, _, dt = ex.train(steps=30, **kw)
print(f"host floor: {1e3*dt:.2f} ms / step")
pr = cProfile.Profile(); pr.enable(); ex.train(steps=30, **kw); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
