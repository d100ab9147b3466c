"""Where does the example training step spend its time?  (host vs device; per phase)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("ex", os.path.join(os.path.dirname(__file__), "..", "examples", "train_dynamic_step.py"))
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
from torch.profiler import profile, ProfilerActivity
ex.train(steps=5, verbose=False)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    ex.train(steps=5, verbose=False)
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
