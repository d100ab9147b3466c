import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the fp64 torch oracle is thousands of small CPU ops: on the GPU box's 256-thread host the default (one thread per core) spends
    # its time in thread hand-offs (bench.py measured one cfg1 oracle frame at 205-831 s on 256 threads against ~3 s on 16)
    import torch

    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Write the achieved-error table of the GPU parity tests (tests/util.record)."""
    from tests.util import PARITY_LOG

    if not PARITY_LOG:
        return
    import json

    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_table.json"), "w") as f:
        json.dump(PARITY_LOG, f, indent=1)
    with open(os.path.join(out, "parity_table.md"), "w") as f:
        f.write("| case | tensor | elements | max err / max ref | frac > 1e-4 | frac > 1e-3 | asserted (tol, flips) |\n"
                "|---|---|---:|---:|---:|---:|---|\n")
        for r in PARITY_LOG:
            a = f"{r['asserted_tol']:g}, {r['asserted_flips']:g}" if "asserted_tol" in r else "-"
            f.write(f"| {r['case']} | {r['tensor']} | {r['n']} | {r['rel_err']:.2e} | {r['frac_bad_1e4']:.2e} | "
                    f"{r['frac_bad_1e3']:.2e} | {a} |\n")
