import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- what the at-size tests really ran: every one appends a record; the session prints them (also under -q) and leaves them in
# gpurun_out/at_size_runs.json, so a green run proves the sizes it covered
AT_SIZE_RUNS = []


def record_at_size(test, **fields):
    rec = dict(test=test, **fields)
    AT_SIZE_RUNS.append(rec)
    try:
        import json
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "at_size_runs.json"), "w") as f:
            json.dump(AT_SIZE_RUNS, f, indent=1)
    except Exception:
        pass
    return rec


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not AT_SIZE_RUNS:
        return
    import json
    terminalreporter.write_line("at-size runs (sizes that actually ran; also in gpurun_out/at_size_runs.json):")
    for r in AT_SIZE_RUNS:
        terminalreporter.write_line("  " + json.dumps(r))
