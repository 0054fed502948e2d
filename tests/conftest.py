import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a ROCm device skips every gpu-marked test instead of failing in it
    (the marker alone only selects / deselects with -m)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a ROCm device (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def emu():
    """Host SIMT-emulator build of the kernel sources (tests only, see tests/hipemu)."""
    from tests.emu_util import emu_lib
    return emu_lib()


# ---------------------------------------------------------------------------------------------------------------
# Parity report: tests that measure an error against the oracle / the reference fixtures record it here; the numbers
# are printed in the terminal summary (also under -q, where captured stdout of passing tests is not shown) and
# appended to gpurun_out/parity_report.jsonl when that directory exists (GPU box runs; xdist-safe).
_PARITY = []


@pytest.fixture
def parity_log():
    def add(name, **fields):
        rec = {"test": name, **fields}
        _PARITY.append(rec)
        # written at once (one O_APPEND line): under pytest-xdist the workers' records never reach the controller's summary
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out):
            try:
                import json
                with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
                    f.write(json.dumps(rec) + "\n")
            except OSError:
                pass
    return add


def pytest_terminal_summary(terminalreporter):
    if not _PARITY:
        return
    import json
    terminalreporter.section("parity report (achieved errors)")
    for rec in _PARITY:
        terminalreporter.write_line(json.dumps(rec))
