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
        _order_gpu_run(items)
        return
    skip = pytest.mark.skip(reason="needs a ROCm device (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_FULL_SIZE = ("test_psmnet_config1_eval", "test_cost_volume_full_size_properties", "test_full_size_eval_parity",
              "test_gwcnet_gc_full_size_train_step_parity", "test_acvnet_full_size_train_step_parity")
_KERNEL_FILES = ("test_capi_symbols.py", "test_kernels.py", "test_conv2d.py", "test_f1ops.py", "test_hygiene.py",
                 "test_igev_preprocess.py", "test_metrics.py")
_LAST_FILES = ("test_torch_ext.py",)       # the torch.utils.cpp_extension door (8 of the entry points): after everything else, so a
                                           # build-system problem there can never again keep a model test from running
_TOY_TRAIN = ("train_grads", "frozen_attention_train", "_train_")          # incl. the isolated (deterministic) hand-written-path tests
_TOY_TRAIN_WHOLE = ("train_parity", "train_step")                          # whole models incl. the stock 2-D CNN: the very last


def _order_gpu_run(items):
    """Order of a run on a GPU box (the driver's round-end `pytest -x -m gpu`): kernel tests -> the five BASELINE.json
    full-size configurations -> everything else -> the small-shape train-step tests -> the torch.utils.cpp_extension door.
    Under `-x` nothing may keep the headline shapes from running (round 4: an ill-conditioned 64x128 train test; round 5: a
    build baton).  The small-shape train tests now run at 128x256 against reference fixtures (tests/test_models.py)."""
    def phase(item):
        path, name = item.nodeid.split("::")[0], item.name
        if any(path.endswith(f) for f in _LAST_FILES):
            return 5
        if any(name.startswith(n) for n in _FULL_SIZE):
            return 1
        if any(path.endswith(f) for f in _KERNEL_FILES):
            return 0
        if any(t in name for t in _TOY_TRAIN_WHOLE):
            return 4
        if any(t in name for t in _TOY_TRAIN):
            return 3
        return 2
    items.sort(key=phase)              # stable: the collection order is kept inside a phase


# ---------------------------------------------------------------------------------------------------------------
# Per-test wall-clock guard: one stuck test costs minutes, not the run.  Two stages, no plugin needed:
#   * SIGALRM after STX_TEST_TIMEOUT s (default 240) raises in the test -- covers anything that spins in Python
#     (torch.utils.file_baton.FileBaton.wait, a rendezvous that never completes): the test FAILS with a traceback;
#   * faulthandler.dump_traceback_later(+60 s, exit=True) ends the process with every thread's stack on stderr when the
#     interpreter never gets to run the signal handler (a kernel that does not return inside hipStreamSynchronize).
TEST_TIMEOUT_S = int(os.environ.get("STX_TEST_TIMEOUT", "240"))


class TestTimeout(Exception):
    pass


@pytest.fixture(autouse=True)
def _wall_clock_guard(request):
    import faulthandler
    import signal
    import threading
    if TEST_TIMEOUT_S <= 0 or threading.current_thread() is not threading.main_thread() or not hasattr(signal, "SIGALRM"):
        yield
        return

    def on_alarm(signum, frame):
        raise TestTimeout(f"{request.node.nodeid} exceeded {TEST_TIMEOUT_S} s (STX_TEST_TIMEOUT)")
    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(TEST_TIMEOUT_S)
    faulthandler.dump_traceback_later(TEST_TIMEOUT_S + 60, exit=True)
    try:
        yield
    finally:
        signal.alarm(0)
        faulthandler.cancel_dump_traceback_later()
        signal.signal(signal.SIGALRM, old)


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
