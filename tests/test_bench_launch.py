"""bench.py's launch contract: `--gpus N` spawns N ranks itself when no torchrun environment is present, refuses to
oversubscribe devices, and rejects a WORLD_SIZE that disagrees with --gpus.  The N = 2 path runs here on CPU: gloo
backend + the host-emulator build of the kernels (tests/bench_emu.py hands bench.main() that platform), tiny shapes."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
BENCH_EMU = os.path.join(ROOT, "tests", "bench_emu.py")


def _run(args, env_extra=None, timeout=900, script=BENCH):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, script] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="needs a box with < 2 devices")
def test_bench_gpus2_refuses_without_devices():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing to run 2 ranks" in (r.stderr + r.stdout)


def test_bench_world_size_mismatch():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_gpus2_spawns_two_gloo_ranks_emulated():
    """`python bench.py --gpus 2` with no torchrun environment: self-spawn through torch.distributed.run, rank binding,
    broadcast, overlapped 2-range gradient all-reduce, barrier / max-over-ranks timing, one JSON line from rank 0."""
    from tests.emu_util import emu_lib
    emu_lib()                                       # build the emulator library once, before the ranks race for it
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--height", "16", "--width", "64", "--maxdisp", "32",
              "--no-cpu-baseline"], {"OMP_NUM_THREADS": "2"}, script=BENCH_EMU)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["scaling"] == "weak" and out["value"] > 0 and out["steps"] == 1
    assert "overlapped" in out["config"]["grad_sync"]
    assert len(out["per_rank"]["ms_per_step"]) == 2 and max(out["per_rank"]["ms_per_step"]) <= out["ms_per_step"] * 1.001


def test_bench_acv_train_two_ranks_emulated():
    """BASELINE.json configs[3]'s exact command form (`bench.py --config acv_train --gpus N`, global batch 16 over 8 GPUs = 2 per
    rank) end to end on two gloo ranks before hardware sees it: ACVNet train step with a batch of two per
    rank, batch sharding as trainer_torchrun.py:130-136 (DistributedSampler) does, one flat averaged gradient exchange
    (:116-121), per-rank times and the exposed all-reduce time in the line."""
    from tests.emu_util import emu_lib
    emu_lib()
    r = _run(["--config", "acv_train", "--gpus", "2", "--batch", "2", "--steps", "1", "--warmup", "0", "--height", "16", "--width", "64",
              "--maxdisp", "64", "--no-cpu-baseline"], {"OMP_NUM_THREADS": "2"}, script=BENCH_EMU)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["config"]["name"] == "acv_train" and out["n_gpus"] == 2 and out["config"]["global_batch"] == 4
    assert out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["per_rank"]["ms_per_step"]) == 2 and "grad_sync_ms_on_stream" in out["per_rank"]


def test_bench_psm_volume_config_emulated():
    r = _run(["--config", "psm_volume", "--steps", "1", "--warmup", "0", "--height", "16", "--width", "64", "--maxdisp", "32",
              "--no-cpu-baseline"], script=BENCH_EMU)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["config"]["name"] == "psm_volume" and out["unit"] == "volumes/s" and out["n_gpus"] == 1


@pytest.mark.gpu
def test_bench_gpus2_rccl():
    """Two ranks over RCCL on a node with >= 2 GPUs (skipped on the 1-GPU test box, where the refusal test runs instead)."""
    if torch.cuda.device_count() < 2:
        r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
        assert r.returncode != 0 and "refusing to run 2 ranks" in (r.stderr + r.stdout)
        return
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "64", "--width", "128", "--maxdisp", "64",
              "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["value"] > 0


def test_committed_pmc_traffic_is_stamped_with_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` constants: bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB of the FIRST kernel
    of each counter pass in a committed summary; a summary whose sha256 stamp does not match this tree's kernel sources (or has
    none) yields None instead of a stale number; the committed r06 summaries match the tree."""
    import bench
    sha = bench.kernel_source_sha("conv3d.hip")
    assert len(sha) == 64 and sha != bench.kernel_source_sha("cost_volume_mfma.hip")
    prof = tmp_path / "profiles"
    prof.mkdir()
    body = ("kernel A\n   FETCH_SIZE                         n=   5 avg=         1000.0\n"
            "kernel B\n   FETCH_SIZE                         n=   5 avg=            7.0\n"
            "kernel A\n   WRITE_SIZE                         n=   5 avg=         3000.0\n"
            "kernel B\n   WRITE_SIZE                         n=   5 avg=            9.0\n")
    (prof / "stamped.txt").write_text(body + f"kernel_source_sha {sha}\n")
    (prof / "stale.txt").write_text(body + "kernel_source_sha " + "0" * 64 + "\n")
    (prof / "unstamped.txt").write_text(body)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    # (kernel_source_sha reads the sources under ROOT: give it the real ones)
    monkeypatch.setattr(bench, "kernel_source_sha", lambda *rel: sha)
    assert bench.committed_pmc_traffic("stamped.txt", ("conv3d.hip",)) == int((2 * 1000.0 + 3000.0) * 1024)
    assert bench.committed_pmc_traffic("stale.txt", ("conv3d.hip",)) is None
    assert bench.committed_pmc_traffic("unstamped.txt", ("conv3d.hip",)) is None
    assert bench.committed_pmc_traffic("unstamped.txt") == int(5000.0 * 1024)       # (round-5 files: no stamp asked)
    assert bench.committed_pmc_traffic("missing.txt", ("conv3d.hip",)) is None
    monkeypatch.undo()
    # the summaries committed this round carry the stamps of THIS tree's sources
    assert bench.committed_pmc_traffic("r06_pmc_conv3d_marchw.txt", ("conv3d.hip",)) is not None
    assert bench.committed_pmc_traffic("r06_pmc_cost_volume_fwd.txt", ("cost_volume_mfma.hip",)) is not None
