#!/usr/bin/env python
"""Test-side launcher for bench.py's N > 1 path on a box without GPUs: the same main(), handed a platform whose ranks
are CPU processes over gloo and whose kernels are the host-emulator build (tests/hipemu).  bench.py itself and the
package know nothing about this file."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


class EmulatedPlatform(bench.Platform):
    dist_backend = "gloo"
    script = os.path.abspath(__file__)
    event_timing = False

    def device_count(self):
        return 1 << 10                       # CPU processes: no device to oversubscribe

    def bind(self, local):
        from tests.emu_util import emu_product_path
        emu_product_path().__enter__()
        return torch.device("cpu")

    def init_process_group(self, dev):
        dist.init_process_group("gloo")

    def sync(self):
        pass

    def tuning_db(self):
        return None


if __name__ == "__main__":
    bench.main(platform=EmulatedPlatform())
