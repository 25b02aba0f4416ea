"""The runner the CPU rehearsals of bench.py's rank logic put in the GPU runner's place (tests/test_abi_and_host.py):
it checks the broadcast blob, takes its shard, sleeps instead of analysing and reports fixed stage times.  bench.py
has no dry-run path of its own; this lives with the tests and is named explicitly (--runner / make_runner)."""
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubRunner:
    def __init__(self, a, blob, dev, rank, world):
        import bench
        from vorbis_amd import sharding
        self.bench = bench
        ref = np.fromfile(os.path.join(ROOT, "vorbis_amd", "data", "setup_%s.bin" % a.setup), dtype=np.uint8)
        assert np.array_equal(np.asarray(blob), ref), "rank %d received a different setup blob" % rank
        nb = a.blocks or 131072
        self.lo, self.hi = sharding.shard_range(nb * world, rank, world)
        self.units, self.unit_name, self.rank, self.steps_run = self.hi - self.lo, "stereo blocks/s", rank, 0

    def step(self):
        time.sleep(0.01 * (1 + self.rank))   # rank 1 is the slow one: the reported time must be ITS time
        self.steps_run += 1

    def sync(self):
        pass

    def timed_begin(self):
        self.t0 = self.steps_run

    def timed_end(self):
        self.timed = self.steps_run - self.t0

    def stage_ms(self, steps):
        assert self.timed == steps
        return {"transform": 1.0, "noisemask": 2.0}

    def parity_sample(self, count):
        return count, 0, "stub"

    def stage_bytes_total(self, stage):
        return self.bench.stage_bytes(stage, 2048) * self.units

    def workload_text(self):
        return "rank-logic rehearsal (no GPU work)"
