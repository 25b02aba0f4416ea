"""GPU suite: multi-GPU where the C host lives (VERDICT r05 missing 2 / next 6).  tests/c/multi_gpu.c is a
plain-C program (no HIP runtime on its link line: vamd_device_count() asks the library) that spreads a vamd_feed and a
vamd_batcher over every GPU the runtime shows and checks that every lane / thread returns the same packets.  On a one-GPU
box the device list names the one GPU twice -- the same code path (lanes dealt round a device list, a context, stream and
arenas per lane on ITS device), short of a second physical device: no box this project has run on holds two, and the
figure for N GPUs stays unmeasured."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_host_spreads_feed_and_batcher_over_all_devices(tmp_path):
    exe = tmp_path / "multi_gpu"
    libdir = os.path.join(ROOT, "vorbis_amd")
    subprocess.run(["gcc", "-O2", "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "multi_gpu.c"), "-o", str(exe),
                    "-L" + libdir, "-lvorbis_amd", "-Wl,-rpath," + libdir, "-lpthread"], check=True)
    r = subprocess.run([str(exe), os.path.join(ROOT, "vorbis_amd", "data", "setup_44k_stereo_q4.bin")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "multi_gpu OK" in r.stdout, r.stdout + r.stderr
    assert "identical on every lane: yes" in r.stdout and "to vamd_encode_block: yes" in r.stdout
