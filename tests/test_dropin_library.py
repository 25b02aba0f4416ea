"""The drop-in as a deliverable (integration/Makefile): build/dropin/libvorbis.so.0.4.9 +
libvorbisenc.so.2.0.12 with the reference's SONAMEs and export list, no test harness inside, and an application
(integration/encode_loop.c: the call sequence of examples/encoder_example.c:140-236) that produces byte-identical
packets on it and on the unmodified reference built beside it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "dropin")
LIB = os.path.join(OUT, "libvorbis.so.0.4.9")
ENC = os.path.join(OUT, "libvorbisenc.so.2.0.12")
needs_build = pytest.mark.skipif(not os.path.exists(LIB), reason="build/dropin not built (integration/Makefile needs the reference's sources)")


def _names(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3}


@needs_build
def test_sonames_and_exports():
    for path, soname in ((LIB, "libvorbis.so.0"), (ENC, "libvorbisenc.so.2"),
                         (os.path.join(OUT, "ref", "libvorbis.so.0.4.9"), "libvorbis.so.0")):
        dyn = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, check=True).stdout
        assert "soname: [%s]" % soname in dyn, dyn
    want = [ln.strip() for ln in open(os.path.join(ROOT, "integration", "exports.txt")) if ln.strip() and not ln.startswith("#")]
    assert len(want) == 45
    have = _names(LIB) | _names(ENC)
    assert not [n for n in want if n not in have]
    # the encoder-setup six live in libvorbisenc, the rest in libvorbis (the reference's ELF packaging, lib/Makefile.am)
    assert {n for n in want if n.startswith("vorbis_encode_")} <= _names(ENC)
    assert {n for n in want if not n.startswith("vorbis_encode_")} <= _names(LIB)
    # nothing of the test harness, and the GPU library is a dependency, not a copy
    assert not [n for n in have if n.startswith("ref_")]
    dyn = subprocess.run(["readelf", "-d", LIB], capture_output=True, text=True, check=True).stdout
    assert "libvorbis_amd.so" in dyn
    # the drop-in exports what the unmodified reference exports (plus the binding's own vamd_* helpers)
    ref_names = _names(os.path.join(OUT, "ref", "libvorbis.so.0.4.9"))
    extra = {n for n in _names(LIB) - ref_names if not n.startswith("vamd_")}
    assert ref_names <= _names(LIB) | {"_ve_envelope_search_cpu"} and not {n for n in extra if not n.endswith("_cpu")}, extra


def _loop(libdirs, *args, detector=None):
    env = dict(os.environ)
    env.pop("VAMD_DETECTOR", None)        # the binding's default (auto), not the suite's (tests/conftest.py forces the GPU's)
    if detector:
        env["VAMD_DETECTOR"] = detector
    env["LD_LIBRARY_PATH"] = os.pathsep.join(libdirs + [env.get("LD_LIBRARY_PATH", "")])
    return subprocess.run([os.path.join(OUT, "encode_loop")] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=600)


@needs_build
def test_application_loop_on_the_reference_build(tmp_path):
    """The driver itself, on the unmodified reference: BASELINE config 1 gives 52 blocks; a second run is identical."""
    ref = [os.path.join(OUT, "ref"), OUT]
    f = tmp_path / "c1.pkts"
    r = _loop(ref, 1024, 0.4, 44100, "write", f)
    assert r.returncode == 0 and "52 blocks" in r.stderr, r.stderr
    r = _loop(ref, 1024, 0.4, 44100, "check", f)
    assert r.returncode == 0 and "identical" in r.stderr, r.stderr


@needs_build
@pytest.mark.gpu
@pytest.mark.parametrize("detector", [None, "gpu", "host"])
@pytest.mark.parametrize("read,quality,frames", [(1024, 0.4, 44100), (65536, 0.4, 44100), (4096, 0.9, 100000), (1024, 0.1, 30000)])
def test_application_on_the_drop_in_matches_the_reference(tmp_path, read, quality, frames, detector):
    """C1 (1 s stereo white noise, 16-bit, q 0.4) at the example's READ 1024 and at 65 536-frame writes (the look-ahead's
    case), and two more: the application's packets -- headers and audio -- are the reference's, byte for byte, whichever
    detector serves the stream (VAMD_DETECTOR: the binding's own choice by the size of the first writes, the GPU's, libvorbis')."""
    f = tmp_path / "ref.pkts"
    r = _loop([os.path.join(OUT, "ref"), OUT], read, quality, frames, "write", f)
    assert r.returncode == 0, r.stderr
    r = _loop([OUT], read, quality, frames, "check", f, detector=detector)
    assert r.returncode == 0 and "identical" in r.stderr, r.stdout + r.stderr
