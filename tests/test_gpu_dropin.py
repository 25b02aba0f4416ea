"""GPU suite: the drop-in boundary end to end.

oracle/_ref/libvorbis_hybrid.so is the reference's own libvorbis objects with lib/mapping0.c replaced
by the binding integration/mapping0_vamd.c (all of mapping0_forward -> one call into libvorbis_amd.so
that returns the block's packet bytes; for modes whose residue the GPU does not search, the numeric
section only and the reference's bit-writing half unchanged) and lib/envelope.c replaced by
integration/envelope_vamd.c (the block-switching detector's steps -> libvorbis_amd.so, mark/cursor
bookkeeping unchanged), so on the gated streams below the GPU also decides where the short blocks go.
Driving the unmodified application loop (vorbis_analysis_buffer / _wrote / _blockout / vorbis_analysis,
examples/encoder_example.c:179-236) through it must yield the very packets the pure CPU reference
emits -- the strongest parity statement the domain offers: every bit of every packet was produced on
the GPU and is identical to what the reference's own analysis, Huffman and VQ code writes.
"""
import numpy as np
import pytest

from oracle import ref

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (ref.available() and ref.hybrid_available()),
                                 reason="oracle/_ref libraries not built (needs /root/reference at build time)")]


def _stream(ch, seconds, kind, seed):
    rng = np.random.default_rng(seed)
    frames = int(44100 * seconds)
    x = (rng.random((ch, frames), dtype=np.float32) - 0.5)
    if kind == "gated":   # transients -> short blocks, transitions (SURVEY.md 8d "C5")
        t = np.arange(frames)
        x *= 2 * np.where((t % 11025) < 1102, 0.5, 0.0005).astype(np.float32)
    elif kind == "s16":   # encoder_example.c:197-202: s16 samples / 32768
        x = np.round(x * 32767).astype(np.int16).astype(np.float32) / 32768.0
    return np.ascontiguousarray(x, dtype=np.float32)


@pytest.mark.parametrize("ch,q,kind", [(2, 0.4, "s16"), (2, 0.9, "gated"), (2, 0.1, "gated"), (1, 0.5, "gated")])
def test_hybrid_encode_emits_reference_packets(ch, q, kind):
    assert hasattr(ref.lib(hybrid=True), "_ve_envelope_search_cpu")  # the detector binding is linked in
    assert hasattr(ref.lib(hybrid=True), "vamd_encode_block")        # ... and packets come from the GPU library
    pcm = _stream(ch, 1.0 if kind == "s16" else 2.0, kind, seed=12345)
    want = ref.RefEncoder(ch, 44100, q).encode_stream(pcm)
    got = ref.RefEncoder(ch, 44100, q, hybrid=True).encode_stream(pcm)
    assert len(want) == len(got) > 40
    assert [(b["lW"], b["W"], b["nW"], b["blocktype"]) for b in want] == \
           [(b["lW"], b["W"], b["nW"], b["blocktype"]) for b in got]
    for k, (a, b) in enumerate(zip(want, got)):
        assert np.float32(a["ampmax_in"]) == np.float32(b["ampmax_in"]), k     # the ampmax chain stays in step
        assert np.float32(a["ampmax_out"]) == np.float32(b["ampmax_out"]), k
        assert a["packet"] == b["packet"], "packet %d differs (W=%d)" % (k, a["W"])
    if kind == "gated":
        assert sum(1 for b in want if b["W"] == 0) > 10


@pytest.mark.parametrize("ch,q,kind,write", [(2, 0.4, "s16", 65536), (2, 0.9, "gated", 65536), (2, 0.4, "gated", 8192),
                                             (2, 0.1, "gated", 30000), (1, 0.5, "gated", 16384), (6, 0.3, "gated", 50000)])
def test_look_ahead_inside_one_stream_emits_reference_packets(ch, q, kind, write):
    """An application that writes more than a block's worth per vorbis_analysis_wrote() (the API takes any amount,
    lib/block.c:390,470): the binding plans the blocks its buffer already determines, runs them as ONE batch
    (vamd_encode_blocks, the ampmax chain on the device) and serves the following vorbis_analysis() calls from those
    packets -- each only after the block the reference's own blockout produced has been held against the planned one,
    sample for sample.  Every packet and every ampmax equals the pure CPU reference fed the same way; and the cache
    really was what served them (hits >> batches)."""
    import ctypes as C
    L = ref.lib(hybrid=True)
    h0 = [C.c_long(0) for _ in range(3)]
    L.vamd_ahead_stats(*[C.byref(v) for v in h0])
    pcm = _stream(ch, 6.0, kind, seed=4242)
    want = ref.RefEncoder(ch, 44100, q).encode_stream(pcm, write_frames=write)
    got = ref.RefEncoder(ch, 44100, q, hybrid=True).encode_stream(pcm, write_frames=write)
    h1 = [C.c_long(0) for _ in range(3)]
    L.vamd_ahead_stats(*[C.byref(v) for v in h1])
    hits, misses, batches = (b.value - a.value for a, b in zip(h0, h1))
    assert len(want) == len(got) > 200
    for k, (a, b) in enumerate(zip(want, got)):
        assert (a["lW"], a["W"], a["nW"], a["blocktype"]) == (b["lW"], b["W"], b["nW"], b["blocktype"]), k
        assert np.float32(a["ampmax_in"]) == np.float32(b["ampmax_in"]), k
        assert np.float32(a["ampmax_out"]) == np.float32(b["ampmax_out"]), k
        assert a["packet"] == b["packet"], "packet %d differs (W=%d)" % (k, a["W"])
    if kind == "gated":
        assert sum(1 for b in want if b["W"] == 0) > 30
    assert hits > 0.8 * len(got) and batches < len(got) / 4 and misses <= batches, (hits, misses, batches, len(got))


@pytest.mark.parametrize("q,kind,write,drain", [(0.9, "gated", 5000, 2), (0.4, "gated", 3000, 1), (0.4, "s16", 20000, 7)])
def test_look_ahead_survives_writes_between_the_blocks(q, kind, write, drain):
    """An application may write more samples before it has pulled every block the buffer holds -- here at most `drain`
    blocks are pulled after each write, so blocks pile up.  The binding's plan was made on less data than the reference's
    blockout later decides on; it only ever covers blocks the data at hand already determines (it stops where the blockout
    would wait), and whatever it planned is verified against the block that really comes, so the packets are the
    reference's all the same."""
    pcm = _stream(2, 8.0, kind, seed=99)
    want = ref.RefEncoder(2, 44100, q).encode_stream(pcm, write_frames=write, drain=drain)
    got = ref.RefEncoder(2, 44100, q, hybrid=True).encode_stream(pcm, write_frames=write, drain=drain)
    assert len(want) == len(got) > 300
    for k, (a, b) in enumerate(zip(want, got)):
        assert (a["lW"], a["W"], a["nW"], a["blocktype"]) == (b["lW"], b["W"], b["nW"], b["blocktype"]), k
        assert np.float32(a["ampmax_out"]) == np.float32(b["ampmax_out"]), k
        assert a["packet"] == b["packet"], "packet %d differs (W=%d)" % (k, a["W"])


@pytest.mark.parametrize("ch,q,managed,kind,write,drain,seed", [
    (2, 0.4, None, "s16", 40000, 6, 11), (2, 0.9, None, "gated", 9000, 3, 12), (1, 0.5, None, "gated", 70000, 40, 13),
    (6, 0.3, None, "gated", 30000, 5, 14), (2, None, (-1, 128000, -1), "gated", 25000, 4, 15)])
def test_look_ahead_under_random_writes_and_pulls(ch, q, managed, kind, write, drain, seed):
    """Every write a pseudo-random 1 .. `write` samples, every pull a pseudo-random 0 .. `drain` blocks (the harness's
    jitter: the same sequence for the reference and the hybrid): hits, stale plans, single blocks and batches in whatever
    order they fall -- and the packets are the reference's."""
    pcm = _stream(ch, 6.0, kind, seed=seed)
    kw = dict(managed=managed) if managed else {}
    args = (ch, 44100) if managed else (ch, 44100, q)
    want = ref.RefEncoder(*args, **kw).encode_stream(pcm, write_frames=write, drain=drain, jitter=seed)
    got = ref.RefEncoder(*args, hybrid=True, **kw).encode_stream(pcm, write_frames=write, drain=drain, jitter=seed)
    assert len(want) == len(got) > 200
    for k, (a, b) in enumerate(zip(want, got)):
        assert (a["lW"], a["W"], a["nW"], a["blocktype"]) == (b["lW"], b["W"], b["nW"], b["blocktype"]), k
        assert np.float32(a["ampmax_out"]) == np.float32(b["ampmax_out"]), k
        assert a["packet"] == b["packet"], "packet %d differs (W=%d)" % (k, a["W"])


def _ahead_stats():
    import ctypes as C
    h, m, b = C.c_long(), C.c_long(), C.c_long()
    ref.lib(hybrid=True).vamd_ahead_stats(C.byref(h), C.byref(m), C.byref(b))
    return h.value, m.value, b.value


@pytest.mark.parametrize("ch,q,managed,kind", [(2, 0.4, None, "gated"), (1, 0.9, None, "gated"), (2, None, (-1, 96000, -1), "s16")])
def test_look_ahead_runs_to_the_end_of_a_stream(ch, q, managed, kind):
    """An application that writes a whole clip, signals the end (vorbis_analysis_wrote(v, 0)) and only then pulls its
    blocks: everything is determined at the first pull, the end of the stream included (lib/block.c:561-568: no mark
    found and no more to come means a short next block; :664-670: the block centred past the last real sample is the
    last), so all of it is looked ahead -- and equals the reference driven the same way, to the last packet."""
    pcm = _stream(ch, 5.0, kind, seed=5)
    kw = dict(managed=managed) if managed else {}
    args = (ch, 44100) if managed else (ch, 44100, q)
    want = ref.RefEncoder(*args, **kw).encode_stream(pcm, write_frames=pcm.shape[1], drain=1)
    h0, m0, b0 = _ahead_stats()
    got = ref.RefEncoder(*args, hybrid=True, **kw).encode_stream(pcm, write_frames=pcm.shape[1], drain=1)
    h1, m1, b1 = _ahead_stats()
    assert len(want) == len(got) > 150
    for k, (a, b) in enumerate(zip(want, got)):
        assert (a["lW"], a["W"], a["nW"], a["blocktype"]) == (b["lW"], b["W"], b["nW"], b["blocktype"]), k
        assert np.float32(a["ampmax_out"]) == np.float32(b["ampmax_out"]), k
        assert a["packet"] == b["packet"], "packet %d differs (W=%d)" % (k, a["W"])
    per_batch = 63 if managed else 255
    assert m1 == m0 and h1 - h0 >= len(got) - 2 - (len(got) // per_batch + 1)   # all but the batches' own first blocks


_BATCH_WORKER = r'''
import sys, threading, time, json
sys.path.insert(0, @ROOT@)
import numpy as np
from oracle import ref
from tests.test_gpu_dropin import _stream
N, q = int(sys.argv[1]), float(sys.argv[2])
streams = [_stream(2, 1.5, "gated" if k % 2 else "s16", seed=100 + k) for k in range(N)]
want = [ref.RefEncoder(2, 44100, q).encode_stream(x) for x in streams]
got = [None] * N
encs = [ref.RefEncoder(2, 44100, q, hybrid=True) for _ in range(N)]
def work(k):
    got[k] = encs[k].encode_stream(streams[k])     # one long C call: the GIL is released, the threads run together
th = [threading.Thread(target=work, args=(k,)) for k in range(N)]
t0 = time.time()
[t.start() for t in th]
[t.join() for t in th]
wall = time.time() - t0
bad = 0
for k in range(N):
    if len(want[k]) != len(got[k]):
        bad += 1
        continue
    for a, b in zip(want[k], got[k]):
        if a["packet"] != b["packet"] or np.float32(a["ampmax_out"]) != np.float32(b["ampmax_out"]):
            bad += 1
import ctypes as C
L = ref.lib(hybrid=True)
print(json.dumps({"streams": N, "blocks": sum(len(w) for w in want), "bad": bad, "wall": wall,
                  "short_blocks": sum(1 for w in want for b in w if b["W"] == 0)}))
'''


def test_batch_mode_many_encoder_threads_emit_reference_packets():
    """VAMD_BATCH: sixteen application threads, each driving its own vorbis_dsp_state through the unmodified
    application loop of the hybrid libvorbis, have their blocks coalesced by the vamd_batcher (one context, batched
    launches); every packet of every stream equals the pure CPU reference's."""
    import json, os, subprocess, sys
    from tests import checker
    env = dict(os.environ, VAMD_BATCH="64", VAMD_BATCH_WAIT_US="3000")
    out = subprocess.run([sys.executable, "-c", _BATCH_WORKER.replace("@ROOT@", repr(checker.ROOT)), "16", "0.4"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["streams"] == 16 and r["blocks"] > 16 * 60 and r["short_blocks"] > 50
    assert r["bad"] == 0, r


@pytest.mark.parametrize("threads,max_batch,lanes", [(96, 256, 8), (48, 8, 2)])
def test_batch_mode_under_load(threads, max_batch, lanes):
    """The round-4 batcher (a library thread per lane, lock-free submission, a futex word per request) with more streams than a
    lane takes at once: 96 threads on eight lanes (the second kind of lane joins), and 48 threads against batches capped at
    eight blocks on two lanes (a lane takes a surplus and pushes it back).  Mixed block sizes; every packet of every stream and
    every ampmax equal the CPU reference's."""
    import json, os, subprocess, sys
    from tests import checker
    env = dict(os.environ, VAMD_BATCH=str(max_batch), VAMD_BATCH_LANES=str(lanes))
    out = subprocess.run([sys.executable, "-c", _BATCH_WORKER.replace("@ROOT@", repr(checker.ROOT)), str(threads), "0.4"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["streams"] == threads and r["blocks"] > threads * 60 and r["short_blocks"] > 50
    assert r["bad"] == 0, r
