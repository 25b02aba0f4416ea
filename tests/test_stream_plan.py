"""GPU suite: device-resident stream control (vamd_plan_streams / vamd_gather_blocks, k_blockout.h) against the
reference's own vorbis_analysis_blockout() -- block sizes, window flags, block types and block contents of whole
streams -- and the planned streams analysed through vamd_analyze_streams_mixed against the reference's packets'
source values."""
import numpy as np
import pytest

from tests import checker

pytestmark = pytest.mark.gpu


def gated_noise(rng, ch, frames, period, burst, loud=0.5, quiet=0.0005):
    t = np.arange(frames)
    gate = np.where((t % period) < burst, loud, quiet).astype(np.float32)
    return ((rng.random((ch, frames), dtype=np.float32) - 0.5) * 2 * gate).astype(np.float32)


@pytest.mark.parametrize("setup,q", [("44k_stereo_q9", 0.9), ("44k_stereo_q4", 0.4)])
def test_plan_matches_reference_blockout(setup, q):
    import torch
    import vorbis_amd
    from oracle import ref
    if not ref.available():
        pytest.skip("needs the reference build to cut genuine streams")
    ch, rate = 2, 44100
    rng = np.random.default_rng(77)
    frames = 44100 * 2
    raws = [gated_noise(rng, ch, frames, 11025, 1102), gated_noise(rng, ch, frames, 7000, 300, 0.9, 0.002),
            ((rng.random((ch, frames), dtype=np.float32) - 0.5) * 0.3).astype(np.float32),     # steady noise: long blocks only
            gated_noise(rng, ch, frames, 1500, 200, 0.7, 0.0001)]                               # dense transients
    an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob(setup), 0)
    bs = an.blocksizes
    # the encoder's own PCM buffer (pre-extrapolated start included) = what envelope_feed reports as seen
    bufs, refs = [], []
    for raw in raws:
        # (fed in the application's 1024-sample chunks: the pre-extrapolation of the first half block is an LPC fit
        # over whatever has arrived when it runs, lib/block.c:524-526)
        e = ref.RefEncoder(ch, rate, q)
        for k in range(0, frames, 1024):
            seen = e.envelope_feed(raw[:, k:k + 1024])["pcm"]
        bufs.append(seen)
        refs.append(ref.RefEncoder(ch, rate, q).encode_stream(raw))
    n = min(b.shape[1] for b in bufs) & ~3
    streams = torch.from_numpy(np.stack([b[:, :n] for b in bufs])).cuda()
    plan, _ = an.plan_streams(streams)
    L = an.plan_lists(plan)
    pcm_blocks = [an.gather_blocks(plan, W, streams).cpu().numpy() for W in (0, 1)]
    assert plan.nblocks[0] > 50 and plan.nblocks[1] > 50
    for s, blocks in enumerate(refs):
        lo, hi = int(L["stream_start"][s]), int(L["stream_start"][s + 1])
        got = L["order"][lo:hi]
        assert hi - lo > 20
        # every planned block is the reference's block of the same index (the reference goes on into its
        # end-of-stream tail, which a plan over the unclosed buffer does not reach)
        assert hi - lo <= len(blocks)
        assert len(blocks) - (hi - lo) <= 12, "the plan stops more than a few blocks before the reference's end of stream"
        for k, o in enumerate(got):
            W, i = (int(o) >> 30) & 1, int(o) & 0x3fffffff
            b = blocks[k]
            assert (W, int(L["lW"][W][i]), int(L["nW"][W][i]), int(L["blocktype"][W][i])) == \
                (b["W"], b["lW"], b["nW"], b["blocktype"]), (s, k)
            assert int(L["src"][W][i]) // (ch * n) == s
            assert np.array_equal(pcm_blocks[W][i], b["pcm"]), (s, k)
    # ... and the planned blocks through the analysis, ampmax chain included, against the reference's own taps
    want = ("mdct", "posts", "post_valid", "iwork", "nonzero", "ampmax_out")
    outs = [an.alloc_outputs(W, plan.nblocks[W], want) for W in (0, 1)]
    states = torch.full((len(raws),), -9999.0, device="cuda")
    dev_blocks = [torch.from_numpy(pcm_blocks[W]).cuda() for W in (0, 1)]
    an.analyze_plan(plan, dev_blocks, outs, states)
    torch.cuda.synchronize()
    host = [{k: v.cpu().numpy() for k, v in outs[W].items()} for W in (0, 1)]
    chk = checker.Checker(setup)
    for s, blocks in enumerate(refs):
        lo, hi = int(L["stream_start"][s]), int(L["stream_start"][s + 1])
        for k, o in enumerate(L["order"][lo:hi]):
            if k % 3 and k < hi - lo - 1:
                continue   # every third block and the last one through the float taps; the chain is checked on all
            W, i = (int(o) >> 30) & 1, int(o) & 0x3fffffff
            b = blocks[k]
            r = chk.tap_block(b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], b["ampmax_in"])
            g = {kk: v[i] for kk, v in host[W].items()}
            assert checker.compare_block(r, g, an.posts[W], keys=("mdct", "post_valid", "iwork", "nonzero"), verbose=True) == 0, (s, k)
        for k, o in enumerate(L["order"][lo:hi]):
            W, i = (int(o) >> 30) & 1, int(o) & 0x3fffffff
            assert np.float32(host[W]["ampmax_out"][i]) == np.float32(blocks[k]["ampmax_out"]), (s, k)
        assert np.float32(states[s].item()) == np.float32(blocks[hi - lo - 1]["ampmax_out"])
    # the same plan analysed IN PLACE -- the blocks read out of the stream buffers through the plan's offsets
    # (vamd_batch_io::pcm_src), no gathered copy -- gives the gathered run's tensors bit for bit, chain states included
    outs2 = [an.alloc_outputs(W, plan.nblocks[W], want) for W in (0, 1)]
    states2 = torch.full((len(raws),), -9999.0, device="cuda")
    an.analyze_plan(plan, None, outs2, states2, streams=streams)
    torch.cuda.synchronize()
    for W in (0, 1):
        for k in want:
            assert torch.equal(outs[W][k], outs2[W][k]), (W, k)
    assert torch.equal(states, states2)
    an.close()
