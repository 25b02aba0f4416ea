/* integration/envelope_vamd.c -- reference-side binding for the block-switching detector
 * (SURVEY.md 8f rank 1).  A maintainer builds this INSTEAD of lib/envelope.c:
 *
 *   - the reference's own envelope.c is pulled in by path (init / mark / shift are untouched;
 *     nothing is copied), with its _ve_envelope_search and _ve_envelope_clear renamed (clear is
 *     wrapped only to release the GPU context with the state it belongs to);
 *   - _ve_envelope_search() below keeps the reference's bookkeeping -- which steps are
 *     due (lib/envelope.c:224-232), how a step's flags land in ve->mark[] (:241-258) and the
 *     cursor walk that turns marks into a block-size decision (:262-325) -- but the
 *     detector itself, _ve_amp() for every due step and channel (:89-215, :234-239), is ONE
 *     call into libvorbis_amd.so.  The detector's running state (envelope_filter_state,
 *     ve->stretch) lives in a vamd_envelope_state next to the GPU context.
 *
 * Linked into oracle/_ref/libvorbis_hybrid.so together with mapping0_vamd.c;
 * tests/test_gpu_dropin.py checks that streams with transients (short blocks, transitions)
 * still encode to byte-identical packets.
 */
#define _ve_envelope_search _ve_envelope_search_cpu
#define _ve_envelope_clear _ve_envelope_clear_cpu
#include "envelope.c" /* the reference's lib/envelope.c, found through -I$(REF)/lib */
#undef _ve_envelope_search
#undef _ve_envelope_clear

#include "vorbis_amd.h"
#include <string.h>
#include <stdlib.h>

/* mapping0_vamd.c owns the per-vorbis_dsp_state side table */
extern vamd_ctx *vamd_ctx_for(vorbis_dsp_state *vd);
extern vamd_envelope_state *vamd_envelope_state_for(vorbis_dsp_state *vd);
extern void vamd_release_key(const void *key);
extern int vamd_batching(void);
extern void vamd_poison(vorbis_dsp_state *vd, int kind); /* mapping0_vamd.c: 1 = non-finite input, 2 = GPU failure */
extern int vamd_detector_dead(vorbis_dsp_state *vd, int set);
extern int vamd_detector_mode(vorbis_dsp_state *vd, int set);

/* Which detector serves a stream: VAMD_DETECTOR=gpu | host | auto (default auto; read once).
 * A detector call is a GPU round trip (~43 us) whatever its size.  An application that writes 1024 frames at a time
 * (the example's READ) asks for sixteen steps per block -- libvorbis' own _ve_amp does those in less time than the trip
 * takes -- while one that writes 65 536 frames asks for a thousand at once.  auto: a stream whose FIRST detector call
 * brings VAMD_DETECTOR_GPU_FROM (256) steps or more is served by the GPU for life, any other by the reference's own
 * detector (this file includes lib/envelope.c unmodified; the same choice VAMD_BATCH mode makes for every stream).  The
 * two keep their running state in different places, so a stream never changes sides. */
static int vamd_detector_policy(void) {
  static int policy = -1; /* 0 auto, 1 host, 2 gpu */
  if (policy < 0) {
    const char *v = getenv("VAMD_DETECTOR");
    policy = !v ? 0 : (!strcmp(v, "host") ? 1 : (!strcmp(v, "gpu") ? 2 : 0));
  }
  return policy;
}
static long vamd_detector_gpu_from(void) {
  static long n = -1;
  if (n < 0) {
    const char *v = getenv("VAMD_DETECTOR_GPU_FROM");
    n = v ? atol(v) : 256;
    if (n < 1) n = 1;
  }
  return n;
}

/* vorbis_dsp_clear() tears the detector down here (lib/block.c:325-328): the GPU context that
 * was created for this analysis state goes with it */
void _ve_envelope_clear(envelope_lookup *e) {
  vamd_release_key(e);
  _ve_envelope_clear_cpu(e);
}

long _ve_envelope_search(vorbis_dsp_state *v) {
  /* batch mode (VAMD_BATCH, mapping0_vamd.c): the shared context belongs to the batcher's leader, and a detector
     round trip per blockout call is exactly what batching is there to avoid -- the reference's own detector runs */
  if (vamd_batching()) return _ve_envelope_search_cpu(v);
  {
    int mode = vamd_detector_mode(v, 0);
    if (!mode) {
      envelope_lookup *ve0 = ((private_state *)(v->backend_state))->ve;
      long first0 = ve0->current / ve0->searchstep;
      const long last0 = v->pcm_current / ve0->searchstep - VE_WIN;
      if (first0 < 0) first0 = 0;
      if (last0 > first0) /* the stream's first steps: decide */
        mode = vamd_detector_mode(v, vamd_detector_policy() ? vamd_detector_policy() : (last0 - first0 >= vamd_detector_gpu_from() ? 2 : 1));
    }
    if (mode != 2) return _ve_envelope_search_cpu(v); /* (undecided: no step is due, the bookkeeping is the same either way) */
  }
  {
  vorbis_info *vi = v->vi;
  codec_setup_info *ci = vi->codec_setup;
  envelope_lookup *ve = ((private_state *)(v->backend_state))->ve;
  const long step = ve->searchstep;
  long first = ve->current / step;
  const long last = v->pcm_current / step - VE_WIN;
  long j;

  if (first < 0) first = 0;
  if (last + VE_WIN + VE_POST > ve->storage) { /* :229-232 */
    ve->storage = last + VE_WIN + VE_POST;
    ve->mark = _ogg_realloc(ve->mark, ve->storage * sizeof(*ve->mark));
  }

  if (last > first) {
    vamd_ctx *ctx = vamd_ctx_for(v);
    vamd_envelope_state *st = vamd_envelope_state_for(v);
    const long nsteps = last - first;
    unsigned char *flags = _ogg_malloc(nsteps);
    const float **chan = alloca(sizeof(*chan) * ve->ch);
    int i, err = -1;
    for (i = 0; i < ve->ch; i++) chan[i] = v->pcm[i] + step * first;
    if (vamd_detector_dead(v, 0)) {
      memset(flags, 0, nsteps); /* a non-finite sample lies behind: the stream ends at its block, no more steps are taken */
      err = 0;
    } else if (ctx && st)
      err = vamd_envelope_search(ctx, chan, nsteps, st, flags);
    if (err == VAMD_ENONFINITE) {
      /* A NaN / Inf somewhere in these steps (include/vorbis_amd.h, "Input domain").  The blocks IN FRONT of it are
         valid and their block-switching decisions need the marks of the clean steps: find the longest clean prefix
         (the call leaves the state alone when it fails, so halving costs a dozen calls, once per stream), keep its
         flags, and stop the detector.  The stream is not poisoned here: the block that holds the sample is refused by
         its own verdict (mapping0_vamd.c: vamd_domain_verdict) -- and with it every later block. */
      long lo = 0, hi = nsteps; /* [0, lo) is clean, step `hi - 1` (or an earlier one) is not */
      vamd_envelope_state probe;
      while (lo + 1 < hi) {
        const long mid = (lo + hi) / 2;
        probe = *st;
        if (vamd_envelope_search(ctx, chan, mid, &probe, flags) == VAMD_OK) lo = mid; else hi = mid;
      }
      memset(flags, 0, nsteps);
      err = lo > 0 ? vamd_envelope_search(ctx, chan, lo, st, flags) : 0;
      vamd_detector_dead(v, 1);
    }
    if (err) {
      /* This entry point has no error return (1 / 0 / -1 all mean something), and a shared library does not end its
         host process.  The stream is flagged instead (mapping0_vamd.c: vamd_poison) and the NEXT vorbis_analysis()
         reports it: OV_EFAULT -- no context, device lost, out of memory, a HIP fault; the text stays with
         vamd_last_error().  No marks come from these steps and the bookkeeping below carries on, so that the blocks
         keep flowing and the caller meets the error at once, not at end of stream (returning -1, "need more data",
         from here on would stall vorbis_analysis_blockout() until the input ends: lib/block.c:558-563). */
      memset(flags, 0, nsteps);
      vamd_poison(v, err == VAMD_ENONFINITE ? 1 : 2);
    }
    for (j = first; j < last; j++) { /* :241-258 */
      const int ret = flags[j - first];
      ve->mark[j + VE_POST] = 0;
      if (ret & 1) {
        ve->mark[j] = 1;
        ve->mark[j + 1] = 1;
      }
      if (ret & 2) {
        ve->mark[j] = 1;
        if (j > 0) ve->mark[j - 1] = 1;
      }
    }
    if (!err && st) ve->stretch = st->stretch;
    _ogg_free(flags);
  }
  ve->current = last * step;

  { /* :262-325: first marked step between the cursor and the decision horizon */
    const long centerW = v->centerW;
    const long testW = centerW + ci->blocksizes[v->W] / 4 + ci->blocksizes[1] / 2 + ci->blocksizes[0] / 4;
    for (j = ve->cursor; j < ve->current - step; j += step) {
      if (j >= testW) return 1;
      ve->cursor = j;
      if (ve->mark[j / step] && j > centerW) {
        ve->curmark = j;
        return j >= testW ? 1 : 0;
      }
    }
  }
  return -1;
  }
}
