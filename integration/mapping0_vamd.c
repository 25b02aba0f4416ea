/* integration/mapping0_vamd.c -- the reference-side binding for the drop-in boundary.
 *
 * libvorbis has no runtime hook for the mapping back-end: _mapping_P[] and
 * mapping0_exportbundle are const objects (reference lib/registry.c:42-44,
 * lib/mapping0.c:802-808), so the hook is at link level.  This
 * translation unit is what a maintainer builds INSTEAD of lib/mapping0.c:
 *
 *   - it pulls the reference's own mapping0.c in by path (pack/unpack/free_info/
 *     inverse and the CPU forward stay exactly as they are; nothing is copied),
 *     renaming only the exported bundle;
 *   - it defines mapping0_forward_vamd(): all of mapping0_forward (lib/mapping0.c:254-687)
 *     is ONE call into libvorbis_amd.so that returns the block's packet bytes
 *     (vamd_encode_block), copied into vbi->packetblob[]; for the few modes whose residue
 *     the GPU does not search, the numeric section (:254-576,613-646) is one call and the
 *     bit-writing half (packet header, floor1_encode's Huffman writes, res*_class /
 *     res*_forward) remains the reference's unchanged host code;
 *   - it exports a mapping0_exportbundle whose .forward is that function.
 *
 * Everything else in libvorbis / libvorbisenc links unchanged.  oracle/Makefile
 * builds exactly this into oracle/_ref/libvorbis_hybrid.so, and
 * tests/test_gpu_dropin.py checks that an encode through the hybrid library
 * emits byte-identical packets to the pure reference.
 *
 * Bitrate-managed encoders get all PACKETBLOBS candidate packets the same way;
 * the bitrate manager that picks one is untouched host code.
 * With VAMD_BATCH=<n> in the environment the VBR blocks of ALL encoder states that share a setup go through one
 * vamd_batcher (include/vorbis_amd.h): many application threads, each driving its own vorbis_dsp_state as libvorbis
 * allows, get their blocks analysed in shared GPU batches of up to n.  (There is no gathering timer any more: blocks
 * gather while the batcher's lanes are busy and leave at once when one is idle; VAMD_BATCH_WAIT_US is still read and
 * handed on, and ignored by the batcher.)  The block-switching detector then stays on the host (envelope_vamd.c).
 * Channel counts above VAMD_MAX_CH are refused with OV_EIMPL (see mapping0_forward_vamd); errors
 * travel as OV_* return codes like everywhere else in libvorbis -- nothing is printed.
 */
#define mapping0_exportbundle mapping0_exportbundle_cpu
#include "mapping0.c" /* the reference's lib/mapping0.c, found through -I$(REF)/lib */
#undef mapping0_exportbundle

#include "vorbis_amd.h"

extern long vamd_pack_setup(vorbis_dsp_state *vd, void *dst, long cap);

/* one GPU context per analysis state, created on first use (vorbis_analysis_init has
 * already built every lookup by then).  A real integration would hang the pointer off
 * private_state; a side table keeps this file self-contained.  The key is the state's
 * envelope_lookup (private_state.ve, lib/block.c:304): it is allocated by
 * vorbis_analysis_init and handed to _ve_envelope_clear() by vorbis_dsp_clear()
 * (lib/block.c:325-328), which is where envelope_vamd.c releases the entry -- so an entry
 * never outlives its stream even though no reference file is edited.
 *
 * libvorbis lets independent vorbis_dsp_states live on different threads, so the table is
 * guarded by a mutex, grows on demand, and holds POINTERS to heap entries: an entry's
 * address (its detector state is handed out by pointer) stays put while the table moves.
 * One state is still used by one thread at a time, as libvorbis itself requires. */
#include <pthread.h>
#include <stdlib.h>
typedef struct vamd_shared { /* one batcher per distinct setup (VAMD_BATCH mode) */
  void *blob;
  long bytes;
  vamd_batcher *batcher;
  int users;
} vamd_shared;
/* ---- look-ahead inside one stream (round 5).  An application that hands vorbis_analysis_wrote() more than a block's
 * worth of samples (the API takes any amount, lib/block.c:390,470; the example's READ 1024 is a choice,
 * examples/encoder_example.c:38) has, by the time it asks for the first block, determined every block its buffer
 * covers: the marks of _ve_envelope_search are there for all the buffered steps, and what vorbis_analysis_blockout will
 * decide for the following blocks follows from them (lib/block.c:534-693, lib/envelope.c:262-353).  On a cache miss the
 * binding therefore replays those decisions ahead of the reference's own blockout (vamd_plan_ahead: the same walk, in
 * the buffer's current coordinates, nothing shifted), sends the block at hand AND the blocks to come through ONE
 * launch sequence (vamd_encode_blocks: the ampmax chain between them on the device) and keeps the packets.  Each later
 * vorbis_analysis() is then served from that cache -- after the block the reference's blockout really produced has been
 * held against the planned one: sequence number, window flags, block type, the incoming ampmax bit for bit and every
 * sample.  A packet is a pure function of exactly those, so a hit is the packet the single-block path would have
 * produced; anything else (more data written in between and a decision changed, end of stream, a replay that went
 * astray) is a miss, which costs the cache and nothing else.  One GPU round trip per buffered stretch instead of one per
 * block.  VBR and bitrate-managed encoders alike (a managed block's fifteen candidate packets ride the same batch; which
 * one goes out is the bitrate manager's business afterwards, as before).  VAMD_LOOKAHEAD=0 switches it off; VAMD_BATCH
 * mode does not use it. */
#define VAMD_AHEAD_MAX 256
typedef struct vamd_ahead_block {
  long sequence;
  int32_t lW, W, nW, blocktype;
  float ampmax_in, ampmax_out;
  int32_t verdict;
  long pcm_at; /* offset (floats) of the block's [ch][n] samples in vamd_ahead.pcm */
} vamd_ahead_block;
typedef struct vamd_ahead {
  int count, next;                /* planned blocks held, and the one expected next */
  vamd_ahead_block blk[VAMD_AHEAD_MAX];
  float *pcm;                     /* the planned blocks' samples as they were sent (what a hit is verified against) */
  long pcm_cap;
  unsigned char *packets;         /* [VAMD_AHEAD_MAX][K][stride]: K = 1, or a bitrate-managed block's PACKETBLOBS candidates */
  int32_t *bits;                  /* [VAMD_AHEAD_MAX][K] */
  long stride;
  long rows;                      /* packet rows (x K) allocated: grows with the plans the stream really makes */
  int K;
  long hits, misses, batches;     /* (diagnostics: vamd_ahead_stats) */
} vamd_ahead;

typedef struct vamd_entry {
  const void *key; /* private_state.ve */
  vamd_ctx *ctx;   /* this state's own context (always in per-state mode; in batch mode only for what is not batched) */
  vamd_shared *shared; /* batch mode: the batcher this state submits to */
  vamd_envelope_state env; /* the block-switching detector's running state (envelope_vamd.c) */
  int poisoned;            /* the stream is over (vamd_poison): VAMD_POISON_NONFINITE -> every later block is OV_EINVAL,
                              VAMD_POISON_FAULT -> OV_EFAULT */
  int detector_mode;       /* envelope_vamd.c: 0 undecided, 1 libvorbis' own detector on the host, 2 the GPU's -- per stream, for life */
  int detector_dead;       /* a non-finite sample has reached the detector (envelope_vamd.c): no more steps are taken;
                              the stream ends at the block that holds the sample (its own verdict poisons it) */
  vamd_ahead *ahead;       /* the look-ahead cache (allocated on first use) */
} vamd_entry;
static pthread_mutex_t vamd_lock = PTHREAD_MUTEX_INITIALIZER;
static long vamd_ahead_total[3]; /* look-ahead hits / misses / batches of the streams already closed (vamd_ahead_stats) */
static vamd_entry **vamd_table = NULL;
static int vamd_count = 0, vamd_cap = 0;
static vamd_shared *vamd_shares[16];
static int vamd_nshares = 0;

static const void *vamd_key(vorbis_dsp_state *vd) { return ((private_state *)vd->backend_state)->ve; }

/* VAMD_BATCH=<max blocks per batch> switches the process to batch mode (read once) */
int vamd_batching(void) {
  static int mode = -1;
  if (mode < 0) {
    const char *v = getenv("VAMD_BATCH");
    mode = v ? atoi(v) : 0;
    if (mode < 0) mode = 0;
  }
  return mode;
}

/* the batcher for this setup blob, created on first use (vamd_lock held); takes ownership of `blob` when it keeps it */
static vamd_shared *vamd_share_for(void *blob, long bytes, int *kept) {
  int i;
  *kept = 0;
  for (i = 0; i < vamd_nshares; i++)
    if (vamd_shares[i]->bytes == bytes && !memcmp(vamd_shares[i]->blob, blob, bytes)) return vamd_shares[i];
  if (vamd_nshares < (int)(sizeof(vamd_shares) / sizeof(vamd_shares[0]))) {
    vamd_shared *sh = _ogg_calloc(1, sizeof(*sh));
    const char *w = getenv("VAMD_BATCH_WAIT_US");
    if (sh && vamd_batcher_create(&sh->batcher, blob, (size_t)bytes, -1, vamd_batching(), w ? atoi(w) : 200) == VAMD_OK) {
      sh->blob = blob;
      sh->bytes = bytes;
      *kept = 1;
      vamd_shares[vamd_nshares++] = sh;
      return sh;
    }
    if (sh) _ogg_free(sh);
  }
  return NULL;
}

/* the entry of `state`, created (context or batcher and all) on first use; NULL if the GPU side cannot be set up */
static vamd_entry *vamd_entry_for(vorbis_dsp_state *state) {
  const void *key = vamd_key(state);
  vamd_entry *e = NULL;
  int i;
  if (!key) return NULL;
  pthread_mutex_lock(&vamd_lock);
  for (i = 0; i < vamd_count; i++)
    if (vamd_table[i]->key == key) {
      e = vamd_table[i];
      break;
    }
  if (!e) {
    /* (held across vamd_create: two threads opening encoders at once serialise here, once per stream) */
    long need = vamd_pack_setup(state, NULL, 0);
    if (need >= 0) {
      void *blob = _ogg_malloc(need);
      vamd_ctx *ctx = NULL;
      vamd_shared *sh = NULL;
      int kept = 0, ok = 0;
      if (blob && vamd_pack_setup(state, blob, need) == need) {
        if (vamd_batching())
          ok = (sh = vamd_share_for(blob, need, &kept)) != NULL;
        else
          ok = vamd_create(&ctx, blob, (size_t)need, -1) == VAMD_OK;
      }
      if (ok) {
        if (vamd_count == vamd_cap) {
          int ncap = vamd_cap ? 2 * vamd_cap : 16;
          vamd_entry **nt = _ogg_realloc(vamd_table, ncap * sizeof(*nt));
          if (nt) vamd_table = nt, vamd_cap = ncap;
        }
        if (vamd_count < vamd_cap && (e = _ogg_calloc(1, sizeof(*e)))) { /* env all-zero: a fresh stream, lib/envelope.c:71 */
          e->key = key;
          e->ctx = ctx;
          e->shared = sh;
          if (sh) {
            sh->users++;
            vamd_batcher_attach(sh->batcher);
          }
          vamd_table[vamd_count++] = e;
        } else if (ctx) {
          vamd_destroy(ctx);
        }
      }
      if (blob && !kept) _ogg_free(blob);
    }
  }
  pthread_mutex_unlock(&vamd_lock);
  return e;
}

/* this state's OWN context: what every per-state call uses; in batch mode it is made only when something that is
 * not batched asks for it (a bitrate-managed encoder's fifteen candidates) */
vamd_ctx *vamd_ctx_for(vorbis_dsp_state *state) {
  vamd_entry *e = vamd_entry_for(state);
  if (!e) return NULL;
  if (!e->ctx && e->shared) {
    pthread_mutex_lock(&vamd_lock);
    if (!e->ctx && vamd_create(&e->ctx, e->shared->blob, (size_t)e->shared->bytes, -1) != VAMD_OK) e->ctx = NULL;
    pthread_mutex_unlock(&vamd_lock);
  }
  return e->ctx;
}

vamd_envelope_state *vamd_envelope_state_for(vorbis_dsp_state *state) {
  vamd_entry *e = vamd_entry_for(state);
  return e ? &e->env : NULL;
}

/* A stream that cannot go on.  Two ways in (include/vorbis_amd.h, "Input domain"):
 *   VAMD_POISON_NONFINITE  a NaN / Inf sample: the reference's own ampmax chain and detector history are not numbers
 *                          from here on, so the encode ends -- vorbis_analysis() returns OV_EINVAL for the block that
 *                          held the sample and for every block after it;
 *   VAMD_POISON_FAULT      the GPU side failed under the block-switching detector (device lost, out of memory, a HIP
 *                          fault): OV_EFAULT from the next vorbis_analysis() on.
 * The detector has no error return (envelope_vamd.c), so for a GPU failure it records the fact here and lets the next
 * vorbis_analysis() report it.  A non-finite sample under the DETECTOR does not poison the stream by itself (round 5 did:
 * with a 65 536-frame write the detector scans ~60 blocks ahead, and the valid blocks in front of the sample were
 * refused with it -- ADVICE r05): the detector keeps the marks of the clean steps in front of the sample and stops, and
 * the stream ends where include/vorbis_amd.h says -- at the block that holds the sample, by that block's own verdict.
 * (A FINITE block beyond the integer bound is neither: OV_EINVAL for that block alone, below.) */
#define VAMD_POISON_NONFINITE 1
#define VAMD_POISON_FAULT 2
void vamd_poison(vorbis_dsp_state *state, int kind) {
  vamd_entry *e = vamd_entry_for(state);
  if (e && !e->poisoned) e->poisoned = kind;
}
/* envelope_vamd.c: which detector serves this stream (set != 0: decide) */
int vamd_detector_mode(vorbis_dsp_state *state, int set) {
  vamd_entry *e = vamd_entry_for(state);
  if (e && set && !e->detector_mode) e->detector_mode = set;
  return e ? e->detector_mode : 1;
}
/* envelope_vamd.c: the detector met a non-finite sample (1: from now on it takes no steps) / is it dead? */
int vamd_detector_dead(vorbis_dsp_state *state, int set) {
  vamd_entry *e = vamd_entry_for(state);
  if (e && set) e->detector_dead = 1;
  return e ? e->detector_dead : 0;
}
static int vamd_poisoned(vorbis_dsp_state *state) {
  vamd_entry *e = vamd_entry_for(state);
  return e ? e->poisoned : 0;
}

/* what vorbis_analysis() returns for a block the library found outside the input domain.  A non-finite sample ends
 * the stream; a finite block beyond the integer bound has no defined result, but the stream's state does: the ampmax
 * chain is carried over it (vamd_* deliver ampmax_out with the error) and the next block is the reference's again. */
static int vamd_domain_verdict(vorbis_block *vb, int ret, float ampmax_out) {
  if (ret == VAMD_ENONFINITE) {
    vamd_poison(vb->vd, VAMD_POISON_NONFINITE);
    return OV_EINVAL;
  }
  if (ret == VAMD_EDOMAIN) {
    ((vorbis_block_internal *)vb->internal)->ampmax = ampmax_out; /* lib/mapping0.c:576 */
    return OV_EINVAL;
  }
  return ret;
}

/* called by _ve_envelope_clear() (envelope_vamd.c) with the envelope_lookup being torn down */
void vamd_release_key(const void *key) {
  vamd_entry *e = NULL;
  vamd_shared *dead = NULL;
  int i;
  if (!key) return;
  pthread_mutex_lock(&vamd_lock);
  for (i = 0; i < vamd_count; i++)
    if (vamd_table[i]->key == key) {
      e = vamd_table[i];
      vamd_table[i] = vamd_table[--vamd_count];
      break;
    }
  if (e && e->shared) {
    vamd_batcher_detach(e->shared->batcher);
    if (--e->shared->users == 0) { /* the last stream of this setup takes the batcher with it */
      dead = e->shared;
      for (i = 0; i < vamd_nshares; i++)
        if (vamd_shares[i] == dead) vamd_shares[i] = vamd_shares[--vamd_nshares];
    }
  }
  if (e && e->ahead) vamd_ahead_total[0] += e->ahead->hits, vamd_ahead_total[1] += e->ahead->misses, vamd_ahead_total[2] += e->ahead->batches;
  pthread_mutex_unlock(&vamd_lock);
  if (dead) {
    vamd_batcher_destroy(dead->batcher);
    _ogg_free(dead->blob);
    _ogg_free(dead);
  }
  if (e) {
    if (e->ctx) vamd_destroy(e->ctx);
    if (e->ahead) {
      if (e->ahead->pcm) _ogg_free(e->ahead->pcm);
      if (e->ahead->packets) _ogg_free(e->ahead->packets);
      if (e->ahead->bits) _ogg_free(e->ahead->bits);
      _ogg_free(e->ahead);
    }
    _ogg_free(e);
  }
}


/* batches run and blocks carried so far, over all batchers (diagnostics) */
void vamd_batch_stats(long *batches, long *blocks, double *run_seconds) {
  int i;
  *batches = *blocks = 0;
  *run_seconds = 0.;
  pthread_mutex_lock(&vamd_lock);
  for (i = 0; i < vamd_nshares; i++) {
    long a = 0, b = 0;
    double t = 0.;
    vamd_batcher_stats(vamd_shares[i]->batcher, &a, &b, &t);
    *batches += a;
    *blocks += b;
    *run_seconds += t;
  }
  pthread_mutex_unlock(&vamd_lock);
}

/* the first batcher's own account of where its batches' time went (diagnostics) */
long vamd_batch_trace(char *buf, long cap) {
  long n = 0;
  pthread_mutex_lock(&vamd_lock);
  if (vamd_nshares) n = vamd_batcher_report(vamd_shares[0]->batcher, buf, cap);
  pthread_mutex_unlock(&vamd_lock);
  return n;
}

/* for a build WITHOUT envelope_vamd.c: call from vorbis_dsp_clear() before b->ve is freed */
void vamd_release_state(vorbis_dsp_state *state) { vamd_release_key(vamd_key(state)); }

/* VAMD_LOOKAHEAD=0 switches the look-ahead off (read once) */
static int vamd_lookahead_on(void) {
  static int mode = -1;
  if (mode < 0) {
    const char *v = getenv("VAMD_LOOKAHEAD");
    mode = v ? (atoi(v) != 0) : 1;
  }
  return mode;
}
/* VAMD_LOOKAHEAD_MAX=<blocks> bounds how far one plan reaches (default: as far as the cache's slots go -- 255 blocks, 63
 * for a bitrate-managed encoder); what a stream's cache holds follows from it: packet rows and a copy of the planned
 * blocks' samples for that many blocks, allocated for the plans the stream really makes (read once) */
static int vamd_lookahead_max(void) {
  static int cap = -1;
  if (cap < 0) {
    const char *v = getenv("VAMD_LOOKAHEAD_MAX");
    cap = v ? atoi(v) : VAMD_AHEAD_MAX - 1;
    if (cap < 1) cap = 1;
    if (cap > VAMD_AHEAD_MAX - 1) cap = VAMD_AHEAD_MAX - 1;
  }
  return cap;
}

/* What vorbis_analysis_blockout() will decide for the blocks AFTER the one it has just handed out, as far as the buffered
 * samples determine them: the state in `v` is already the next block's (lib/block.c:649-685), the marks of every buffered
 * step are in ve->mark (the blockout that produced the current block ran _ve_envelope_search over all of them), and
 * nothing is shifted here, so every position stays in the buffer's current coordinates -- shifting subtracts the same
 * amount from both sides of every comparison below.  The cursor walk is lib/envelope.c:262-325, the size / window /
 * blocktype decisions lib/block.c:552-611 with _ve_envelope_mark lib/envelope.c:329-353.  Stops where blockout would
 * say "not enough data", or at the stream's last block once the application has written its last samples (an
 * application that writes a whole clip, signals the end and only then pulls its blocks gets them all looked ahead).
 * begin[k] = first sample of planned block k's window in v->pcm[]. */
static int vamd_plan_ahead(vorbis_dsp_state *v, int max, vamd_ahead_block *out, long *begin) {
  codec_setup_info *ci = v->vi->codec_setup;
  envelope_lookup *ve = ((private_state *)v->backend_state)->ve;
  const long step = ve->searchstep, current = ve->current;
  const long bs[2] = {ci->blocksizes[0], ci->blocksizes[1]};
  int W = (int)v->W, lW = (int)v->lW, n = 0;
  long centerW = v->centerW, cursor = ve->cursor, curmark = ve->curmark, j;
  const long eof = v->eofflag; /* > 0: the last real sample's position, once the application has written its last (:664-673) */
  if (eof < 0 || !v->preextrapolate) return 0;
  while (n < max) {
    const long testW = centerW + bs[W] / 4 + bs[1] / 2 + bs[0] / 4;
    int bp = -1, nW, blocktype;
    long centerNext;
    for (j = cursor; j < current - step; j += step) {
      if (j >= testW) {
        bp = 1;
        break;
      }
      cursor = j;
      if (ve->mark[j / step] && j > centerW) {
        curmark = j;
        bp = j >= testW ? 1 : 0;
        break;
      }
    }
    if (bp < 0 && !eof) break; /* lib/block.c:558-560: not enough data to search a full long block -- unless no more will come */
    nW = (bp < 0 || bs[0] == bs[1]) ? 0 : bp; /* :561-568 */
    centerNext = centerW + bs[W] / 4 + bs[nW] / 4;
    if (v->pcm_current < centerNext + bs[nW] / 2) break; /* :574-583 */
    if (W) {
      blocktype = (!lW || !nW) ? BLOCKTYPE_TRANSITION : BLOCKTYPE_LONG;
    } else {
      const long beginW = centerW - bs[0] / 4 - bs[0] / 4, endW = centerW + bs[0] / 4 + bs[0] / 4;
      int hit = curmark >= beginW && curmark < endW;
      long i;
      for (i = beginW / step; !hit && i < endW / step; i++) hit = i >= 0 && i < ve->storage && ve->mark[i];
      blocktype = hit ? BLOCKTYPE_IMPULSE : BLOCKTYPE_PADDING;
    }
    out[n].lW = lW, out[n].W = W, out[n].nW = nW, out[n].blocktype = blocktype;
    begin[n] = centerW - bs[W] / 2;
    if (begin[n] < 0 || begin[n] + bs[W] > v->pcm_current) break;
    n++;
    if (eof && centerW >= eof) break; /* the stream's last block, :664-670 */
    lW = W;
    W = nW;
    centerW = centerNext;
  }
  return n;
}

/* hits, misses and batches of every stream's look-ahead since the process started, closed streams included (diagnostics) */
void vamd_ahead_stats(long *hits, long *misses, long *batches) {
  int i;
  pthread_mutex_lock(&vamd_lock);
  *hits = vamd_ahead_total[0], *misses = vamd_ahead_total[1], *batches = vamd_ahead_total[2];
  for (i = 0; i < vamd_count; i++)
    if (vamd_table[i]->ahead) {
      *hits += vamd_table[i]->ahead->hits;
      *misses += vamd_table[i]->ahead->misses;
      *batches += vamd_table[i]->ahead->batches;
    }
  pthread_mutex_unlock(&vamd_lock);
}

/* The bit-writing half for one candidate packet k where the GPU does not assemble packets (the mode's
 * residue back-end is not covered there): unchanged host code, lib/mapping0.c:596-687 -- packet header,
 * floor1_encode's Huffman words from the posts the GPU fitted, res*_class / res*_forward.
 *   posts / post_valid / iwork / nonzero  this candidate's [ch][...] rows */
static int vamd_write_packet(vorbis_block *vb, int k, int *posts, const int *post_valid, int *iwork,
                             const int *nonzero, int *scratch) {
  vorbis_dsp_state *vd = vb->vd;
  vorbis_info *vi = vd->vi;
  codec_setup_info *ci = vi->codec_setup;
  private_state *b = vd->backend_state;
  vorbis_block_internal *vbi = (vorbis_block_internal *)vb->internal;
  const int n = vb->pcmend, ch = vi->channels, modenumber = vb->W;
  vorbis_info_mapping0 *info = ci->map_param[modenumber];
  oggpack_buffer *opb = vbi->packetblob[k];
  int **couple_bundle = alloca(sizeof(*couple_bundle) * ch);
  int *zerobundle = alloca(sizeof(*zerobundle) * ch);
  int i, j;

  oggpack_write(opb, 0, 1);
  oggpack_write(opb, modenumber, b->modebits);
  if (vb->W) {
    oggpack_write(opb, vb->lW, 1);
    oggpack_write(opb, vb->nW, 1);
  }
  for (i = 0; i < ch; i++) {
    int submap = info->chmuxlist[i];
    /* the integer curve floor1_encode renders as a side effect is identical to the one the GPU
       already divided out, and is discarded */
    floor1_encode(opb, vb, b->flr[info->floorsubmap[submap]], post_valid[i] ? posts + i * VAMD_POSTS_STRIDE : NULL,
                  scratch);
  }
  for (i = 0; i < info->submaps; i++) {
    int ch_in_bundle = 0;
    long **classifications;
    int resnum = info->residuesubmap[i];
    for (j = 0; j < ch; j++)
      if (info->chmuxlist[j] == i) {
        zerobundle[ch_in_bundle] = nonzero[j] ? 1 : 0; /* already carries the coupling fix-up */
        couple_bundle[ch_in_bundle++] = iwork + j * (n / 2);
      }
    classifications = _residue_P[ci->residue_type[resnum]]->class(vb, b->residue[resnum], couple_bundle, zerobundle,
                                                                 ch_in_bundle);
    ch_in_bundle = 0;
    for (j = 0; j < ch; j++)
      if (info->chmuxlist[j] == i) couple_bundle[ch_in_bundle++] = iwork + j * (n / 2);
    _residue_P[ci->residue_type[resnum]]->forward(opb, vb, b->residue[resnum], couple_bundle, zerobundle,
                                                   ch_in_bundle, classifications, i);
  }
  return 0;
}

static int mapping0_forward_vamd(vorbis_block *vb) {
  vorbis_dsp_state *vd = vb->vd;
  vorbis_info *vi = vd->vi;
  vorbis_block_internal *vbi = (vorbis_block_internal *)vb->internal;
  const int n = vb->pcmend, ch = vi->channels;
  const int managed = vorbis_bitrate_managed(vb) ? 1 : 0, nk = managed ? PACKETBLOBS : 1;
  vamd_ctx *ctx;
  float *mdct;
  int *iwork, *posts, *post_valid, *nonzero, *scratch;
  float ampmax_out = 0.f;
  int k, ret, pkcap;

  /* more than VAMD_MAX_CH (8) channels -- no layout Vorbis I assigns a channel order to -- is outside the GPU
     path.  That is an error here (OV_EIMPL out of vorbis_analysis()), not a quiet detour: a maintainer who wants
     such streams on the host's own mapping0_forward says so at build time. */
  if (ch > VAMD_MAX_CH) {
#ifdef VAMD_HOST_FORWARD_ABOVE_MAX_CH
    return mapping0_forward(vb);
#else
    return OV_EIMPL;
#endif
  }
  vb->mode = vb->W;
  if ((k = vamd_poisoned(vd))) return k == VAMD_POISON_FAULT ? OV_EFAULT : OV_EINVAL;

  /* ---- batch mode (VAMD_BATCH): the block joins whatever the other encoder threads have pending and comes back
     as its packet, exactly as from vamd_encode_block below */
  if (vamd_batching() && !managed) {
    vamd_entry *e = vamd_entry_for(vd);
    if (!e || !e->shared) return OV_EFAULT;
    pkcap = vamd_packet_capacity(vamd_batcher_context(e->shared->batcher), vb->W);
    if (pkcap > 0) {
      unsigned char *packet = _vorbis_block_alloc(vb, pkcap);
      int32_t bits = 0;
      ret = vamd_batcher_encode_block(e->shared->batcher, (const float *const *)vb->pcm, vb->lW, vb->W, vb->nW,
                                      vbi->blocktype, vbi->ampmax, &ampmax_out, packet, pkcap, &bits);
      if (ret) return vamd_domain_verdict(vb, ret, ampmax_out);
      vbi->ampmax = ampmax_out; /* lib/mapping0.c:576 */
      oggpack_writecopy(vbi->packetblob[PACKETBLOBS / 2], packet, bits);
      return 0;
    }
  }

  ctx = vamd_ctx_for(vd);
  if (!ctx) return OV_EFAULT; /* no silent fallback: a missing GPU is an error */

  /* ---- look-ahead inside one stream (the comment at vamd_ahead): serve this block from the packets planned earlier,
     or plan the blocks the buffer already determines and run them with this one */
  if (vamd_lookahead_on() && !vamd_batching() && vamd_packet_capacity(ctx, 0) > 0 && vamd_packet_capacity(ctx, 1) > 0) {
    vamd_entry *e = vamd_entry_for(vd);
    vamd_ahead *A = e ? e->ahead : NULL;
    const long bsz[2] = {((codec_setup_info *)vi->codec_setup)->blocksizes[0], ((codec_setup_info *)vi->codec_setup)->blocksizes[1]};
    const int maxplan = (managed && vamd_lookahead_max() > 63) ? 63 : vamd_lookahead_max(); /* (a managed block's packets are fifteen rows) */
    int i;
    if (A && A->next < A->count && A->K == nk) {
      const vamd_ahead_block *p = &A->blk[A->next];
      const long slot = (long)(p - A->blk);
      int same = p->sequence == (long)vb->sequence && p->W == vb->W && p->lW == vb->lW && p->nW == vb->nW &&
                 p->blocktype == vbi->blocktype && !memcmp(&p->ampmax_in, &vbi->ampmax, sizeof(float));
      for (i = 0; same && i < ch; i++) same = !memcmp(A->pcm + p->pcm_at + (long)i * n, vb->pcm[i], (size_t)n * sizeof(float));
      if (same) {
        A->next++;
        A->hits++;
        if (p->verdict) return vamd_domain_verdict(vb, p->verdict, p->ampmax_out);
        vbi->ampmax = p->ampmax_out; /* lib/mapping0.c:576 */
        for (k = 0; k < nk; k++)
          oggpack_writecopy(vbi->packetblob[managed ? k : PACKETBLOBS / 2], A->packets + (slot * nk + k) * A->stride,
                            A->bits[slot * nk + k]);
        return 0;
      }
      A->count = A->next = 0; /* the stream went another way than planned: the cache is worth nothing */
      A->misses++;
    }
    if (e) {
      vamd_ahead_block planned[VAMD_AHEAD_MAX];
      long begin[VAMD_AHEAD_MAX];
      const int np = vamd_plan_ahead(vd, maxplan, planned + 1, begin + 1);
      if (np > 0) {
        const int nb = np + 1;
        const float **ptr = _vorbis_block_alloc(vb, (long)nb * ch * sizeof(*ptr));
        int32_t *dW = _vorbis_block_alloc(vb, 4L * nb * sizeof(*dW)), *dlW = dW + nb, *dnW = dW + 2 * nb, *dbt = dW + 3 * nb;
        int32_t *verdict = _vorbis_block_alloc(vb, (long)nb * sizeof(*verdict));
        float *ain = _vorbis_block_alloc(vb, 2L * nb * sizeof(*ain)), *aout = ain + nb;
        long need = 0, at = 0, stride = vamd_packet_capacity(ctx, 0);
        int j;
        if (vamd_packet_capacity(ctx, 1) > stride) stride = vamd_packet_capacity(ctx, 1);
        if (!A) A = e->ahead = _ogg_calloc(1, sizeof(*A));
        for (j = 1; j < nb; j++) need += (long)ch * bsz[planned[j].W];
        if (A && A->pcm_cap < need) {
          if (A->pcm) _ogg_free(A->pcm);
          A->pcm = _ogg_malloc((size_t)(need + need / 2) * sizeof(float));
          A->pcm_cap = A->pcm ? need + need / 2 : 0;
        }
        if (A && (A->stride != stride || A->K != nk || A->rows < nb)) {
          /* rows for the plans this stream really makes (round 5 took the worst case -- 256 rows, 64 x 15 for a managed
             encoder: 5.6 MB a stream -- on the first plan of two blocks): what is needed now, doubled while it grows */
          long rows = nb > 2 * A->rows ? nb : 2 * A->rows;
          if (rows > maxplan + 1) rows = maxplan + 1;
          if (A->packets) _ogg_free(A->packets);
          if (A->bits) _ogg_free(A->bits);
          A->packets = _ogg_malloc((size_t)rows * nk * stride);
          A->bits = _ogg_malloc((size_t)rows * nk * sizeof(*A->bits));
          A->stride = (A->packets && A->bits) ? stride : 0;
          A->rows = A->stride ? rows : 0;
          A->K = nk;
          A->count = A->next = 0;
        }
        if (A && A->pcm_cap >= need && A->stride == stride && A->K == nk && A->rows >= nb) {
          planned[0].lW = vb->lW, planned[0].W = vb->W, planned[0].nW = vb->nW, planned[0].blocktype = vbi->blocktype;
          for (i = 0; i < ch; i++) ptr[i] = vb->pcm[i];
          for (j = 1; j < nb; j++) { /* the samples as they lie in the encoder's own buffer, kept for the comparison later */
            const long nj = bsz[planned[j].W];
            planned[j].pcm_at = at;
            for (i = 0; i < ch; i++) {
              memcpy(A->pcm + at, vd->pcm[i] + begin[j], (size_t)nj * sizeof(float));
              ptr[(long)j * ch + i] = A->pcm + at;
              at += nj;
            }
          }
          for (j = 0; j < nb; j++) dlW[j] = planned[j].lW, dW[j] = planned[j].W, dnW[j] = planned[j].nW, dbt[j] = planned[j].blocktype;
          A->count = A->next = 0;
          ret = vamd_encode_blocks(ctx, nb, ptr, dlW, dW, dnW, dbt, vbi->ampmax, managed, ain, aout, A->packets, stride, A->bits,
                                   verdict);
          if (ret) return ret;
          A->batches++;
          for (j = 1; j < nb; j++) { /* (slot j of the cache = block j of the batch: its packet rows are rows j * nk ..) */
            vamd_ahead_block *q = &A->blk[j];
            *q = planned[j];
            q->sequence = (long)vb->sequence + j;
            q->ampmax_in = ain[j], q->ampmax_out = aout[j], q->verdict = verdict[j];
          }
          A->next = 1;
          A->count = nb;
          if (verdict[0]) return vamd_domain_verdict(vb, verdict[0], aout[0]);
          vbi->ampmax = aout[0]; /* lib/mapping0.c:576 */
          for (k = 0; k < nk; k++) {
            if (A->bits[k] > 8 * stride) return OV_EFAULT; /* cannot happen: stride is the worst case */
            oggpack_writecopy(vbi->packetblob[managed ? k : PACKETBLOBS / 2], A->packets + (long)k * stride, A->bits[k]);
          }
          return 0;
        }
      }
    }
  }

  /* ---- the whole of mapping0_forward in one call (lib/mapping0.c:254-687): the block's finished
     packet -- all PACKETBLOBS candidates for a bitrate-managed encoder, which lets
     vorbis_bitrate_addblock() choose (lib/mapping0.c:593-595) -- comes back as bytes */
  pkcap = vamd_packet_capacity(ctx, vb->W);
  if (pkcap > 0) {
    unsigned char *packets = _vorbis_block_alloc(vb, (long)nk * pkcap);
    int32_t *bits = _vorbis_block_alloc(vb, nk * sizeof(*bits));
    ret = vamd_encode_block(ctx, (const float *const *)vb->pcm, vb->lW, vb->W, vb->nW, vbi->blocktype, vbi->ampmax,
                            managed, &ampmax_out, packets, pkcap, bits);
    if (ret) return vamd_domain_verdict(vb, ret, ampmax_out); /* an OV_* code, out through vorbis_analysis(); the text stays with vamd_last_error(ctx) */
    vbi->ampmax = ampmax_out; /* lib/mapping0.c:576 */
    for (k = 0; k < nk; k++) {
      if (bits[k] > 8 * pkcap) return OV_EFAULT; /* cannot happen: pkcap is the worst case */
      oggpack_writecopy(vbi->packetblob[managed ? k : PACKETBLOBS / 2], packets + (long)k * pkcap, bits[k]);
    }
    return 0;
  }

  /* ---- modes whose residue the GPU does not search (none that libvorbisenc sets up; hand-built ones): the numeric
     section through couple/quantise is one call (lib/mapping0.c:254-576,613-646; managed: +:507-573
     for each candidate), the bit-writing half stays the reference's host code */
  mdct = _vorbis_block_alloc(vb, ch * (n / 2) * sizeof(*mdct));
  scratch = _vorbis_block_alloc(vb, (n / 2) * sizeof(*scratch));
  iwork = _vorbis_block_alloc(vb, nk * ch * (n / 2) * sizeof(*iwork));
  posts = _vorbis_block_alloc(vb, nk * ch * VAMD_POSTS_STRIDE * sizeof(*posts));
  post_valid = _vorbis_block_alloc(vb, nk * ch * sizeof(*post_valid));
  nonzero = _vorbis_block_alloc(vb, nk * ch * sizeof(*nonzero));
  if (managed)
    ret = vamd_analyze_block_managed(ctx, (const float *const *)vb->pcm, vb->lW, vb->W, vb->nW, vbi->blocktype,
                                     vbi->ampmax, mdct, &ampmax_out, posts, post_valid, iwork, nonzero, NULL, NULL,
                                     NULL);
  else
    ret = vamd_analyze_block(ctx, (const float *const *)vb->pcm, vb->lW, vb->W, vb->nW, vbi->blocktype, vbi->ampmax,
                             mdct, NULL, posts, post_valid, iwork, nonzero, &ampmax_out);
  if (ret) return vamd_domain_verdict(vb, ret, ampmax_out);
  /* the residue coder below is the host's own: the input domain's test on the values it will search (vorbis_amd.h,
     "Input domain" (2)) is then the caller's -- every quantised value at a bin the residue codes within the setup's bound */
  for (k = 0; k < nk; k++) {
    int c;
    for (c = 0; c < ch; c++) {
      int lo = 0, hi = 0, j;
      const int q = vamd_quant_limit(ctx, vb->W, c, &lo, &hi, NULL);
      const int *v = iwork + ((long)k * ch + c) * (n / 2);
      for (j = lo; j < hi; j++)
        if (v[j] > q || v[j] < -q) return vamd_domain_verdict(vb, VAMD_EDOMAIN, ampmax_out);
    }
  }
  vbi->ampmax = ampmax_out; /* lib/mapping0.c:576 */
  for (k = 0; k < nk; k++) {
    ret = vamd_write_packet(vb, managed ? k : PACKETBLOBS / 2, posts + k * ch * VAMD_POSTS_STRIDE, post_valid + k * ch,
                            iwork + k * ch * (n / 2), nonzero + k * ch, scratch);
    if (ret) return ret;
  }
  return 0;
}

const vorbis_func_mapping mapping0_exportbundle = {&mapping0_pack, &mapping0_unpack, &mapping0_free_info,
                                                   &mapping0_forward_vamd, &mapping0_inverse};
