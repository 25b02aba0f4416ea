/* oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin C entry points over the UNMODIFIED reference sources, which
 * oracle/Makefile compiles in place from /root/reference/lib into
 * oracle/_ref/libvorbis_ref.so.  Three jobs:
 *
 *  1. ref_open()/ref_pack_setup(): run the reference's own libvorbisenc +
 *     vorbis_analysis_init() and serialise the derived lookups with the
 *     reference-side packer (integration/vamd_pack_setup.c).
 *  2. ref_tap_block(): re-state mapping0_forward's VBR call sequence
 *     (reference lib/mapping0.c:230-696) using only the reference's *extern*
 *     functions, copying every intermediate out ("taps" at the #if 0
 *     _analysis_output sites), and cross-check the resulting
 *     packet bytes against the real vorbis_analysis() on the same block.
 *  3. ref_encode_stream(): the application loop of
 *     examples/encoder_example.c:179-236 (minus libogg framing) recording the
 *     genuine per-block (lW,W,nW,blocktype,ampmax_in) sequence and packets.
 *
 * Plain C types only so Python ctypes / the C tests can call it.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <time.h>
#include "vorbis/codec.h"
#include "vorbis/vorbisenc.h"
#include "codec_internal.h"
#include "registry.h"
#include "window.h"
#include "mdct.h"
#include "smallft.h"
#include "psy.h"
#include "envelope.h"
#include "scales.h"
#include "misc.h"

extern long vamd_pack_setup(vorbis_dsp_state *vd, void *dst, long cap);

typedef struct ref_enc {
  vorbis_info vi;
  vorbis_dsp_state vd;
  vorbis_block vb;    /* used for tap / real single-block analysis */
  int channels;
  float quality;
} ref_enc;

ref_enc *ref_open(int channels, long rate, float quality) {
  ref_enc *e = (ref_enc *)calloc(1, sizeof(*e));
  if (!e) return NULL;
  vorbis_info_init(&e->vi);
  if (vorbis_encode_init_vbr(&e->vi, channels, rate, quality)) {
    vorbis_info_clear(&e->vi);
    free(e);
    return NULL;
  }
  vorbis_analysis_init(&e->vd, &e->vi);
  vorbis_block_init(&e->vd, &e->vb);
  e->channels = channels;
  e->quality = quality;
  return e;
}

/* a bitrate-managed encoder (vorbis_encode_init: ABR/CBR): mapping0_forward then builds all
 * PACKETBLOBS candidate packets per block (lib/mapping0.c:507-573,596-687) */
ref_enc *ref_open_managed(int channels, long rate, long max_bitrate, long nominal_bitrate, long min_bitrate) {
  ref_enc *e = (ref_enc *)calloc(1, sizeof(*e));
  if (!e) return NULL;
  vorbis_info_init(&e->vi);
  if (vorbis_encode_init(&e->vi, channels, rate, max_bitrate, nominal_bitrate, min_bitrate)) {
    vorbis_info_clear(&e->vi);
    free(e);
    return NULL;
  }
  vorbis_analysis_init(&e->vd, &e->vi);
  vorbis_block_init(&e->vd, &e->vb);
  e->channels = channels;
  e->quality = -999.f;
  return e;
}

/* VBR with channel coupling switched off (vorbis_encode_ctl OV_ECTL_COUPLING_SET = 0): stereo as two
 * independent channels, residue type 1 over both */
ref_enc *ref_open_uncoupled(int channels, long rate, float quality) {
  ref_enc *e = (ref_enc *)calloc(1, sizeof(*e));
  int zero = 0;
  if (!e) return NULL;
  vorbis_info_init(&e->vi);
  if (vorbis_encode_setup_vbr(&e->vi, channels, rate, quality) ||
      vorbis_encode_ctl(&e->vi, OV_ECTL_COUPLING_SET, &zero) || vorbis_encode_setup_init(&e->vi)) {
    vorbis_info_clear(&e->vi);
    free(e);
    return NULL;
  }
  vorbis_analysis_init(&e->vd, &e->vi);
  vorbis_block_init(&e->vd, &e->vb);
  e->channels = channels;
  e->quality = quality;
  return e;
}

/* VBR with the other vorbis_encode_ctl() settings that reach the analysis path (lib/vorbisenc.c:1167-1191):
 * OV_ECTL_LOWPASS_SET (kHz; moves floor1's n / the sliding lowpass, lib/vorbisenc.c:529,866-880) and OV_ECTL_IBLOCK_SET
 * (the impulse blocks' noise tuning, :1183-1190, :797-812).  lowpass_khz <= 0 / iblock > 0: that one is left alone. */
ref_enc *ref_open_ctl(int channels, long rate, float quality, double lowpass_khz, double iblock) {
  ref_enc *e = (ref_enc *)calloc(1, sizeof(*e));
  int bad;
  if (!e) return NULL;
  vorbis_info_init(&e->vi);
  bad = vorbis_encode_setup_vbr(&e->vi, channels, rate, quality);
  if (!bad && lowpass_khz > 0.) bad = vorbis_encode_ctl(&e->vi, OV_ECTL_LOWPASS_SET, &lowpass_khz);
  if (!bad && iblock <= 0.) bad = vorbis_encode_ctl(&e->vi, OV_ECTL_IBLOCK_SET, &iblock);
  if (!bad) bad = vorbis_encode_setup_init(&e->vi);
  if (bad) {
    vorbis_info_clear(&e->vi);
    free(e);
    return NULL;
  }
  vorbis_analysis_init(&e->vd, &e->vi);
  vorbis_block_init(&e->vd, &e->vb);
  e->channels = channels;
  e->quality = quality;
  return e;
}

int ref_is_managed(ref_enc *e) { return vorbis_bitrate_managed(&e->vb) ? 1 : 0; }

void ref_close(ref_enc *e) {
  if (!e) return;
  vorbis_block_clear(&e->vb);
  vorbis_dsp_clear(&e->vd);
  vorbis_info_clear(&e->vi);
  free(e);
}

long ref_pack_setup(ref_enc *e, void *dst, long cap) { return vamd_pack_setup(&e->vd, dst, cap); }

int ref_blocksize(ref_enc *e, int W) {
  codec_setup_info *ci = (codec_setup_info *)e->vi.codec_setup;
  return (int)ci->blocksizes[W];
}

int ref_floor_posts(ref_enc *e, int W) {
  codec_setup_info *ci = (codec_setup_info *)e->vi.codec_setup;
  private_state *b = (private_state *)e->vd.backend_state;
  vorbis_info_mapping0 *info = (vorbis_info_mapping0 *)ci->map_param[W];
  return ((vorbis_look_floor1 *)b->flr[info->floorsubmap[0]])->posts;
}

/* ---- single-function taps (unit-level parity) ---------------------------- */

void ref_apply_window(ref_enc *e, float *d, int lW, int W, int nW) {
  codec_setup_info *ci = (codec_setup_info *)e->vi.codec_setup;
  private_state *b = (private_state *)e->vd.backend_state;
  _vorbis_apply_window(d, b->window, ci->blocksizes, lW, W, nW);
}

void ref_mdct_forward(ref_enc *e, int W, const float *in, float *out) {
  private_state *b = (private_state *)e->vd.backend_state;
  int n = ref_blocksize(e, W);
  float *tmp = (float *)malloc(sizeof(float) * n);
  memcpy(tmp, in, sizeof(float) * n);
  mdct_forward((mdct_lookup *)b->transform[W][0], tmp, out);
  free(tmp);
}

void ref_drft_forward(ref_enc *e, int W, float *data) {
  private_state *b = (private_state *)e->vd.backend_state;
  drft_forward(&b->fft_look[W], data);
}

void ref_noisemask(ref_enc *e, int psy, const float *logmdct, float *noise) {
  private_state *b = (private_state *)e->vd.backend_state;
  int n = b->psy[psy].n;
  float *tmp = (float *)malloc(sizeof(float) * n);
  memcpy(tmp, logmdct, sizeof(float) * n);
  _vp_noisemask(b->psy + psy, tmp, noise);
  free(tmp);
}

void ref_tonemask(ref_enc *e, int psy, const float *logfft, float *tone, float global_ampmax,
                  float local_ampmax) {
  private_state *b = (private_state *)e->vd.backend_state;
  int n = b->psy[psy].n;
  float *tmp = (float *)malloc(sizeof(float) * n);
  memcpy(tmp, logfft, sizeof(float) * n);
  _vp_tonemask(b->psy + psy, tmp, tone, global_ampmax, local_ampmax);
  free(tmp);
}

float ref_ampmax_decay(ref_enc *e, float amp, int W) {
  long save = e->vd.W;
  float r;
  e->vd.W = W;
  r = _vp_ampmax_decay(amp, &e->vd);
  e->vd.W = save;
  return r;
}

/* ---- whole-block taps ----------------------------------------------------- */

typedef struct ref_taps {
  /* all optional (NULL = skip); float/int arrays are [ch][...] contiguous */
  float *windowed;     /* [ch][n]   after _vorbis_apply_window */
  float *mdct_raw;     /* [ch][n/2] mdct_forward output (pre AoTuV-M1) */
  float *fft_packed;   /* [ch][n]   drft_forward output, FFTPACK order */
  float *logfft;       /* [ch][n/2] */
  float *logmdct;      /* [ch][n/2] */
  float *noise;        /* [ch][n/2] */
  float *tone;         /* [ch][n/2] */
  float *logmask;      /* [ch][n/2] after _vp_offset_and_mix(select 1) */
  float *mdct;         /* [ch][n/2] post-M1 spectrum (what gets quantised) */
  int *posts;          /* [ch][65]  floor1_fit output (bit 15 = unused flag) */
  int *post_valid;     /* [ch]      0 when floor1_fit returned NULL */
  int *ilogmask;       /* [ch][n/2] integer floor curve from floor1_encode */
  int *iwork;          /* [ch][n/2] quantised + coupled residue */
  int *nonzero;        /* [ch]      after the coupling fix-up */
  float *local_ampmax; /* [ch] */
  float *ampmax_out;   /* [1] */
  unsigned char *packet; /* packet bytes of the tap run */
  long packet_cap;
  long packet_bytes;   /* out */
  int packet_matches_real; /* out: 1 when the real vorbis_analysis() produced identical bytes+ampmax */
  /* residue back-end (submap 0): what res*_class decided and, in call order, every codebook entry
     res*_forward emitted through vorbis_book_encode() except the phrase-book words */
  int *res_class;            /* [res_class_cap] the partition classes, submap after submap (see tap_core) */
  long res_class_cap;
  long res_partvals;         /* out: partitions classified (0 when the class function returned NULL) */
  unsigned short *res_entries;
  long res_entries_cap;
  long res_count;            /* out */
} ref_taps;

/* managed-mode taps: everything that exists once per candidate packet (blob) */
typedef struct ref_mtaps {
  int *posts;        /* [PACKETBLOBS][ch][65] floor_posts[i][k]; all-zero + post_valid 0 where NULL */
  int *post_valid;   /* [PACKETBLOBS][ch] */
  int *ilogmask;     /* [PACKETBLOBS][ch][n/2] */
  int *iwork;        /* [PACKETBLOBS][ch][n/2] */
  int *nonzero;      /* [PACKETBLOBS][ch] */
  unsigned char *packets; /* the PACKETBLOBS candidate packets back to back */
  long packets_cap;
  long packet_bytes[PACKETBLOBS]; /* out */
  int packets_match_real;         /* out */
} ref_mtaps;

/* lib/res0.c is compiled with -Dvorbis_book_encode=ref_tap_book_encode (oracle/Makefile): the
 * reference source is untouched, its calls to the bit-writer just pass through here first */
#undef vorbis_book_encode
extern int vorbis_book_encode(codebook *book, int a, oggpack_buffer *b);
static struct {
  int armed;
  const codebook *skip; /* the phrase book */
  unsigned short *out;
  long cap, count;
} res_tap;
int ref_tap_book_encode(codebook *book, int a, oggpack_buffer *b) {
  if (res_tap.armed && book != res_tap.skip) {
    if (res_tap.out && res_tap.count < res_tap.cap) res_tap.out[res_tap.count] = (unsigned short)a;
    res_tap.count++;
  }
  return vorbis_book_encode(book, a, b);
}

static void load_block(ref_enc *e, const float *pcm, int lW, int W, int nW, int blocktype,
                       float ampmax_in) {
  vorbis_block *vb = &e->vb;
  vorbis_block_internal *vbi = (vorbis_block_internal *)vb->internal;
  int n = ref_blocksize(e, W), i;
  /* what vorbis_analysis_blockout does to hand a block over, lib/block.c:592-643 */
  _vorbis_block_ripcord(vb);
  vb->lW = lW;
  vb->W = W;
  vb->nW = nW;
  vbi->blocktype = blocktype;
  vb->vd = &e->vd;
  vb->pcmend = n;
  vb->eofflag = 0;
  vbi->ampmax = ampmax_in;
  vb->pcm = (float **)_vorbis_block_alloc(vb, sizeof(*vb->pcm) * e->channels);
  for (i = 0; i < e->channels; i++) {
    vb->pcm[i] = (float *)_vorbis_block_alloc(vb, n * sizeof(float));
    memcpy(vb->pcm[i], pcm + (size_t)i * n, n * sizeof(float));
  }
}

/* the real thing on one pre-cut block; returns packet bytes (or <0) */
long ref_real_block(ref_enc *e, const float *pcm, int lW, int W, int nW, int blocktype,
                    float ampmax_in, unsigned char *pkt, long cap, float *ampmax_out) {
  ogg_packet op;
  int ret;
  load_block(e, pcm, lW, W, nW, blocktype, ampmax_in);
  ret = vorbis_analysis(&e->vb, &op);
  if (ret) return ret;
  if (ampmax_out) *ampmax_out = ((vorbis_block_internal *)e->vb.internal)->ampmax;
  if (pkt) {
    if (op.bytes > cap) return -1;
    memcpy(pkt, op.packet, op.bytes);
  }
  return op.bytes;
}

/* mapping0_forward restated over the reference's extern functions, VBR branch only */
static int tap_core(ref_enc *e, const float *pcm, int lW, int W, int nW, int blocktype,
                    float ampmax_in, ref_taps *t, int with_residue, ref_mtaps *m) {
  vorbis_block *vb = &e->vb;
  vorbis_info *vi = &e->vi;
  codec_setup_info *ci = (codec_setup_info *)vi->codec_setup;
  private_state *b = (private_state *)e->vd.backend_state;
  vorbis_block_internal *vbi = (vorbis_block_internal *)vb->internal;
  int ch = vi->channels, n = (int)ci->blocksizes[W], n2 = n / 2;
  int i, j, k = PACKETBLOBS / 2;
  vorbis_info_mapping0 *info = (vorbis_info_mapping0 *)ci->map_param[W];
  vorbis_look_psy *psy_look = b->psy + blocktype + (W ? 2 : 0);
  float global_ampmax = ampmax_in;
  float local_ampmax[8];
  int nonzero[8];
  float *gmdct[8];
  int *iwork[8];
  int *posts[8];
  int *mposts[8][PACKETBLOBS];
  float *noise, *tone;
  oggpack_buffer *opb;
  const int managed = vorbis_bitrate_managed(vb) ? 1 : 0;

  if ((managed && !m) || (!managed && m) || ch > 8) return OV_EIMPL;

  load_block(e, pcm, lW, W, nW, blocktype, ampmax_in);
  for (i = 0; i < PACKETBLOBS; i++) oggpack_reset(vbi->packetblob[i]);
  vb->mode = W;
  opb = vbi->packetblob[k];

  for (i = 0; i < ch; i++) {
    float scale = 4.f / n;
    float scale_dB;
    float *p = vb->pcm[i];
    float *logfft = p;
    iwork[i] = (int *)_vorbis_block_alloc(vb, n2 * sizeof(int));
    gmdct[i] = (float *)_vorbis_block_alloc(vb, n2 * sizeof(float));
    scale_dB = todB(&scale) + .345;
    _vorbis_apply_window(p, b->window, ci->blocksizes, lW, W, nW);
    if (t->windowed) memcpy(t->windowed + (size_t)i * n, p, n * sizeof(float));
    mdct_forward((mdct_lookup *)b->transform[W][0], p, gmdct[i]);
    if (t->mdct_raw) memcpy(t->mdct_raw + (size_t)i * n2, gmdct[i], n2 * sizeof(float));
    drft_forward(&b->fft_look[W], p);
    if (t->fft_packed) memcpy(t->fft_packed + (size_t)i * n, p, n * sizeof(float));
    logfft[0] = scale_dB + todB(p) + .345;
    local_ampmax[i] = logfft[0];
    for (j = 1; j < n - 1; j += 2) {
      float temp = p[j] * p[j] + p[j + 1] * p[j + 1];
      temp = logfft[(j + 1) >> 1] = scale_dB + .5f * todB(&temp) + .345;
      if (temp > local_ampmax[i]) local_ampmax[i] = temp;
    }
    if (local_ampmax[i] > 0.f) local_ampmax[i] = 0.f;
    if (local_ampmax[i] > global_ampmax) global_ampmax = local_ampmax[i];
    if (t->logfft) memcpy(t->logfft + (size_t)i * n2, logfft, n2 * sizeof(float));
    if (t->local_ampmax) t->local_ampmax[i] = local_ampmax[i];
  }

  noise = (float *)_vorbis_block_alloc(vb, n2 * sizeof(float));
  tone = (float *)_vorbis_block_alloc(vb, n2 * sizeof(float));
  for (i = 0; i < ch; i++) {
    int submap = info->chmuxlist[i];
    float *mdct = gmdct[i];
    float *logfft = vb->pcm[i];
    float *logmdct = logfft + n2;
    float *logmask = logfft;
    for (j = 0; j < n2; j++) logmdct[j] = todB(mdct + j) + .345;
    if (t->logmdct) memcpy(t->logmdct + (size_t)i * n2, logmdct, n2 * sizeof(float));
    _vp_noisemask(psy_look, logmdct, noise);
    if (t->noise) memcpy(t->noise + (size_t)i * n2, noise, n2 * sizeof(float));
    _vp_tonemask(psy_look, logfft, tone, global_ampmax, local_ampmax[i]);
    if (t->tone) memcpy(t->tone + (size_t)i * n2, tone, n2 * sizeof(float));
    _vp_offset_and_mix(psy_look, noise, tone, 1, logmask, mdct, logmdct);
    if (t->logmask) memcpy(t->logmask + (size_t)i * n2, logmask, n2 * sizeof(float));
    if (t->mdct) memcpy(t->mdct + (size_t)i * n2, mdct, n2 * sizeof(float));
    if (ci->floor_type[info->floorsubmap[submap]] != 1) return -1;
    posts[i] = floor1_fit(vb, (vorbis_look_floor1 *)b->flr[info->floorsubmap[submap]], logmdct, logmask);
    if (t->post_valid) t->post_valid[i] = posts[i] ? 1 : 0;
    if (t->posts) {
      int np = ((vorbis_look_floor1 *)b->flr[info->floorsubmap[submap]])->posts;
      memset(t->posts + (size_t)i * 65, 0, 65 * sizeof(int));
      if (posts[i]) memcpy(t->posts + (size_t)i * 65, posts[i], np * sizeof(int));
    }
    if (managed) { /* lib/mapping0.c:507-573 */
      vorbis_look_floor1 *fl = (vorbis_look_floor1 *)b->flr[info->floorsubmap[submap]];
      int kk;
      for (kk = 0; kk < PACKETBLOBS; kk++) mposts[i][kk] = NULL;
      mposts[i][PACKETBLOBS / 2] = posts[i];
      if (posts[i]) {
        _vp_offset_and_mix(psy_look, noise, tone, 2, logmask, mdct, logmdct);
        mposts[i][PACKETBLOBS - 1] = floor1_fit(vb, fl, logmdct, logmask);
        _vp_offset_and_mix(psy_look, noise, tone, 0, logmask, mdct, logmdct);
        mposts[i][0] = floor1_fit(vb, fl, logmdct, logmask);
        for (kk = 1; kk < PACKETBLOBS / 2; kk++)
          mposts[i][kk] = floor1_interpolate_fit(vb, fl, mposts[i][0], mposts[i][PACKETBLOBS / 2],
                                                 kk * 65536 / (PACKETBLOBS / 2));
        for (kk = PACKETBLOBS / 2 + 1; kk < PACKETBLOBS - 1; kk++)
          mposts[i][kk] = floor1_interpolate_fit(vb, fl, mposts[i][PACKETBLOBS / 2], mposts[i][PACKETBLOBS - 1],
                                                 (kk - PACKETBLOBS / 2) * 65536 / (PACKETBLOBS / 2));
      }
      for (kk = 0; kk < PACKETBLOBS; kk++) {
        if (m->post_valid) m->post_valid[kk * ch + i] = mposts[i][kk] ? 1 : 0;
        if (m->posts) {
          int *dst = m->posts + ((size_t)kk * ch + i) * 65;
          memset(dst, 0, 65 * sizeof(int));
          if (mposts[i][kk]) memcpy(dst, mposts[i][kk], fl->posts * sizeof(int));
        }
      }
    }
  }
  vbi->ampmax = global_ampmax;
  if (t->ampmax_out) *t->ampmax_out = global_ampmax;

  if (managed) { /* lib/mapping0.c:596-687, every candidate packet */
    int **couple_bundle = (int **)alloca(sizeof(*couple_bundle) * ch);
    int *zerobundle = (int *)alloca(sizeof(*zerobundle) * ch);
    long used = 0;
    int kk;
    for (kk = 0; kk < PACKETBLOBS; kk++) {
      opb = vbi->packetblob[kk];
      oggpack_write(opb, 0, 1);
      oggpack_write(opb, W, b->modebits);
      if (W) {
        oggpack_write(opb, lW, 1);
        oggpack_write(opb, nW, 1);
      }
      for (i = 0; i < ch; i++) {
        int submap = info->chmuxlist[i];
        nonzero[i] = floor1_encode(opb, vb, (vorbis_look_floor1 *)b->flr[info->floorsubmap[submap]], mposts[i][kk],
                                   iwork[i]);
        if (m->ilogmask) memcpy(m->ilogmask + ((size_t)kk * ch + i) * n2, iwork[i], n2 * sizeof(int));
      }
      _vp_couple_quantize_normalize(kk, &ci->psy_g_param, psy_look, info, gmdct, iwork, nonzero,
                                    ci->psy_g_param.sliding_lowpass[W][kk], ch);
      for (i = 0; i < ch; i++) {
        if (m->iwork) memcpy(m->iwork + ((size_t)kk * ch + i) * n2, iwork[i], n2 * sizeof(int));
        if (m->nonzero) m->nonzero[kk * ch + i] = nonzero[i];
      }
      for (i = 0; i < info->submaps; i++) {
        int ch_in_bundle = 0;
        long **classifications;
        int resnum = info->residuesubmap[i];
        for (j = 0; j < ch; j++)
          if (info->chmuxlist[j] == i) {
            zerobundle[ch_in_bundle] = nonzero[j] ? 1 : 0;
            couple_bundle[ch_in_bundle++] = iwork[j];
          }
        classifications = _residue_P[ci->residue_type[resnum]]->class(vb, b->residue[resnum], couple_bundle,
                                                                     zerobundle, ch_in_bundle);
        ch_in_bundle = 0;
        for (j = 0; j < ch; j++)
          if (info->chmuxlist[j] == i) couple_bundle[ch_in_bundle++] = iwork[j];
        _residue_P[ci->residue_type[resnum]]->forward(opb, vb, b->residue[resnum], couple_bundle, zerobundle,
                                                       ch_in_bundle, classifications, i);
      }
      m->packet_bytes[kk] = oggpack_bytes(opb);
      if (m->packets && used + m->packet_bytes[kk] <= m->packets_cap)
        memcpy(m->packets + used, oggpack_get_buffer(opb), m->packet_bytes[kk]);
      used += m->packet_bytes[kk];
    }
    t->packet_bytes = 0;
    return 0;
  }

  oggpack_write(opb, 0, 1);
  oggpack_write(opb, W, b->modebits);
  if (W) {
    oggpack_write(opb, lW, 1);
    oggpack_write(opb, nW, 1);
  }
  for (i = 0; i < ch; i++) {
    int submap = info->chmuxlist[i];
    nonzero[i] = floor1_encode(opb, vb, (vorbis_look_floor1 *)b->flr[info->floorsubmap[submap]],
                               posts[i], iwork[i]);
    if (t->ilogmask) memcpy(t->ilogmask + (size_t)i * n2, iwork[i], n2 * sizeof(int));
  }
  _vp_couple_quantize_normalize(k, &ci->psy_g_param, psy_look, info, gmdct, iwork, nonzero,
                                ci->psy_g_param.sliding_lowpass[W][k], ch);
  for (i = 0; i < ch; i++) {
    if (t->iwork) memcpy(t->iwork + (size_t)i * n2, iwork[i], n2 * sizeof(int));
    if (t->nonzero) t->nonzero[i] = nonzero[i];
  }
  if (with_residue) {
    int **couple_bundle = (int **)alloca(sizeof(*couple_bundle) * ch);
    int *zerobundle = (int *)alloca(sizeof(*zerobundle) * ch);
    for (i = 0; i < info->submaps; i++) {
      int ch_in_bundle = 0;
      long **classifications;
      int resnum = info->residuesubmap[i];
      for (j = 0; j < ch; j++)
        if (info->chmuxlist[j] == i) {
          zerobundle[ch_in_bundle] = nonzero[j] ? 1 : 0;
          couple_bundle[ch_in_bundle++] = iwork[j];
        }
      classifications = _residue_P[ci->residue_type[resnum]]->class(vb, b->residue[resnum], couple_bundle,
                                                                   zerobundle, ch_in_bundle);
      {
        /* classes of every submap one after the other; a type-1 residue classifies each coded channel
           (partition-major, channel-minor: the order _01forward walks them in) */
        vorbis_info_residue0 *ri = (vorbis_info_residue0 *)ci->residue_param[resnum];
        long pv = (ri->end - ri->begin) / ri->grouping, p;
        int streams = 1, q;
        if (ci->residue_type[resnum] != 2) {
          streams = 0;
          for (q = 0; q < ch_in_bundle; q++) streams += zerobundle[q] ? 1 : 0;
        }
        if (i == 0) {
          t->res_partvals = 0;
          res_tap.out = t->res_entries;
          res_tap.cap = t->res_entries_cap;
          res_tap.count = 0;
        }
        if (classifications)
          for (p = 0; p < pv; p++)
            for (q = 0; q < streams; q++) {
              if (t->res_class && t->res_partvals < t->res_class_cap) t->res_class[t->res_partvals] = (int)classifications[q][p];
              t->res_partvals++;
            }
        res_tap.armed = 1;
        res_tap.skip = ci->fullbooks + ri->groupbook;
      }
      ch_in_bundle = 0;
      for (j = 0; j < ch; j++)
        if (info->chmuxlist[j] == i) couple_bundle[ch_in_bundle++] = iwork[j];
      _residue_P[ci->residue_type[resnum]]->forward(opb, vb, b->residue[resnum], couple_bundle, zerobundle,
                                                     ch_in_bundle, classifications, i);
      res_tap.armed = 0;
      t->res_count = res_tap.count;
    }
  }
  t->packet_bytes = oggpack_bytes(opb);
  if (t->packet && t->packet_cap >= t->packet_bytes)
    memcpy(t->packet, oggpack_get_buffer(opb), t->packet_bytes);
  return 0;
}

int ref_tap_block(ref_enc *e, const float *pcm, int lW, int W, int nW, int blocktype,
                  float ampmax_in, ref_taps *t) {
  /* run the real analysis first so its packet can be compared */
  unsigned char *realpkt = (unsigned char *)malloc(1 << 17);
  float real_ampmax = 0.f, tap_ampmax = 0.f;
  float *user_ampmax = t->ampmax_out;
  long realbytes = ref_real_block(e, pcm, lW, W, nW, blocktype, ampmax_in, realpkt, 1 << 17, &real_ampmax);
  int ret;
  if (realbytes < 0) { free(realpkt); return (int)realbytes; }
  t->ampmax_out = &tap_ampmax;
  ret = tap_core(e, pcm, lW, W, nW, blocktype, ampmax_in, t, 1, NULL);
  t->ampmax_out = user_ampmax;
  if (user_ampmax) *user_ampmax = tap_ampmax;
  if (ret == 0) {
    vorbis_block_internal *vbi = (vorbis_block_internal *)e->vb.internal;
    oggpack_buffer *opb = vbi->packetblob[PACKETBLOBS / 2];
    t->packet_matches_real = (t->packet_bytes == realbytes && real_ampmax == tap_ampmax &&
                              memcmp(oggpack_get_buffer(opb), realpkt, realbytes) == 0);
  }
  free(realpkt);
  return ret;
}

/* ---- the application loop ------------------------------------------------- */

typedef struct ref_block_rec {
  int lW, W, nW, blocktype;
  float ampmax_in, ampmax_out;
  long pcm_offset;     /* offset (floats) of this block's [ch][n] PCM in pcm_out */
  long packet_offset;  /* offset of this block's packet in packets_out */
  long packet_bytes;
  long granulepos;     /* ogg_packet.granulepos as vorbis_analysis() / vorbis_bitrate_flushpacket() set it */
  long eos;            /* ogg_packet.e_o_s */
} ref_block_rec;

/* Managed-mode counterpart of ref_tap_block: runs the real vorbis_analysis(vb, NULL) first (the
 * only legal call in managed mode, lib/analysis.c:50-53) and keeps its PACKETBLOBS candidate
 * packets, then the tap restatement, and compares all of them byte for byte. */
int ref_tap_block_managed(ref_enc *e, const float *pcm, int lW, int W, int nW, int blocktype, float ampmax_in,
                          ref_taps *t, ref_mtaps *m) {
  vorbis_block_internal *vbi = (vorbis_block_internal *)e->vb.internal;
  unsigned char *real = (unsigned char *)malloc(1 << 20);
  long realbytes[PACKETBLOBS], used = 0, at = 0;
  float real_ampmax, tap_ampmax = 0.f, *user_ampmax = t->ampmax_out;
  int ret, k, same = 1;
  if (!ref_is_managed(e)) { free(real); return OV_EINVAL; }
  load_block(e, pcm, lW, W, nW, blocktype, ampmax_in);
  ret = vorbis_analysis(&e->vb, NULL);
  if (ret) { free(real); return ret; }
  real_ampmax = vbi->ampmax;
  for (k = 0; k < PACKETBLOBS; k++) {
    realbytes[k] = oggpack_bytes(vbi->packetblob[k]);
    if (used + realbytes[k] > (1 << 20)) { free(real); return -1; }
    memcpy(real + used, oggpack_get_buffer(vbi->packetblob[k]), realbytes[k]);
    used += realbytes[k];
  }
  t->ampmax_out = &tap_ampmax;
  ret = tap_core(e, pcm, lW, W, nW, blocktype, ampmax_in, t, 1, m);
  t->ampmax_out = user_ampmax;
  if (user_ampmax) *user_ampmax = tap_ampmax;
  if (ret == 0) {
    for (k = 0; k < PACKETBLOBS; k++) {
      if (m->packet_bytes[k] != realbytes[k] ||
          memcmp(oggpack_get_buffer(vbi->packetblob[k]), real + at, realbytes[k]) != 0)
        same = 0;
      at += realbytes[k];
    }
    m->packets_match_real = same && real_ampmax == tap_ampmax;
  }
  free(real);
  return ret;
}

/* Feed planar PCM pcm[ch][frames] in 1024-frame chunks through
 * vorbis_analysis_buffer/_wrote/_blockout/vorbis_analysis exactly as
 * examples/encoder_example.c:179-236 does; record each block.  Returns the
 * number of blocks (may exceed max_blocks: then only the first max_blocks are
 * recorded), or <0 on error.  The encoder state is consumed: open a fresh
 * ref_enc per stream. */
/* `write_frames`: samples handed over per vorbis_analysis_wrote() call -- the example's READ is 1024
 * (examples/encoder_example.c:38), but the API takes any amount (lib/block.c:390,470).  `tolerate`: a
 * vorbis_analysis() that fails does not end the run; the block is recorded with packet_bytes = the error code
 * (negative) and no packet, and the loop goes on as an application that ignores return codes would. */
/* ref_stream_set_drain(k): after each write the loop below pulls at most k blocks (0 = all of them, the default), so that
 * blocks pile up in the encoder's buffer while more samples arrive -- an application is free to do that (the reference's
 * own output depends on the write pattern only in the stream's first block, whose pre-extrapolation takes in whatever
 * is buffered when it is cut, lib/block.c:420-465,519-524). */
static long ref_drain_limit = 0;
void ref_stream_set_drain(long k) { ref_drain_limit = k; }
/* ref_stream_set_jitter(seed): seed != 0 makes every write a pseudo-random 1 .. write_frames samples and every pull a
 * pseudo-random 0 .. drain blocks (the same sequence for the same seed: the reference and the hybrid are driven alike) */
static unsigned long ref_jitter_seed = 0;
void ref_stream_set_jitter(unsigned long seed) { ref_jitter_seed = seed; }
static unsigned long ref_jitter_next(unsigned long *st) {
  *st = *st * 6364136223846793005UL + 1442695040888963407UL;
  return *st >> 33;
}
long ref_encode_stream_ex(ref_enc *e, const float *pcm, long frames, long write_frames, int tolerate, ref_block_rec *recs,
                          long max_blocks, float *pcm_out, long pcm_cap, unsigned char *packets_out, long packets_cap) {
  vorbis_block vb;
  long fed = 0, nblocks = 0, pcm_used = 0, pkt_used = 0;
  int ch = e->channels, eos = 0, i;
  unsigned long jit = ref_jitter_seed;
  if (write_frames < 1) write_frames = 1024;
  vorbis_block_init(&e->vd, &vb);
  while (!eos) {
    long chunk = frames - fed, this_write = write_frames, this_drain = ref_drain_limit;
    if (ref_jitter_seed) {
      this_write = 1 + (long)(ref_jitter_next(&jit) % (unsigned long)write_frames);
      if (ref_drain_limit > 0) this_drain = (long)(ref_jitter_next(&jit) % (unsigned long)(ref_drain_limit + 1));
    }
    if (chunk > this_write) chunk = this_write;
    if (chunk > 0) {
      float **buf = vorbis_analysis_buffer(&e->vd, (int)write_frames);
      for (i = 0; i < ch; i++) memcpy(buf[i], pcm + (size_t)i * frames + fed, chunk * sizeof(float));
      vorbis_analysis_wrote(&e->vd, (int)chunk);
      fed += chunk;
    } else {
      vorbis_analysis_wrote(&e->vd, 0);
    }
    long pulled = 0;
    while ((ref_drain_limit <= 0 || chunk <= 0 || pulled < this_drain) && vorbis_analysis_blockout(&e->vd, &vb) == 1) {
      ogg_packet op;
      vorbis_block_internal *vbi = (vorbis_block_internal *)vb.internal;
      int n = vb.pcmend;
      float ampmax_in = vbi->ampmax;
      int ret;
      pulled++;
      if (nblocks < max_blocks && recs) {
        ref_block_rec *r = recs + nblocks;
        r->lW = (int)vb.lW; r->W = (int)vb.W; r->nW = (int)vb.nW; r->blocktype = vbi->blocktype;
        r->ampmax_in = ampmax_in;
        r->pcm_offset = -1;
        if (pcm_out && pcm_used + (long)ch * n <= pcm_cap) {
          r->pcm_offset = pcm_used;
          for (i = 0; i < ch; i++) memcpy(pcm_out + pcm_used + (size_t)i * n, vb.pcm[i], n * sizeof(float));
          pcm_used += (long)ch * n;
        }
      }
      if (vorbis_bitrate_managed(&vb)) {
        /* the general application loop, examples/encoder_example.c:211-217: the bitrate manager
           picks one of the block's candidate packets */
        ret = vorbis_analysis(&vb, NULL);
        if (!ret) ret = vorbis_bitrate_addblock(&vb);
        if (!ret && vorbis_bitrate_flushpacket(&e->vd, &op) != 1) ret = -1;
      } else {
        ret = vorbis_analysis(&vb, &op);
      }
      if (ret && !tolerate) { vorbis_block_clear(&vb); return ret; }
      if (nblocks < max_blocks && recs) {
        ref_block_rec *r = recs + nblocks;
        r->ampmax_out = vbi->ampmax;
        r->granulepos = ret ? -1 : (long)op.granulepos;
        r->eos = ret ? 0 : (long)op.e_o_s;
        r->packet_bytes = ret ? ret : op.bytes;
        r->packet_offset = -1;
        if (!ret && packets_out && pkt_used + op.bytes <= packets_cap) {
          r->packet_offset = pkt_used;
          memcpy(packets_out + pkt_used, op.packet, op.bytes);
          pkt_used += op.bytes;
        }
      }
      nblocks++;
      if (vb.eofflag) eos = 1;
    }
    if (chunk <= 0 && e->vd.eofflag == -1) eos = 1;
    if (chunk <= 0 && !eos && e->vd.eofflag == 0) eos = 1; /* defensive: nothing more will come */
  }
  vorbis_block_clear(&vb);
  return nblocks;
}
long ref_encode_stream(ref_enc *e, const float *pcm, long frames, ref_block_rec *recs, long max_blocks,
                       float *pcm_out, long pcm_cap, unsigned char *packets_out, long packets_cap) {
  return ref_encode_stream_ex(e, pcm, frames, 1024, 0, recs, max_blocks, pcm_out, pcm_cap, packets_out, packets_cap);
}

/* ---- many encoder threads, timed in C (profiles/rNN_batcher.txt) -----------------------
 * `nthreads` application threads, each with its own encoder state, each pushing the same `frames`-sample planar
 * signal through the unmodified application loop above (nothing recorded), `passes` times over (a fresh state per
 * pass).  The states of the first pass are opened before the clock starts; the clock runs from the moment every
 * thread stands at the start line until the last one is through.  Linked into libvorbis_hybrid.so the same function
 * times the GPU back-end (VAMD_BATCH in the environment: the batcher).  Python drives it with ONE call, so no
 * interpreter lock, allocation or copy of Python's sits inside the timed region (round 3's figures were taken around
 * ref_encode_stream() calls from Python threads, whose per-block post-processing serialises on the interpreter lock).
 * Returns wall seconds (< 0: an encode failed); *blocks_out = blocks encoded, cpu_out[2] = user, system seconds. */
#include <pthread.h>
#include <sys/resource.h>
typedef struct ref_tt_arg {
  int ch, passes;
  long rate, frames, blocks, write_frames;
  float q;
  const float *pcm;
  ref_enc *first;
  pthread_barrier_t *line;
  int failed;
} ref_tt_arg;
/* ref_time_set_managed(nominal bitrate): the timed encoders are bitrate-managed (ABR at that rate) instead of VBR at q; 0 = VBR */
static long ref_tt_nominal = 0;
void ref_time_set_managed(long nominal) { ref_tt_nominal = nominal; }
static ref_enc *ref_tt_open(int ch, long rate, float q) {
  return ref_tt_nominal > 0 ? ref_open_managed(ch, rate, -1, ref_tt_nominal, -1) : ref_open(ch, rate, q);
}
static void *ref_tt_run(void *v) {
  ref_tt_arg *a = (ref_tt_arg *)v;
  int p;
  pthread_barrier_wait(a->line);
  for (p = 0; p < a->passes; p++) {
    ref_enc *e = p == 0 ? a->first : ref_tt_open(a->ch, a->rate, a->q);
    long nb = e ? ref_encode_stream_ex(e, a->pcm, a->frames, a->write_frames, 0, NULL, 0, NULL, 0, NULL, 0) : -1;
    if (e) ref_close(e);
    if (nb < 0) { a->failed = 1; break; }
    a->blocks += nb;
  }
  return NULL;
}
/* write_frames: samples per vorbis_analysis_wrote() call (1024 = the example's READ) */
double ref_time_threads_w(int nthreads, int ch, long rate, float q, const float *pcm, long frames, long write_frames, int passes,
                          long *blocks_out, double *cpu_out) {
  pthread_t *th = (pthread_t *)calloc(nthreads, sizeof(*th));
  ref_tt_arg *args = (ref_tt_arg *)calloc(nthreads, sizeof(*args));
  pthread_barrier_t line;
  struct timespec t0, t1;
  struct rusage r0, r1;
  pthread_attr_t attr;
  int i, bad = 0;
  long blocks = 0;
  pthread_barrier_init(&line, NULL, nthreads + 1);
  pthread_attr_init(&attr);
  pthread_attr_setstacksize(&attr, 1 << 20); /* mapping0_forward's alloca()s are a few hundred KB at most */
  for (i = 0; i < nthreads; i++) {
    args[i].ch = ch, args[i].rate = rate, args[i].q = q, args[i].pcm = pcm, args[i].frames = frames;
    args[i].passes = passes, args[i].line = &line, args[i].write_frames = write_frames;
    args[i].first = ref_tt_open(ch, rate, q);
    pthread_create(&th[i], &attr, ref_tt_run, &args[i]);
  }
  getrusage(RUSAGE_SELF, &r0);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_barrier_wait(&line);
  for (i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  getrusage(RUSAGE_SELF, &r1);
  for (i = 0; i < nthreads; i++) blocks += args[i].blocks, bad |= args[i].failed;
  if (blocks_out) *blocks_out = blocks;
  if (cpu_out) {
    cpu_out[0] = (r1.ru_utime.tv_sec - r0.ru_utime.tv_sec) + 1e-6 * (r1.ru_utime.tv_usec - r0.ru_utime.tv_usec);
    cpu_out[1] = (r1.ru_stime.tv_sec - r0.ru_stime.tv_sec) + 1e-6 * (r1.ru_stime.tv_usec - r0.ru_stime.tv_usec);
  }
  pthread_barrier_destroy(&line);
  pthread_attr_destroy(&attr);
  free(th);
  free(args);
  return bad ? -1. : (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
double ref_time_threads(int nthreads, int ch, long rate, float q, const float *pcm, long frames, int passes,
                        long *blocks_out, double *cpu_out) {
  return ref_time_threads_w(nthreads, ch, rate, q, pcm, frames, 1024, passes, blocks_out, cpu_out);
}

/* ---- the block-switching detector ------------------------------
 * ref_envelope_feed(): append `frames` samples per channel exactly as an application does
 * (vorbis_analysis_buffer + vorbis_analysis_wrote, which also runs the start-of-stream
 * pre-extrapolation, lib/block.c:398-458), then run the reference's own
 * _ve_envelope_search() (lib/envelope.c:217-327) over everything buffered so far.
 * vorbis_analysis_blockout() is never called, so nothing is shifted out and ve->mark[j]
 * stays addressable for every step j since the start of the stream.
 * Returns the number of detector steps done so far (ve->current / ve->searchstep). */
long ref_envelope_feed(ref_enc *e, const float *pcm, long frames) {
  private_state *b = (private_state *)e->vd.backend_state;
  int i;
  if (frames > 0) {
    float **buf = vorbis_analysis_buffer(&e->vd, (int)frames);
    for (i = 0; i < e->channels; i++) memcpy(buf[i], pcm + (size_t)i * frames, frames * sizeof(float));
    if (vorbis_analysis_wrote(&e->vd, (int)frames)) return -1;
  }
  _ve_envelope_search(&e->vd);
  return b->ve->current / b->ve->searchstep;
}

#define REF_ENV_MAX_CH 8
typedef struct ref_env_state {
  int stretch;
  int ampptr[REF_ENV_MAX_CH][VE_BANDS];
  float ampbuf[REF_ENV_MAX_CH][VE_BANDS][VE_AMP];
  int nearptr[REF_ENV_MAX_CH];
  float nearDC[REF_ENV_MAX_CH][VE_NEARDC];
  float nearDC_acc[REF_ENV_MAX_CH], nearDC_partialacc[REF_ENV_MAX_CH];
} ref_env_state;

/* Copy out what the detector saw and decided: the PCM ring as it stands (pcm_seen[ch][cap],
 * returns samples per channel via *pcm_len), the marks of steps [0, nmarks) and the filter
 * state.  Returns ve->current / ve->searchstep. */
long ref_envelope_get(ref_enc *e, float *pcm_seen, long cap, long *pcm_len, int *marks, long nmarks,
                      ref_env_state *st) {
  private_state *b = (private_state *)e->vd.backend_state;
  envelope_lookup *ve = b->ve;
  int i, j;
  long n = e->vd.pcm_current;
  if (pcm_len) *pcm_len = n;
  if (pcm_seen) {
    if (n > cap) n = cap;
    for (i = 0; i < e->channels; i++) memcpy(pcm_seen + (size_t)i * cap, e->vd.pcm[i], n * sizeof(float));
  }
  if (marks)
    for (j = 0; j < nmarks; j++) marks[j] = j < ve->storage ? ve->mark[j] : -1;
  if (st) {
    memset(st, 0, sizeof(*st));
    st->stretch = ve->stretch;
    for (i = 0; i < e->channels && i < REF_ENV_MAX_CH; i++) {
      envelope_filter_state *f = ve->filter + i * VE_BANDS;
      st->nearptr[i] = f->nearptr;
      st->nearDC_acc[i] = f->nearDC_acc;
      st->nearDC_partialacc[i] = f->nearDC_partialacc;
      memcpy(st->nearDC[i], f->nearDC, sizeof(f->nearDC));
      for (j = 0; j < VE_BANDS; j++) {
        st->ampptr[i][j] = f[j].ampptr;
        memcpy(st->ampbuf[i][j], f[j].ampbuf, sizeof(f[j].ampbuf));
      }
    }
  }
  return ve->current / ve->searchstep;
}

/* ---- CPU baseline timing ---------------------------------------------------- */

/* Time `reps` passes of the real vorbis_analysis() over `nblocks` pre-cut
 * blocks pcm[nblocks][ch][n] (all (lW,W,nW)=(1,1,1), LONG, ampmax_in=-9999 -- the C3/C4
 * batch convention).  Returns seconds of wall-clock for the analysis calls. */
double ref_time_analysis(ref_enc *e, const float *pcm, long nblocks, int reps) {
  struct timespec t0, t1;
  int n = ref_blocksize(e, 1), r;
  long k;
  ogg_packet op;
  double total = 0.;
  for (r = 0; r < reps; r++)
    for (k = 0; k < nblocks; k++) {
      load_block(e, pcm + (size_t)k * e->channels * n, 1, 1, 1, BLOCKTYPE_LONG, -9999.f);
      clock_gettime(CLOCK_MONOTONIC, &t0);
      vorbis_analysis(&e->vb, &op);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      total += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    }
  return total;
}

/* Time the DSP part only: the tap sequence up to and including
 * _vp_couple_quantize_normalize (what the GPU path computes), skipping the
 * residue VQ / Huffman bit-writing that stays on the host. */
double ref_time_dsp(ref_enc *e, const float *pcm, long nblocks, int reps) {
  struct timespec t0, t1;
  int n = ref_blocksize(e, 1), r;
  long k;
  ref_taps t;
  double total = 0.;
  memset(&t, 0, sizeof(t));
  for (r = 0; r < reps; r++)
    for (k = 0; k < nblocks; k++) {
      const float *blk = pcm + (size_t)k * e->channels * n;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      tap_core(e, blk, 1, 1, 1, BLOCKTYPE_LONG, -9999.f, &t, 0, NULL);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      total += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    }
  return total;
}

/* ---- the reference's own test grid (test/test.c:30-75) ----------------------------------
 * One cell: write_vorbis_data_or_die()'s encode sequence (test/write_read.c:30-126: the whole
 * signal handed over in ONE vorbis_analysis_buffer/_wrote pair, the same samples in every channel,
 * vorbis_analysis(vb, NULL) + vorbis_bitrate_addblock + vorbis_bitrate_flushpacket) and
 * read_vorbis_data_or_die()'s decode (:130-300: three header packets into vorbis_synthesis_headerin,
 * then vorbis_synthesis / _blockin / _pcmout, channel 0 kept), with the packets handed from one to the
 * other directly instead of through libogg pages (libogg is not part of the reference tree and adds no
 * arithmetic).  Linked into libvorbis_hybrid.so the same function runs the encode on the GPU back-end.
 *   packets_out / sizes  every packet in order, the three headers first
 *   decoded              channel 0 of the decoder's output, at most `count` samples
 * Returns the number of packets (headers included), or < 0. */
long ref_matrix_case(int ch, long rate, float q, const float *data, int count, unsigned char *packets_out,
                     long cap, long *sizes, long max_packets, float *decoded, long *decoded_total) {
  vorbis_info vi, vi2;
  vorbis_comment vc, vc2;
  vorbis_dsp_state vd, vd2;
  vorbis_block vb, vb2;
  ogg_packet hdr[3], op;
  ogg_packet *ops = NULL;
  long np = 0, used = 0, k, ret = 0, read_total = 0;
  int i;

  vorbis_info_init(&vi);
  if (vorbis_encode_init_vbr(&vi, ch, rate, q)) {
    vorbis_info_clear(&vi);
    return -1;
  }
  vorbis_comment_init(&vc);
  vorbis_comment_add_tag(&vc, "ENCODER", "test/util.c");
  vorbis_analysis_init(&vd, &vi);
  vorbis_block_init(&vd, &vb);
  ops = (ogg_packet *)calloc(max_packets > 0 ? max_packets : 1, sizeof(*ops));
  if (!ops) {
    vorbis_block_clear(&vb);
    vorbis_dsp_clear(&vd);
    vorbis_comment_clear(&vc);
    vorbis_info_clear(&vi);
    return -7;
  }
  vorbis_analysis_headerout(&vd, &vc, &hdr[0], &hdr[1], &hdr[2]);
#define KEEP(P)                                                   \
  do {                                                            \
    if (np >= max_packets || used + (P).bytes > cap) {            \
      ret = -2;                                                   \
      goto done_encode;                                           \
    }                                                             \
    memcpy(packets_out + used, (P).packet, (P).bytes);            \
    ops[np] = (P);                                                \
    ops[np].packet = packets_out + used;                          \
    sizes[np++] = (P).bytes;                                      \
    used += (P).bytes;                                            \
  } while (0)
  for (i = 0; i < 3; i++) KEEP(hdr[i]);
  {
    float **buffer = vorbis_analysis_buffer(&vd, count);
    for (i = 0; i < ch; i++) memcpy(buffer[i], data, count * sizeof(float));
    vorbis_analysis_wrote(&vd, count);
    vorbis_analysis_wrote(&vd, 0);
  }
  while (vorbis_analysis_blockout(&vd, &vb) == 1) {
    if (vorbis_analysis(&vb, NULL) || vorbis_bitrate_addblock(&vb)) {
      ret = -3;
      goto done_encode;
    }
    while (vorbis_bitrate_flushpacket(&vd, &op)) KEEP(op);
  }
#undef KEEP
done_encode:
  vorbis_block_clear(&vb);
  vorbis_dsp_clear(&vd);
  vorbis_comment_clear(&vc);
  vorbis_info_clear(&vi);
  if (ret) {
    free(ops);
    return ret;
  }

  vorbis_info_init(&vi2);
  vorbis_comment_init(&vc2);
  for (k = 0; k < 3; k++)
    if (vorbis_synthesis_headerin(&vi2, &vc2, &ops[k]) < 0) ret = -4;
  if (!ret && vi2.rate != rate) ret = -5;
  if (!ret && vorbis_synthesis_init(&vd2, &vi2)) ret = -6;
  if (!ret) {
    vorbis_block_init(&vd2, &vb2);
    for (k = 3; k < np; k++) {
      float **pcm;
      int samples;
      if (vorbis_synthesis(&vb2, &ops[k]) == 0) vorbis_synthesis_blockin(&vd2, &vb2);
      while ((samples = vorbis_synthesis_pcmout(&vd2, &pcm)) > 0 && read_total < count) {
        int bout = samples < count ? samples : count;
        bout = read_total + bout > count ? count - read_total : bout;
        memcpy(decoded + read_total, pcm[0], bout * sizeof(float));
        vorbis_synthesis_read(&vd2, bout);
        read_total += bout;
      }
    }
    vorbis_block_clear(&vb2);
    vorbis_dsp_clear(&vd2);
  }
  vorbis_comment_clear(&vc2);
  vorbis_info_clear(&vi2);
  free(ops);
  *decoded_total = read_total;
  return ret ? ret : np;
}
