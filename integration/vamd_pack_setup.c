/* integration/vamd_pack_setup.c -- the reference-side half of the boundary.
 *
 * This file is what a libvorbis maintainer adds to lib/ (next to block.c): it
 * compiles against libvorbis' own internal headers and serialises the lookups
 * vorbis_analysis_init() built (reference lib/block.c:170-293,296-311) into the
 * POD blob of include/vamd_setup.h.  The GPU layer (libvorbis_amd.so) never sees
 * a libvorbis struct.  It is also compiled into oracle/_ref/libvorbis_ref.so so
 * tests and the committed setup blobs are produced by exactly this code.
 *
 *   long vamd_pack_setup(vorbis_dsp_state *vd, void *dst, long cap)
 *     returns the blob size in bytes (call with dst==NULL to size it),
 *     or a negative OV_* code: OV_EINVAL (not an analysis state),
 *     OV_EIMPL (a setup the GPU path does not cover yet).
 */
#include <string.h>
#include <stdint.h>
#include "vorbis/codec.h"
#include "codec_internal.h"
#include "registry.h"
#include "window.h"
#include "mdct.h"
#include "smallft.h"
#include "psy.h"
#include "envelope.h"
#include "codebook.h"
#include "misc.h"
#include "vamd_setup.h"

static uint32_t place(uint32_t *cursor, uint32_t bytes) {
  uint32_t at = (*cursor + 15u) & ~15u;
  *cursor = at + bytes;
  return at;
}

static void put(void *dst, uint32_t off, const void *src, uint32_t bytes) {
  if (dst) memcpy((char *)dst + off, src, bytes);
}

long vamd_pack_setup(vorbis_dsp_state *vd, void *dst, long cap) {
  vorbis_info *vi;
  codec_setup_info *ci;
  private_state *b;
  vamd_setup_header h;
  uint32_t cur = sizeof(vamd_setup_header);
  int W, p, i, j;

  if (!vd || !vd->analysisp || !vd->vi || !vd->backend_state) return OV_EINVAL;
  vi = vd->vi;
  ci = (codec_setup_info *)vi->codec_setup;
  b = (private_state *)vd->backend_state;
  /* two block sizes: four psy looks and two modes; the single-size setups (8 / 11 kHz) have two looks
     and one mode, and only ever produce W = 0 blocks (lib/block.c:572-573): their W = 1 slots are
     filled with copies so the blob keeps one shape */
  if (!ci || (ci->psys != 4 && ci->psys != 2) || ci->modes < 1) return OV_EINVAL;
  if (vi->channels < 1 || vi->channels > VAMD_MAX_CH) return OV_EIMPL;

  /* a sizing pass (dst == NULL) always runs before anything is written */
  if (dst) {
    long need = vamd_pack_setup(vd, NULL, 0);
    if (need < 0) return need;
    if (cap < need) return OV_EINVAL;
    memset(dst, 0, (size_t)need);
  }
  memset(&h, 0, sizeof(h));
  h.magic = VAMD_SETUP_MAGIC;
  h.version = VAMD_SETUP_VERSION;
  h.channels = vi->channels;
  h.rate = (int32_t)vi->rate;
  h.blocksizes[0] = (int32_t)ci->blocksizes[0];
  h.blocksizes[1] = (int32_t)ci->blocksizes[1];
  h.managed = b->bms.managed ? 1 : 0;

  for (W = 0; W < 2; W++) {
    vamd_xform_tab *x = &h.xform[W];
    mdct_lookup *m = (mdct_lookup *)b->transform[W][0];
    drft_lookup *f = &b->fft_look[W];
    int n = m->n;
    x->n = n;
    x->log2n = m->log2n;
    x->mdct_scale = m->scale;
    x->fft_nf = f->splitcache[1];
    if (x->fft_nf > 16 || f->n != n) return OV_EIMPL;
    for (i = 0; i < x->fft_nf; i++) x->fft_fac[i] = f->splitcache[2 + i];
    x->off_mdct_trig = place(&cur, (uint32_t)(n + n / 4) * 4u);
    put(dst, x->off_mdct_trig, m->trig, (uint32_t)(n + n / 4) * 4u);
    x->off_mdct_bitrev = place(&cur, (uint32_t)(n / 4) * 4u);
    put(dst, x->off_mdct_bitrev, m->bitrev, (uint32_t)(n / 4) * 4u);
    x->off_fft_wa = place(&cur, (uint32_t)(2 * n) * 4u);
    put(dst, x->off_fft_wa, f->trigcache + n, (uint32_t)(2 * n) * 4u);
    x->off_window = place(&cur, (uint32_t)(n / 2) * 4u);
    /* b->window[W] indexes vwin[] exactly as _vorbis_apply_window does
       (lib/window.c:2102-2110) */
    put(dst, x->off_window, _vorbis_window_get(b->window[W]), (uint32_t)(n / 2) * 4u);
  }

  for (p = 0; p < 4; p++) {
    vamd_psy_tab *t = &h.psy[p];
    vorbis_look_psy *l = b->psy + (ci->psys == 4 ? p : (p & 1));
    vorbis_info_psy *pi = l->vi;
    int n = l->n;
    uint32_t off;
    t->n = n;
    t->blockflag = pi->blockflag;
    t->firstoc = (int32_t)l->firstoc;
    t->shiftoc = (int32_t)l->shiftoc;
    t->eighth_octave_lines = l->eighth_octave_lines;
    t->total_octave_lines = l->total_octave_lines;
    t->m_val = l->m_val;
    t->ath_adjatt = pi->ath_adjatt;
    t->ath_maxatt = pi->ath_maxatt;
    for (i = 0; i < P_NOISECURVES; i++) t->tone_masteratt[i] = pi->tone_masteratt[i];
    t->tone_abs_limit = pi->tone_abs_limit;
    t->noisemaxsupp = pi->noisemaxsupp;
    t->noisewindowfixed = pi->noisewindowfixed;
    t->max_curve_dB = pi->max_curve_dB;
    for (i = 0; i < NOISE_COMPAND_LEVELS; i++) t->noisecompand[i] = pi->noisecompand[i];
    t->normal_p = pi->normal_p;
    t->normal_start = pi->normal_start;
    t->normal_partition = pi->normal_partition;
    t->normal_thresh = pi->normal_thresh;

    t->off_ath = place(&cur, (uint32_t)n * 4u);
    put(dst, t->off_ath, l->ath, (uint32_t)n * 4u);

    /* octave[] and bark[] are `long` in the reference; their values fit int32 */
    t->off_octave = place(&cur, (uint32_t)n * 4u);
    t->off_bark = place(&cur, (uint32_t)n * 4u);
    if (dst)
      for (i = 0; i < n; i++) {
        int32_t oc = (int32_t)l->octave[i], bk = (int32_t)l->bark[i];
        memcpy((char *)dst + t->off_octave + 4u * i, &oc, 4);
        memcpy((char *)dst + t->off_bark + 4u * i, &bk, 4);
      }

    t->off_noiseoffset = place(&cur, (uint32_t)(P_NOISECURVES * n) * 4u);
    for (i = 0; i < P_NOISECURVES; i++)
      put(dst, t->off_noiseoffset + (uint32_t)(i * n) * 4u, l->noiseoffset[i], (uint32_t)n * 4u);

    off = t->off_tonecurves = place(&cur, (uint32_t)(P_BANDS * P_LEVELS * (EHMER_MAX + 2)) * 4u);
    for (i = 0; i < P_BANDS; i++)
      for (j = 0; j < P_LEVELS; j++) {
        put(dst, off, l->tonecurves[i][j], (EHMER_MAX + 2) * 4u);
        off += (EHMER_MAX + 2) * 4u;
      }
  }

  {
    vorbis_info_psy_global *g = &ci->psy_g_param;
    h.psy_g.ampmax_att_per_sec = g->ampmax_att_per_sec;
    for (i = 0; i < PACKETBLOBS; i++) {
      h.psy_g.coupling_pointlimit[0][i] = g->coupling_pointlimit[0][i];
      h.psy_g.coupling_pointlimit[1][i] = g->coupling_pointlimit[1][i];
      h.psy_g.coupling_prepointamp[i] = g->coupling_prepointamp[i];
      h.psy_g.coupling_postpointamp[i] = g->coupling_postpointamp[i];
      h.psy_g.sliding_lowpass[0][i] = g->sliding_lowpass[0][i];
      h.psy_g.sliding_lowpass[1][i] = g->sliding_lowpass[1][i];
    }
  }

  for (W = 0; W < 2; W++) {
    /* mode number == W in every libvorbisenc setup (lib/mapping0.c:248) */
    vamd_mode_tab *m = &h.mode[W];
    vorbis_info_mapping0 *map;
    int sm;
    const int mode = W < ci->modes ? W : 0;
    if (ci->map_type[ci->mode_param[mode]->mapping] != 0) return OV_EIMPL;
    map = (vorbis_info_mapping0 *)ci->map_param[ci->mode_param[mode]->mapping];
    m->submaps = map->submaps;
    m->coupling_steps = map->coupling_steps;
    if (map->submaps < 1 || map->submaps > VAMD_MAX_SUBMAPS || map->coupling_steps > VAMD_MAX_COUPLING) return OV_EIMPL;
    for (i = 0; i < map->coupling_steps; i++) {
      m->coupling_mag[i] = map->coupling_mag[i];
      m->coupling_ang[i] = map->coupling_ang[i];
    }
    for (i = 0; i < vi->channels; i++) m->chmuxlist[i] = map->submaps > 1 ? map->chmuxlist[i] : 0;
    for (sm = 0; sm < map->submaps; sm++) {
      vamd_floor1_tab *f = &m->floor[sm];
      const int fidx = map->floorsubmap[sm];
      vorbis_look_floor1 *fl;
      vorbis_info_floor1 *fi;
      if (ci->floor_type[fidx] != 1) return OV_EIMPL; /* lib/mapping0.c:498 */
      fl = (vorbis_look_floor1 *)b->flr[fidx];
      fi = fl->vi;
      f->posts = fl->posts;
      f->look_n = fl->n;
      f->quant_q = fl->quant_q;
      f->mult = fi->mult;
      f->info_n = fi->n;
      f->maxover = fi->maxover;
      f->maxunder = fi->maxunder;
      f->maxerr = fi->maxerr;
      f->twofitweight = fi->twofitweight;
      f->twofitatten = fi->twofitatten;
      for (i = 0; i < fl->posts && i < VAMD_POSIT; i++) {
        f->postlist[i] = fi->postlist[i];
        f->sorted_index[i] = fl->sorted_index[i];
        f->forward_index[i] = fl->forward_index[i];
        f->reverse_index[i] = fl->reverse_index[i];
      }
      for (i = 0; i < fl->posts - 2 && i < VIF_POSIT; i++) {
        f->hineighbor[i] = fl->hineighbor[i];
        f->loneighbor[i] = fl->loneighbor[i];
      }
      /* what the bit-writing half of floor1_encode walks (lib/floor1.c:833-921) */
      if (fi->partitions > VAMD_FLOOR_PARTS) return OV_EIMPL;
      f->partitions = fi->partitions;
      for (i = 0; i < fi->partitions; i++) f->partitionclass[i] = fi->partitionclass[i];
      for (i = 0; i < VAMD_FLOOR_CLASSES; i++) {
        f->class_dim[i] = fi->class_dim[i];
        f->class_subs[i] = fi->class_subs[i];
        f->class_book[i] = fi->class_book[i];
        for (j = 0; j < 8; j++) f->class_subbook[i][j] = fi->class_subbook[i][j];
      }
    }
  }
  h.modebits = b->modebits;
  h.modes = ci->modes;

  {
    /* the block-switching detector's lookup (lib/envelope.c:30-74) */
    envelope_lookup *ve = b->ve;
    vorbis_info_psy_global *g = &ci->psy_g_param;
    vamd_envelope_tab *t = &h.env;
    int n;
    if (!ve) return OV_EINVAL;
    n = ve->winlength;
    if (n != ve->mdct.n || n < 64 || ve->ch != vi->channels) return OV_EIMPL;
    t->winlength = n;
    t->searchstep = ve->searchstep;
    t->log2n = ve->mdct.log2n;
    t->mdct_scale = ve->mdct.scale;
    t->minenergy = ve->minenergy;
    t->stretch_penalty = g->stretch_penalty;
    for (i = 0; i < VE_BANDS; i++) {
      t->preecho_thresh[i] = g->preecho_thresh[i];
      t->postecho_thresh[i] = g->postecho_thresh[i];
      t->band_begin[i] = ve->band[i].begin;
      t->band_end[i] = ve->band[i].end;
      t->band_total[i] = ve->band[i].total;
      if (ve->band[i].end > VAMD_VE_BANDWIN || ve->band[i].begin + ve->band[i].end > n / 4) return OV_EIMPL;
      for (j = 0; j < ve->band[i].end; j++) t->band_window[i][j] = ve->band[i].window[j];
    }
    t->off_mdct_trig = place(&cur, (uint32_t)(n + n / 4) * 4u);
    put(dst, t->off_mdct_trig, ve->mdct.trig, (uint32_t)(n + n / 4) * 4u);
    t->off_mdct_bitrev = place(&cur, (uint32_t)(n / 4) * 4u);
    put(dst, t->off_mdct_bitrev, ve->mdct.bitrev, (uint32_t)(n / 4) * 4u);
    t->off_window = place(&cur, (uint32_t)n * 4u);
    put(dst, t->off_window, ve->mdct_win, (uint32_t)n * 4u);
  }

  {
    /* residue back-ends and the codebooks they search (lib/res0.c:175-260, lib/sharedbook.c) */
    vamd_book_tab *books = NULL;
    if (ci->books < 0 || ci->books > 256) return OV_EIMPL;
    h.nbooks = ci->books;
    h.off_books = place(&cur, (uint32_t)(ci->books * sizeof(vamd_book_tab)));
    if (dst) books = (vamd_book_tab *)((char *)dst + h.off_books);
    for (i = 0; i < ci->books; i++) {
      const codebook *cb = ci->fullbooks + i;
      uint32_t off = place(&cur, (uint32_t)cb->entries);
      uint32_t offc = place(&cur, (uint32_t)cb->entries * 4u);
      if (!cb->codelist) return OV_EINVAL; /* (vorbis_book_init_encode builds it) */
      if (dst) {
        vamd_book_tab *t = books + i;
        t->dim = (int32_t)cb->dim;
        t->entries = (int32_t)cb->entries;
        t->minval = cb->minval;
        t->delta = cb->delta;
        t->quantvals = cb->quantvals;
        t->off_lengths = off;
        t->off_codes = offc;
        put(dst, offc, cb->codelist, (uint32_t)cb->entries * 4u);
        for (j = 0; j < cb->entries; j++) ((signed char *)dst)[off + j] = (signed char)cb->c->lengthlist[j];
      }
    }
    for (W = 0; W < 2; W++) {
     int sm;
     vorbis_info_mapping0 *map = (vorbis_info_mapping0 *)ci->map_param[ci->mode_param[W < ci->modes ? W : 0]->mapping];
     for (sm = 0; sm < map->submaps; sm++) {
      int resnum = map->residuesubmap[sm], acc = 0, maxstage = 0;
      vorbis_info_residue0 *ri = (vorbis_info_residue0 *)ci->residue_param[resnum];
      vamd_residue_tab *t = &h.res[W][sm];
      t->type = ci->residue_type[resnum];
      t->begin = (int32_t)ri->begin;
      t->end = (int32_t)ri->end;
      t->grouping = ri->grouping;
      t->partitions = ri->partitions;
      t->groupbook = ri->groupbook;
      t->groupbook_dim = (int32_t)ci->fullbooks[ri->groupbook].dim;
      if (ri->partitions > VAMD_RES_MAXCLASS) return OV_EIMPL;
      for (i = 0; i < ri->partitions; i++) {
        int stages = 0, v = ri->secondstages[i];
        while (v) { stages++; v >>= 1; } /* ilog, lib/res0.c:210 */
        if (stages > VAMD_RES_MAXSTAGE) return OV_EIMPL;
        if (stages > maxstage) maxstage = stages;
        t->secondstages[i] = ri->secondstages[i];
        t->classmetric1[i] = ri->classmetric1[i];
        t->classmetric2[i] = ri->classmetric2[i];
        for (j = 0; j < VAMD_RES_MAXSTAGE; j++) t->partbooks[i][j] = -1;
        for (j = 0; j < stages; j++)
          if (ri->secondstages[i] & (1 << j)) t->partbooks[i][j] = ri->booklist[acc++]; /* lib/res0.c:214-222 */
      }
      t->stages = maxstage;
     }
    }
  }

  cur = (cur + 15u) & ~15u;
  h.total_bytes = cur;
  if (dst) memcpy(dst, &h, sizeof(h));
  return (long)cur;
}
