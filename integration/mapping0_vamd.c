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
 *   - it defines mapping0_forward_vamd(): the numeric section of mapping0_forward
 *     (lib/mapping0.c:254-576 and the floor render + couple/quantise of :613-646)
 *     is ONE call into libvorbis_amd.so; the bit-writing half (packet header,
 *     floor1_encode's Huffman writes, res*_class / res*_forward) is the
 *     reference's unchanged host code;
 *   - it exports a mapping0_exportbundle whose .forward is that function.
 *
 * Everything else in libvorbis / libvorbisenc links unchanged.  oracle/Makefile
 * builds exactly this into oracle/_ref/libvorbis_hybrid.so, and
 * tests/test_gpu_dropin.py checks that an encode through the hybrid library
 * emits byte-identical packets to the pure reference.
 *
 * Bitrate-managed encoders get all PACKETBLOBS candidate packets the same way
 * (vamd_analyze_block_managed); the bitrate manager that picks one is untouched host code.
 * Only channel counts above VAMD_MAX_CH fall through to the reference's CPU forward (host
 * code choosing its own CPU implementation -- the GPU library itself has no CPU path).
 */
#include <stdio.h>
#define mapping0_exportbundle mapping0_exportbundle_cpu
#include "mapping0.c" /* the reference's lib/mapping0.c, found through -I$(REF)/lib */
#undef mapping0_exportbundle

#include "vorbis_amd.h"

extern long vamd_pack_setup(vorbis_dsp_state *vd, void *dst, long cap);
/* res0_vamd.c: res2_forward for classes / entries the GPU already chose */
extern int vamd_res2_forward(oggpack_buffer *opb, vorbis_block *vb, vorbis_look_residue *vl, const int32_t *res_class,
                             long partvals, const uint16_t *entries, long nentries);

/* one GPU context per analysis state, created on first use (vorbis_analysis_init has
 * already built every lookup by then).  A real integration would hang the pointer off
 * private_state; a side table keeps this file self-contained.  The key is the state's
 * envelope_lookup (private_state.ve, lib/block.c:304): it is allocated by
 * vorbis_analysis_init and handed to _ve_envelope_clear() by vorbis_dsp_clear()
 * (lib/block.c:325-328), which is where envelope_vamd.c releases the entry -- so an entry
 * never outlives its stream even though no reference file is edited. */
#define VAMD_MAX_STATES 64
static struct {
  const void *vd; /* key: private_state.ve */
  vamd_ctx *ctx;
  vamd_envelope_state env; /* the block-switching detector's running state (envelope_vamd.c) */
} vamd_states[VAMD_MAX_STATES];

static const void *vamd_key(vorbis_dsp_state *vd) { return ((private_state *)vd->backend_state)->ve; }

vamd_ctx *vamd_ctx_for(vorbis_dsp_state *state) {
  const void *vd = vamd_key(state);
  int i;
  if (!vd) return NULL;
  for (i = 0; i < VAMD_MAX_STATES; i++)
    if (vamd_states[i].vd == vd) return vamd_states[i].ctx;
  for (i = 0; i < VAMD_MAX_STATES; i++)
    if (!vamd_states[i].vd) {
      long need = vamd_pack_setup(state, NULL, 0);
      void *blob;
      vamd_ctx *ctx = NULL;
      if (need < 0) return NULL;
      blob = _ogg_malloc(need);
      if (vamd_pack_setup(state, blob, need) != need || vamd_create(&ctx, blob, (size_t)need, -1) != VAMD_OK)
        ctx = NULL;
      _ogg_free(blob);
      if (ctx) {
        vamd_states[i].vd = vd;
        vamd_states[i].ctx = ctx;
        memset(&vamd_states[i].env, 0, sizeof(vamd_states[i].env)); /* fresh stream, lib/envelope.c:71 */
      }
      return ctx;
    }
  return NULL;
}

vamd_envelope_state *vamd_envelope_state_for(vorbis_dsp_state *state) {
  int i;
  if (!vamd_ctx_for(state)) return NULL;
  for (i = 0; i < VAMD_MAX_STATES; i++)
    if (vamd_states[i].vd == vamd_key(state)) return &vamd_states[i].env;
  return NULL;
}

/* called by _ve_envelope_clear() (envelope_vamd.c) with the envelope_lookup being torn down */
void vamd_release_key(const void *key) {
  int i;
  for (i = 0; i < VAMD_MAX_STATES; i++)
    if (key && vamd_states[i].vd == key) {
      vamd_destroy(vamd_states[i].ctx);
      vamd_states[i].vd = NULL;
      vamd_states[i].ctx = NULL;
    }
}


/* for a build WITHOUT envelope_vamd.c: call from vorbis_dsp_clear() before b->ve is freed */
void vamd_release_state(vorbis_dsp_state *state) { vamd_release_key(vamd_key(state)); }

/* The bit-writing half for one candidate packet k, unchanged host code (lib/mapping0.c:596-687):
 * packet header, floor1_encode's Huffman words from the posts the GPU fitted, residue.
 *   posts / post_valid / iwork / nonzero  this candidate's [ch][...] rows
 *   res_*   this candidate's residue decisions, or rescap == 0 for the host's own res*_class/forward */
static int vamd_write_packet(vorbis_block *vb, int k, int *posts, const int *post_valid, int *iwork,
                             const int *nonzero, int rescap, const int32_t *res_class, const uint16_t *res_entries,
                             const int32_t *res_count, int *scratch) {
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
  if (rescap > 0) {
    /* lib/mapping0.c:673-683 with the search done: the reference's _01forward writes the bits */
    if (vamd_res2_forward(opb, vb, b->residue[info->residuesubmap[0]], res_class, res_count[0], res_entries,
                          res_count[1]))
      return OV_EFAULT;
    return 0;
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
  codec_setup_info *ci = vi->codec_setup;
  vorbis_block_internal *vbi = (vorbis_block_internal *)vb->internal;
  const int n = vb->pcmend, ch = vi->channels;
  const int managed = vorbis_bitrate_managed(vb) ? 1 : 0, nk = managed ? PACKETBLOBS : 1;
  vorbis_info_mapping0 *info = ci->map_param[vb->W];
  vamd_ctx *ctx;
  float *mdct;
  int *iwork, *posts, *post_valid, *nonzero, *scratch;
  int32_t *res_class = NULL, *res_count = NULL;
  uint16_t *res_entries = NULL;
  float ampmax_out;
  int k, ret, rescap;

  if (ch > VAMD_MAX_CH) return mapping0_forward(vb); /* not covered: the host's own CPU code */
  ctx = vamd_ctx_for(vd);
  if (!ctx) return OV_EFAULT; /* no silent fallback: a missing GPU is an error */

  vb->mode = vb->W;
  mdct = _vorbis_block_alloc(vb, ch * (n / 2) * sizeof(*mdct));
  scratch = _vorbis_block_alloc(vb, (n / 2) * sizeof(*scratch));
  iwork = _vorbis_block_alloc(vb, nk * ch * (n / 2) * sizeof(*iwork));
  posts = _vorbis_block_alloc(vb, nk * ch * VAMD_POSTS_STRIDE * sizeof(*posts));
  post_valid = _vorbis_block_alloc(vb, nk * ch * sizeof(*post_valid));
  nonzero = _vorbis_block_alloc(vb, nk * ch * sizeof(*nonzero));
  /* where the mode's residue is covered (type 2 stereo / type 1 mono, one submap), its classification
     and lattice search come back with the same call and the host only writes bits */
  rescap = info->submaps == 1 ? vamd_residue_capacity(ctx, vb->W) : 0;
  if (rescap > 0) {
    res_class = _vorbis_block_alloc(vb, nk * VAMD_RES_CLASS_STRIDE * sizeof(*res_class));
    res_entries = _vorbis_block_alloc(vb, nk * rescap * sizeof(*res_entries));
    res_count = _vorbis_block_alloc(vb, nk * 2 * sizeof(*res_count));
  }

  /* ---- the numeric section: window, MDCT, FFT, masking, floor fit(s), floor curve(s),
     couple/quantise, residue search -- one call (lib/mapping0.c:254-576,613-646; managed: +:507-573
     and :613-646 for each of the PACKETBLOBS candidates) */
  if (managed)
    ret = vamd_analyze_block_managed(ctx, (const float *const *)vb->pcm, vb->lW, vb->W, vb->nW, vbi->blocktype,
                                     vbi->ampmax, mdct, &ampmax_out, posts, post_valid, iwork, nonzero, res_class,
                                     res_entries, res_count);
  else
    ret = vamd_analyze_block_res(ctx, (const float *const *)vb->pcm, vb->lW, vb->W, vb->nW, vbi->blocktype,
                                 vbi->ampmax, mdct, NULL, posts, post_valid, iwork, nonzero, &ampmax_out, res_class,
                                 res_entries, res_count);
  if (ret) {
    fprintf(stderr, "vorbis_amd: block analysis failed (%d): %s\n", ret, vamd_last_error(ctx));
    return ret;
  }
  vbi->ampmax = ampmax_out; /* lib/mapping0.c:576 */

  /* ---- the bit-writing half: VBR writes candidate PACKETBLOBS/2 only, a managed encoder all of
     them and lets vorbis_bitrate_addblock() choose (lib/mapping0.c:593-595) */
  for (k = 0; k < nk; k++) {
    ret = vamd_write_packet(vb, managed ? k : PACKETBLOBS / 2, posts + k * ch * VAMD_POSTS_STRIDE, post_valid + k * ch,
                            iwork + k * ch * (n / 2), nonzero + k * ch, rescap,
                            rescap > 0 ? res_class + k * VAMD_RES_CLASS_STRIDE : NULL,
                            rescap > 0 ? res_entries + (long)k * rescap : NULL, rescap > 0 ? res_count + 2 * k : NULL,
                            scratch);
    if (ret) return ret;
  }
  return 0;
}

const vorbis_func_mapping mapping0_exportbundle = {&mapping0_pack, &mapping0_unpack, &mapping0_free_info,
                                                   &mapping0_forward_vamd, &mapping0_inverse};
