/* integration/res0_vamd.c -- reference-side binding for the residue back-end.
 * A maintainer builds this INSTEAD of lib/res0.c.  It pulls the reference's own res0.c in by path,
 * unchanged (every residue type, the decoder, the export bundles), and adds ONE function:
 *
 *   vamd_res2_forward(): res2_forward (lib/res0.c:783-809) -- or res1_forward (:729-746) with one
 *   channel in use, which reaches _01forward the same way -- for a block whose partition classes
 *   and lattice-VQ entries libvorbis_amd.so has already chosen.  It hands the reference's own
 *   _01forward (:534-640) -- phrase-book words, stage/partition interleaving, statistics -- an
 *   encode callback that, instead of searching (_encodepart / local_book_besterror, :322-410),
 *   writes the next precomputed entries through the unchanged vorbis_book_encode.
 *
 * mapping0_vamd.c calls it in place of _residue_P[2]->class / ->forward when the mode's residue is
 * covered (vamd_residue_capacity() > 0).  Bits out are byte-identical (tests/test_gpu_dropin.py).
 */
#include <stdint.h>
#include "res0.c" /* the reference's lib/res0.c, found through -I$(REF)/lib */

/* the entry list of the block being written; one encoder state is driven by one thread at a time
 * (libvorbis' own rule), different states may run on different threads */
static __thread struct {
  const uint16_t *next;
  long left;
} vamd_replay;

static int vamd_encode_replay(oggpack_buffer *opb, int *vec, int n, codebook *book) {
  int i, bits = 0;
  const int step = n / book->dim; /* _encodepart, :392-393 */
  (void)vec;
  for (i = 0; i < step; i++) {
    if (vamd_replay.left <= 0) return bits; /* cannot happen: the counts come from the same class table */
    vamd_replay.left--;
    bits += vorbis_book_encode(book, *vamd_replay.next++, opb);
  }
  return bits;
}

/* res_class[partvals], entries[nentries] as vamd_analyze_block_res() returned them.
 * Returns what res2_forward returns; -1 if the entry list and the class table disagree. */
int vamd_res2_forward(oggpack_buffer *opb, vorbis_block *vb, vorbis_look_residue *vl, const int32_t *res_class,
                      long partvals, const uint16_t *entries, long nentries) {
  vorbis_look_residue0 *look = (vorbis_look_residue0 *)vl;
  vorbis_info_residue0 *info = look->info;
  long **partword, i;
  int *work, ret;
  if (partvals <= 0) return 0; /* res2_class returned NULL: nothing is written (:766-777,:799-808) */
  if (partvals != (info->end - info->begin) / info->grouping) return -1;
  partword = _vorbis_block_alloc(vb, sizeof(*partword));
  partword[0] = _vorbis_block_alloc(vb, partvals * sizeof(*partword[0]));
  for (i = 0; i < partvals; i++) partword[0][i] = res_class[i];
  /* _01forward only forms in[0]+offset pointers for the callback, which ignores them */
  work = _vorbis_block_alloc(vb, (info->end + 1) * sizeof(*work));
  vamd_replay.next = entries;
  vamd_replay.left = nentries;
  look->frames++; /* _2class's statistic, :529 */
  ret = _01forward(opb, vl, &work, 1, partword, vamd_encode_replay);
  return (ret == 0 && vamd_replay.left != 0) ? -1 : ret;
}
