// k_pack.h -- packet assembly: the bit-writing half of mapping0_forward for one block
// (reference lib/mapping0.c:593-606 header bits, lib/floor1.c:833-921 floor1_encode's writes,
// lib/res0.c:534-640 _01forward's phrase words and codewords, lib/codebook.c:146-151
// vorbis_book_encode, libogg oggpack_write's LSb-first packing); SURVEY.md 8f rank 4.
//
// Bit packing looks serial -- every field starts where the previous one ended -- but the fields
// themselves are known up front: each is a (codeword, length) pair looked up from tables by values
// the earlier stages left in HBM (fitted posts, partition classes, codebook entries).  So a wave
// takes the fields 64 at a time in emission order, a prefix sum over the lengths gives every lane
// its bit offset, and the lanes OR their fields into a small ring of 32-bit words in LDS; full words
// leave for HBM a wave-width at a time (coalesced).  One wave per packet.
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"
#include "k_floor.h"
#include "k_residue.h"

namespace vamd {

#define VAMD_PK_RING 256  // words; a power of two, > 2 * 64 + 2 (one batch of fields spans <= 65 words)

struct BitRing {
  int *ring;          // LDS [VAMD_PK_RING], all zero between packets
  unsigned *out;      // HBM row of the packet
  int out_words;      // its length; words past it are counted but dropped
  long bitpos;        // bits written so far (wave-uniform)
  long flushed;       // words already in HBM
};

// words [flushed, upto) are complete: move them out and hand their ring slots back
VAMD_DEV void ring_flush(BitRing &r, long upto) {
  WAVE_SYNC();
  for (long w = r.flushed + LANE; w < upto; w += NLANES) {
    const int slot = (int)(w & (VAMD_PK_RING - 1));
    const unsigned u = (unsigned)r.ring[slot];
    r.ring[slot] = 0;
    if (w < r.out_words) r.out[w] = u;
  }
  r.flushed = upto;
  WAVE_SYNC();
}

// oggpack_write(code, len) for every lane's field, in lane order.  Collective; len 0 = no field.
VAMD_DEV void ring_put(BitRing &r, unsigned code, int len) {
  const int incl = wave_scan_sum(len);
  const int total = wave_last(incl);
  const long start = r.bitpos + incl - len;
  if (((r.bitpos + total + 31) >> 5) - r.flushed > VAMD_PK_RING) ring_flush(r, r.bitpos >> 5);
  if (len > 0) {
    if (len < 32) code &= (1u << len) - 1u;
    const long w = start >> 5;
    const int sh = (int)(start & 31);
    lds_atomic_or(r.ring + (w & (VAMD_PK_RING - 1)), (int)(code << sh));
    if (sh + len > 32) lds_atomic_or(r.ring + ((w + 1) & (VAMD_PK_RING - 1)), (int)(code >> (32 - sh)));
  }
  r.bitpos += total;
}

// vorbis_book_encode's lookup (lib/codebook.c:146-151): no field for an out-of-range or unused entry
VAMD_DEV void book_word(const PackP &K, int booknum, int entry, unsigned &code, int &len) {
  code = 0;
  len = 0;
  if (booknum < 0) return;
  const vamd_book_tab &bk = K.books[booknum];
  if (entry < 0 || entry >= bk.entries) return;
  const int l = ((const signed char *)(K.base + bk.off_lengths))[entry];
  if (l <= 0) return;
  len = l;
  code = ((const uint32_t *)(K.base + bk.off_codes))[entry];
}

// the cascade choice of one post value (lib/floor1.c:866-876): first sub-book it fits in
VAMD_DEV int floor_subclass(const PackP &K, const vamd_floor1_tab &f, int cls, int val) {
  const int csub = 1 << f.class_subs[cls];
  for (int l = 0; l < csub; l++) {
    const int bn = f.class_subbook[cls][l];
    const int maxval = bn < 0 ? 1 : K.books[bn].entries;
    if (val < maxval) return l;
  }
  return 0;
}

// One channel's floor1_encode writes.  outv LDS [VAMD_POSTS_STRIDE].
VAMD_DEV void pack_floor(const PackP &K, int sm, const FloorP &F, const int *__restrict__ posts, int valid, int *outv,
                         BitRing &r) {
  if (!valid) {  // "oggpack_write(opb,0,1)", lib/floor1.c:948-952
    ring_put(r, 0u, LANE == 0 ? 1 : 0);
    return;
  }
  const vamd_floor1_tab &f = *(sm ? K.ftab[1] : K.ftab[0]);  // (no dynamic index into the by-value parameter struct)
  {
    LaneInts fitted, postlist, post, wrapped;
    fitted.load(posts, F.posts);
    postlist.load(F.postlist, F.posts);
    wrapped.fill(0);
    floor_quantise_predict(F, fitted, postlist, post, &wrapped);
    WAVE_FOR(i, F.posts) outv[i] = wrapped.at(i);
    WAVE_SYNC();
  }
  // the nontrivial-floor flag and the two end posts (:833-841)
  for (int t0 = 0; t0 < 3; t0 += NLANES) {
    const int t = t0 + LANE;
    ring_put(r, t == 0 ? 1u : (unsigned)outv[t < 3 ? t - 1 : 0], t == 0 ? 1 : (t < 3 ? (sm ? K.qbits[1] : K.qbits[0]) : 0));
  }
  // partition by partition (:845-917): slot 0 of a partition is its cascade word, slots 1..8 its posts
  const int slots = f.partitions * 9;
  for (int t0 = 0; t0 < slots; t0 += NLANES) {
    const int t = t0 + LANE;
    unsigned code = 0;
    int len = 0;
    if (t < slots) {
      const int i = t / 9, k = t - i * 9 - 1;
      int j = 2;
      for (int q = 0; q < i; q++) j += f.class_dim[f.partitionclass[q]];
      const int cls = f.partitionclass[i], cdim = f.class_dim[cls], csubbits = f.class_subs[cls];
      if (k < 0) {
        if (csubbits) {
          int cval = 0;
          for (int q = 0; q < cdim; q++) cval |= floor_subclass(K, f, cls, outv[j + q]) << (q * csubbits);
          book_word(K, f.class_book[cls], cval, code, len);
        }
      } else if (k < cdim) {
        const int val = outv[j + k];
        const int sub = csubbits ? floor_subclass(K, f, cls, val) : 0;
        book_word(K, f.class_subbook[cls][sub], val, code, len);
      }
    }
    ring_put(r, code, len);
  }
  WAVE_SYNC();  // outv is reused by the next channel
}

// The residue of one submap (lib/res0.c:534-640 with the search already done).
//   res_class / res_entries / res_count: what residue_block left for this submap (k_residue.h)
//   cls LDS [slots]; off LDS [stages*slots + 1]; info LDS [stages*slots]
VAMD_DEV void pack_residue(const PackP &K, const ResP &R, const int *__restrict__ res_class,
                           const unsigned short *__restrict__ res_entries, const int *__restrict__ res_count,
                           int *cls, int *off, int *info, BitRing &r) {
  const vamd_residue_tab &t = *R.tab;
  const int slots = res_count[0];
  if (slots <= 0) return;  // nothing to code: res*_forward writes nothing
  const int partvals = R.partvals, ns = slots / partvals;  // streams: 1 (type 2) or the coded channels (type 1)
  WAVE_FOR(i, slots) cls[i] = res_class[i];
  WAVE_SYNC();
  residue_offsets(R, slots, cls, off, info);
  const int ppw = t.groupbook_dim;
  for (int s = 0; s < t.stages; s++) {
    const int *so = off + s * slots;
    const int base = so[0];
    // stage 0 also carries the phrase words: one per stream ahead of every group of ppw partitions
    const int total = so[slots] - base + (s == 0 ? ns * ((partvals + ppw - 1) / ppw) : 0);
    for (int v0 = 0; v0 < total; v0 += NLANES) {
      const int v = v0 + LANE;
      unsigned code = 0;
      int len = 0;
      if (v < total) {
        // fields are numbered in emission order; slot q = (partition i, stream j)'s start at so[q]-base
        // plus, in stage 0, the phrase words of the groups begun before it (its own group's once j > 0)
        int lo = 0, hi = slots - 1;
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          const int mi = mid / ns, mj = mid - mi * ns;
          const int at = so[mid] - base + (s == 0 ? ns * ((mi + (mj > 0) + ppw - 1) / ppw) : 0);
          if (at <= v) lo = mid; else hi = mid - 1;
        }
        const int q = lo, i = q / ns, j = q - i * ns;
        int k = v - (so[q] - base + (s == 0 ? ns * ((i + (j > 0) + ppw - 1) / ppw) : 0));
        const bool leads = s == 0 && j == 0 && i % ppw == 0;
        if (leads && k < ns) {  // stream k's classes of the group as one number, lib/res0.c:589-598
          int val = cls[i * ns + k];
          for (int p = 1; p < ppw; p++) {
            val *= t.partitions;
            if (i + p < partvals) val += cls[(i + p) * ns + k];
          }
          book_word(K, t.groupbook, val, code, len);
        } else {
          if (leads) k -= ns;
          const int e = so[q] + k;
          book_word(K, info[s * slots + q], e < R.cap ? (int)res_entries[e] : -1, code, len);
        }
      }
      ring_put(r, code, len);
    }
  }
}

// One packet: header, the channels' floors, the submaps' residues.
//   posts [ch][VAMD_POSTS_STRIDE], post_valid [ch]  as floor_encode_render left them
//   res_class [submaps][VAMD_RES_CLASS_STRIDE], res_entries [row], res_count [submaps][2]: one block's rows
//   packet HBM [out_words] words; bits_out <- oggpack_bits()
//   LDS: ring [VAMD_PK_RING] zeroed here, outv [VAMD_POSTS_STRIDE], cls/off/info as pack_residue
VAMD_DEV void pack_block(const PackP &K, const FloorP &F0, const FloorP &F1, const ResP &R0, const ResP &R1, const ChMap &cm,
                         int ch, int W, int lW, int nW, const int *__restrict__ posts, const int *__restrict__ post_valid,
                         const int *__restrict__ res_class, const unsigned short *__restrict__ res_entries,
                         const int *__restrict__ res_count, int *ring, int *outv, int *cls, int *off, int *info,
                         unsigned *__restrict__ packet, int out_words, int *__restrict__ bits_out) {
  WAVE_FOR(i, VAMD_PK_RING) ring[i] = 0;
  WAVE_SYNC();
  BitRing r;
  r.ring = ring;
  r.out = packet;
  r.out_words = out_words;
  r.bitpos = 0;
  r.flushed = 0;
  {  // lib/mapping0.c:598-604: packet type 0 (audio), the mode number, and for a long block its neighbours' sizes
    unsigned hdr = (unsigned)W << 1;  // "int modenumber=vb->W", lib/mapping0.c:248
    int len = 1 + K.modebits;
    if (W) {
      hdr |= (unsigned)(lW ? 1 : 0) << len;
      hdr |= (unsigned)(nW ? 1 : 0) << (len + 1);
      len += 2;
    }
    ring_put(r, hdr, LANE == 0 ? len : 0);
  }
  for (int c = 0; c < ch; c++) {
    const int sm = cm.sub[c];
    pack_floor(K, sm, sm ? F1 : F0, posts + c * VAMD_POSTS_STRIDE, post_valid[c], outv, r);
  }
  for (int sm = 0; sm < cm.submaps; sm++) {
    const ResP &R = sm ? R1 : R0;
    pack_residue(K, R, res_class + R.cls_base, res_entries + R.ent_base, res_count + 2 * sm, cls, off, info, r);
  }
  ring_flush(r, (r.bitpos + 31) >> 5);
  if (LANE == 0) *bits_out = (int)r.bitpos;
}

}  // namespace vamd
