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

#define VAMD_PK_RING 1024  // words; a power of two, > 64 * VAMD_PK_NF + 2 (one batch of fields spans <= 513 words)

struct BitRing {
  int *ring;          // LDS [VAMD_PK_RING], all zero between packets
  unsigned *out;      // HBM row of the packet
  int out_words;      // its length; words past it are counted but dropped
  int bitpos;         // bits written so far (wave-uniform; a packet is far below 2^31 bits: K.capacity bytes)
  int flushed;        // words already in HBM
};

// words [flushed, upto) are complete: move them out and hand their ring slots back
VAMD_DEV void ring_flush(BitRing &r, int upto) {
  WAVE_SYNC();
  for (int w = r.flushed + LANE; w < upto; w += NLANES) {
    const int slot = w & (VAMD_PK_RING - 1);
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
  const int start = r.bitpos + incl - len;
  if (((r.bitpos + total + 31) >> 5) - r.flushed > VAMD_PK_RING) ring_flush(r, r.bitpos >> 5);
  if (len > 0) {
    if (len < 32) code &= (1u << len) - 1u;
    const int w = start >> 5;
    const int sh = start & 31;
    lds_atomic_or(r.ring + (w & (VAMD_PK_RING - 1)), (int)(code << sh));
    if (sh + len > 32) lds_atomic_or(r.ring + ((w + 1) & (VAMD_PK_RING - 1)), (int)(code >> (32 - sh)));
  }
  r.bitpos += total;
}

// What the fields are looked up in, copied into LDS when a packet begins: a lone wave assembling one packet (the
// per-block entry points, the batcher's small batches) spends its time waiting for dependent reads -- class of the
// partition -> dimension of the class -> sub-book -> size of the book -> codeword took five trips to L2 per field, the
// walk to a partition's first post (a loop over the partitions before it) two more per partition; from LDS only the
// codeword itself is still a trip (k_pack alone: 36 -> 14 us per stereo packet, tools/gpu_block_phases.py).
#define VAMD_PK_FTAB_INTS (32 + 16 + 16 + 16 + 128 + 128)
struct PackTabs {
  int *part;     // [32]  floor partition i: its first post | class << 8 | class_dim << 12 | class_subs << 16
  int *cdim;     // [16]  class_dim
  int *csub;     // [16]  class_subs
  int *cbook;    // [16]  class_book
  int *subbook;  // [16][8] class_subbook
  int *maxval;   // [16][8] what floor_subclass compares a value with: the sub-book's entries (1 for none)
  int *books;    // [nbooks][3] entries, off_lengths, off_codes
  int floor_of;  // the submap whose floor the first six hold (-1: none yet)
  VAMD_MEM void at(int *base) {
    part = base, cdim = part + 32, csub = cdim + 16, cbook = csub + 16, subbook = cbook + 16, maxval = subbook + 128;
    books = maxval + 128;
    floor_of = -1;
  }
};

// The same for VAMD_PK_NF consecutive fields per lane (lane l holds fields NF*l .. NF*l + NF-1 of the batch): one scan
// for NF times the fields, and NF independent lookups per lane in flight -- a long block's later residue stages are some
// two thousand codewords, thirty-odd trips of the wave at one field per lane.
#define VAMD_PK_NF 8
VAMD_DEV void ring_putn(BitRing &r, const unsigned *code, const int *len) {
  int mine = 0;
#pragma unroll
  for (int k = 0; k < VAMD_PK_NF; k++) mine += len[k];
  const int incl = wave_scan_sum(mine);
  const int total = wave_last(incl);
  int start = r.bitpos + incl - mine;
  if (((r.bitpos + total + 31) >> 5) - r.flushed > VAMD_PK_RING) ring_flush(r, r.bitpos >> 5);
#pragma unroll
  for (int k = 0; k < VAMD_PK_NF; k++) {
    if (len[k] > 0) {
      unsigned c = code[k];
      if (len[k] < 32) c &= (1u << len[k]) - 1u;
      const int w = start >> 5;
      const int sh = start & 31;
      lds_atomic_or(r.ring + (w & (VAMD_PK_RING - 1)), (int)(c << sh));
      if (sh + len[k] > 32) lds_atomic_or(r.ring + ((w + 1) & (VAMD_PK_RING - 1)), (int)(c >> (32 - sh)));
    }
    start += len[k];
  }
  r.bitpos += total;
}

// vorbis_book_encode's lookup (lib/codebook.c:146-151): no field for an out-of-range or unused entry
VAMD_DEV void book_word(const PackP &K, const PackTabs &T, int booknum, int entry, unsigned &code, int &len) {
  code = 0;
  len = 0;
  if (booknum < 0) return;
  const int *bk = T.books + 3 * booknum;
  if (entry < 0 || entry >= bk[0]) return;
  const int l = ((const signed char *)(K.base + (unsigned)bk[1]))[entry];
  const unsigned c = ((const uint32_t *)(K.base + (unsigned)bk[2]))[entry];  // (asked for beside the length, not after it)
  if (l <= 0) return;
  len = l;
  code = c;
}

VAMD_DEV void pack_book_table(const PackP &K, PackTabs &T) {
  WAVE_FOR(b, K.nbooks) {
    const vamd_book_tab &bk = K.books[b];
    T.books[3 * b] = bk.entries;
    T.books[3 * b + 1] = (int)bk.off_lengths;
    T.books[3 * b + 2] = (int)bk.off_codes;
  }
  WAVE_SYNC();
}

// the floor's class tables (vorbis_info_floor1, lib/backends.h:60-72) and each partition's first post (:845-864)
VAMD_DEV void pack_floor_table(const vamd_floor1_tab &f, int sm, PackTabs &T) {
  if (T.floor_of == sm) return;
  T.floor_of = sm;
  WAVE_FOR(c, VAMD_FLOOR_CLASSES) {
    T.cdim[c] = f.class_dim[c];
    T.csub[c] = f.class_subs[c];
    T.cbook[c] = f.class_book[c];
  }
  WAVE_FOR(x, VAMD_FLOOR_CLASSES * 8) {
    const int bn = f.class_subbook[x >> 3][x & 7];
    T.subbook[x] = bn;
    T.maxval[x] = bn < 0 ? 1 : T.books[3 * bn];
  }
  WAVE_SYNC();
  WAVE_FOR(i, f.partitions) {
    const int cls = f.partitionclass[i];
    T.part[i] = cls << 8 | T.cdim[cls] << 12 | T.csub[cls] << 16;
  }
  WAVE_SYNC();
  WAVE_FOR(i, f.partitions) {
    int j = 2;
    for (int q = 0; q < i; q++) j += (T.part[q] >> 12) & 15;
    T.part[i] |= j;
  }
  WAVE_SYNC();
}

// the cascade choice of one post value (lib/floor1.c:866-876): first sub-book it fits in
VAMD_DEV int floor_subclass(const PackTabs &T, int cls, int csubbits, int val) {
  const int csub = 1 << csubbits;
  for (int l = 0; l < csub; l++)
    if (val < T.maxval[cls * 8 + l]) return l;
  return 0;
}

// One channel's floor1_encode writes.  outv LDS [VAMD_POSTS_STRIDE].
//   wrapped  HBM [posts] or null: floor1_encode's out[] as the floor stage left it (floor_quantise_predict's second
//            result); without it the values are formed here again
VAMD_DEV void pack_floor(const PackP &K, PackTabs &T, int sm, const FloorP &F, const int *__restrict__ posts,
                         const int *__restrict__ wrapped_in, int valid, int *outv, BitRing &r, PhaseClock &pc) {
  if (!valid) {  // "oggpack_write(opb,0,1)", lib/floor1.c:948-952
    ring_put(r, 0u, LANE == 0 ? 1 : 0);
    return;
  }
  const vamd_floor1_tab &f = *(sm ? K.ftab[1] : K.ftab[0]);  // (no dynamic index into the by-value parameter struct)
  if (wrapped_in) {
    WAVE_FOR(i, F.posts) outv[i] = wrapped_in[i];
  } else {
    LaneInts fitted, postlist, post, wrapped;
    fitted.load(posts, F.posts);
    postlist.load(F.postlist, F.posts);
    wrapped.fill(0);
    floor_quantise_predict(F, fitted, postlist, post, &wrapped);
    WAVE_FOR(i, F.posts) outv[i] = wrapped.at(i);
  }
  pack_floor_table(f, sm, T);  // (ends with a WAVE_SYNC)
  WAVE_SYNC();
  pc.mark(2);
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
      const int pw = T.part[i];
      const int j = pw & 0xff, cls = (pw >> 8) & 15, cdim = (pw >> 12) & 15, csubbits = pw >> 16;
      if (k < 0) {
        if (csubbits) {
          int cval = 0;
          for (int q = 0; q < cdim; q++) cval |= floor_subclass(T, cls, csubbits, outv[j + q]) << (q * csubbits);
          book_word(K, T, T.cbook[cls], cval, code, len);
        }
      } else if (k < cdim) {
        const int val = outv[j + k];
        const int sub = csubbits ? floor_subclass(T, cls, csubbits, val) : 0;
        book_word(K, T, T.subbook[cls * 8 + sub], val, code, len);
      }
    }
    ring_put(r, code, len);
  }
  WAVE_SYNC();  // outv is reused by the next channel
  pc.mark(7);
}

// The residue of one submap (lib/res0.c:534-640 with the search already done).
//   res_class / res_entries / res_count: what residue_block left for this submap (k_residue.h)
//   cls LDS [slots]; off LDS [stages*slots + 1]; info LDS [stages*slots]
//   res_books: the book of every entry as residue_block left it, or null (then it is searched for here)
//   rtab: R.fast as the caller holds it in LDS (k_pack_waves), or null: the (class, stage) rows then come from the image
VAMD_DEV void pack_residue(const PackP &K, const PackTabs &T, const ResP &R, const int *__restrict__ res_class,
                           const unsigned short *__restrict__ res_entries, const unsigned char *__restrict__ res_books,
                           const int *__restrict__ res_count, int *cls, int *off, int *info, BitRing &r, PhaseClock &pc,
                           const int *rtab = nullptr) {
  const vamd_residue_tab &t = *R.tab;
  const int slots = res_count[0];
  if (slots <= 0) return;  // nothing to code: res*_forward writes nothing
  const int partvals = R.partvals, ns = slots / partvals;  // streams: 1 (type 2) or the coded channels (type 1)
  WAVE_FOR(i, slots) cls[i] = res_class[i];
  WAVE_SYNC();
  if (rtab) {
    // residue_offsets out of the LDS copy of the rows: no trip to the image per (stage, slot)
    const ResStage *rows = (const ResStage *)(rtab + ((2 * R.nparts + 3) & ~3));
    const int stages = R.nstages, items = stages * slots;
    int carry = 0;
    for (int base = 0; base < items; base += NLANES) {
      const int it = base + LANE;
      int c = 0;
      if (it < items) {
        const int s = it / slots, i = it - s * slots;
        const ResStage &st = rows[cls[i] * stages + s];
        info[it] = st.bn;
        c = st.bn >= 0 ? st.nv : 0;
      }
      const int incl = wave_scan_sum(c);
      if (it < items) off[it] = carry + incl - c;
      carry += wave_last(incl);
    }
    if (LANE == 0) off[items] = carry;
    WAVE_SYNC();
  } else {
    residue_offsets<true>(R, slots, cls, off, info);  // (one wave: k_pack's only one, k_pack_pair's first)
  }
  pc.mark(3);
  const int ppw = R.groupbook_dim;
  {
    // stage 0 also carries the phrase words: one per stream ahead of every group of ppw partitions
    const int *so = off;
    const int total = so[slots] + ns * ((partvals + ppw - 1) / ppw);
    for (int v0 = 0; v0 < total; v0 += NLANES) {
      const int v = v0 + LANE;
      unsigned code = 0;
      int len = 0;
      if (v < total) {
        // fields are numbered in emission order; slot q = (partition i, stream j)'s start at so[q]
        // plus the phrase words of the groups begun before it (its own group's once j > 0)
        int lo = 0, hi = slots - 1;
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          const int mi = mid / ns, mj = mid - mi * ns;
          const int at = so[mid] + ns * ((mi + (mj > 0) + ppw - 1) / ppw);
          if (at <= v) lo = mid; else hi = mid - 1;
        }
        const int q = lo, i = q / ns, j = q - i * ns;
        int k = v - (so[q] + ns * ((i + (j > 0) + ppw - 1) / ppw));
        const bool leads = j == 0 && i % ppw == 0;
        if (leads && k < ns) {  // stream k's classes of the group as one number, lib/res0.c:589-598
          int val = cls[i * ns + k];
          for (int p = 1; p < ppw; p++) {
            val *= R.nparts;
            if (i + p < partvals) val += cls[(i + p) * ns + k];
          }
          book_word(K, T, R.groupbook, val, code, len);
        } else {
          if (leads) k -= ns;
          const int e = so[q] + k;
          book_word(K, T, info[q], e < R.cap ? (int)res_entries[e] : -1, code, len);
        }
      }
      ring_put(r, code, len);
    }
    pc.mark(4);
  }
  {
    // The later stages are plain: off[] is the emission order itself, stage after stage, so field v of them is entry
    // off[slots] + v, and one run of trips covers them all (a stage of its own costs at least a trip, and there are up to
    // seven).  Its book comes with it (res_books) -- or is that of the last (stage, slot) pair beginning at or before
    // it, a binary search of nine dependent LDS reads per field: a long block has some two thousand of them.  With the
    // book at hand a lane takes VAMD_PK_NF fields a trip, their lookups all in flight together.
    const int first = slots, last = R.nstages * slots, base = off[first], total = off[last] - base;
    for (int v0 = 0; v0 < total; v0 += NLANES * VAMD_PK_NF) {
      unsigned code[VAMD_PK_NF];
      int len[VAMD_PK_NF], entry[VAMD_PK_NF], book[VAMD_PK_NF];
#pragma unroll
      for (int k = 0; k < VAMD_PK_NF; k++) {
        const int v = v0 + LANE * VAMD_PK_NF + k, e = base + v;
        const bool in = v < total && e < R.cap;
        entry[k] = in ? (int)res_entries[e] : -1;
        book[k] = in && res_books ? (int)res_books[e] : -1;
      }
      if (!res_books) {
#pragma unroll
        for (int k = 0; k < VAMD_PK_NF; k++) {
          const int v = v0 + LANE * VAMD_PK_NF + k, e = base + v;
          if (v < total) {
            int lo = first, hi = last - 1;
            while (lo < hi) {
              const int mid = (lo + hi + 1) >> 1;
              if (off[mid] <= e) lo = mid; else hi = mid - 1;
            }
            book[k] = info[lo];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < VAMD_PK_NF; k++) book_word(K, T, book[k], entry[k], code[k], len[k]);
      ring_putn(r, code, len);
    }
    pc.mark(5);
  }
}

// The packet itself, the ring zeroed and the tables in place (pack_block below; k_pack_waves, whose waves keep both across
// packets).  rtab: see pack_residue.
VAMD_DEV void pack_block_body(const PackP &K, PackTabs &T, const FloorP &F0, const FloorP &F1, const ResP &R0, const ResP &R1, const ChMap &cm,
                              int ch, int W, int lW, int nW, const int *__restrict__ posts, const int *__restrict__ wrapped,
                              const int *__restrict__ post_valid, const int *__restrict__ res_class, const unsigned short *__restrict__ res_entries,
                              const unsigned char *__restrict__ res_books, const int *__restrict__ res_count, int *ring, int *outv, int *cls, int *off, int *info,
                              unsigned *__restrict__ packet, int out_words, int *__restrict__ bits_out, PhaseClock &pc, const int *rtab) {
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
    pack_floor(K, T, sm, sm ? F1 : F0, posts + c * VAMD_POSTS_STRIDE, wrapped ? wrapped + c * VAMD_POSTS_STRIDE : nullptr,
               post_valid[c], outv, r, pc);
  }
  for (int sm = 0; sm < cm.submaps; sm++) {
    const ResP &R = sm ? R1 : R0;
    pack_residue(K, T, R, res_class + R.cls_base, res_entries + R.ent_base, res_books ? res_books + R.ent_base : nullptr,
                 res_count + 2 * sm, cls, off, info, r, pc, sm == 0 ? rtab : nullptr);
  }
  ring_flush(r, (r.bitpos + 31) >> 5);
  if (LANE == 0) *bits_out = r.bitpos;
  pc.mark(6);
}

// One packet: header, the channels' floors, the submaps' residues.
//   posts [ch][VAMD_POSTS_STRIDE], post_valid [ch]  as floor_encode_render left them
//   res_class [submaps][VAMD_RES_CLASS_STRIDE], res_entries [row], res_count [submaps][2]: one block's rows;
//   res_books [row] or null (pack_residue)
//   packet HBM [out_words] words; bits_out <- oggpack_bits()
//   wrapped [ch][VAMD_POSTS_STRIDE] or null: see pack_floor
//   LDS: ring [VAMD_PK_RING] zeroed here, outv [VAMD_POSTS_STRIDE], cls/off/info as pack_residue,
//        tabs [VAMD_PK_FTAB_INTS + 3 * K.nbooks] (PackTabs)
VAMD_DEV void pack_block(const PackP &K, const FloorP &F0, const FloorP &F1, const ResP &R0, const ResP &R1, const ChMap &cm,
                         int ch, int W, int lW, int nW, const int *__restrict__ posts, const int *__restrict__ wrapped,
                         const int *__restrict__ post_valid, const int *__restrict__ res_class, const unsigned short *__restrict__ res_entries,
                         const unsigned char *__restrict__ res_books, const int *__restrict__ res_count, int *ring, int *outv, int *cls, int *off, int *info, int *tabs,
                         unsigned *__restrict__ packet, int out_words, int *__restrict__ bits_out, PhaseClock &pc) {
  WAVE_FOR(i, VAMD_PK_RING) ring[i] = 0;
  PackTabs T;
  T.at(tabs);
  pack_book_table(K, T);  // (ends with a WAVE_SYNC)
  pack_block_body(K, T, F0, F1, R0, R1, cm, ch, W, lW, nW, posts, wrapped, post_valid, res_class, res_entries, res_books, res_count, ring,
                  outv, cls, off, info, packet, out_words, bits_out, pc, nullptr);
}

}  // namespace vamd
