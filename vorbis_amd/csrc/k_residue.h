// k_residue.h -- the residue back-end's numeric half: partition classes and the lattice-VQ search
// for type-2 residues (a bundle's channels interleaved: res2_class / res2_forward) and type-1
// residues (every coded channel on its own: res1_class / res1_forward) (reference lib/res0.c:
// _2class :479-532, _01class :412-470, res1_forward :729-746, res2_forward :783-809, _01forward
// :534-640, _encodepart :384-410, local_book_besterror :322-382); SURVEY.md 8f rank 2.  Looking the
// chosen entries' codewords up and packing them, with the phrase-book words, is k_pack.h.
//
// One team (workgroup of up to four waves) per block and submap.  A partition's class needs only its own 2 x 16 values; a (partition,
// stage, vector) search touches only its own `dim` values of the running work vector and is
// integer arithmetic, so within a stage all vectors of all partitions are searched at once and
// the stages -- which refine the same values -- follow each other with a wave sync.  Entries are
// written in the order _01forward emits them, (stage, partition, vector), at offsets from a
// prefix sum over the static per-class vector counts, so the host walks one packed list.
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"

namespace vamd {

// local_book_besterror, lib/res0.c:322-382.  `a` -> the vector in the work buffer (LDS).
VAMD_DEV int residue_besterror(const ResP &R, const vamd_book_tab &bk, int *a) {
  const signed char *len = (const signed char *)(R.base + bk.off_lengths);
  const int dim = bk.dim, minval = bk.minval, del = bk.delta, qv = bk.quantvals, ze = qv >> 1;
  const float rcp = div_rcp(del);
  int index = 0;
  int p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int av[8];
#pragma unroll
  for (int i = 0; i < 8; i++) av[i] = i < dim ? a[i] : 0;
#pragma unroll
  for (int o = 7; o >= 0; o--) {
    if (o >= dim) continue;
    // C's (a - minval + (del>>1)) / del truncates toward zero; del == 1 needs no division
    const int num = av[o] - minval + (del != 1 ? (del >> 1) : 0);
    int v;
    if (del == 1) {
      v = num;
    } else {
      const int mag = num < 0 ? -num : num;
      const int q = mag < (1 << 24) ? div_small(mag, del, rcp) : mag / del;
      v = num < 0 ? -q : q;
    }
    const int m = (v < ze ? ((ze - v) << 1) - 1 : ((v - ze) << 1));
    index = index * qv + (m < 0 ? 0 : (m >= qv ? qv - 1 : m));
    p[o] = v * del + minval;
  }
  if (len[index] <= 0) {
    // the lattice point is not a populated entry: exhaustive nearest search, in entry order (:349-376)
    int best = -1;
    int ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int maxval = minval + del * (qv - 1);
    for (int i = 0; i < bk.entries; i++) {
      if (len[i] > 0) {
        int dist = 0;
        for (int j = 0; j < dim; j++) {
          const int val = ev[j] - av[j];
          dist += val * val;
        }
        if (best == -1 || dist < best) {
          for (int j = 0; j < 8; j++) p[j] = ev[j];
          best = dist;
          index = i;
        }
      }
      int j = 0;  // odometer over the lattice, lib/res0.c:371-375 (the guard only matters after the last entry)
      while (j < 8 && ev[j] >= maxval) ev[j++] = 0;
      if (j < 8) {
        if (ev[j] >= 0) ev[j] += del;
        ev[j] = -ev[j];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (i < dim) a[i] = av[i] - p[i];
  return index;
}

// Emission offsets of a block's codebook entries, (stage, slot) order, from the slots' classes
// (`partvals` = number of slots): off[s*partvals + i] = entries written before slot i's stage-s vectors,
// off[stages*partvals] = the total (returned); `info` keeps each pair's book (-1 = the class skips
// the stage) so that nothing downstream goes back to the class tables.  cls/off/info in LDS.
//   WAVE_ONLY: the caller is one wave of a larger workgroup (k_pack_pair): its lanes alone, no workgroup barrier
template <bool WAVE_ONLY = false>
VAMD_DEV int residue_offsets(const ResP &R, int partvals, const int *cls, int *off, int *info) {
  const vamd_residue_tab &t = *R.tab;
  const int spp = t.grouping, items = t.stages * partvals;
  auto item = [&](int it) {
    const int s = it / partvals, i = it - s * partvals;
    const int c = cls[i];
    const int bn = ((t.secondstages[c] >> s) & 1) ? t.partbooks[c][s] : -1;
    info[it] = bn;
    off[it] = bn >= 0 ? spp / R.books[bn].dim : 0;
  };
  if constexpr (WAVE_ONLY) {
    WAVE_FOR(it, items) item(it);
    WAVE_SYNC();
  } else {
    TEAM_FOR(it, items) item(it);
    TEAM_SYNC();
  }
  if (WAVE_ONLY || TEAM_FIRST_WAVE) {
    int carry = 0;  // exclusive prefix sum over the <= 8 x 256 counts, a wave-width at a time
    for (int base = 0; base < items; base += NLANES) {
      const int it = base + LANE;
      const int c = it < items ? off[it] : 0;
      const int incl = wave_scan_sum(c);
      if (it < items) off[it] = carry + incl - c;
      carry += wave_last(incl);
    }
    if (LANE == 0) off[items] = carry;
  }
  if constexpr (WAVE_ONLY) {
    WAVE_SYNC();
  } else {
    TEAM_SYNC();
  }
  return off[items];
}

// One submap's residue: classification and search.
//   iwork[c]   HBM [n2]   quantised (and coupled) residue of the bundle's channel c; nonzero[c] its flag
//   work       LDS [bundle*n2]; cls LDS [slots]; off LDS [stages*slots + 1]; info LDS [stages*slots]
//   class_out  HBM [VAMD_RES_CLASS_STRIDE]; entries_out HBM [R.cap]; count_out HBM [2] = {classes, entries}
//   books_out  HBM [R.cap] or null: the book each entry belongs to, for the packet stage (k_pack.h: its fields then
//              need no search for the (stage, slot) pair they come from)
//   over       (optional) the input domain's integer edge, second half (include/vorbis_amd.h; the first is k_couple's):
//              bit c is set in a lane that loaded, for the bundle's channel c, a value the search below will see --
//              one at a position in [begin, end) -- beyond R.qmax, the bound up to which local_book_besterror's
//              integer arithmetic is defined by C for this residue's codebooks (derive_quant_limit, vamd_bind.h)
// A type-2 residue codes the bundle's channels interleaved as ONE stream (res2_class / res2_forward);
// type 1 codes every channel whose floor is not all zero as a stream of its own (res1_class /
// res1_forward, lib/res0.c:729-762), and _01forward then walks (stage, partition, stream, vector): a
// "slot" below is a (partition, stream) pair, numbered partition-major.
VAMD_DEV void residue_block(const ResP &R, int n2, const int *const *iwork, const int *nonzero, int *work, int *cls, int *off,
                            int *info, int *__restrict__ class_out, unsigned short *__restrict__ entries_out,
                            int *__restrict__ count_out, PhaseClock &pc, unsigned char *__restrict__ books_out = nullptr,
                            unsigned *over = nullptr) {
  const vamd_residue_tab &t = *R.tab;
  const int ch = R.bundle, spp = t.grouping, nparts = t.partitions, partvals = R.partvals, stages = t.stages;
  int ns = 0;  // streams
  if (t.type == 2) {
    int used = 0;
    for (int c = 0; c < ch; c++) used |= nonzero[c];
    ns = used ? 1 : 0;
  } else {
    for (int c = 0; c < ch; c++) ns += nonzero[c] ? 1 : 0;
  }
  if (!ns) {  // res*_class returns NULL and res*_forward writes nothing (:740-744,:766-777,:799-808)
    if (TEAM_LEADER) {
      count_out[0] = 0;
      count_out[1] = 0;
    }
    return;
  }
  if (t.type == 2) {
    // the interleaved work vector of res2_forward (:791-797)
    unsigned bad = 0;
    TEAM_FOR(j, n2)
      for (int c = 0; c < ch; c++) {
        const int v = iwork[c][j], i = j * ch + c;
        work[i] = v;
        if (i >= t.begin && i < t.end && (v > R.qmax || v < -R.qmax)) bad |= 1u << c;
      }
    if (over) *over = bad;
  } else {
    int sidx = 0;  // coded channels, packed in order (:738-739)
    unsigned bad = 0;
    for (int c = 0; c < ch; c++)
      if (nonzero[c]) {
        TEAM_FOR(j, n2) {
          const int v = iwork[c][j];
          work[sidx * n2 + j] = v;
          if (j >= t.begin && j < t.end && (v > R.qmax || v < -R.qmax)) bad |= 1u << c;
        }
        sidx++;
      }
    if (over) *over = bad;
  }
  TEAM_SYNC();
  const int slots = partvals * ns;
  // _01class (:436-453): peak against classmetric1, scaled mean against classmetric2, stream by stream
  if (t.type == 1) {
    const float scale = (float)(100. / spp);
    TEAM_FOR(q, slots) {
      const int i = q / ns, strm = q - i * ns;
      const int *w = work + strm * n2 + t.begin + i * spp;
      int mx = 0, ent = 0;
      for (int k = 0; k < spp; k++) {
        const int a = w[k] < 0 ? -w[k] : w[k];
        if (a > mx) mx = a;
        ent += a;
      }
      ent = (int)((float)ent * scale);  // "ent*=scale" with a float scale
      int k = 0;
      for (; k < nparts - 1; k++)
        if (mx <= t.classmetric1[k] && (t.classmetric2[k] < 0 || ent < t.classmetric2[k])) break;
      cls[q] = k;
      class_out[q] = k;
    }
  }
  // _2class (:501-518): channel 0 of the bundle against classmetric1, the rest against classmetric2
  if (t.type != 1) TEAM_FOR(i, partvals) {
    int magmax = 0, angmax = 0;
    const int *w = work + t.begin + i * spp;  // begin/ch bins in, interleaved
    for (int j = 0; j < spp; j += ch) {
      const int m = w[j] < 0 ? -w[j] : w[j];
      if (m > magmax) magmax = m;
      for (int k = 1; k < ch; k++) {
        const int a = w[j + k] < 0 ? -w[j + k] : w[j + k];
        if (a > angmax) angmax = a;
      }
    }
    int j = 0;
    for (; j < nparts - 1; j++)
      if (magmax <= t.classmetric1[j] && angmax <= t.classmetric2[j]) break;
    cls[i] = j;
    class_out[i] = j;
  }
  TEAM_SYNC();
  const int carry = residue_offsets(R, slots, cls, off, info);
  if (TEAM_LEADER) {
    count_out[0] = slots;
    count_out[1] = carry;
  }
  pc.mark(0);
  // the search, stage by stage (_01forward's s loop outermost, :585).  A stage's vectors are
  // numbered densely (its slice of the emission order), so every lane has one to search.
  for (int s = 0; s < stages; s++) {
    const int *so = off + s * slots;
    const int base = so[0], total = so[slots] - base;  // (off[] is stage-major: the next stage starts there)
    TEAM_FOR(v, total) {
      int lo = 0, hi = slots - 1;  // the slot whose vectors include v: last q with so[q] - base <= v
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (so[mid] - base <= v) lo = mid; else hi = mid - 1;
      }
      const int q = lo, k = v - (so[q] - base);
      const int i = q / ns, strm = q - i * ns;
      const vamd_book_tab &bk = R.books[info[s * slots + q]];
      const int entry = residue_besterror(R, bk, work + strm * n2 + t.begin + i * spp + k * bk.dim);
      if (base + v < R.cap) {
        entries_out[base + v] = (unsigned short)entry;
        if (books_out) books_out[base + v] = (unsigned char)info[s * slots + q];  // (< 256 books: vamd_bind.h)
      }
    }
    TEAM_SYNC();
  }
  pc.mark(1);
}

#if VAMD_GPU
// ---- round 6: the stereo type-2 residue out of registers ---------------------------------------------------------------
// residue_block above moves a block's values into LDS, classifies a partition per lane (32 reads each, all lanes on the
// same banks), and searches stage after stage with a workgroup barrier between stages and a binary search per vector for
// the partition it belongs to: 51 k + 28 k cycles per wave and block (tools/res_profile.py), most of them waiting.  None
// of that is needed where the vectors tile runs of EIGHT interleaved values (every book's dim divides 8, the partitions
// are 8, 16 or 32 values: ResP::chunked, vamd_bind.h): a stage only ever refines values another vector of the same run
// left behind, so a lane that owns a run -- four bins of both channels, two 16-byte loads -- takes it through all the
// stages in registers.  The partition's class is a max over the 1, 2 or 4 lanes that hold it (DPP quad_perm), the
// emission offsets are the same prefix sum over the classes as before (residue_offsets), and a run's vectors of a stage
// go to consecutive places behind its partition's.  No work vector in LDS, no barrier at all (a wave per block), no search.
VAMD_DEV int quad_max(int v, int lanes) {  // max over the aligned group of `lanes` (1, 2, 4) lanes; every lane active
  if (lanes >= 2) {
    const int t = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    v = v > t ? v : t;
  }
  if (lanes >= 4) {
    const int t = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    v = v > t ? v : t;
  }
  return v;
}

// local_book_besterror (lib/res0.c:322-382) on DIM values held in registers (`a`: constant indices once unrolled);
// the book as the (class, stage) row of the LDS table holds it
template <int DIM>
VAMD_DEV int residue_besterror_regs(const ResP &R, const ResStage &bk, int *a) {
  const int minval = bk.minval, del = bk.delta, qv = bk.quantvals, ze = qv >> 1;
  const float rcp = div_rcp(del);
  int index = 0;
  int p[DIM];
#pragma unroll
  for (int o = DIM - 1; o >= 0; o--) {
    const int num = a[o] - minval + (del != 1 ? (del >> 1) : 0);
    int v;
    if (del == 1) {
      v = num;
    } else {  // C's division truncates toward zero
      const int mag = num < 0 ? -num : num;
      const int q = mag < (1 << 24) ? div_small(mag, del, rcp) : mag / del;
      v = num < 0 ? -q : q;
    }
    const int m = (v < ze ? ((ze - v) << 1) - 1 : ((v - ze) << 1));
    index = index * qv + (m < 0 ? 0 : (m >= qv ? qv - 1 : m));
    p[o] = v * del + minval;
  }
  if (!bk.full) {
    const signed char *len = (const signed char *)(R.base + (unsigned)bk.off_lengths);
    if (len[index] <= 0) {
      // the lattice point is not a populated entry: exhaustive nearest search, in entry order (:349-376)
      int best = -1;
      int ev[DIM];
#pragma unroll
      for (int j = 0; j < DIM; j++) ev[j] = 0;
      const int maxval = minval + del * (qv - 1);
      for (int i = 0; i < bk.entries; i++) {
        if (len[i] > 0) {
          int dist = 0;
#pragma unroll
          for (int j = 0; j < DIM; j++) {
            const int val = ev[j] - a[j];
            dist += val * val;
          }
          if (best == -1 || dist < best) {
#pragma unroll
            for (int j = 0; j < DIM; j++) p[j] = ev[j];
            best = dist;
            index = i;
          }
        }
        // the odometer over the lattice (:371-375): digits at their maximum wrap to zero, the first one that is not steps
        bool carry = true;
#pragma unroll
        for (int j = 0; j < DIM; j++)
          if (carry) {
            if (ev[j] >= maxval) {
              ev[j] = 0;
            } else {
              if (ev[j] >= 0) ev[j] += del;
              ev[j] = -ev[j];
              carry = false;
            }
          }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < DIM; i++) a[i] -= p[i];
  return index;
}

// One block's search by ONE wave (k_residue_chunks: persistent waves, a block each at a time -- nothing to wait for but the
// wave's own loads, and a CU holds as many blocks in flight as it has wave slots).
//   iw0 / iw1  HBM [n2] the two channels' quantised (and coupled) residue
//   tab        LDS [R.fast_ints]: R.fast, copied there by the caller once per workgroup
//   cls [partvals], off [stages*partvals + 1]: this wave's LDS
// The runs are fetched twice, 64 at a time -- once for the classes, once for the search (the second time out of the
// caches): held in registers across the offsets' prefix sum, four passes of them cost the wave half its neighbours.
// Returns the input domain's integer edge, second half (include/vorbis_amd.h): bit c set when channel c holds a value beyond R.qmax.
VAMD_DEV void residue_run_fetch(const ResP &R, const int *__restrict__ iw0, const int *__restrict__ iw1, int it, int chunks, int *v) {
  I4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (it < chunks) {
    const int bin = (R.begin >> 1) + 4 * it;
    a = *(const I4 *)(iw0 + bin);
    b = *(const I4 *)(iw1 + bin);
  }
  v[0] = a.x, v[1] = b.x, v[2] = a.y, v[3] = b.y, v[4] = a.z, v[5] = b.z, v[6] = a.w, v[7] = b.w;
}
VAMD_DEV unsigned residue_wave_chunks(const ResP &R, const int *__restrict__ iw0, const int *__restrict__ iw1, int nz, const int *tab,
                                      int *cls, int *off, int *__restrict__ class_out, unsigned short *__restrict__ entries_out,
                                      int *__restrict__ count_out, PhaseClock &pc, unsigned char *__restrict__ books_out) {
  if (!nz) {  // res2_class returns NULL and res2_forward writes nothing
    if (LANE == 0) {
      count_out[0] = 0;
      count_out[1] = 0;
    }
    return 0;
  }
  const int spp = R.tab_grouping, nparts = R.nparts, partvals = R.partvals, stages = R.nstages;
  const int g = spp >> 3, chunks = partvals * g;  // lanes per partition (1, 2 or 4); runs of eight in [begin, end)
  const int *metric1 = tab, *metric2 = tab + nparts;
  const ResStage *rows = (const ResStage *)(tab + ((2 * nparts + 3) & ~3));
  unsigned bad = 0;
  for (int base = 0; base < chunks; base += NLANES) {
    const int it = base + LANE;
    int v[8];
    residue_run_fetch(R, iw0, iw1, it, chunks, v);
    int mag = 0, ang = 0;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const int m = v[k] < 0 ? -v[k] : v[k], an = v[k + 1] < 0 ? -v[k + 1] : v[k + 1];
      mag = m > mag ? m : mag;
      ang = an > ang ? an : ang;
    }
    // (zeros of the lanes past the end are inside the bound)
    if (mag > R.qmax) bad |= 1u;
    if (ang > R.qmax) bad |= 2u;
    // _2class (:501-518): channel 0 of the bundle against classmetric1, channel 1 against classmetric2
    mag = quad_max(mag, g);
    ang = quad_max(ang, g);
    int j = 0;
    for (; j < nparts - 1; j++)
      if (mag <= metric1[j] && ang <= metric2[j]) break;
    if (it < chunks && (it & (g - 1)) == 0) {
      cls[it / g] = j;
      class_out[it / g] = j;
    }
  }
  WAVE_SYNC();
  // where a partition's vectors of a stage go: _01forward emits (stage, partition, vector) -- an exclusive prefix sum over
  // the (stage, partition) counts, a wave-width at a time
  const int items = stages * partvals;
  int carry = 0;
  for (int base = 0; base < items; base += NLANES) {
    const int itx = base + LANE;
    int c = 0;
    if (itx < items) {
      const int s = itx / partvals, i = itx - s * partvals;
      const ResStage &st = rows[cls[i] * stages + s];
      c = st.bn >= 0 ? st.nv : 0;
    }
    const int incl = wave_scan_sum(c);
    if (itx < items) off[itx] = carry + incl - c;
    carry += wave_last(incl);
  }
  if (LANE == 0) {
    count_out[0] = partvals;
    count_out[1] = carry;
  }
  WAVE_SYNC();
  pc.mark(0);
  // the search: a run through every stage its partition's class has a book for
  for (int base = 0; base < chunks; base += NLANES) {
    const int it = base + LANE;
    int v[8];
    residue_run_fetch(R, iw0, iw1, it, chunks, v);
    if (it >= chunks) continue;
    const int part = it / g, r = it & (g - 1);
    const int mycls = cls[part];
    for (int s = 0; s < stages; s++) {
      const ResStage st = rows[mycls * stages + s];
      if (st.bn < 0) continue;
      const int dim = st.dim;
      const int e0 = off[s * partvals + part] + r * (8 / dim);
      auto emit = [&](int k, int entry) {
        if (e0 + k < R.cap) {
          entries_out[e0 + k] = (unsigned short)entry;
          if (books_out) books_out[e0 + k] = (unsigned char)st.bn;  // (< 256 books: vamd_bind.h)
        }
      };
      if (dim == 2) {
#pragma unroll
        for (int k = 0; k < 4; k++) emit(k, residue_besterror_regs<2>(R, st, v + 2 * k));
      } else if (dim == 4) {
#pragma unroll
        for (int k = 0; k < 2; k++) emit(k, residue_besterror_regs<4>(R, st, v + 4 * k));
      } else if (dim == 8) {
        emit(0, residue_besterror_regs<8>(R, st, v));
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) emit(k, residue_besterror_regs<1>(R, st, v + k));
      }
    }
  }
  WAVE_SYNC();  // (cls / off are the next block's from here)
  pc.mark(1);
  return bad;
}
// The same for ONE block spread over a workgroup (a handful of units: the per-block entry points, the batcher's small
// batches -- a lone block's latency is the caller's): a thread a run, the classes and offsets through LDS between two
// barriers, the search out of registers as above.  chunks <= blockDim.x.
//   tab [R.fast_ints], cls [partvals], off [stages * partvals + 1]: the workgroup's LDS
VAMD_DEV unsigned residue_team_chunks(const ResP &R, const int *__restrict__ iw0, const int *__restrict__ iw1, int nz, int *tab, int *cls,
                                      int *off, int *__restrict__ class_out, unsigned short *__restrict__ entries_out,
                                      int *__restrict__ count_out, PhaseClock &pc, unsigned char *__restrict__ books_out) {
  if (!nz) {
    if (TEAM_LEADER) {
      count_out[0] = 0;
      count_out[1] = 0;
    }
    return 0;
  }
  const int spp = R.tab_grouping, nparts = R.nparts, partvals = R.partvals, stages = R.nstages;
  const int g = spp >> 3, chunks = partvals * g;
  const int it = (int)threadIdx.x;
  TEAM_FOR(i, R.fast_ints >> 2)((I4 *)tab)[i] = ((const I4 *)R.fast)[i];  // the tables and the run: one trip to memory for both
  int v[8];
  residue_run_fetch(R, iw0, iw1, it, chunks, v);
  TEAM_SYNC();
  const int *metric1 = tab, *metric2 = tab + nparts;
  const ResStage *rows = (const ResStage *)(tab + ((2 * nparts + 3) & ~3));
  unsigned bad = 0;
  int mag = 0, ang = 0;
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const int m = v[k] < 0 ? -v[k] : v[k], an = v[k + 1] < 0 ? -v[k + 1] : v[k + 1];
    mag = m > mag ? m : mag;
    ang = an > ang ? an : ang;
  }
  if (mag > R.qmax) bad |= 1u;
  if (ang > R.qmax) bad |= 2u;
  mag = quad_max(mag, g);
  ang = quad_max(ang, g);
  int mycls = 0;
  for (; mycls < nparts - 1; mycls++)
    if (mag <= metric1[mycls] && ang <= metric2[mycls]) break;
  if (it < chunks && (it & (g - 1)) == 0) {
    cls[it / g] = mycls;
    class_out[it / g] = mycls;
  }
  TEAM_SYNC();
  const int items = stages * partvals;
  if (TEAM_FIRST_WAVE) {
    int carry = 0;
    for (int base = 0; base < items; base += NLANES) {
      const int itx = base + LANE;
      int c = 0;
      if (itx < items) {
        const int s = itx / partvals, i = itx - s * partvals;
        const ResStage &st = rows[cls[i] * stages + s];
        c = st.bn >= 0 ? st.nv : 0;
      }
      const int incl = wave_scan_sum(c);
      if (itx < items) off[itx] = carry + incl - c;
      carry += wave_last(incl);
    }
    if (LANE == 0) {
      count_out[0] = partvals;
      count_out[1] = carry;
    }
  }
  TEAM_SYNC();
  pc.mark(0);
  if (it < chunks) {
    const int part = it / g, r = it & (g - 1);
    for (int s = 0; s < stages; s++) {
      const ResStage st = rows[mycls * stages + s];
      if (st.bn < 0) continue;
      const int dim = st.dim;
      const int e0 = off[s * partvals + part] + r * (8 / dim);
      auto emit = [&](int k, int entry) {
        if (e0 + k < R.cap) {
          entries_out[e0 + k] = (unsigned short)entry;
          if (books_out) books_out[e0 + k] = (unsigned char)st.bn;
        }
      };
      if (dim == 2) {
#pragma unroll
        for (int k = 0; k < 4; k++) emit(k, residue_besterror_regs<2>(R, st, v + 2 * k));
      } else if (dim == 4) {
#pragma unroll
        for (int k = 0; k < 2; k++) emit(k, residue_besterror_regs<4>(R, st, v + 4 * k));
      } else if (dim == 8) {
        emit(0, residue_besterror_regs<8>(R, st, v));
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) emit(k, residue_besterror_regs<1>(R, st, v + k));
      }
    }
  }
  pc.mark(1);
  return bad;
}
#endif

}  // namespace vamd
