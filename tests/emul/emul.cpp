// tests/emul/emul.cpp -- TEST BUILD ONLY: the kernel bodies of vorbis_amd/csrc
// (k_*.h) compiled by the host C++ compiler with LANE=0 / NLANES=1
// (vamd_wave.h), so that every stage's arithmetic can be checked bit-for-bit
// against the reference on a machine without a GPU.  Nothing in the product
// links or loads this file; the product fails loudly without its HIP library.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "vamd_bind.h"
#include "k_transform.h"
#include "k_noise.h"
#include "k_tone.h"
#include "k_floor.h"
#include "k_couple.h"
#include "k_envelope.h"
#include "k_blockout.h"
#include "k_residue.h"
#include "k_pack.h"
#include "k_lpc.h"

using namespace vamd;

// the single-lane order of events of _vp_noisemask: all bins in one lane, the five running sums one after the other
struct ScanSerial {
  void before_terms() const {}
  void scan(float *S, int n) const {
    for (int a = 0; a < 5; a++) running_sum_inplace(S + a * VAMD_NZ_STRIDE(n), n);
  }
};
#define VAMD_NZ_HOST_BINS 4096
#define VAMD_HOST_QUADS 2048  // a whole block's quads in the one lane of this build (blocks up to 8192 samples)
static void noisemask_block(const PsyP &P, const float *logmdct, float *out, float *S, PhaseClock &pc) {
  static float lm[VAMD_NZ_HOST_BINS], o[VAMD_NZ_HOST_BINS];
  static int braw[VAMD_NZ_HOST_BINS], bk[VAMD_NZ_HOST_BINS];
  for (int i = 0; i < P.n; i++) lm[i] = logmdct[i];
  noise_bark_fetch<VAMD_NZ_HOST_BINS, 0>(P, braw, 0);
  noise_bark_edges<VAMD_NZ_HOST_BINS, 0>(P, braw, bk, 0);
  noisemask_bins<ScanSerial, VAMD_NZ_HOST_BINS, 0>(P, lm, bk, o, S, [&](int dB) { return P.noisecompand[dB]; }, ScanSerial(), pc, 0);
  for (int i = 0; i < P.n; i++) out[i] = o[i];
}

struct Emul {
  std::vector<unsigned char> image;
  std::vector<uint32_t> doff;
  std::vector<PsyDerived> derived;
  Bound B;
  std::string err;
};

struct EmulTaps {  // arrays [ch][...], any may be null
  float *mdct_raw, *logfft, *logmdct, *noise, *tone, *logmask, *mdct;
  int *posts, *post_valid, *ilogmask, *iwork, *nonzero;
  float *local_ampmax, *ampmax_out;
  int *res_class;               // [VAMD_RES_CLASS_STRIDE]
  unsigned short *res_entries;  // [capacity of the mode]
  int *res_count;               // [2]
  unsigned char *packet;        // [emul_packet_capacity] (needs the three res_* too)
  int *packet_bits;
};

struct EmulMTaps {  // per candidate packet of a bitrate-managed block, [15][ch][...]
  int *posts, *post_valid, *iwork, *nonzero;
  unsigned char *packets;  // [15][emul_packet_capacity] or null
  int *packet_bits;        // [15]
};

// residue search + packet assembly of one packet, the way k_residue / k_pack run them
//   wrapped: floor1_encode's out[] as the floor stage left it ([ch][VAMD_POSTS_STRIDE]) or null.  The packet is assembled the
//   way the GPU library does it -- the floor's out[] handed over, every entry's book from the residue search -- and once
//   more the other way (out[] formed again, books searched for); *packet_bits = -1 if the two differ.
static void residue_and_pack(const Bound &B, int W, int lW, int nW, const int *iwork, const int *nonzero, const int *posts,
                             const int *post_valid, int *res_class, unsigned short *res_entries, int *res_count,
                             unsigned char *packet, int *packet_bits, const int *wrapped = nullptr) {
  const int ch = B.channels, n2 = B.bs[W] / 2;
  const ChMap &cm = B.chmap[W];
  PhaseClock pc;
  pc.start(nullptr);
  std::vector<int> lds(B.res_lds_ints[W] + 16);
  std::vector<unsigned char> books((size_t)B.res_cap[W] + 16, 255);
  for (int sm = 0; sm < cm.submaps; sm++) {
    const ResP &Rp = B.res[W][sm];
    int *work = lds.data(), *cls = work + Rp.bundle * n2, *off = cls + VAMD_RES_CLASS_STRIDE,
        *info = off + (Rp.tab->stages * Rp.slots + 1);
    const int *ip[VAMD_MAX_CH];
    int nz[VAMD_MAX_CH], nb = 0;
    for (int c = 0; c < ch; c++)
      if (cm.sub[c] == sm) ip[nb] = iwork + c * n2, nz[nb] = nonzero[c], nb++;
    residue_block(Rp, n2, ip, nz, work, cls, off, info, res_class + Rp.cls_base, res_entries + Rp.ent_base, res_count + 2 * sm,
                  pc, books.data() + Rp.ent_base);
  }
  if (!packet) return;
  std::vector<int> ring(VAMD_PK_RING), outv(VAMD_POSTS_STRIDE), cls(VAMD_RES_CLASS_STRIDE), off(B.res_off_ints[W]),
      info(B.res_off_ints[W]), tabs(VAMD_PK_FTAB_INTS + 3 * B.pack[W].nbooks);
  pack_block(B.pack[W], B.floor[W][0], B.floor[W][1], B.res[W][0], B.res[W][1], cm, ch, W, lW, nW, posts, wrapped, post_valid, res_class, res_entries,
             books.data(), res_count, ring.data(), outv.data(), cls.data(), off.data(), info.data(), tabs.data(), (unsigned *)packet,
             B.pack[W].capacity / 4, packet_bits, pc);
  std::vector<unsigned> again(B.pack[W].capacity / 4 + 1, 0);
  int bits2 = 0;
  pack_block(B.pack[W], B.floor[W][0], B.floor[W][1], B.res[W][0], B.res[W][1], cm, ch, W, lW, nW, posts, nullptr, post_valid, res_class, res_entries,
             nullptr, res_count, ring.data(), outv.data(), cls.data(), off.data(), info.data(), tabs.data(), again.data(),
             B.pack[W].capacity / 4, &bits2, pc);
  if (bits2 != *packet_bits || memcmp(again.data(), packet, (size_t)((bits2 + 7) / 8)) != 0) *packet_bits = -1;
}

// couple / quantise / normalise with whichever form the layout needs (as launch_couple picks the kernel)
static void couple_any(const CoupleP &C, const PsyP &P, int n2, const float *const *mp, const ilog_t *const *ip, int *const *op,
                       int *nonzero, PhaseClock &pc) {
  std::vector<float> cand(n2), key(n2), sgn(n2), accp(256);
  CoupleLds L = {cand.data(), key.data(), sgn.data(), accp.data()};
  if (C.ch > 2 || C.coupling_steps > 1) {
    std::vector<float> st((size_t)4 * C.ch * n2);
    CoupleState S = {st.data(), st.data() + (size_t)C.ch * n2, st.data() + (size_t)2 * C.ch * n2,
                     (int *)(st.data() + (size_t)3 * C.ch * n2)};
    couple_block_general(C, P, n2, mp, ip, op, nonzero, L, S, pc);
  } else {
    couple_block(C, P, n2, mp, ip, op, nonzero, L, pc,
                 getenv("VAMD_EMUL_COUPLE_BAND_LOG2") ? ldexpf(1.f, atoi(getenv("VAMD_EMUL_COUPLE_BAND_LOG2"))) : VAMD_COUPLE_BAND);
  }
}

static int analyze_core(void *h, const float *pcm, int lW, int W, int nW, int blocktype, float ampmax_in, EmulTaps *t,
                        EmulMTaps *m);

extern "C" {

void *emul_open(const void *blob, size_t bytes) {
  Emul *e = new Emul;
  if (build_image(blob, bytes, &e->image, &e->doff, &e->derived, &e->err) != VAMD_OK) {
    fprintf(stderr, "emul_open: %s\n", e->err.c_str());
    delete e;
    return nullptr;
  }
  bind_params(e->image, e->doff, e->derived, e->image.data(), &e->B);
  return e;
}
void emul_close(void *h) { delete (Emul *)h; }
int emul_residue_capacity(void *h, int W) { return ((Emul *)h)->B.res_cap[W]; }
int emul_submaps(void *h, int W) { return ((Emul *)h)->B.chmap[W].submaps; }
int emul_residue_offset(void *h, int W, int sm) { return ((Emul *)h)->B.res[W][sm].ent_base; }
int emul_packet_capacity(void *h, int W) { return ((Emul *)h)->B.pack[W].capacity; }

// The packed residue table of (W, sm) -- what k_residue_chunks / k_pack_waves copy into LDS (ResP::fast, built by
// build_image) -- held against the blob's own tables, restated here field by field; and ResP::chunked against the
// conditions under which a run of eight values is closed under every stage.  Returns 0, or the number of the first
// check that fails.
int emul_residue_fast_check(void *h, int W, int sm, int *chunked_out) {
  Emul *e = (Emul *)h;
  const unsigned char *img = e->image.data();
  vamd_setup_header hd;
  memcpy(&hd, img, sizeof(hd));
  const ResP &R = e->B.res[W][sm];
  if (chunked_out) *chunked_out = R.chunked;
  if (sm >= hd.mode[W].submaps) return 0;
  const vamd_residue_tab &r = hd.res[W][sm];
  const vamd_book_tab *bk = (const vamd_book_tab *)(img + hd.off_books);
  const int *tab = R.fast;
  if (R.nparts != r.partitions || R.nstages != r.stages || R.begin != r.begin || R.tab_grouping != r.grouping) return 1;
  if (R.groupbook != r.groupbook || R.groupbook_dim != r.groupbook_dim) return 2;
  if (R.fast_ints != ((2 * r.partitions + 3) & ~3) + r.partitions * r.stages * (int)(sizeof(ResStage) / 4)) return 3;
  for (int c = 0; c < r.partitions; c++)
    if (tab[c] != r.classmetric1[c] || tab[r.partitions + c] != r.classmetric2[c]) return 4;
  const ResStage *rows = (const ResStage *)(tab + ((2 * r.partitions + 3) & ~3));
  bool dims_tile = true;
  for (int c = 0; c < r.partitions; c++)
    for (int s = 0; s < r.stages; s++) {
      const ResStage &st = rows[c * r.stages + s];
      const int bn = ((r.secondstages[c] >> s) & 1) ? r.partbooks[c][s] : -1;
      if (st.bn != bn) return 5;
      if (bn < 0) continue;
      const vamd_book_tab &b = bk[bn];
      if (st.dim != b.dim || st.minval != b.minval || st.delta != b.delta || st.quantvals != b.quantvals || st.entries != b.entries ||
          (uint32_t)st.off_lengths != b.off_lengths)
        return 6;
      if (st.nv * b.dim != r.grouping) return 7;
      const signed char *len = (const signed char *)(img + b.off_lengths);
      int used = 0;
      for (int i = 0; i < b.entries; i++) used += len[i] > 0;
      if (st.full != (used == b.entries)) return 8;
      if (8 % b.dim) dims_tile = false;
    }
  const bool want = R.covered && r.type == 2 && R.bundle == 2 && (r.grouping == 8 || r.grouping == 16 || r.grouping == 32) &&
                    r.begin % 8 == 0 && dims_tile;
  if ((R.chunked != 0) != want) return 9;
  return 0;
}

int emul_mdct_forward(void *h, int W, const float *in, float *out) {
  Emul *e = (Emul *)h;
  const XformP &P = e->B.xf[W];
  std::vector<float> A(VAMD_XF_A_FLOATS(P.n)), Bw(VAMD_XF_B_FLOATS(P.n));
  PhaseClock pc;
  pc.start(nullptr);
  static PcmTile<VAMD_HOST_QUADS> tile;
  const WaveTeam tm;
  pcm_fetch(tile, in, P.n, tm);
  window_store(P, W, 1, 1, tile, A.data(), false, tm);
  mdct_forward_wave(P, A.data(), Bw.data(), Bw.data(), pc);
  memcpy(out, Bw.data(), sizeof(float) * (P.n / 2));
  return 0;
}

int emul_analyze_block(void *h, const float *pcm, int lW, int W, int nW, int blocktype, float ampmax_in,
                       EmulTaps *t) {
  return analyze_core(h, pcm, lW, W, nW, blocktype, ampmax_in, t, nullptr);
}
int emul_analyze_block_managed(void *h, const float *pcm, int lW, int W, int nW, int blocktype, float ampmax_in,
                               EmulTaps *t, EmulMTaps *m) {
  return analyze_core(h, pcm, lW, W, nW, blocktype, ampmax_in, t, m);
}
}

static int analyze_core(void *h, const float *pcm, int lW, int W, int nW, int blocktype, float ampmax_in, EmulTaps *t,
                        EmulMTaps *m) {
  Emul *e = (Emul *)h;
  const Bound &B = e->B;
  const int ch = B.channels, n = B.bs[W], n2 = n / 2;
  const XformP &X = B.xf[W];
  const PsyP &P = B.psy[blocktype + (W ? 2 : 0)];
  const CoupleP &C = B.couple[W];
  std::vector<float> A(VAMD_XF_A_FLOATS(n)), Bw(VAMD_XF_B_FLOATS(n));
  std::vector<float> mdct_raw(ch * n2), logfft(ch * n2), logmdct(ch * n2), noise(ch * n2), tone(ch * n2),
      logmask(ch * n2), mdct(ch * n2), lmd(n2), mask(n2);
  std::vector<int> posts(ch * VAMD_POSTS_STRIDE), post_valid(ch), iwork(ch * n2), nonzero(ch), wrapped(ch * VAMD_POSTS_STRIDE);
  std::vector<ilog_t> ilogmask(ch * n2);
  std::vector<float> local(ch);
  std::vector<ilog_t> m_ilog;
  float global = ampmax_in;
  PhaseClock pc;
  pc.start(nullptr);
  for (int i = 0; i < ch; i++) {
    static PcmTile<VAMD_HOST_QUADS> tile;
    const WaveTeam tm;
    pcm_fetch(tile, pcm + (size_t)i * n, n, tm);
    transform_window(X, W, lW, nW, tile, A.data(), pc, tm);
    // the size-specialised instantiations the launcher picks, and the general one for everything else
    std::vector<float> tpack(VAMD_TPACK_FLOATS(n) > 0 ? VAMD_TPACK_FLOATS(n) : 1);
    mdct_tpack_fill(tpack.data(), X.trig, n, 0, 1);
    XformP XP = X;
    XP.tpack = tpack.data();
    switch (fixed_logn(X)) {
#define EMUL_XF(L) case L: local[i] = transform_block<L>(XP, A.data(), Bw.data(), &mdct_raw[i * n2], &logmdct[i * n2], &logfft[i * n2], pc); break;
      EMUL_XF(8) EMUL_XF(9) EMUL_XF(10) EMUL_XF(11) EMUL_XF(12)
#undef EMUL_XF
      default: local[i] = transform_block(X, A.data(), Bw.data(), &mdct_raw[i * n2], &logmdct[i * n2], &logfft[i * n2], pc);
    }
    if (local[i] > global) global = local[i];
  }
  {
    const int nlp = VAMD_LINES_PAD(P.total_octave_lines);
    std::vector<float> S(5 * VAMD_NZ_STRIDE(n2)), nz(n2), wk(n2), seed(seed_pad_lo(P.eighth_octave_lines) + nlp + seed_pad_hi(P.eighth_octave_lines), -9999.f), ampstack(nlp), flr(n2), ring_amp(VAMD_RING);
    std::vector<int> ring_pos(VAMD_RING);
    std::vector<unsigned short> surv(nlp);
    FloorScratch sc;
    for (int i = 0; i < ch; i++) {
      const FloorP &F = B.floor[W][B.chmap[W].sub[i]];
      noisemask_block(P, &logmdct[i * n2], &noise[i * n2], S.data(), pc);
      tonemask_block(P, &logfft[i * n2], &tone[i * n2], global, local[i], seed.data() + seed_pad_lo(P.eighth_octave_lines), ampstack.data(),
                     flr.data(), ring_amp.data(), ring_pos.data(), surv.data(), pc);
      offset_and_mix_wave(P, &noise[i * n2], &tone[i * n2], &mdct_raw[i * n2], &mdct[i * n2],
                          &logmask[i * n2], (unsigned short *)lmd.data(), F.twofitatten, pc);
      if (m) {
        m_ilog.resize((size_t)VAMD_PACKETBLOBS * ch * n2);
        floor_managed_block(P, F, n2, &noise[i * n2], &tone[i * n2], &mdct_raw[i * n2], (unsigned short *)lmd.data(), &sc,
                            m->posts + i * VAMD_POSTS_STRIDE, (long)ch * VAMD_POSTS_STRIDE, m->post_valid + i, ch,
                            m_ilog.data() + (size_t)i * n2, (long)ch * n2, m->nonzero + i, ch, pc);
        continue;
      }
      nonzero[i] = floor_fit_render_block(F, n2, (const unsigned short *)lmd.data(), &sc, &posts[i * VAMD_POSTS_STRIDE],
                                          &post_valid[i], &ilogmask[i * n2], pc, &wrapped[i * VAMD_POSTS_STRIDE]);
    }
  }
  {
    const float *mp[VAMD_MAX_CH];
    const ilog_t *ip[VAMD_MAX_CH];
    int *op[VAMD_MAX_CH];
    for (int i = 0; i < ch; i++) {
      mp[i] = &mdct[i * n2];
      ip[i] = &ilogmask[i * n2];
      op[i] = &iwork[i * n2];
    }
    if (m) {  // every candidate packet with its own coupling parameters, over the same spectrum
      for (int k = 0; k < VAMD_PACKETBLOBS; k++) {
        for (int i = 0; i < ch; i++) {
          ip[i] = m_ilog.data() + ((size_t)k * ch + i) * n2;
          op[i] = m->iwork + ((size_t)k * ch + i) * n2;
        }
        couple_any(B.couple_all[W].c[k], P, n2, mp, ip, op, m->nonzero + k * ch, pc);
        if (m->packets) {
          if (!B.res_cap[W]) return -130;
          std::vector<int> rc(VAMD_MAX_SUBMAPS * VAMD_RES_CLASS_STRIDE), cnt(2 * VAMD_MAX_SUBMAPS);
          std::vector<unsigned short> re(B.res_cap[W]);
          residue_and_pack(B, W, lW, nW, m->iwork + (size_t)k * ch * n2, m->nonzero + k * ch,
                           m->posts + (size_t)k * ch * VAMD_POSTS_STRIDE, m->post_valid + k * ch, rc.data(), re.data(),
                           cnt.data(), m->packets + (size_t)k * B.pack[W].capacity, m->packet_bits + k);
        }
      }
      if (t->mdct) memcpy(t->mdct, mdct.data(), sizeof(float) * mdct.size());
      if (t->logmask) memcpy(t->logmask, logmask.data(), sizeof(float) * logmask.size());
      if (t->ampmax_out) *t->ampmax_out = global;
      return 0;
    }
    couple_any(C, P, n2, mp, ip, op, nonzero.data(), pc);
  }
  if (t->res_entries && t->res_class && t->res_count) {
    if (!B.res_cap[W]) return -130;
    residue_and_pack(B, W, lW, nW, iwork.data(), nonzero.data(), posts.data(), post_valid.data(), t->res_class,
                     t->res_entries, t->res_count, t->packet, t->packet_bits, wrapped.data());
  }
#define OUT(name, vec, type) \
  if (t->name) memcpy(t->name, vec.data(), sizeof(type) * vec.size())
  OUT(mdct_raw, mdct_raw, float);
  OUT(logfft, logfft, float);
  OUT(logmdct, logmdct, float);
  OUT(noise, noise, float);
  OUT(tone, tone, float);
  OUT(logmask, logmask, float);
  OUT(mdct, mdct, float);
  OUT(posts, posts, int);
  OUT(post_valid, post_valid, int);
  if (t->ilogmask)
    for (size_t k = 0; k < ilogmask.size(); k++) t->ilogmask[k] = ilogmask[k];
  OUT(iwork, iwork, int);
  OUT(nonzero, nonzero, int);
  OUT(local_ampmax, local, float);
  if (t->ampmax_out) *t->ampmax_out = global;
  return 0;
}

extern "C" {
// the block-switching detector, one stream: the same four stages the HIP library launches
// (vamd_envelope_search_batch), run in order on the host.  pcm[ch][len].
int emul_envelope_search(void *h, const float *pcm, long len, long nsteps, vamd_envelope_state *st,
                         unsigned char *ret) {
  Emul *e = (Emul *)h;
  const EnvP &E = e->B.env;
  const int ch = e->B.channels, n = E.mdct.n, n2 = n / 2;
  if ((nsteps - 1) * E.searchstep + n > len) return -1;
  std::vector<float> near((size_t)ch * (VAMD_VE_NEAR_HIST + nsteps)), raw((size_t)ch * nsteps * VAMD_VE_SPREAD),
      amp((size_t)ch * (VAMD_VE_AMP_HIST + nsteps) * 8, 0.f);
  std::vector<uint32_t> bits(nsteps);
  const int LOGS = 2, S = 1 << LOGS;  // side-by-side transforms, as the HIP kernel runs them (it uses 16)
  std::vector<float> A((size_t)n * S), Wk((size_t)(n2 + VAMD_PW_SIZE(n2)) * S), spec((size_t)n2 * S);
  PhaseClock pc;
  pc.start(nullptr);
  for (int c = 0; c < ch; c++) {
    float *nr = near.data() + (size_t)c * (VAMD_VE_NEAR_HIST + nsteps);
    float *am = amp.data() + (size_t)c * (VAMD_VE_AMP_HIST + nsteps) * 8;
    for (int i = 0; i < VAMD_VE_NEAR_HIST; i++) nr[i] = st->near_hist[c][i];
    for (int i = 0; i < VAMD_VE_AMP_HIST * 8; i++) am[i] = st->amp_hist[c][i >> 3][i & 7];
    for (long j = 0; j < nsteps; j += S)
      env_spectrum_wave<LOGS>(E, pcm + (size_t)c * len + j * E.searchstep, nsteps - j < S ? (int)(nsteps - j) : S, A.data(),
                              Wk.data(), spec.data(), nr + VAMD_VE_NEAR_HIST + j,
                              raw.data() + ((size_t)c * nsteps + j) * VAMD_VE_SPREAD, pc);
    for (long j = 0; j < nsteps; j++) {
      const float decay = env_decay(nr + VAMD_VE_NEAR_HIST + j, (long)st->steps + j);
      for (int b = 0; b < VAMD_VE_BANDS; b++)
        am[(VAMD_VE_AMP_HIST + j) * 8 + b] =
            env_band_amp(E, raw.data() + ((size_t)c * nsteps + j) * VAMD_VE_SPREAD, decay, b);
    }
  }
  for (long j = 0; j < nsteps; j++) {
    const float *a[VAMD_MAX_CH];
    for (int c = 0; c < ch; c++) a[c] = amp.data() + ((size_t)c * (VAMD_VE_AMP_HIST + nsteps) + VAMD_VE_AMP_HIST + j) * 8;
    bits[j] = env_trigger_bits(E, a, ch, 8);
  }
  st->stretch = env_walk_wave(bits.data(), nsteps, st->stretch, ret);
  st->steps += nsteps;
  for (int c = 0; c < ch; c++) {
    const float *nt = near.data() + (size_t)c * (VAMD_VE_NEAR_HIST + nsteps) + nsteps;
    for (int i = 0; i < VAMD_VE_NEAR_HIST; i++) st->near_hist[c][i] = nt[i];
    const float *at = amp.data() + ((size_t)c * (VAMD_VE_AMP_HIST + nsteps) + nsteps) * 8;
    for (int i = 0; i < VAMD_VE_AMP_HIST * 8; i++) st->amp_hist[c][i >> 3][i & 7] = at[i];
  }
  return 0;
}
}

extern "C" {
// blockout's decisions for one stream from its detector flags: what k_plan_streams does per stream (mark bytes via
// mark_at, then plan_stream).  kind[], begin[] hold maxblocks entries; returns the number of blocks planned.
// eof: 0, or the stream's end as k_blockout.h's BlockoutP::eof; pending (optional): where the walk stopped (centerW)
int emul_plan_stream(void *h, const unsigned char *flags, long nsteps, long nsamples, int maxblocks, int *kind, int *begin, long eof,
                     long *pending) {
  Emul *e = (Emul *)h;
  BlockoutP B;
  B.bs[0] = e->B.bs[0];
  B.bs[1] = e->B.bs[1];
  blockout_set_step(B, e->B.env.searchstep);
  B.nsamples = nsamples;
  B.nsteps = nsteps;
  B.maxblocks = maxblocks;
  B.eof = eof;
  const long last = blockout_steps(B);
  std::vector<unsigned char> marks((size_t)nsteps + 4, 0);
  for (long p = 0; p < nsteps + 4; p++) marks[p] = p < last ? (unsigned char)mark_at(flags, last, p) : 0;
  std::vector<PlannedBlock> out((size_t)maxblocks);
  int n0, n1;
  long pc = 0;
  const int n = plan_stream(B, marks.data(), out.data(), &n0, &n1, &pc);
  if (pending) *pending = pc;
  for (int k = 0; k < n; k++) kind[k] = out[k].kind, begin[k] = out[k].begin;
  return n;
}

// The two ends of a stream as k_lpc_head / k_lpc_tail (vamd_kernels.h) form them, on one channel's buffer x:
//   head: x[0, head) from the first n samples x[head, head + n)   (lib/block.c:417-458; nothing when n <= 32)
//   tail: x[eof, eof + pad) from the last min(eof - start, bs1) samples before eof   (:474-512; zeros when eof - start <= 64)
void emul_lpc_head(float *x, int head, int n) {
  if (n <= 32) return;
  const int order = 16;
  std::vector<float> work((size_t)n + head), coeff(VAMD_LPC_MAX_ORDER);
  std::vector<double> aut(2 * VAMD_LPC_MAX_ORDER + 1);
  for (int j = 0; j < n; j++) work[j] = x[head + n - 1 - j];
  lpc_from_data(work.data(), n, order, aut.data(), coeff.data());
  lpc_predict(coeff.data(), work.data() + n - order, order, work.data() + n, head);
  for (int i = 0; i < head; i++) x[head - 1 - i] = work[n + i];
}
void emul_lpc_tail(float *x, long eof, long start, int bs1, int pad) {
  const int order = 32;
  if (start < 0) start = 0;
  const long have = eof - start;
  if (have > order * 2) {
    const int n = have < bs1 ? (int)have : bs1;
    std::vector<float> coeff(VAMD_LPC_MAX_ORDER), out((size_t)pad);
    std::vector<double> aut(2 * VAMD_LPC_MAX_ORDER + 1);
    lpc_from_data(x + eof - n, n, order, aut.data(), coeff.data());
    lpc_predict(coeff.data(), x + eof - order, order, out.data(), pad);
    for (int i = 0; i < pad; i++) x[eof + i] = out[i];
  } else {
    for (int i = 0; i < pad; i++) x[eof + i] = 0.f;
  }
}
}

extern "C" {
// accumulate_fit's work list (derive_fit_segments, bound as FloorP::fit_segs) against the loops of the reference
// (lib/floor1.c:601-608 calling :393-454): bin i is summed into interval j exactly when sorted_index[j] <= i <=
// min(sorted_index[j+1], n-1).  Returns the number of (bin, interval) pairs on which the two disagree, over every
// floor of both size classes of the setup.
long emul_fit_segments_check(void *h) {
  Emul *e = (Emul *)h;
  long bad = 0;
  for (int W = 0; W < 2; W++)
    for (int sm = 0; sm < VAMD_MAX_SUBMAPS; sm++) {
      const FloorP &F = e->B.floor[W][sm];
      const int n2 = e->B.psy[2 * W].n;
      std::vector<int> got((size_t)n2 * 64, 0), want((size_t)n2 * 64, 0);
      for (int j = 0; j + 1 < F.posts; j++) {
        int x1 = F.sorted_index[j + 1];
        if (x1 >= F.look_n) x1 = F.look_n - 1;
        for (int i = F.sorted_index[j]; i <= x1; i++)
          if (i < n2) want[(size_t)i * 64 + j]++;
      }
      for (int sg = 0; sg < F.fit_nseg; sg++) {
        const unsigned int *r = F.fit_segs + VAMD_FITSEG_WORDS * sg;
        const int chunk = (int)r[0], j = (int)r[1];
        for (int b = 0; b < 16; b++) {
          const unsigned int m = (r[4 + (b >> 1)] >> (16 * (b & 1))) & 0xffffu;
          if (m != 0 && m != 0xffffu) bad++;
          if (m && 16 * chunk + b < n2 && j < 64) got[(size_t)(16 * chunk + b) * 64 + j]++;
        }
      }
      for (size_t k = 0; k < got.size(); k++) bad += got[k] != want[k];
    }
  return bad;
}

// div_magic() (the kernels' one-multiply floor division) against C's "/" over the domain the floor's line walks
// use it on: every divisor den <= VAMD_DIV_MAGIC_MAX, every |dy| <= 1023, the steps k next to the ends and strided
// through the middle.  Returns the number of disagreements.
long emul_div_magic_check(void) {
  const std::vector<uint32_t> m = derive_div_magic();
  long bad = 0;
  for (int den = 1; den <= VAMD_DIV_MAGIC_MAX; den++)
    for (int ady = 0; ady <= 1023; ady++) {
      const int ks[] = {0, 1, den / 3, den / 2, den - 2, den - 1};
      for (int k : ks)
        if (k >= 0 && k < den && div_magic(k * ady, m[den]) != (k * ady) / den) bad++;
      if ((ady & 63) == 63 || ady == 1)
        for (int k = 0; k < den; k += 7)
          if (div_magic(k * ady, m[den]) != (k * ady) / den) bad++;
    }
  return bad;
}

// quant_energy() (two fused multiply-adds around a float square root) against the fp64 expression of the reference
// (quant_energy_f64), on the values where they could part: the boundaries (k + 1/2)^2 and their float neighbours,
// the squares, and a sweep of random floats.  Returns the number of disagreements.
long emul_quant_energy_check(void) {
  long bad = 0;
  auto probe = [&](float ve) {
    if (!(ve >= 0.f)) return;
    if (quant_energy(ve, 1.f) != quant_energy_f64(ve, 1.f)) bad++;
    if (quant_energy(ve, -1.f) != quant_energy_f64(ve, -1.f)) bad++;
    if (ve < 1.7e13f) {  // the GPU's square root may be one ulp off: a candidate one beside the answer must do too
      const int k = quant_energy_f64(ve, 1.f);
      if (quant_energy_from(ve, 1.f, (float)(k + 1)) != k) bad++;
      if (k > 0 && quant_energy_from(ve, 1.f, (float)(k - 1)) != k) bad++;
    }
  };
  auto around = [&](double x) {
    float f = (float)x;
    for (int s = 0; s < 4; s++) f = nextafterf(f, 0.f);
    for (int s = 0; s < 9; s++, f = nextafterf(f, INFINITY)) probe(f);
  };
  for (long k = 0; k < 6000000; k += (k < 70000 ? 1 : 997)) {
    around(((double)k + .5) * ((double)k + .5));
    around((double)k * (double)k);
  }
  uint32_t x = 12345u;
  for (int i = 0; i < 4000000; i++) {
    x = x * 1664525u + 1013904223u;
    uint32_t bits = (x >> 1) % 0x5f000000u;  // non-negative floats up to ~9e18
    float f;
    memcpy(&f, &bits, 4);
    probe(f);
  }
  return bad;
}

// k_couple's division-free forms (chan_bin_sure, couple_bin_sure) against the exact ones, aimed at the decision steps:
// |m| / f at the coupling point, at the half-integers and at their float neighbours, the coupled ratio at the squares
// of the half-integers; every estimate moved down, left alone, moved up, and by the hash.  A bin the fast form calls
// sure must equal the exact form in every field the stage reads.  Returns the disagreements; *unsure_ppm <- how often
// a random bin is sent to the exact path (parts per million).
long emul_couple_estimate_check(long *unsure_ppm) {
  long bad = 0;
  CoupleP C;
  memset(&C, 0, sizeof C);
  C.ch = 2, C.coupling_steps = 1, C.mag[0] = 0, C.ang[0] = 1;
  C.pointlimit = 100, C.prepoint = 0.35f, C.postpoint = 1.49f, C.sliding_lowpass = 1 << 30;  // (values are arbitrary floats)
  const float band = VAMD_COUPLE_BAND;
  auto same_bin = [&](const ChanBin &a, const ChanBin &b) {
    return a.fg == b.fg && a.out == b.out && !memcmp(&a.qe, &b.qe, 4) && !memcmp(&a.fl2, &b.fl2, 4) && a.re == b.re;
  };
  auto probe = [&](float m, int ilog, int b) {
    const ChanBin e = chan_bin(1, m, ilog, b, 0x7fffffff, C);
    bool unsure = false;
    const ChanBin s = chan_bin_sure(1, m, ilog, b, C, band, unsure);
    if (!unsure && !same_bin(e, s)) bad++;
    return unsure;
  };
  auto around = [&](double x, int ilog, int b) {
    float f = (float)x;
    for (int s = 0; s < 6; s++) f = nextafterf(f, 0.f);
    for (int s = 0; s < 13; s++, f = nextafterf(f, INFINITY)) probe(f, ilog, b), probe(-f, ilog, b);
  };
  auto probe_pair = [&](float qeM, float fl2M, float qeA, float fl2A, int fgM, int fgA, float sM, float sA, int b) {
    ChanBin M, A;
    M.re = sM * qeM, M.qe = qeM, M.fl2 = fl2M, M.fg = fgM, M.out = 3, M.cand = -1.f;
    A.re = sA * qeA, A.qe = qeA, A.fl2 = fl2A, A.fg = fgA, A.out = -2, A.cand = -1.f;
    ChanBin M2 = M, A2 = A;
    int iM = M.out, iA = A.out, jM = iM, jA = iA;
    couple_bin(M, A, iM, iA, b, 0x7fffffff, C);
    bool unsure = false;
    couple_bin_sure(M2, A2, jM, jA, b, C, band, unsure);
    if (!unsure && (iM != jM || iA != jA)) bad++;
    return unsure;
  };
  for (int mode = -1; mode <= 2; mode++) {
    g_estimate_nudge = mode;
    for (int ilog = 0; ilog < 256; ilog += (mode == 2 ? 1 : 5)) {
      const double f = floor1_fromdB(ilog);
      for (int b = 0; b < 200; b += 150) {
        around(f * C.prepoint, ilog, b);
        around(f * C.postpoint, ilog, b);
        for (long k = 0; k < 300000; k += (k < 3000 ? 1 : 4001)) around(f * ((double)k + .5), ilog, b);
      }
    }
    uint32_t x = 777u + mode;
    for (int i = 0; i < 300000; i++) {  // the coupled ratio at the steps: qe = (k + 1/2)^2 * (fl2M + fl2A), and neighbours
      x = x * 1664525u + 1013904223u;
      const float fa = floor1_fromdB((x >> 8) & 255), fb = floor1_fromdB((x >> 16) & 255);
      const float fl2M = fa * fa, fl2A = fb * fb, sum = fl2M + fl2A;
      const long k = (x >> 24) < 200 ? (x >> 3) % 40 : (x >> 3) % 70000;
      float q = (float)(((double)k + .5) * ((double)k + .5) * (double)sum);
      for (int s = 0; s < 4; s++) q = nextafterf(q, 0.f);
      for (int s = 0; s < 9; s++, q = nextafterf(q, INFINITY)) {
        // lossy below the point limit (M.re += A.re, qe = |re|): give A a zero so that qe is q itself
        probe_pair(q, fl2M, 0.f, fl2A, 0, 0, (x & 1) ? 1.f : -1.f, 1.f, 10);
        // lossy above it: qe = |M.re| + |A.re|
        probe_pair(q * .5f, fl2M, q * .5f, fl2A, 0, 0, (x & 1) ? 1.f : -1.f, (x & 2) ? 1.f : -1.f, 150);
        probe_pair(q, fl2M, q * .25f, fl2A, (x >> 2) & 1, 1, 1.f, -1.f, 150);  // lossless: integers only
      }
    }
  }
  g_estimate_nudge = 2;
  long unsure = 0, total = 0;
  uint32_t x = 4242u;
  for (int i = 0; i < 2000000; i++) {
    x = x * 1664525u + 1013904223u;
    const int ilog = (x >> 8) & 255;
    // |m| / f log-uniform over 2^-6 .. 2^10
    uint32_t y = x * 2654435761u;
    y ^= y >> 13;
    const float rho = ldexpf(1.f + (float)(y & 0x7fffffu) / 8388608.f, (int)((x >> 3) & 15) - 6);
    unsure += probe(rho * floor1_fromdB(ilog), ilog, (x & 1) ? 10 : 150) ? 1 : 0;
    total++;
  }
  if (unsure_ppm) *unsure_ppm = unsure * 1000000 / total;
  return bad;
}

// what the chunks of one block add up to, chunk after chunk (k_tone_chase_wave runs chase_chunk one per lane and
// combines the chunks with wave operations)
static int chase_chunks_host(const float *seeds, int linesper, int n, unsigned short *surv, int *accepted, int *rounds) {
  const int cs = (n + VAMD_CHASE_CHUNKS - 1) / VAMD_CHASE_CHUNKS;
  float ring_amp[VAMD_RING];
  int ring_pos[VAMD_RING];
  ChaseChunk r[VAMD_CHASE_CHUNKS];
  uint32_t used[VAMD_CHASE_CHUNKS];
  int nc = 0;
  for (int c = 0; c < VAMD_CHASE_CHUNKS && c * cs < n; c++, nc++) {
    const int s0 = c * cs, e0 = s0 + cs < n ? s0 + cs : n;
    // (eight lines per window, every libvorbisenc setup: the register form k_tone_seed_chase runs)
    r[c] = linesper == 8 ? chase_chunk_regs<8>(seeds, n, s0, e0, cs, VAMD_CHASE_WARM * linesper, 0)
                         : chase_chunk(seeds, linesper, n, s0, e0, VAMD_CHASE_WARM * linesper, 0, ring_amp, ring_pos, 1, 0);
    used[c] = r[c].sig_in;
  }
  int ok = 0, rd = 0;
  {  // a long run of equal values: straight to the serial walk
    int run = 0, longest = 0;
    for (int c = 0; c < nc; c++) {
      const int s0 = c * cs, e0 = s0 + cs < n ? s0 + cs : n;
      run = chase_flat_chunk(seeds, s0, e0) ? run + 1 : 0;
      if (run > longest) longest = run;
    }
    if (longest > VAMD_CHASE_FLAT_MAX) {
      *accepted = 0;
      *rounds = -1;
      return 0;
    }
  }
  for (; rd <= VAMD_CHASE_ROUNDS; rd++) {
    // (all chunks of a round look at the previous round's exits, as the lanes of a wave do)
    uint32_t prev[VAMD_CHASE_CHUNKS];
    for (int c = 0; c < nc; c++) prev[c] = c ? r[c - 1].sig_out : 0;
    int need_any = 0;
    for (int c = 0; c < nc; c++) {
      if (r[c].exact || used[c] == prev[c]) continue;
      need_any = 1;
      if (rd == VAMD_CHASE_ROUNDS) break;
      const int s0 = c * cs, e0 = s0 + cs < n ? s0 + cs : n;
      const ChaseChunk t = linesper == 8 ? chase_chunk_regs<8>(seeds, n, s0, e0, cs, -1, prev[c])
                                         : chase_chunk(seeds, linesper, n, s0, e0, -1, prev[c], ring_amp, ring_pos, 1, 0);
      used[c] = prev[c];
      r[c].popped = t.popped;
      r[c].sig_out = t.sig_out;
    }
    if (!need_any) {
      ok = 1;
      break;
    }
  }
  int ns = 0;
  for (int c = 0; c < nc; c++) {
    const int s0 = c * cs, e0 = s0 + cs < n ? s0 + cs : n;
    for (int k = 0; k < e0 - s0; k++)
      if (!((r[c].popped >> k) & 1u)) surv[ns++] = (unsigned short)(s0 + k);
  }
  *accepted = ok;
  *rounds = rd;
  return ns;
}

// seed_chase part 1 two ways over the same seed lines: the serial walk (tone_chase_thread) and the chunked one
// (chase_chunks_host); returns 1 if the survivor lists agree, and through *accepted whether the chunks verified.
int emul_chase_compare(const float *seeds, int linesper, int n, int *accepted, int *nsurv_out, int *rounds) {
  std::vector<float> ring_amp(VAMD_RING);
  std::vector<int> ring_pos(VAMD_RING);
  std::vector<unsigned short> a(n + 16), b(n + 16);
  const int na = tone_chase_thread(seeds, linesper, n, ring_amp.data(), ring_pos.data(), 1, 0, a.data());
  const int nb = chase_chunks_host(seeds, linesper, n, b.data(), accepted, rounds);
  *nsurv_out = na;
  if (!*accepted) return 1;  // (the kernel would take the serial walk)
  if (na != nb) return 0;
  for (int k = 0; k < na; k++)
    if (a[k] != b[k]) return 0;
  return 1;
}
}
