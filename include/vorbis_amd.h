/* vorbis_amd.h -- C ABI of libvorbis_amd.so, the MI355X (gfx950) implementation of
 * libvorbis' per-block encode analysis.
 *
 * Drop-in boundary: libvorbis reaches the analysis
 * through vorbis_analysis() (reference lib/analysis.c:29-63) ->
 * _mapping_P[0]->forward == mapping0_forward() (lib/mapping0.c:230-696).  The
 * entry points below replace mapping0_forward -- the numeric section
 * (lib/mapping0.c:254-576, :613-646), the residue back-end's classification and
 * VQ search and the bit-writing of the packet (:596-687) -- and the step loop of
 * the block-switching detector (lib/envelope.c:217-262); the host keeps blockout,
 * bitrate management and Ogg framing.  INTEGRATION.md shows the patch a libvorbis
 * maintainer applies.  Covered: 1 to 8 channels (mono, stereo, the 5.1 layout with
 * its two submaps and four coupling steps, and libvorbisenc's uncoupled template
 * for the rest), VBR and bitrate-managed encoders, block sizes up to 4096.
 *
 * Conventions follow libvorbis: plain C types, caller-allocated outputs, int
 * return codes with the OV_* values of include/vorbis/codec.h:221-235, no
 * callbacks, no exceptions.  A vamd_ctx, like a vorbis_dsp_state, must not be
 * used from two threads at once; distinct contexts are independent.
 */
#ifndef VORBIS_AMD_H
#define VORBIS_AMD_H

#include <stddef.h>
#include <stdint.h>
#include "vamd_setup.h"

#ifdef __cplusplus
extern "C" {
#endif

/* return codes == libvorbis OV_* (include/vorbis/codec.h:221-235) */
#define VAMD_OK        0
#define VAMD_EFAULT   (-129) /* OV_EFAULT: HIP runtime failure; see vamd_last_error() */
#define VAMD_EIMPL    (-130) /* OV_EIMPL: setup/feature outside the covered path */
#define VAMD_EINVAL   (-131) /* OV_EINVAL: bad argument */
#define VAMD_EVERSION (-134) /* OV_EVERSION: setup blob version mismatch */
/* NOT libvorbis codes: the input (not the call) was outside the domain the reference's own arithmetic is defined on ("Input
 * domain" below).  Kept apart from VAMD_EINVAL so that a binding can tell such a block -- which it reports as OV_EINVAL out
 * of vorbis_analysis() -- from a programming error in its own arguments, which it should not swallow.
 *   VAMD_EDOMAIN     finite samples, but a quantised value beyond the bound up to which the reference's integer arithmetic
 *                    is defined by C (vamd_quant_limit()): THIS block has no defined result; the stream goes on
 *   VAMD_ENONFINITE  a NaN / Inf sample: the reference's own state (ampmax chain, detector history) is not a number from
 *                    here on, and the stream is over */
#define VAMD_EDOMAIN    (-140)
#define VAMD_ENONFINITE (-141)

/* lib/codec_internal.h:23-26 */
#define VAMD_BLOCKTYPE_IMPULSE    0
#define VAMD_BLOCKTYPE_PADDING    1
#define VAMD_BLOCKTYPE_TRANSITION 0
#define VAMD_BLOCKTYPE_LONG       1

#define VAMD_POSTS_STRIDE 32   /* ints per channel in the posts[] output (>= 29 posts at 44.1 kHz) */
#define VAMD_AMPMAX_FLOOR (-9999.f) /* a fresh vorbis_block's ampmax, lib/block.c:86 */

typedef struct vamd_ctx vamd_ctx;

/* Replaces the table-building half of vorbis_analysis_init() for the GPU side
 * (lib/block.c:296-311 -> _vds_shared_init :170-293): validates the setup blob
 * produced by vamd_pack_setup() (integration/vamd_pack_setup.c), copies it to
 * HBM on `device` and derives the static index tables the kernels use.
 * `device` < 0 keeps the calling thread's current HIP device. */
int vamd_create_abi(vamd_ctx **ctx, const void *setup_blob, size_t blob_bytes, int device, int caller_abi_version);
/* The entry point is vamd_create(); it hands the library the VAMD_ABI_VERSION of the header the CALLER was compiled
 * against, and the library refuses any other than its own (VAMD_EVERSION): the descriptor structs below grow between
 * releases, and a caller built against a shorter one would have the library read past the end of what it wrote.  (There
 * is deliberately no plain `vamd_create` symbol: a binary built before this rule fails to link instead of to behave.) */
#define vamd_create(ctx, setup_blob, blob_bytes, device) \
  vamd_create_abi((ctx), (setup_blob), (blob_bytes), (device), VAMD_ABI_VERSION)

/* Counterpart of vorbis_dsp_clear() (lib/block.c:340-388) for the GPU state. */
void vamd_destroy(vamd_ctx *ctx);

/* Text of the last failure on this context ("" if none). */
const char *vamd_last_error(const vamd_ctx *ctx);

/* All launches of this context go to `hip_stream` (a hipStream_t; NULL = the
 * null stream).  The caller owns the stream. */
int vamd_set_stream(vamd_ctx *ctx, void *hip_stream);

/* Pre-size the internal HBM workspace for batches of up to `max_blocks` blocks
 * of size class W so that no allocation happens inside a timed region. */
int vamd_reserve(vamd_ctx *ctx, int W, long max_blocks);

/* Measurement hook (no libvorbis counterpart): when enabled, a HIP event is
 * recorded on the context's stream before the first stage kernel of a batch and
 * after every stage.  vamd_stage_ms() synchronises the stream and returns, per
 * stage, the summed elapsed milliseconds of all batches issued since the last
 * call -- eight stages: 0 transform, 1 ampmax, 2 noisemask, 3 tonemask, 4 floor,
 * 5 couple, 6 residue search, 7 packet assembly (stages a batch did not run stay
 * 0; with nstages < 8 the later ones are dropped) -- and the number of batches in
 * *runs (a mixed-size call counts once; both size classes add to the same stage). */
int vamd_profile(vamd_ctx *ctx, int enable);
int vamd_stage_ms(vamd_ctx *ctx, float *ms, int nstages, int *runs);

/* Finer measurement hook: arm (enable=1) / disarm (0) an in-kernel stopwatch; while armed lane 0
 * of every wave adds the shader-clock ticks each phase took to one of 80 slots (16 per stage:
 * transform 0.., noisemask 16.., tonemask 32.., floor 48.., couple 64..).  If out80 is non-NULL
 * the slots accumulated so far are copied out first. */
int vamd_debug_cycles(vamd_ctx *ctx, int enable, unsigned long long *out80);

/* Measurement aid (no libvorbis counterpart): the shader clock while the chip is busy.  While `acc3` (device memory,
 * three 64-bit words zeroed by the caller) is set, every batch of more than a few thousand blocks at the masking level
 * or above has the first wave of its tone stack walk -- which runs beside the noise mask, the path's longest stage --
 * add the shader ticks of its own life (s_memtime) to acc3[0], the ticks of the chip-wide 100 MHz clock (s_memrealtime)
 * to acc3[1] and 1 to acc3[2].  acc3[0] / acc3[1] x 100 MHz is the clock the vector units ran at -- what bench.py prices
 * roofline.valu with instead of a nominal figure.  NULL switches it off.  No launch, no stream of its own. */
int vamd_clock_probe(vamd_ctx *ctx, unsigned long long *acc3);

/* Calibration aid for counter passes (no libvorbis counterpart): copy `bytes` (a multiple of 16) from `src` to `dst`
 * (device pointers) with the library's own kernel k_calib_copy, 16 bytes per lane -- exactly `bytes` read and
 * `bytes` written under a name a profile can find, so that FETCH_SIZE / WRITE_SIZE are scaled by a measured factor
 * (tools/prof_run.py, tools/make_profiles.py) instead of an assumed one. */
int vamd_calib_copy(vamd_ctx *ctx, void *dst, const void *src, size_t bytes);

int vamd_channels(const vamd_ctx *ctx);
int vamd_blocksize(const vamd_ctx *ctx, int W);
int vamd_posts(const vamd_ctx *ctx, int W);

/* ---- batched device API (all pointers are DEVICE pointers) ------------------
 *
 * One batch = `nblocks` blocks of the same size class W, block-major:
 * pcm[nblocks][ch][n], every per-bin output [nblocks][ch][n/2].
 */

/* mdct_forward(lookup,in,out), reference lib/mdct.c:492-562, for `nframes`
 * independent n-sample frames (no window applied; BASELINE config 2).
 * in[nframes][n] -> out[nframes][n/2]. */
int vamd_mdct_forward_batch(vamd_ctx *ctx, int W, const float *in, float *out, long nframes);

/* Per-block descriptors.  The arrays (device, length nblocks) may be NULL, in
 * which case the uniform_* value applies to every block. */
typedef struct vamd_batch_desc {
  int         W;                 /* size class of every block in the batch (vb->W) */
  long        nblocks;
  const int32_t *lW, *nW;        /* vb->lW, vb->nW (lib/block.c:593-595) */
  const int32_t *blocktype;      /* vbi->blocktype (lib/block.c:597-615) */
  const float   *ampmax_in;      /* vbi->ampmax on entry (lib/mapping0.c:244) */
  int         uniform_lW, uniform_nW, uniform_blocktype;
  float       uniform_ampmax_in;
} vamd_batch_desc;

/* Outputs; any pointer may be NULL (that tensor is then kept internal or not
 * produced).  Names follow mapping0_forward's variables / its #if 0
 * _analysis_output taps (lib/mapping0.c:279-657). */
typedef struct vamd_batch_io {
  const float *pcm;       /* in  [nb][ch][n]   vb->pcm, un-windowed; not modified */
  float   *mdct_raw;      /* out [nb][ch][n/2] mdct_forward output (tap "mdct", pre-M1) */
  float   *logfft;        /* out [nb][ch][n/2] (tap "fft") */
  float   *logmdct;       /* out [nb][ch][n/2] (tap "mdct" in dB) */
  float   *noise;         /* out [nb][ch][n/2] _vp_noisemask (tap "noise") */
  float   *tone;          /* out [nb][ch][n/2] _vp_tonemask (tap "tone") */
  float   *logmask;       /* out [nb][ch][n/2] _vp_offset_and_mix select 1 (tap "mask1") */
  float   *mdct;          /* out [nb][ch][n/2] gmdct after AoTuV-M1: the spectrum that is quantised */
  int32_t *posts;         /* out [nb][ch][VAMD_POSTS_STRIDE] floor1_fit result, bit 15 = unused flag */
  int32_t *post_valid;    /* out [nb][ch] 0 where floor1_fit returns NULL (all-zero floor) */
  int32_t *ilogmask;      /* out [nb][ch][n/2] integer floor curve of floor1_encode (tap "maskI") */
  int32_t *iwork;         /* out [nb][ch][n/2] quantised, coupled residue (tap "res") */
  int32_t *nonzero;       /* out [nb][ch] after _vp_couple_quantize_normalize's fix-up */
  float   *local_ampmax;  /* out [nb][ch] */
  float   *ampmax_out;    /* out [nb] vbi->ampmax on exit (lib/mapping0.c:576) */
  /* residue back-end, level FULL only, all three or none: what res2_class
   * decides and, in the order res2_forward emits them, the codebook entries its lattice search
   * picks (lib/res0.c:479-532,783-809,534-640,322-410).  Available when vamd_residue_capacity() > 0. */
  int32_t  *res_class;    /* out [nb][S][VAMD_RES_CLASS_STRIDE] class of each partition (S = vamd_submaps(ctx, W);
                             type 1 over several channels: index = partition * coded channels + channel) */
  uint16_t *res_entries;  /* out [nb][vamd_residue_capacity(ctx, W)] entry numbers in emission order, (stage,
                             partition, channel, vector); a second submap's start at vamd_residue_offset(ctx, W, 1) */
  int32_t  *res_count;    /* out [nb][S][2] {classes (0 = nothing to code), entries} of each submap */
  /* packet assembly, level FULL only, both or none: the block's finished audio
   * packet -- header bits, floor1_encode's writes, the residue's phrase words and codewords, packed
   * LSb first exactly as oggpack_write would (lib/mapping0.c:598-606, lib/floor1.c:833-921,
   * lib/res0.c:534-640).  Available when vamd_packet_capacity() > 0. */
  uint8_t  *packets;      /* out [nb][packet_stride] bytes; the packet is the first (bits+7)/8 of a row */
  int32_t  *packet_bits;  /* out [nb] oggpack_bits(); > 8*packet_stride: the row was too short, packet cut off */
  int64_t   packet_stride;/* row length in bytes, a multiple of 4 (vamd_packet_capacity() always suffices) */
  uint8_t  *status;       /* out [nb][ch] 0, or a set of VAMD_STATUS_* bits where the channel-block was outside the input
                             domain (below) */
  /* blocks read in place (ABI 7).  With pcm_src non-NULL the batch is NOT a packed [nb][ch][n] array: block b, channel c
   * starts at pcm + pcm_src[b] + c * pcm_channel_stride (floats; both multiples of 4, pcm 16-byte aligned) -- e.g. a
   * vamd_stream_plan's src[W] over the stream buffers themselves, which spares vamd_gather_blocks and its copy of every
   * sample.  Blocks may overlap (consecutive blocks of a stream share half their samples). */
  const int64_t *pcm_src;       /* in  device [nb], or NULL: packed */
  int64_t   pcm_channel_stride; /* in  floats between a block's channels (only with pcm_src) */
} vamd_batch_io;
/* vamd_batch_desc / vamd_batch_io / vamd_managed_io MUST be zero-initialised by the caller (memset, = {0}) before the
 * fields it uses are set: members are appended at the END between releases (`status` came in with ABI 6, `pcm_src` with
 * 7), and a member the caller does not set is then a null pointer, which every entry point reads as "not wanted".  A
 * caller built against another header than the library it loads is refused by vamd_create() (VAMD_EVERSION) and can ask
 * beforehand with vamd_abi_version() != VAMD_ABI_VERSION. */
#define VAMD_ABI_VERSION 9
int vamd_abi_version(void);
/* GPUs the HIP runtime shows this process (hipGetDeviceCount; < 0: an OV_*-valued error) -- for plain-C hosts that spread
 * a vamd_feed / vamd_batcher over all of them without linking the runtime themselves. */
int vamd_device_count(void);
/* The environment knobs in force for a context, as "NAME=value" words (read once, at vamd_create; vorbis_amd/csrc/vamd_knobs.h).  The
 * operating knobs (VAMD_VERBOSE, VAMD_BATCH_LANES / _EAGER / _JOIN / _SPIN_BELOW) are always honoured; the test knobs --
 * kernel choices turned the other way, widened margins, occupancy caps, injected failures -- only beside
 * VAMD_TEST_KNOBS=1, so that an inherited environment cannot change what a drop-in library does. */
const char *vamd_config_string(const vamd_ctx *ctx);

/* ---- Input domain ---------------------------------------------------------------------------------
 * libvorbis does not validate PCM: whatever floats arrive go through mapping0_forward.  Inside the domain below this
 * library reproduces the reference bit for bit -- denormals, signed zeros and signals thousands of times over full
 * scale included (tests/soak_lib.py kinds 8-13).  Outside it the reference's own result is not defined by C, there is
 * nothing to be identical to, and the library REPORTS such input instead of inventing an answer.  The domain has two
 * edges, and they are the reference's own:
 *
 *   (1) finite arithmetic.  A NaN or +-Inf sample inside a block's windowed span (a sample the window zeroes --
 *       lib/window.c:2117-2118 -- never enters the arithmetic, in the reference or here) reaches the reference's
 *       float -> int conversions (lib/psy.c:958-962) and stays in its ampmax chain and detector history for the rest
 *       of the stream.  Finite samples do the same only beyond ~3e16 x full scale, where the reference's own fp32
 *       power spectrum overflows to Inf (lib/mapping0.c:323-343).  Test: the block's spectral peak on the reference's
 *       logfft scale (0 dB = a full-scale sine, before the clamp of :345) above +330 dB -- any NaN / Inf puts it
 *       there, todB() reads a float's bits -- one compare per channel-block.  VAMD_STATUS_NONFINITE.
 *   (2) the reference's integers.  Its quantised values are float -> int conversions (lib/psy.c:958-962): defined by C
 *       below 2^31 (VAMD_QUANT_LIMIT_INT).  Where noise normalisation is at work (q < 0.4 at 44.1 kHz; bins from
 *       normal_start on) it squares them in an `int` (:985): defined up to VAMD_QUANT_LIMIT_SQUARE = 46 340.  Its
 *       residue search sums up to eight squared differences in an `int` (lib/res0.c:361-364): defined while every
 *       value AT A POSITION THE RESIDUE CODES stays within a bound Q that depends only on the setup's codebooks --
 *       floor(sqrt(INT_MAX / dim)) less the lattice reach of the last residue class's cascade:
 *       vorbis_amd/csrc/vamd_bind.h, derive_quant_limit(), holds the proof, and vamd_quant_limit() returns Q with the
 *       bins it holds at (13 000 - 32 000 for the libvorbisenc setups: spectra ~ +85 ... +90 dB over full scale;
 *       un-normalised int16-scale floats are beyond it).  Tests: the coupling stage holds every value it writes
 *       against the first two bounds (level FULL; the levels below it form no integers); the residue search holds
 *       every value it loads from a coded position against the third (wherever it runs: res_* or packets asked for --
 *       a caller that takes `iwork` to a residue coder of its own makes that test itself, as the binding does).
 *       VAMD_STATUS_RANGE.  Up to the bounds everything is exact -- the "+60 dB" line of ABI 7 was a sufficient
 *       margin, not the edge; the edge is now the arithmetic's own.  (Ordinary input gets nowhere near Q where a
 *       codebook is searched; elsewhere it does: the reference's own test signal, a full-scale sine, quantises to
 *       ~66 000 at q 0.85 in a 5.1 stream's LFE channel, whose floor and residue end at bin 12 -- coded by nobody, squared
 *       by nobody, and the reference's defined result: tests/test_reference_matrix.py.)
 *
 *   - the host-pointer calls (vamd_analyze_block*, vamd_encode_block, vamd_batcher_encode_block) return
 *     VAMD_ENONFINITE for (1), VAMD_EDOMAIN for (2) (their argument errors stay VAMD_EINVAL); *ampmax_out is
 *     delivered either way (it comes out of the block's FFT).  vamd_envelope_search knows (1) only: a finite stream is
 *     cut into the reference's blocks whatever its level.  Through the binding (integration/mapping0_vamd.c)
 *     vorbis_analysis() returns OV_EINVAL for a block outside the domain.  For (2) that is THIS block only: the
 *     ampmax chain is carried over it and every later block of the stream is again the reference's, bit for bit.  For
 *     (1) it is this block and every later one -- the deviation from upstream that remains deliberate: libvorbis goes
 *     on encoding with NaN in its state; this back-end ends the stream;
 *   - the device-pointer calls are asynchronous: they fill `status` (when given) and count; vamd_input_status()
 *     synchronises the context's stream, returns VAMD_ENONFINITE / VAMD_EDOMAIN if anything issued since the previous
 *     call was outside the domain (how many channel-blocks of either kind / detector steps: the two optional outputs;
 *     the fifteen candidate packets of a bitrate-managed block may count one channel-block up to fifteen times) and
 *     resets the counts.  Outputs of such blocks are deterministic but unspecified; every other block of the batch is
 *     unaffected. */
#define VAMD_STATUS_RANGE     1 /* bits of status[] */
#define VAMD_STATUS_NONFINITE 2
/* the bounds of (2) for size class W and channel `channel` (any pointer may be NULL): the return value Q holds at the
 * bins [*first_bin, *end_bin) of the channel that its residue codes; VAMD_QUANT_LIMIT_SQUARE at the bins from
 * *square_bin on (n/2: nowhere -- noise normalisation is off for this size class); VAMD_QUANT_LIMIT_INT everywhere. */
#define VAMD_QUANT_LIMIT_SQUARE 46340
#define VAMD_QUANT_LIMIT_INT 0x7fffff80
int vamd_quant_limit(const vamd_ctx *ctx, int W, int channel, int *first_bin, int *end_bin, int *square_bin);
int vamd_input_status(vamd_ctx *ctx, long *bad_channel_blocks, long *bad_detector_steps);

#define VAMD_RES_CLASS_STRIDE 512 /* ints per block and submap in res_class[] (>= classified partitions) */

/* Entries one block of size class W can emit at most (the row length of res_entries), or 0 when the
 * mode's residue is not covered on the GPU (residue types 1 and 2 with vectors of <= 8 dimensions are). */
int vamd_residue_capacity(const vamd_ctx *ctx, int W);
/* Submaps of the mode (1; 2 for the 5.1 layout: the full-range channels, then the LFE), and where in a
 * block's res_entries row submap `sm`'s entries start. */
int vamd_submaps(const vamd_ctx *ctx, int W);
int vamd_residue_offset(const vamd_ctx *ctx, int W, int submap);

/* Bytes the longest possible packet of size class W takes (a multiple of 4; worst case over every
 * field's longest codeword), or 0 when packets of this mode are not assembled on the GPU (they are
 * wherever vamd_residue_capacity() > 0). */
int vamd_packet_capacity(const vamd_ctx *ctx, int W);

/* how far down mapping0_forward the batch runs */
#define VAMD_LEVEL_TRANSFORM 1 /* window, MDCT, FFT, logfft/logmdct, local ampmax (lib/mapping0.c:254-360,384) */
#define VAMD_LEVEL_PSY       2 /* + _vp_noisemask, _vp_tonemask (:417-440)           [BASELINE config 3] */
#define VAMD_LEVEL_FULL      3 /* + offset_and_mix, floor1_fit, floor render, couple/quantise (:463-646) [config 4] */

int vamd_analyze_batch(vamd_ctx *ctx, const vamd_batch_desc *desc, const vamd_batch_io *io, int level);

/* A real stream: blocks in stream order whose ampmax chains from block to block
 * (lib/block.c:626-628, lib/psy.c:837-848, lib/mapping0.c:244,346,576).  Same
 * as vamd_analyze_batch(level FULL) except desc->ampmax_in is ignored: block 0
 * starts from `ampmax_state` (host float, in/out: vorbis_look_psy_global.ampmax
 * semantics) and block k+1 receives decay(ampmax_out[k]).  W of block k is
 * desc->W (one size class per call); mixed streams are issued as consecutive
 * calls per run of equal W. */
int vamd_analyze_stream(vamd_ctx *ctx, const vamd_batch_desc *desc, const vamd_batch_io *io,
                        float *ampmax_state);

/* A real stream that mixes both block sizes (BASELINE config 5): the host's
 * vorbis_analysis_blockout() decides (lW, W, nW, blocktype) per block as before; the blocks are
 * handed over bucketed by size class, and `order[k]` (device, length nblocks_total) names the k-th
 * block of the stream: bit 30 = its size class W, bits 0..29 = its index inside that class's
 * batch.  The ampmax chain runs through the blocks in stream order with each block's own decay
 * (lib/psy.c:842: secs = blocksize[W]/2/rate).  Either batch may be empty. */
int vamd_analyze_stream_mixed(vamd_ctx *ctx, const vamd_batch_desc *desc_short, const vamd_batch_io *io_short,
                              const vamd_batch_desc *desc_long, const vamd_batch_io *io_long, const int32_t *order,
                              long nblocks_total, float *ampmax_state);

/* The same for MANY streams in one call (an encoder farm's batch): the blocks of all streams are bucketed
 * by size class as above; order[] lists them stream after stream, each stream's in its own stream order,
 * and stream s owns order[stream_start[s] .. stream_start[s+1]) (device int64 [nstreams+1], stream_start[0]
 * = 0, stream_start[nstreams] = nblocks_total).  ampmax_states (device float [nstreams]) holds every
 * stream's vorbis_look_psy_global.ampmax on entry and is updated in place; the chains run one thread per
 * stream.  Asynchronous on the context's stream like vamd_analyze_batch. */
int vamd_analyze_streams_mixed(vamd_ctx *ctx, const vamd_batch_desc *desc_short, const vamd_batch_io *io_short,
                               const vamd_batch_desc *desc_long, const vamd_batch_io *io_long, const int32_t *order,
                               const int64_t *stream_start, long nstreams, long nblocks_total, float *ampmax_states);


/* ---- per-block host API: the compatibility path behind vorbis_analysis() -----
 * Host pointers.  pcm[ch] -> n samples each (vb->pcm); outputs sized as above for
 * nblocks == 1.  Latency-bound by design (one launch sequence + two PCIe
 * copies per block); throughput users batch. */
int vamd_analyze_block(vamd_ctx *ctx, const float *const *pcm, int lW, int W, int nW, int blocktype,
                       float ampmax_in, float *mdct /*[ch][n/2]*/, float *logmask /*[ch][n/2] or NULL*/,
                       int32_t *posts /*[ch][VAMD_POSTS_STRIDE]*/, int32_t *post_valid /*[ch]*/,
                       int32_t *iwork /*[ch][n/2]*/, int32_t *nonzero /*[ch]*/, float *ampmax_out);

/* ---- the block-switching detector --------------------------------
 * Replaces the step loop of _ve_envelope_search() (lib/envelope.c:217-262) with its _ve_amp()
 * calls (:89-215): one detector step per `searchstep` (64) samples, each reading `winlength`
 * (128) samples of every channel.  Per step the reference ORs three flags: 1|4 = pre-echo
 * trigger, 2 = post-echo trigger; the caller applies them to ve->mark[] exactly as
 * lib/envelope.c:241-258 does (INTEGRATION.md shows the binding) -- cursor/testW logic, which
 * decides block sizes from the marks, stays host code.
 *
 * vamd_envelope_state carries what the reference keeps in envelope_filter_state + ve->stretch
 * (lib/envelope.h:34-45,66), as plain histories: all-zero == a fresh stream (the reference
 * calloc's its state).  A stream may be fed in calls of any length. */
#define VAMD_VE_NEAR_HIST 30  /* near-DC terms a step can reach back to (two refresh periods of 15) */
#define VAMD_VE_AMP_HIST  16  /* band amplitudes a step can reach back to (13), padded */
typedef struct vamd_envelope_state {
  int64_t steps;    /* detector steps consumed so far (nearptr == steps % 15) */
  int32_t stretch;  /* ve->stretch */
  int32_t pad;
  float near_hist[VAMD_MAX_CH][VAMD_VE_NEAR_HIST];   /* oldest first */
  float amp_hist[VAMD_MAX_CH][VAMD_VE_AMP_HIST][8];  /* oldest first; 7 bands + 1 pad */
} vamd_envelope_state;

/* Batch form, everything device-resident.  Stream s, channel c, step j reads
 * pcm[s*stream_stride + c*channel_stride + j*searchstep .. + winlength).  `states` [nstreams]
 * is read and updated in place; ret[s*nsteps + j] receives the step's flags (0..7). */
int vamd_envelope_search_batch(vamd_ctx *ctx, const float *pcm, long stream_stride, long channel_stride,
                               long nstreams, long nsteps, vamd_envelope_state *states, unsigned char *ret);

/* One stream from host memory (the per-call compatibility form used by the libvorbis binding):
 * pcm[c] points at the first sample of the first step; state and ret are host memory. */
int vamd_envelope_search(vamd_ctx *ctx, const float *const *pcm, long nsteps, vamd_envelope_state *state,
                         unsigned char *ret);

/* ---- bitrate-managed blocks ----------------------------------------------
 * With a bitrate manager (vorbis_encode_init; vorbis_bitrate_managed()) mapping0_forward prepares
 * PACKETBLOBS = 15 candidate packets per block and lets vorbis_bitrate_addblock() pick one: two more
 * floor fits on the lower / higher noise curves (offset_select 0 / 2), twelve interpolated floors
 * (floor1_interpolate_fit), and for every candidate its own floor curve, couple/quantise with that
 * candidate's coupling parameters, and residue (lib/mapping0.c:507-573,596-687).  The spectrum, the
 * select-1 mask and ampmax are shared.  Outputs are laid out [block][candidate][channel][...]. */
typedef struct vamd_managed_io {
  int32_t  *posts;       /* out [nb][15][ch][VAMD_POSTS_STRIDE] floor_posts[i][k]                (required) */
  int32_t  *post_valid;  /* out [nb][15][ch] 0 where floor_posts[i][k] is NULL                   (required) */
  int32_t  *iwork;       /* out [nb][15][ch][n/2] quantised, coupled residue of candidate k      (required) */
  int32_t  *nonzero;     /* out [nb][15][ch]                                                     (required) */
  int32_t  *res_class;   /* out [nb][15][S][VAMD_RES_CLASS_STRIDE]   optional, all three or none */
  uint16_t *res_entries; /* out [nb][15][vamd_residue_capacity(ctx, W)] */
  int32_t  *res_count;   /* out [nb][15][S][2] */
  uint8_t  *packets;     /* out [nb][15][packet_stride]           optional, with packet_bits (as vamd_batch_io) */
  int32_t  *packet_bits; /* out [nb][15] */
  int64_t   packet_stride;
} vamd_managed_io;

/* vamd_analyze_batch(level FULL) for bitrate-managed blocks: `io` carries pcm and the shared outputs
 * (mdct, logmask, ampmax_out, the psy taps; its per-candidate fields posts/post_valid/ilogmask/iwork/
 * nonzero/res_* are ignored), `m` the per-candidate ones.  Works on any setup blob: the fifteen
 * candidates' parameters are part of every libvorbisenc setup. */
int vamd_analyze_batch_managed(vamd_ctx *ctx, const vamd_batch_desc *desc, const vamd_batch_io *io,
                               const vamd_managed_io *m);

/* One managed block from host memory (the binding's call): mdct [ch][n/2] and ampmax_out are shared,
 * posts [15][ch][VAMD_POSTS_STRIDE], post_valid / nonzero [15][ch], iwork [15][ch][n/2]; the res_*
 * arrays ([15][...] as above) may be NULL. */
int vamd_analyze_block_managed(vamd_ctx *ctx, const float *const *pcm, int lW, int W, int nW, int blocktype,
                               float ampmax_in, float *mdct, float *ampmax_out, int32_t *posts,
                               int32_t *post_valid, int32_t *iwork, int32_t *nonzero, int32_t *res_class,
                               uint16_t *res_entries, int32_t *res_count);

/* vamd_analyze_block() plus the residue back-end's decisions for the block (host pointers; any may be
 * NULL).  res_entries must hold vamd_residue_capacity(ctx, W) entries, res_class S * VAMD_RES_CLASS_STRIDE
 * ints, res_count S * 2 ints (S = vamd_submaps(ctx, W)).  Fails with VAMD_EIMPL when res_* are asked for a mode that is not covered. */
int vamd_analyze_block_res(vamd_ctx *ctx, const float *const *pcm, int lW, int W, int nW, int blocktype,
                           float ampmax_in, float *mdct, float *logmask, int32_t *posts, int32_t *post_valid,
                           int32_t *iwork, int32_t *nonzero, float *ampmax_out, int32_t *res_class,
                           uint16_t *res_entries, int32_t *res_count);

/* One block from host memory all the way to its packet(s): what mapping0_forward leaves in
 * vbi->packetblob[] (lib/mapping0.c:593-687).  managed == 0: one packet (candidate PACKETBLOBS/2, the
 * VBR case); managed != 0: all 15 candidates of a bitrate-managed block.  packets [1 or 15][packet_stride]
 * and packet_bits [1 or 15] are host memory; rows need vamd_packet_capacity(ctx, W) bytes.  Fails with
 * VAMD_EIMPL where vamd_packet_capacity() is 0. */
int vamd_encode_block(vamd_ctx *ctx, const float *const *pcm, int lW, int W, int nW, int blocktype, float ampmax_in,
                      int managed, float *ampmax_out, uint8_t *packets, long packet_stride, int32_t *packet_bits);

/* The same for `nblocks` CONSECUTIVE blocks of one stream, in stream order, in one launch sequence -- for a caller that
 * already holds the samples of several blocks: a libvorbis application that writes more than a block's worth per
 * vorbis_analysis_wrote() (lib/block.c:470-532) has determined every block its buffer covers, and the binding looks ahead
 * (integration/mapping0_vamd.c).  managed == 0: one packet per block (VBR); managed != 0: the fifteen
 * candidate packets of a bitrate-managed block, as vamd_encode_block.  Host memory throughout:
 *   pcm[b * ch + c]   channel c of block b (blocksize[W[b]] samples), read during the call only
 *   lW / W / nW / blocktype [nblocks]   what vorbis_analysis_blockout decides per block (lib/block.c:589-611)
 *   ampmax_in_first   vbi->ampmax of block 0 as blockout left it; block b > 0 receives _vp_ampmax_decay of block b-1's
 *                     result exactly as the next blockout would hand it on (lib/block.c:626-628, lib/psy.c:837-848) --
 *                     ampmax_in[b] (optional out) is what each block received, ampmax_out[b] (optional) what it left
 *   packets [nblocks][1 or 15][packet_stride], packet_bits [nblocks][1 or 15]   as vamd_encode_block (rows need
 *                     vamd_packet_capacity of the larger size class)
 *   verdict [nblocks] VAMD_OK, or VAMD_EDOMAIN / VAMD_ENONFINITE for a block outside the input domain (no packet; its
 *                     ampmax still feeds the chain, as the reference's would)
 * Returns VAMD_OK when the batch ran; per-block trouble is in verdict[]. */
int vamd_encode_blocks(vamd_ctx *ctx, long nblocks, const float *const *pcm, const int32_t *lW, const int32_t *W,
                       const int32_t *nW, const int32_t *blocktype, float ampmax_in_first, int managed, float *ampmax_in,
                       float *ampmax_out, uint8_t *packets, long packet_stride, int32_t *packet_bits, int32_t *verdict);

/* winlength / searchstep of the detector (128 / 64 in every libvorbis setup). */
int vamd_envelope_geometry(const vamd_ctx *ctx, int *winlength, int *searchstep);

/* ---- device-resident stream control (reference lib/block.c:534-693 vorbis_analysis_blockout's decisions,
 * lib/envelope.c:262-353 cursor walk and _ve_envelope_mark): whole streams in, block lists out, no host round
 * trip per block.
 *
 * vamd_plan_streams: `nstreams` streams of `nsamples` samples per channel, device-resident -- stream s, channel c
 * at pcm + s*stream_stride + c*channel_stride -- each laid out as the encoder's own PCM buffer would be had
 * nothing been shifted out of it: what vorbis_analysis_buffer()/vorbis_analysis_wrote() accumulate, the
 * start-of-stream pre-extrapolation included (lib/block.c:398-458: host code in the reference, the caller's
 * here), so the first block is centred at blocksizes[1]/2.  Runs the block-switching detector over every
 * stream (as vamd_envelope_search_batch; `states` [nstreams], device, all-zero = fresh streams, updated),
 * then one thread per stream replays blockout's size / window / blocktype decisions and the plan's device
 * arrays are filled.  A plan covers the blocks the reference hands out while the given data suffices
 * (eofflag == 0); a caller closing a stream appends the reference's end-of-stream tail (lib/block.c:486-532)
 * to the buffer first.  Synchronises the context's stream once (the block counts come back to size the
 * outputs).  The plan's arrays belong to the context and stay valid until the next vamd_plan_streams on it.
 *
 * vamd_gather_blocks: the planned blocks of size class W copied out of the streams into a batch
 * pcm_blocks[nblocks[W]][ch][blocksize[W]] (device), the layout vamd_analyze_streams_mixed takes -- whose
 * descriptor arrays, order[] and stream_start[] are the plan's. */
typedef struct vamd_stream_plan {
  int64_t nstreams;
  int64_t nblocks[2];            /* blocks of each size class over all streams */
  const int32_t *lW[2], *nW[2], *blocktype[2];   /* device, per size class [nblocks[W]] */
  const int64_t *src[2];         /* device [nblocks[W]]: offset of the block's first sample from `pcm` (channel 0) */
  const int32_t *order;          /* device [nblocks[0] + nblocks[1]]: W << 30 | index, stream after stream */
  const int64_t *stream_start;   /* device [nstreams + 1] into order[] */
} vamd_stream_plan;

int vamd_plan_streams(vamd_ctx *ctx, const float *pcm, long stream_stride, long channel_stride, long nstreams,
                      long nsamples, vamd_envelope_state *states, vamd_stream_plan *plan);
int vamd_gather_blocks(vamd_ctx *ctx, const vamd_stream_plan *plan, int W, const float *pcm, long channel_stride,
                       float *pcm_blocks);
/* vamd_plan_streams for COMPLETE streams, the two ends included (ABI 9): what the application loop of
 * examples/encoder_example.c:179-236 makes of a stream it writes 1024 frames at a time and then closes with
 * vorbis_analysis_wrote(v, 0).  Stream s, channel c occupies pcm + s*stream_stride + c*channel_stride, laid out
 *   [ blocksizes[1]/2 samples of room | nframes real samples | 3 * blocksizes[1] samples of room ]
 * (channel_stride >= their sum).  The call fills the room in front with the reference's backward LPC extrapolation of
 * the stream's first samples (_preextrapolate_helper, lib/block.c:417-458: order 16, from the first blocksizes[1] + 1024
 * samples), the room behind with its forward extrapolation of the last ones (lib/block.c:474-512: order 32, from as much
 * of the last long block as the encoder still holds when the stream is closed -- which depends on where its block walk
 * stands, so the walk is run up to there first), takes the detector over all of it and plans every block up to and
 * including the stream's last (vb->eofflag, lib/block.c:664-670).  lib/lpc.c's arithmetic -- serial fp64 sums,
 * Levinson-Durbin, the fp32 predictor chain -- is reproduced operation for operation.  `states` as vamd_plan_streams
 * (all-zero on entry). */
int vamd_plan_streams_whole(vamd_ctx *ctx, float *pcm, long stream_stride, long channel_stride, long nstreams,
                            long nframes, vamd_envelope_state *states, vamd_stream_plan *plan);
/* The same for streams of UNEQUAL length: nframes[s] (HOST array [nstreams], each 1 .. max_frames) real samples in stream s,
 * every buffer laid out for max_frames (the room behind a shorter stream's samples starts where ITS samples end and must be
 * zero up to the buffer's end).  One launch sequence for all: every stream gets its own extrapolations, its own share of the
 * detector's steps (the state a stream is left in is the state after exactly its steps) and its own walk. */
int vamd_plan_streams_whole_v(vamd_ctx *ctx, float *pcm, long stream_stride, long channel_stride, long nstreams,
                              long max_frames, const int64_t *nframes, vamd_envelope_state *states, vamd_stream_plan *plan);
/* A plan's lists copied to host arrays (any may be NULL): per size class W lW / nW / blocktype / src [nblocks[W]],
 * order [nblocks[0] + nblocks[1]], stream_start [nstreams + 1].  Synchronises. */
int vamd_plan_fetch(vamd_ctx *ctx, const vamd_stream_plan *plan, int32_t *const lW[2], int32_t *const nW[2],
                    int32_t *const blocktype[2], int64_t *const src[2], int32_t *order, int64_t *stream_start);

/* ---- the host shim for encoders that submit ONE block at a time from many threads.
 * libvorbis' unit of work is one block of one stream (mapping0_forward, reference lib/mapping0.c:233-687); a
 * batcher coalesces concurrent vamd_batcher_encode_block() calls -- same contract as vamd_encode_block() for a VBR
 * encoder, callable from any number of threads, one stream per thread as libvorbis itself requires -- into batched
 * launches.  It owns a few LANES (VAMD_BATCH_LANES in the environment, default 8, at most 16): a context, a HIP stream, a
 * pinned staging arena and one library thread each.  A caller queues its block and sleeps; a lane that is idle takes everything pending
 * of one size class (at most `max_batch` blocks) at once, runs it as a single vamd_analyze_batch() with packet output
 * and wakes exactly the owners of those blocks; blocks that arrive while every lane is busy gather for the next one
 * (a lane that finds both size classes waiting takes the one it did not take last time, so neither waits for ever).
 * There is no timer (`max_wait_us` is accepted and unused since round 4: waiting for stragglers cost more than it
 * gathered).  vamd_batcher_attach / _detach announce a stream (a vorbis_dsp_state); they are bookkeeping only.
 * Errors: OV_*-valued as everywhere; the text of the last one with vamd_batcher_last_error().  vamd_batcher_context()
 * is the first lane's context (capacities, geometry; NOT for launches).  Reference-side use:
 * integration/mapping0_vamd.c with VAMD_BATCH set in the environment. */
typedef struct vamd_batcher vamd_batcher;
int vamd_batcher_create(vamd_batcher **out, const void *setup_blob, size_t blob_bytes, int device, int max_batch,
                        int max_wait_us);
/* The same over several GPUs (ABI 9): lanes are dealt round `devices` (HIP ordinals; at least one lane per device), each with
 * its context, stream, arena and thread on its own device; a batch runs wherever its lane lives -- blocks are independent
 * (SURVEY.md 8e), so nothing crosses between devices. */
int vamd_batcher_create_multi(vamd_batcher **out, const void *setup_blob, size_t blob_bytes, const int *devices, int ndevices,
                              int max_batch, int max_wait_us);
void vamd_batcher_destroy(vamd_batcher *b);
void vamd_batcher_attach(vamd_batcher *b);
void vamd_batcher_detach(vamd_batcher *b);
int vamd_batcher_encode_block(vamd_batcher *b, const float *const *pcm, int lW, int W, int nW, int blocktype,
                              float ampmax_in, float *ampmax_out, uint8_t *packet, long packet_cap,
                              int32_t *packet_bits);
const char *vamd_batcher_last_error(const vamd_batcher *b);
/* batches run, blocks carried, seconds spent inside the batched GPU calls (any pointer may be NULL) */
void vamd_batcher_stats(vamd_batcher *b, long *batches, long *blocks, double *run_seconds);
/* where the batches' time went, as text (diagnostics): gather / staging / GPU / hand-out / wake-up per batch, why gathers
 * ended, batch-size histogram.  Returns the length written. */
long vamd_batcher_report(vamd_batcher *b, char *buf, long cap);
vamd_ctx *vamd_batcher_context(vamd_batcher *b);

/* ---- the host-fed farm (ABI 9): whole streams in from HOST memory, finished packets back to host memory, over one
 * or several GPUs (SURVEY.md 8d "report separately H2D/D2H-inclusive", 8e).  Everything above assumes samples that are
 * already in HBM, or one block per call; an encoder farm holds decoded audio in host memory -- 16-bit interleaved, as
 * examples/encoder_example.c:179-202 reads it -- and wants Ogg payloads back.  A vamd_feed owns LANES (per device: a
 * context, a HIP stream, a library thread, pinned input and output arenas, the streams' HBM buffers); a GROUP of streams
 * travels through one lane:
 *     upload (one copy command out of the pinned arena) -> 16-bit to float on the device (x / 32768.f, exactly
 *     encoder_example.c:197-202) -> vamd_plan_streams_whole (both stream ends, detector, block walk) -> full analysis of
 *     every block where it lies in the stream buffers, 50 % overlap read in place, ampmax chains per stream ->
 *     residue search + packet assembly -> the packets laid end to end, written straight into the pinned output arena
 * and while one lane computes, the next group's upload runs beside it on another lane's stream (the link and the
 * shader array are separate resources).  Only samples cross the link upwards (4 bytes per stereo frame: 4 KB per long
 * stereo block) and only packet bytes downwards.  The call sequence mirrors libvorbis' own:
 *     slot = vamd_feed_buffer(f, &pcm)     like vorbis_analysis_buffer(): where to put the next group's samples
 *                                          (blocks while every lane is busy); interleaved [stream][frame][channel]
 *     vamd_feed_wrote(f, slot, ...)        like vorbis_analysis_wrote(): the group is the library's; returns at once
 *     vamd_feed_packets(f, slot, &out)     waits for the group; pointers into the lane's pinned output arena
 *     vamd_feed_release(f, slot)           the lane may be handed out again
 * Every stream of a group is complete; vamd_feed_wrote() takes streams of one length, vamd_feed_wrote_v() of any lengths.
 * Per stream the packets are byte for byte what the reference encoder emits for the same samples written 1024 frames
 * at a time and closed with vorbis_analysis_wrote(v, 0) -- first block to last (tests/test_feed.py).  VBR setups whose
 * packets the GPU assembles (vamd_packet_capacity() > 0).  Thread rules: one thread drives a feed (or several, each
 * with its own slots); the lanes' threads are the library's. */
typedef struct vamd_feed vamd_feed;
#define VAMD_FEED_S16 0 /* int16_t, interleaved; sample = x / 32768.f */
#define VAMD_FEED_F32 1 /* float, interleaved, already scaled to +-1 */
/* devices: HIP ordinals (NULL / 0: the calling thread's current device); lanes_per_device >= 1 (2 or 3 overlap upload,
 * compute and hand-back; 4-5 keep the shader array fed while the chains of small kernels at a group's start and end
 * run); a group holds at most max_streams streams of at most max_frames frames each, all in `format`.  Pinned host
 * memory per lane: the group's samples plus about a quarter of that for its packets. */
int vamd_feed_create(vamd_feed **out, const void *setup_blob, size_t blob_bytes, const int *devices, int ndevices,
                     int lanes_per_device, long max_streams, long max_frames, int format /* VAMD_FEED_S16 / _F32 */);
void vamd_feed_destroy(vamd_feed *f);
int vamd_feed_lanes(const vamd_feed *f);
int vamd_feed_device(const vamd_feed *f, int slot);   /* the device lane `slot` runs on */
int vamd_feed_buffer(vamd_feed *f, void **pcm);       /* >= 0: the slot; < 0: an OV_*-valued error */
int vamd_feed_wrote(vamd_feed *f, int slot, long nstreams, long frames);
/* streams of unequal length: frames[s] (host array, each 1 .. max_frames, their sum <= max_streams * max_frames) frames of
 * stream s, the streams laid back to back in the arena (stream s starts at frame frames[0] + ... + frames[s-1]) */
int vamd_feed_wrote_v(vamd_feed *f, int slot, long nstreams, const int64_t *frames);
typedef struct vamd_feed_result {
  int64_t nstreams, nblocks;      /* blocks == packets, all streams */
  const int64_t *stream_start;    /* [nstreams + 1]: stream s owns packets [stream_start[s], stream_start[s+1]), in stream order */
  const int64_t *offset;          /* [nblocks] where the packet starts in bytes[] (a multiple of 4) */
  const int32_t *bits;            /* [nblocks] oggpack_bits() of the packet: (bits + 7) / 8 bytes; -1: no packet, see info */
  const int64_t *granulepos;      /* [nblocks] ogg_packet.granulepos (vb->granulepos, lib/block.c:620) */
  const uint8_t *info;            /* [nblocks] bit 0: vb->W; bit 1: last packet of its stream (op.e_o_s); bits 2-3: VAMD_STATUS_*
                                     of a block outside the input domain (no packet) */
  const uint8_t *bytes;           /* the packets, end to end */
  int64_t total_bytes;
  double upload_ms, device_ms, total_ms; /* of this group: the upload alone; upload to last kernel; wrote() to ready */
} vamd_feed_result;
int vamd_feed_packets(vamd_feed *f, int slot, vamd_feed_result *out);
int vamd_feed_release(vamd_feed *f, int slot);
const char *vamd_feed_last_error(const vamd_feed *f);

#ifdef __cplusplus
}
#endif
#endif /* VORBIS_AMD_H */
