/* vamd_setup.h -- the POD "setup blob" handed to vamd_create().
 *
 * libvorbis derives every lookup the per-block analysis needs once per stream
 * in vorbis_analysis_init() -> _vds_shared_init() (reference lib/block.c:170-293):
 * mdct_init (lib/mdct.c:51-90), drft_init (lib/smallft.c:1241), _vp_psy_init
 * (lib/psy.c:266-362), floor1_look (lib/floor1.c:178-255), plus the
 * quality-interpolated parameters libvorbisenc wrote into codec_setup_info
 * (lib/vorbisenc.c:682-861).  The GPU layer never recomputes any of them
 * (host libm cos/sin/log results are part of the bit-exact answer); the host
 * serialises them into this little-endian, pointer-free blob
 * (integration/vamd_pack_setup.c shows the reference-side packer) and the GPU
 * layer copies the blob to HBM verbatim.  In a multi-GPU job rank 0 packs it and
 * broadcasts the bytes (RCCL) -- see DESIGN.md "multi-GPU".
 *
 * All `off_*` fields are byte offsets from the start of the blob, 16-byte
 * aligned.  Scalars keep the reference's types (float stays float, the one
 * double stays double).
 */
#ifndef VAMD_SETUP_H
#define VAMD_SETUP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAMD_SETUP_MAGIC   0x31544553444d4156ULL /* "VAMDSET1" little-endian */
#define VAMD_SETUP_VERSION 7u

#define VAMD_PACKETBLOBS   15  /* lib/codec_internal.h:28 */
#define VAMD_P_BANDS       17  /* lib/psy.h:28 */
#define VAMD_P_LEVELS      8   /* lib/psy.h:29 */
#define VAMD_EHMER_MAX     56  /* lib/psy.h:24 */
#define VAMD_EHMER_OFFSET  16  /* lib/masking.h:22 */
#define VAMD_P_NOISECURVES 3   /* lib/psy.h:31 */
#define VAMD_NOISE_COMPAND_LEVELS 40 /* lib/psy.h:33 */
#define VAMD_POSIT         65  /* VIF_POSIT+2, lib/backends.h:57 */
#define VAMD_FLOOR_PARTS   31  /* VIF_PARTS, lib/backends.h:59 */
#define VAMD_FLOOR_CLASSES 16  /* VIF_CLASS, lib/backends.h:58 */
#define VAMD_MAX_CH        8   /* channel counts covered: every layout Vorbis I assigns an order to (mono .. 7.1) */
#define VAMD_MAX_SUBMAPS   2   /* 5.1: the five full-range channels, and the LFE on its own floor and residue */
#define VAMD_MAX_COUPLING  4   /* coupling steps (5.1: L-R, surround L-R, L-C, L-SL; lib/modes/residue_44p51.h:283) */
#define VAMD_VE_BANDS      7   /* lib/envelope.h:28 */
#define VAMD_VE_NEARDC     15  /* lib/envelope.h:29 */
#define VAMD_VE_AMP        17  /* VE_PRE+VE_POST-1, lib/envelope.h:26 */
#define VAMD_VE_MINSTRETCH 2   /* lib/envelope.h:31 */
#define VAMD_VE_MAXSTRETCH 12  /* lib/envelope.h:32 */
#define VAMD_VE_BANDWIN    8   /* widest band window, lib/envelope.c:52-58 */

/* one per block size W (0 = short, 1 = long): mdct_lookup (lib/mdct.h:55-63),
 * drft_lookup (lib/smallft.h:22-26) and the vwin table (lib/window.c) */
typedef struct vamd_xform_tab {
  int32_t  n;              /* block size (samples) */
  int32_t  log2n;
  float    mdct_scale;     /* 4.f/n */
  int32_t  fft_nf;         /* number of FFTPACK factors (splitcache[1]) */
  int32_t  fft_fac[16];    /* the factors, splitcache[2..] */
  uint32_t off_mdct_trig;  /* float[n + n/4] */
  uint32_t off_mdct_bitrev;/* int32[n/4] */
  uint32_t off_fft_wa;     /* float[2n] = trigcache + n (twiddles) */
  uint32_t off_window;     /* float[n/2] rising half-window vwin[n] */
  uint32_t pad[3];
} vamd_xform_tab;

/* one per psy look, index = blocktype + 2*W (lib/mapping0.c:250):
 * vorbis_look_psy (lib/psy.h:94-113) + the vorbis_info_psy fields the path reads */
typedef struct vamd_psy_tab {
  int32_t  n;                    /* bins = blocksize/2 */
  int32_t  blockflag;
  int32_t  firstoc, shiftoc;
  int32_t  eighth_octave_lines, total_octave_lines;
  float    m_val;
  float    ath_adjatt, ath_maxatt;
  float    tone_masteratt[VAMD_P_NOISECURVES];
  float    tone_abs_limit;
  float    noisemaxsupp;
  int32_t  noisewindowfixed;
  float    max_curve_dB;
  float    noisecompand[VAMD_NOISE_COMPAND_LEVELS];
  int32_t  normal_p, normal_start, normal_partition;
  int32_t  pad0;
  double   normal_thresh;
  uint32_t off_ath;              /* float[n] */
  uint32_t off_octave;           /* int32[n] */
  uint32_t off_bark;             /* int32[n]  ((lo-1)<<16)+(hi-1), lib/psy.c:319 */
  uint32_t off_noiseoffset;      /* float[3][n] */
  uint32_t off_tonecurves;       /* float[17][8][58], first two = fence posts */
  uint32_t pad1[3];
} vamd_psy_tab;

/* vorbis_info_psy_global slices (lib/psy.h:65-83) */
typedef struct vamd_psy_global_tab {
  float   ampmax_att_per_sec;
  int32_t coupling_pointlimit[2][VAMD_PACKETBLOBS];
  int32_t coupling_prepointamp[VAMD_PACKETBLOBS];
  int32_t coupling_postpointamp[VAMD_PACKETBLOBS];
  int32_t sliding_lowpass[2][VAMD_PACKETBLOBS];
  int32_t pad[3];
} vamd_psy_global_tab;

/* vorbis_look_floor1 (lib/codec_internal.h:138-154) + vorbis_info_floor1
 * encode-side fields (lib/backends.h:60-84) */
typedef struct vamd_floor1_tab {
  int32_t posts;
  int32_t look_n;          /* look->n = postlist[1] */
  int32_t quant_q;
  int32_t mult;
  int32_t info_n;          /* info->n, the lowpass-limited fit range */
  float   maxover, maxunder, maxerr;
  float   twofitweight, twofitatten;
  int32_t pad[2];
  int32_t postlist[VAMD_POSIT];
  int32_t sorted_index[VAMD_POSIT];
  int32_t forward_index[VAMD_POSIT];
  int32_t reverse_index[VAMD_POSIT];
  int32_t hineighbor[VAMD_POSIT];
  int32_t loneighbor[VAMD_POSIT];
  int32_t pad2[2];
  /* the bit-writing half of floor1_encode (lib/floor1.c:833-921): vorbis_info_floor1's partition and
   * class tables (lib/backends.h:60-72) */
  int32_t partitions;
  int32_t partitionclass[VAMD_FLOOR_PARTS];
  int32_t class_dim[VAMD_FLOOR_CLASSES];
  int32_t class_subs[VAMD_FLOOR_CLASSES];
  int32_t class_book[VAMD_FLOOR_CLASSES];
  int32_t class_subbook[VAMD_FLOOR_CLASSES][8];   /* book number, -1 = none */
} vamd_floor1_tab;

/* one per mode W: vorbis_info_mapping0 (lib/backends.h:130-141) with its floor */
typedef struct vamd_mode_tab {
  int32_t submaps;                 /* 1 or 2 */
  int32_t coupling_steps;          /* 0 .. VAMD_MAX_COUPLING, applied in order */
  int32_t coupling_mag[VAMD_MAX_COUPLING], coupling_ang[VAMD_MAX_COUPLING];
  int32_t chmuxlist[VAMD_MAX_CH];  /* submap of each channel */
  vamd_floor1_tab floor[VAMD_MAX_SUBMAPS];   /* the floor of each submap (floorsubmap[]) */
} vamd_mode_tab;

/* the block-switching detector: envelope_lookup (lib/envelope.h:54-74, built by
 * _ve_envelope_init lib/envelope.c:30-74) + the vorbis_info_psy_global fields _ve_amp reads
 * (lib/psy.h:69-72) */
typedef struct vamd_envelope_tab {
  int32_t  winlength;                       /* 128 */
  int32_t  searchstep;                      /* 64 */
  int32_t  log2n;
  float    mdct_scale;
  float    minenergy;                       /* gi->preecho_minenergy */
  float    stretch_penalty;
  float    preecho_thresh[VAMD_VE_BANDS];
  float    postecho_thresh[VAMD_VE_BANDS];
  int32_t  band_begin[VAMD_VE_BANDS];
  int32_t  band_end[VAMD_VE_BANDS];
  float    band_total[VAMD_VE_BANDS];       /* 1/sum(window) */
  float    band_window[VAMD_VE_BANDS][VAMD_VE_BANDWIN];
  uint32_t off_mdct_trig;                   /* float[n + n/4] */
  uint32_t off_mdct_bitrev;                 /* int32[n/4] */
  uint32_t off_window;                      /* float[n]  mdct_win = sin^2 */
} vamd_envelope_tab;

/* residue back-end of one mode: vorbis_info_residue0 (lib/backends.h:103-118)
 * with its book list expanded per class and stage as res0_look does (lib/res0.c:201-224) */
#define VAMD_RES_MAXCLASS 64
#define VAMD_RES_MAXSTAGE 8
typedef struct vamd_residue_tab {
  int32_t type;          /* ci->residue_type[]; the GPU covers type 2 (interleaved channels) */
  int32_t begin, end;    /* in interleaved samples for type 2 */
  int32_t grouping;      /* samples per partition */
  int32_t partitions;    /* classes */
  int32_t stages;        /* max ilog(secondstages[]) */
  int32_t groupbook;     /* phrase book number */
  int32_t groupbook_dim; /* partitions per phrase word */
  int32_t secondstages[VAMD_RES_MAXCLASS];
  int32_t classmetric1[VAMD_RES_MAXCLASS];
  int32_t classmetric2[VAMD_RES_MAXCLASS];
  int32_t partbooks[VAMD_RES_MAXCLASS][VAMD_RES_MAXSTAGE]; /* book number, -1 = none */
} vamd_residue_tab;

/* the encode side of one codebook (lib/codebook.h:57-71): a centred integer lattice */
typedef struct vamd_book_tab {
  int32_t  dim, entries;
  int32_t  minval, delta, quantvals;
  uint32_t off_lengths;  /* int8[entries] codeword lengths (<= 0: unused entry) */
  uint32_t off_codes;    /* uint32[entries] codebook.codelist: the codewords as written, LSb first */
  int32_t  pad;
} vamd_book_tab;

typedef struct vamd_setup_header {
  uint64_t magic;
  uint32_t version;
  uint32_t total_bytes;
  int32_t  channels;
  int32_t  rate;
  int32_t  blocksizes[2];
  int32_t  managed;                /* the host runs a bitrate manager: blocks want all 15 candidate packets */
  int32_t  modebits;               /* private_state.modebits: width of a packet's mode number */
  int32_t  modes;                  /* ci->modes (1: both size classes share mode 0) */
  int32_t  pad;
  vamd_xform_tab      xform[2];
  vamd_psy_tab        psy[4];
  vamd_psy_global_tab psy_g;
  vamd_mode_tab       mode[2];
  vamd_envelope_tab   env;
  vamd_residue_tab    res[2][VAMD_MAX_SUBMAPS];  /* per mode W and submap (residuesubmap[]) */
  int32_t             nbooks;      /* ci->books */
  uint32_t            off_books;   /* vamd_book_tab[nbooks] */
} vamd_setup_header;

#ifdef __cplusplus
}
#endif
#endif /* VAMD_SETUP_H */
