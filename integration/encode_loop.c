/* integration/encode_loop.c -- an APPLICATION, linked the way applications link (-lvorbisenc -lvorbis -logg), that
 * drives libvorbis through the call sequence of the reference's examples/encoder_example.c:140-236:
 *
 *   vorbis_info_init -> vorbis_encode_init_vbr -> vorbis_comment_init / _add_tag -> vorbis_analysis_init ->
 *   vorbis_block_init -> vorbis_analysis_headerout -> { vorbis_analysis_buffer, 16-bit -> float, vorbis_analysis_wrote,
 *   while (vorbis_analysis_blockout == 1) { vorbis_analysis(&vb, NULL); vorbis_bitrate_addblock;
 *   while (vorbis_bitrate_flushpacket) <packet> } } -> vorbis_analysis_wrote(0) -> ... -> the clears
 *
 * and writes every ogg_packet it is handed -- the three headers, then the audio packets: what ogg_stream_packetin()
 * would receive -- to a file, or compares them with such a file (memcmp).  Which libvorbis it runs on is the loader's
 * choice (LD_LIBRARY_PATH): the drop-in built by integration/Makefile, or the unmodified reference next to it.
 * tests/test_dropin_library.py runs it on both and holds the two files against each other.
 *
 *   encode_loop READ QUALITY FRAMES write out.pkts        READ = frames per vorbis_analysis_wrote() (the example: 1024)
 *   encode_loop READ QUALITY FRAMES check ref.pkts        exit 0 = every packet byte-identical
 *
 * The signal is BASELINE config 1's: stereo white noise, uniform in [-0.5, 0.5), seed 12345, quantised to 16 bits. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vorbis/vorbisenc.h>

static unsigned long long lcg = 12345;
static double uniform(void) { /* 53 bits of a 64-bit LCG (Knuth's MMIX constants) */
  lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)(lcg >> 11) / 9007199254740992.0;
}

typedef struct { unsigned char *p; size_t n, cap; } blob;
static void put(blob *b, const void *src, size_t n) {
  if (b->n + n > b->cap) {
    b->cap = (b->n + n) * 2 + 4096;
    b->p = (unsigned char *)realloc(b->p, b->cap);
    if (!b->p) { fprintf(stderr, "out of memory\n"); exit(2); }
  }
  memcpy(b->p + b->n, src, n);
  b->n += n;
}
static void record(blob *b, const ogg_packet *op) {
  long long head[4];
  head[0] = op->bytes, head[1] = (long long)op->granulepos, head[2] = op->e_o_s, head[3] = op->b_o_s;
  put(b, head, sizeof(head));
  put(b, op->packet, (size_t)op->bytes);
}

int main(int argc, char **argv) {
  if (argc != 6) {
    fprintf(stderr, "usage: %s READ QUALITY FRAMES write|check FILE\n", argv[0]);
    return 2;
  }
  const long READ = atol(argv[1]), frames = atol(argv[3]);
  const float quality = (float)atof(argv[2]);
  const int check = !strcmp(argv[4], "check");
  if (READ < 1 || frames < 1) return 2;
  short *pcm = (short *)malloc((size_t)frames * 2 * sizeof(short));
  for (long i = 0; i < frames * 2; i++) {
    double x = (uniform() - 0.5) * 32768.0;
    long v = (long)(x < 0 ? x - 0.5 : x + 0.5);
    pcm[i] = (short)(v > 32767 ? 32767 : v < -32768 ? -32768 : v);
  }

  vorbis_info vi;
  vorbis_comment vc;
  vorbis_dsp_state vd;
  vorbis_block vb;
  ogg_packet op;
  blob out = {0, 0, 0};
  long blocks = 0, packets = 0, fed = 0;
  int eos = 0;

  vorbis_info_init(&vi);
  if (vorbis_encode_init_vbr(&vi, 2, 44100, quality)) { fprintf(stderr, "vorbis_encode_init_vbr failed\n"); return 1; }
  vorbis_comment_init(&vc);
  vorbis_comment_add_tag(&vc, "ENCODER", "encode_loop.c");
  if (vorbis_analysis_init(&vd, &vi)) { fprintf(stderr, "vorbis_analysis_init failed\n"); return 1; }
  vorbis_block_init(&vd, &vb);
  {
    ogg_packet header, header_comm, header_code;
    vorbis_analysis_headerout(&vd, &vc, &header, &header_comm, &header_code);
    record(&out, &header);
    record(&out, &header_comm);
    record(&out, &header_code);
  }
  struct timespec t0, t1, tw;
  long warm_blocks = -1; /* blocks out when the steady-state clock started */
  const long WARM = 64;  /* the first blocks pay for start-up once per process: the GPU runtime's initialisation, the code
                            object's load, the context (setup blob up, workspace allocated) -- ~0.15 s that a ten-second clip
                            would book on 437 blocks; the steady rate is the one an encoder of real files sees */
  clock_gettime(CLOCK_MONOTONIC, &t0);
  tw = t0;
  while (!eos) {
    long n = frames - fed;
    if (n > READ) n = READ;
    if (n == 0) {
      vorbis_analysis_wrote(&vd, 0); /* end of stream */
    } else {
      float **buffer = vorbis_analysis_buffer(&vd, (int)READ);
      for (long i = 0; i < n; i++) {
        buffer[0][i] = pcm[(fed + i) * 2] / 32768.f;
        buffer[1][i] = pcm[(fed + i) * 2 + 1] / 32768.f;
      }
      vorbis_analysis_wrote(&vd, (int)n);
      fed += n;
    }
    while (vorbis_analysis_blockout(&vd, &vb) == 1) {
      int r = vorbis_analysis(&vb, NULL);
      if (r) { fprintf(stderr, "vorbis_analysis failed: %d\n", r); return 1; }
      vorbis_bitrate_addblock(&vb);
      blocks++;
      if (blocks == WARM) {
        clock_gettime(CLOCK_MONOTONIC, &tw);
        warm_blocks = blocks;
      }
      while (vorbis_bitrate_flushpacket(&vd, &op)) {
        record(&out, &op);
        packets++;
        if (op.e_o_s) eos = 1;
      }
    }
    if (n == 0 && !eos) eos = 1; /* (nothing more can come) */
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  vorbis_block_clear(&vb);
  vorbis_dsp_clear(&vd);
  vorbis_comment_clear(&vc);
  vorbis_info_clear(&vi);
  const double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  const double steady = (t1.tv_sec - tw.tv_sec) + 1e-9 * (t1.tv_nsec - tw.tv_nsec);
  fprintf(stderr, "%s: READ %ld q %.2f: %ld blocks, %ld audio packets, %zu bytes recorded, %.0f blocks/s over the whole run", vorbis_version_string(), READ,
          quality, blocks, packets, out.n, blocks / (secs > 0 ? secs : 1));
  if (warm_blocks > 0 && blocks > warm_blocks && steady > 0)
    fprintf(stderr, ", %.0f blocks/s after the first %ld (start-up %.0f ms)", (blocks - warm_blocks) / steady, warm_blocks, (secs - steady) * 1e3);
  fprintf(stderr, "\n");

  int rc = 0;
  if (!check) {
    FILE *f = fopen(argv[5], "wb");
    if (!f || fwrite(out.p, 1, out.n, f) != out.n) { fprintf(stderr, "cannot write %s\n", argv[5]); rc = 2; }
    if (f) fclose(f);
  } else {
    FILE *f = fopen(argv[5], "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", argv[5]); return 2; }
    fseek(f, 0, SEEK_END);
    const long want = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *ref = (unsigned char *)malloc((size_t)want + 1);
    if (fread(ref, 1, (size_t)want, f) != (size_t)want) rc = 2;
    fclose(f);
    if (!rc && ((size_t)want != out.n || memcmp(ref, out.p, out.n))) {
      size_t at = 0;
      while (at < out.n && at < (size_t)want && ref[at] == out.p[at]) at++;
      fprintf(stderr, "MISMATCH: %zu bytes here, %ld there, first difference at byte %zu\n", out.n, want, at);
      rc = 1;
    } else if (!rc) {
      fprintf(stderr, "identical: %ld packets (3 headers + %ld audio), %zu bytes\n", packets + 3, packets, out.n);
    }
    free(ref);
  }
  free(out.p);
  free(pcm);
  return rc;
}
