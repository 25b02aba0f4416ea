/* tests/c/multi_gpu.c -- a plain-C host spreading work over every GPU the runtime shows.
 * Built and run by tests/test_multi_gpu_c.py:  multi_gpu <libvorbis_amd.so dir is on the link line> <setup blob file>
 *
 *   1. a vamd_feed over devices {0 .. n-1} (one GPU here: {0, 0} -- the device list is then two entries naming the same GPU,
 *      which walks the same code: lanes dealt round the list, a context / stream / arenas per lane on ITS device): the same
 *      group of 16-bit streams goes through every lane; all lanes must return identical packets, and vamd_feed_device()
 *      must show the lanes dealt round the list.
 *   2. a vamd_batcher over the same list: T threads each encode the same block sequence; every thread must get the packets
 *      thread 0 gets, and those must equal vamd_encode_block()'s on a plain context.
 * Prints one line per check and "multi_gpu OK devices=<n>"; exit status 0 only if everything held. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vorbis_amd.h"

#define NSTREAMS 6
#define FRAMES 20000
#define NTHREADS 8
#define NBLOCKS 12

static unsigned long long lcg = 99;
static double uniform(void) {
  lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)(lcg >> 11) / 9007199254740992.0;
}

static unsigned char *blob;
static size_t blob_bytes;
static vamd_batcher *B;
static float *blocks; /* [NBLOCKS][2][2048] */
static long pkcap;
static unsigned char *thread_pk[NTHREADS];
static int thread_bits[NTHREADS][NBLOCKS], thread_rc[NTHREADS];

static void *encode_thread(void *arg) {
  const long t = (long)arg;
  float amp = VAMD_AMPMAX_FLOOR, out;
  vamd_batcher_attach(B);
  for (int k = 0; k < NBLOCKS; k++) {
    const float *pcm[2] = {blocks + (size_t)k * 4096, blocks + (size_t)k * 4096 + 2048};
    int r = vamd_batcher_encode_block(B, pcm, 1, 1, 1, VAMD_BLOCKTYPE_LONG, amp, &out, thread_pk[t] + (size_t)k * pkcap, pkcap,
                                      &thread_bits[t][k]);
    if (r) thread_rc[t] = r;
    amp = out;
  }
  vamd_batcher_detach(B);
  return NULL;
}

int main(int argc, char **argv) {
  if (argc != 2) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  fseek(f, 0, SEEK_END);
  blob_bytes = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  blob = (unsigned char *)malloc(blob_bytes);
  if (fread(blob, 1, blob_bytes, f) != blob_bytes) return 2;
  fclose(f);

  const int ngpu = vamd_device_count();
  if (ngpu < 1) { fprintf(stderr, "no GPU (vamd_device_count = %d)\n", ngpu); return 1; }
  int devices[64], ndev = ngpu > 1 ? (ngpu > 64 ? 64 : ngpu) : 2;
  for (int i = 0; i < ndev; i++) devices[i] = ngpu > 1 ? i : 0;
  int bad = 0;

  /* ---- 1. the feed over all devices */
  vamd_feed *feed = NULL;
  int r = vamd_feed_create(&feed, blob, blob_bytes, devices, ndev, 2, NSTREAMS, FRAMES, VAMD_FEED_S16);
  if (r) { fprintf(stderr, "vamd_feed_create: %d\n", r); return 1; }
  const int lanes = vamd_feed_lanes(feed);
  short *pcm16 = (short *)malloc((size_t)NSTREAMS * FRAMES * 2 * sizeof(short));
  for (long i = 0; i < (long)NSTREAMS * FRAMES * 2; i++) {
    const long frame = (i / 2) % FRAMES;
    const double gate = (frame % 7000) < 500 ? 0.5 : 0.001;
    pcm16[i] = (short)((uniform() - 0.5) * 2 * gate * 32767.0);
  }
  unsigned char *first = NULL;
  long long first_bytes = 0, first_blocks = 0;
  int *slots = (int *)malloc(sizeof(int) * lanes);
  for (int l = 0; l < lanes; l++) { /* every lane out at once: all of them are in flight together */
    void *p;
    slots[l] = vamd_feed_buffer(feed, &p);
    if (slots[l] < 0) { fprintf(stderr, "vamd_feed_buffer: %d\n", slots[l]); return 1; }
    memcpy(p, pcm16, (size_t)NSTREAMS * FRAMES * 2 * sizeof(short));
    if ((r = vamd_feed_wrote(feed, slots[l], NSTREAMS, FRAMES))) { fprintf(stderr, "vamd_feed_wrote: %d\n", r); return 1; }
  }
  int seen_dev[64] = {0};
  for (int l = 0; l < lanes; l++) {
    vamd_feed_result res;
    if ((r = vamd_feed_packets(feed, slots[l], &res))) { fprintf(stderr, "vamd_feed_packets: %d (%s)\n", r, vamd_feed_last_error(feed)); return 1; }
    const int dev = vamd_feed_device(feed, slots[l]);
    if (dev != devices[slots[l] % ndev]) { printf("lane %d on device %d, expected %d\n", slots[l], dev, devices[slots[l] % ndev]); bad++; }
    if (dev >= 0 && dev < 64) seen_dev[dev]++;
    if (!first) {
      first_bytes = res.total_bytes, first_blocks = res.nblocks;
      first = (unsigned char *)malloc((size_t)res.total_bytes + (size_t)res.nblocks * 12 + 16);
      memcpy(first, res.bytes, (size_t)res.total_bytes);
      memcpy(first + res.total_bytes, res.bits, (size_t)res.nblocks * 4);
      memcpy(first + res.total_bytes + res.nblocks * 4, res.granulepos, (size_t)res.nblocks * 8);
    } else if (res.total_bytes != first_bytes || res.nblocks != first_blocks || memcmp(first, res.bytes, (size_t)first_bytes) ||
               memcmp(first + first_bytes, res.bits, (size_t)first_blocks * 4) ||
               memcmp(first + first_bytes + first_blocks * 4, res.granulepos, (size_t)first_blocks * 8)) {
      printf("lane %d (device %d) returned other packets than lane %d\n", slots[l], dev, slots[0]);
      bad++;
    }
    vamd_feed_release(feed, slots[l]);
  }
  int used = 0;
  for (int d = 0; d < 64; d++) used += seen_dev[d] > 0;
  printf("feed: %d lanes over %d device(s), %lld packets / %lld bytes per group, identical on every lane: %s\n", lanes, used,
         first_blocks, first_bytes, bad ? "NO" : "yes");
  if (used != (ngpu > 1 ? ndev : 1)) { printf("feed used %d devices, expected %d\n", used, ngpu > 1 ? ndev : 1); bad++; }
  vamd_feed_destroy(feed);

  /* ---- 2. the batcher over all devices */
  blocks = (float *)malloc(sizeof(float) * NBLOCKS * 4096);
  for (long i = 0; i < NBLOCKS * 4096; i++) blocks[i] = (float)(uniform() - 0.5);
  if ((r = vamd_batcher_create_multi(&B, blob, blob_bytes, devices, ndev, 64, 0))) { fprintf(stderr, "vamd_batcher_create_multi: %d\n", r); return 1; }
  pkcap = vamd_packet_capacity(vamd_batcher_context(B), 1);
  pthread_t th[NTHREADS];
  for (long t = 0; t < NTHREADS; t++) {
    thread_pk[t] = (unsigned char *)calloc((size_t)NBLOCKS, (size_t)pkcap);
    pthread_create(&th[t], NULL, encode_thread, (void *)t);
  }
  for (int t = 0; t < NTHREADS; t++) pthread_join(th[t], NULL);
  int bbad = 0;
  for (int t = 0; t < NTHREADS; t++) {
    if (thread_rc[t]) { printf("thread %d: error %d (%s)\n", t, thread_rc[t], vamd_batcher_last_error(B)); bbad++; }
    for (int k = 0; k < NBLOCKS; k++)
      if (thread_bits[t][k] != thread_bits[0][k] ||
          memcmp(thread_pk[t] + (size_t)k * pkcap, thread_pk[0] + (size_t)k * pkcap, (size_t)(thread_bits[0][k] + 7) / 8))
        bbad++;
  }
  vamd_ctx *ctx = NULL; /* the plain context's answer */
  if ((r = vamd_create(&ctx, blob, blob_bytes, 0))) { fprintf(stderr, "vamd_create: %d\n", r); return 1; }
  unsigned char *one = (unsigned char *)malloc((size_t)pkcap);
  float amp = VAMD_AMPMAX_FLOOR, out;
  for (int k = 0; k < NBLOCKS; k++) {
    const float *pcm[2] = {blocks + (size_t)k * 4096, blocks + (size_t)k * 4096 + 2048};
    int32_t bits = 0;
    if ((r = vamd_encode_block(ctx, pcm, 1, 1, 1, VAMD_BLOCKTYPE_LONG, amp, 0, &out, one, pkcap, &bits))) { fprintf(stderr, "vamd_encode_block: %d\n", r); return 1; }
    if (bits != thread_bits[0][k] || memcmp(one, thread_pk[0] + (size_t)k * pkcap, (size_t)(bits + 7) / 8)) bbad++;
    amp = out;
  }
  vamd_destroy(ctx);
  long batches = 0, nblk = 0;
  vamd_batcher_stats(B, &batches, &nblk, NULL);
  printf("batcher: %d threads x %d blocks over %d device entr%s, %ld batches: packets identical across threads and to vamd_encode_block: %s\n",
         NTHREADS, NBLOCKS, ndev, ndev == 1 ? "y" : "ies", batches, bbad ? "NO" : "yes");
  vamd_batcher_destroy(B);
  bad += bbad;
  if (!bad) printf("multi_gpu OK devices=%d\n", ngpu);
  return bad ? 1 : 0;
}
