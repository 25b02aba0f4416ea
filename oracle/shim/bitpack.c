/* oracle/shim/bitpack.c -- TEST INFRASTRUCTURE ONLY.
 *
 * From-scratch LSb-first bit packer with the call semantics of libogg's
 * bitwise.c (values masked to `bits`, byte buffer grown on demand, reads past
 * the end return -1).  Only what libvorbis' lib/ sources link against.
 */
#include <string.h>
#include <limits.h>
#include <ogg/ogg.h>

#define GROW 256

static unsigned long lowmask(int bits) {
  return bits >= 32 ? 0xffffffffUL : ((1UL << bits) - 1UL);
}

void oggpack_writeinit(oggpack_buffer *b) {
  memset(b, 0, sizeof(*b));
  b->buffer = (unsigned char *)malloc(GROW);
  b->ptr = b->buffer;
  b->buffer[0] = 0;
  b->storage = GROW;
}

void oggpack_reset(oggpack_buffer *b) {
  if (!b->ptr) return;
  b->ptr = b->buffer;
  b->buffer[0] = 0;
  b->endbit = 0;
  b->endbyte = 0;
}

void oggpack_writeclear(oggpack_buffer *b) {
  if (b->buffer) free(b->buffer);
  memset(b, 0, sizeof(*b));
}

void oggpack_writetrunc(oggpack_buffer *b, long bits) {
  long bytes = bits >> 3;
  if (!b->ptr) return;
  bits -= bytes * 8;
  b->ptr = b->buffer + bytes;
  b->endbit = (int)bits;
  b->endbyte = bytes;
  *b->ptr &= (unsigned char)lowmask((int)bits);
}

void oggpack_write(oggpack_buffer *b, unsigned long value, int bits) {
  if (bits < 0 || bits > 32) goto fail;
  if (b->endbyte >= b->storage - 4) {
    unsigned char *nb;
    if (!b->ptr) return;
    if (b->storage > LONG_MAX - GROW) goto fail;
    nb = (unsigned char *)realloc(b->buffer, b->storage + GROW);
    if (!nb) goto fail;
    b->buffer = nb;
    b->storage += GROW;
    b->ptr = b->buffer + b->endbyte;
  }
  value &= lowmask(bits);
  {
    /* spill the value across up to five bytes starting at the current bit */
    int have = b->endbit;          /* bits already used in *ptr */
    int total = have + bits;
    unsigned long long acc = (unsigned long long)value << have;
    b->ptr[0] |= (unsigned char)(acc & 0xff);
    if (total >= 8)  b->ptr[1] = (unsigned char)((acc >> 8) & 0xff);
    if (total >= 16) b->ptr[2] = (unsigned char)((acc >> 16) & 0xff);
    if (total >= 24) b->ptr[3] = (unsigned char)((acc >> 24) & 0xff);
    if (total >= 32) b->ptr[4] = (unsigned char)((acc >> 32) & 0xff);
    b->endbyte += total / 8;
    b->ptr += total / 8;
    b->endbit = total & 7;
  }
  return;
fail:
  oggpack_writeclear(b);
}

/* libogg's oggpack_writecopy(): append `bits` bits of an LSb-first packed buffer (whole bytes,
 * then the low bits of the last one) */
void oggpack_writecopy(oggpack_buffer *b, void *source, long bits) {
  const unsigned char *src = (const unsigned char *)source;
  long i, bytes = bits / 8;
  for (i = 0; i < bytes; i++) oggpack_write(b, src[i], 8);
  if (bits & 7) oggpack_write(b, src[bytes], (int)(bits & 7));
}

void oggpack_readinit(oggpack_buffer *b, unsigned char *buf, int bytes) {
  memset(b, 0, sizeof(*b));
  b->buffer = b->ptr = buf;
  b->storage = bytes;
}

static long peek(oggpack_buffer *b, int bits, int *ok) {
  unsigned long long acc = 0;
  int need = bits + b->endbit;
  int k;
  *ok = 1;
  if (bits < 0 || bits > 32) { *ok = 0; return -1; }
  if (b->endbyte >= b->storage - 4) {
    /* not the fast path: make sure every bit asked for exists */
    if (b->endbyte > b->storage - ((need + 7) >> 3)) { *ok = 0; return -1; }
    if (!bits) return 0;
  }
  for (k = 0; k * 8 < need; k++)
    acc |= (unsigned long long)b->ptr[k] << (8 * k);
  return (long)((acc >> b->endbit) & lowmask(bits));
}

long oggpack_look(oggpack_buffer *b, int bits) {
  int ok;
  return peek(b, bits, &ok);
}

void oggpack_adv(oggpack_buffer *b, int bits) {
  bits += b->endbit;
  if (b->endbyte > b->storage - ((bits + 7) >> 3)) {
    b->ptr = NULL;
    b->endbyte = b->storage;
    b->endbit = 1;
    return;
  }
  b->ptr += bits / 8;
  b->endbyte += bits / 8;
  b->endbit = bits & 7;
}

long oggpack_read(oggpack_buffer *b, int bits) {
  int ok;
  long v;
  if (!b->ptr) return -1;
  v = peek(b, bits, &ok);
  if (!ok) {
    b->ptr = NULL;
    b->endbyte = b->storage;
    b->endbit = 1;
    return -1;
  }
  oggpack_adv(b, bits);
  return v;
}

long oggpack_bytes(oggpack_buffer *b) { return b->endbyte + (b->endbit + 7) / 8; }
long oggpack_bits(oggpack_buffer *b) { return b->endbyte * 8 + b->endbit; }
unsigned char *oggpack_get_buffer(oggpack_buffer *b) { return b->buffer; }
