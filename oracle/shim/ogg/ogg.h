/* oracle/shim/ogg/ogg.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Minimal stand-in for libogg's <ogg/ogg.h>: the two structs whose layout is
 * part of the libvorbis ABI (oggpack_buffer is embedded by value in
 * vorbis_block, include/vorbis/codec.h:90) and the LSb-first bit-packer entry
 * points the libvorbis encode/decode sources call (12 symbols, see
 * SURVEY.md 8c).  Framing (ogg_stream_*, ogg_sync_*) is deliberately absent:
 * the oracle feeds packets straight from vorbis_analysis().
 */
#ifndef VAMD_ORACLE_OGG_H
#define VAMD_ORACLE_OGG_H

#include <stddef.h>
#include <ogg/os_types.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  long           endbyte;
  int            endbit;
  unsigned char *buffer;
  unsigned char *ptr;
  long           storage;
} oggpack_buffer;

typedef struct {
  unsigned char *packet;
  long           bytes;
  long           b_o_s;
  long           e_o_s;
  ogg_int64_t    granulepos;
  ogg_int64_t    packetno;
} ogg_packet;

void  oggpack_writeinit(oggpack_buffer *b);
void  oggpack_writetrunc(oggpack_buffer *b, long bits);
void  oggpack_reset(oggpack_buffer *b);
void  oggpack_writeclear(oggpack_buffer *b);
void  oggpack_readinit(oggpack_buffer *b, unsigned char *buf, int bytes);
void  oggpack_write(oggpack_buffer *b, unsigned long value, int bits);
void  oggpack_writecopy(oggpack_buffer *b, void *source, long bits);
long  oggpack_look(oggpack_buffer *b, int bits);
void  oggpack_adv(oggpack_buffer *b, int bits);
long  oggpack_read(oggpack_buffer *b, int bits);
long  oggpack_bytes(oggpack_buffer *b);
long  oggpack_bits(oggpack_buffer *b);
unsigned char *oggpack_get_buffer(oggpack_buffer *b);

#ifdef __cplusplus
}
#endif
#endif
