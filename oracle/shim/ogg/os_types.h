/* oracle/shim/ogg/os_types.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Minimal stand-in for libogg's <ogg/os_types.h>.  libogg is an un-vendored
 * third-party dependency of xiph/vorbis (CMakeLists.txt:70-72 find_package(Ogg))
 * and is absent from this image.  It contributes no arithmetic to the encode
 * hot path -- only integer typedefs and allocator macros -- so the oracle build
 * (oracle/Makefile) supplies this shim instead.  Written from scratch.
 */
#ifndef VAMD_ORACLE_OGG_OS_TYPES_H
#define VAMD_ORACLE_OGG_OS_TYPES_H

#include <stdint.h>
#include <stdlib.h>

#define _ogg_malloc  malloc
#define _ogg_calloc  calloc
#define _ogg_realloc realloc
#define _ogg_free    free

typedef int16_t  ogg_int16_t;
typedef uint16_t ogg_uint16_t;
typedef int32_t  ogg_int32_t;
typedef uint32_t ogg_uint32_t;
typedef int64_t  ogg_int64_t;
typedef uint64_t ogg_uint64_t;

#endif
