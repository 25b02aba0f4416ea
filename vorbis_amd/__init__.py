"""vorbis_amd -- MI355X-native per-block Vorbis encode analysis (libvorbis' mapping0_forward
numeric path) behind a C ABI (include/vorbis_amd.h).

This package is only plumbing: it loads libvorbis_amd.so (hand-written HIP for gfx950, built by
`__graft_entry__.build()`: one hipcc command, see there), hands it device pointers of torch tensors and
mirrors the C entry points one-to-one.  There is no CPU fallback: without the HIP library, or
without a GPU, every compute call raises.
"""
from .api import (Analyzer, Feed, FEED_S16, FEED_F32, VamdError, load_library, library_path, default_setup_blob, LEVEL_TRANSFORM,
                  LEVEL_PSY, LEVEL_FULL, POSTS_STRIDE, BLOCKTYPE_IMPULSE, BLOCKTYPE_PADDING,
                  BLOCKTYPE_TRANSITION, BLOCKTYPE_LONG, EXPORTED_SYMBOLS, EnvelopeState, envelope_marks, packet_bytes,
                  VAMD_OK, VAMD_EFAULT, VAMD_EIMPL, VAMD_EINVAL, VAMD_EVERSION, VAMD_EDOMAIN, VAMD_ENONFINITE)

__all__ = ["Analyzer", "Feed", "FEED_S16", "FEED_F32", "VamdError", "load_library", "library_path", "default_setup_blob", "LEVEL_TRANSFORM",
           "LEVEL_PSY", "LEVEL_FULL", "POSTS_STRIDE", "BLOCKTYPE_IMPULSE", "BLOCKTYPE_PADDING",
           "BLOCKTYPE_TRANSITION", "BLOCKTYPE_LONG", "EXPORTED_SYMBOLS", "EnvelopeState", "envelope_marks", "packet_bytes",
           "VAMD_OK", "VAMD_EFAULT", "VAMD_EIMPL", "VAMD_EINVAL", "VAMD_EVERSION", "VAMD_EDOMAIN", "VAMD_ENONFINITE"]
