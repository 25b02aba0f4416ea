// k_floor.h -- _vp_offset_and_mix (reference lib/psy.c:779-835), floor1_fit
// (lib/floor1.c:576-729) and the post-quantise + render_line0 half of
// floor1_encode (lib/floor1.c:766-831,923-946); SURVEY.md 8a rows a11-a13.  One
// wavefront per channel-block.
//
// Parallel form: the mix is per bin.  accumulate_fit's integer sums take one
// lane per post interval.  The greedy split loop is inherently ordered (each
// decision moves the neighbours of later posts); every lane runs it uniformly
// with the small state in LDS, while inspect_error -- the only part that
// touches O(n) bins -- is evaluated by all 64 lanes at once: the Bresenham line
// has the closed form y(x) = y0 + k*base + sgn*floor(k*ady'/adx), the squared
// error is an integer wave sum and the early-outs are an any().  fit_line's
// fp64 sums run in the reference's term order on every lane.
//
// LDS: qc[n] 16-bit words -- everything the fit reads per bin: the mask quantised by vorbis_dBquant
// (bits 0-9; the fit never looks at the float mask) and the "mdct + twofitatten >= mask" class
// (bit 15; the only thing the fit reads logmdct for) --, FloorScratch (interval accumulators + the
// rendered segment list).
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"
#include "k_tone.h"

namespace vamd {

#define VAMD_MAXPOSTS 32

#include "k_floor.inc"

}  // namespace vamd
