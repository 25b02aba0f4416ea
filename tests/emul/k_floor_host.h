// tests/emul/k_floor_host.h -- TEST BUILD ONLY (see vamd_wave_host.h).
//
// The one-lane forms of the two pieces of vorbis_amd/csrc/k_floor.h that deal their work over the lanes of a wave
// with ballots and cross-lane reads (fit_line_pair, floor_render_curve): same inputs, same arithmetic in the same
// order, walked serially, so that the rest of the floor stage's body -- which is shared -- can be checked against the
// oracle without a GPU.  Included by k_floor.h in the host build only; the library never sees it.
#pragma once

// the two fit_line calls of a split (lib/floor1.c:456-514 with both ends unconstrained), out of the per-interval
// rows [xb, yb, x2b, xyb, bn] the shared code left in `term`
static inline int fit_line_rows(const double *term, int first, int fits, int x0, int x1, int *y0, int *y1) {
  double xb = 0, yb = 0, x2b = 0, xyb = 0, bn = 0;
  for (int i = first; i < first + fits; i++) {
    xb += term[i * 5];
    yb += term[i * 5 + 1];
    x2b += term[i * 5 + 2];
    xyb += term[i * 5 + 3];
    bn += term[i * 5 + 4];
  }
  const double denom = (bn * x2b - xb * xb);
  if (denom > 0.) {
    const double aa = (yb * x2b - xyb * xb) / denom;
    const double bb = (bn * xyb - xb * yb) / denom;
    *y0 = (int)rint(aa + bb * x0);
    *y1 = (int)rint(aa + bb * x1);
    if (*y0 > 1023) *y0 = 1023;
    if (*y1 > 1023) *y1 = 1023;
    if (*y0 < 0) *y0 = 0;
    if (*y1 < 0) *y1 = 0;
    return 0;
  }
  *y0 = 0;
  *y1 = 0;
  return 1;
}
VAMD_DEV void fit_line_pair(const double *term, double *, int firstL, int fitsL, int x0L, int x1L, int firstR, int fitsR,
                            int x0R, int x1R, int *ret0, int *ly0, int *ly1, int *ret1, int *hy0, int *hy1) {
  *ret0 = fit_line_rows(term, firstL, fitsL, x0L, x1L, ly0, ly1);
  *ret1 = fit_line_rows(term, firstR, fitsR, x0R, x1R, hy0, hy1);
}

// render_line0 over the used posts in x order (lib/floor1.c:923-946): segment list, then every bin its segment's line
VAMD_DEV void floor_render_curve(const FloorP &F, int posts, int n2, const LaneInts &forward_index, const LaneInts &post,
                                 const LaneInts &postlist, FloorScratch *, ilog_t *ilogmask, PhaseClock &pc) {
  int segx[VAMD_MAXPOSTS + 1], segy[VAMD_MAXPOSTS + 1], nseg;
  {
    int ns = 0;
    segx[0] = 0;
    segy[0] = post.get(0) * F.mult;
    for (int j = 1; j < posts; j++) {
      const int cur = forward_index.get(j);
      const int pc_ = post.get(cur);
      const int hy = pc_ & 0x7fff;
      if (hy == pc_) {
        ns++;
        segx[ns] = postlist.get(cur);
        segy[ns] = hy * F.mult;
      }
    }
    nseg = ns;
  }
  pc.mark(3);
  if (ilogmask) {
    const int ns = nseg;
    for (int x = 0; x < n2; x++) {
      int v;
      if (x >= segx[ns]) {
        v = segy[ns];
      } else {
        int s = 0;
        while (x >= segx[s + 1]) s++;
        const LineStep st = line_step(segx[s], segx[s + 1], segy[s], segy[s + 1], F.div_magic);
        v = line_y(st, segy[s], x - segx[s]);
      }
      ilogmask[x] = (ilog_t)v;
    }
  }
}
