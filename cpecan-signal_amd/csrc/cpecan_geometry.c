/*
 * cpecan_geometry.c -- host-side integer geometry of the banded DP: the band (one [xmyL,xmyR]
 * interval per anti-diagonal) and the split of an alignment at large anchor-free gaps.
 *
 * Reference behaviour reproduced (results are bit-exact, see tests/test_geometry.py):
 *   band_construct            impl/pairwiseAligner.c:132-184  (+ band_setCurrentDiagonal :108)
 *   getSplitPoints            impl/pairwiseAligner.c:1289-1340
 * The band is built in closed form instead of by the reference's incremental parity fix-ups:
 * between two consecutive anchor points P and N (matrix coordinates, i.e. sequence coordinate + 1)
 * the band is the rectangle x in [xLo, xHi], y in [yLo, yHi] with
 *     xLo = clamp(Px - e/2), yLo = clamp(Py - e/2), xHi = clamp(Nx + e/2), yHi = clamp(Ny + e/2)
 * and an anti-diagonal x+y = d crosses it in x in [max(xLo, d - yHi), min(xHi, d - yLo)].
 */
#include "cpecan_hip.h"

#include <math.h>
#include <stddef.h>

static int64_t clampi(int64_t z, int64_t hi) { return z < 0 ? 0 : (z > hi ? hi : z); }

/* floor division by 2 is not what the reference does: diagonal_getXCoordinate is C integer
 * division (truncation toward zero) of (xay + xmy), kept here for negative intermediate values. */
static int64_t half(int64_t v) { return v / 2; }

int cpecan_band_construct(const int64_t *anchors, int64_t n_anchors, int64_t lX, int64_t lY,
                          int64_t expansion, int32_t *xmyL, int32_t *xmyR) {
    if (lX < 0 || lY < 0 || expansion < 0 || (expansion & 1) || !xmyL || !xmyR) return CPECAN_EINVAL;
    if (n_anchors > 0 && !anchors) return CPECAN_EINVAL;
    const int64_t last = lX + lY;
    xmyL[0] = 0;
    xmyR[0] = 0;
    int64_t px = 0, py = 0; /* previous anchor point, matrix coordinates */
    int64_t d = 1;
    for (int64_t a = 0; a <= n_anchors && d <= last; a++) {
        int64_t nx = lX, ny = lY;
        if (a < n_anchors) {
            nx = anchors[2 * a] + 1;
            ny = anchors[2 * a + 1] + 1;
            /* band_construct asserts strictly increasing anchors inside the matrix (:164-169) */
            if (nx <= px || ny <= py || nx > lX || ny > lY) return CPECAN_EBAND;
        }
        const int64_t pxay = px + py, pxmy = px - py, nxay = nx + ny, nxmy = nx - ny;
        const int64_t xLo = clampi(half(pxay + pxmy - expansion), lX);
        const int64_t yHi = clampi(half(nxay - (nxmy - expansion)), lY);
        const int64_t xHi = clampi(half(nxay + nxmy + expansion), lX);
        const int64_t yLo = clampi(half(pxay - (pxmy + expansion)), lY);
        const int64_t end = nxay < last ? nxay : last;
        for (; d <= end; d++) {
            int64_t xmin = d - yHi > xLo ? d - yHi : xLo;
            int64_t xmax = d - yLo < xHi ? d - yLo : xHi;
            if (xmin > xmax) return CPECAN_EBAND;
            xmyL[d] = (int32_t) (2 * xmin - d);
            xmyR[d] = (int32_t) (2 * xmax - d);
        }
        px = nx;
        py = ny;
    }
    if (d <= last) return CPECAN_EBAND;
    return CPECAN_OK;
}

typedef struct {
    int64_t *out, n, cap;
} tuple_sink;

static void emit4(tuple_sink *s, int64_t a, int64_t b, int64_t c, int64_t d) {
    if (s->n < s->cap) {
        int64_t *o = s->out + 4 * s->n;
        o[0] = a; o[1] = b; o[2] = c; o[3] = d;
    }
    s->n++;
}

/* One anchor-free block from (x2,y2) to (x3,y3): if its area exceeds the limit, close the running
 * region half-way into the block (at most sqrt(limit) in) and restart it the same distance before
 * the block's end. */
static int cut_block(int64_t *x1, int64_t *y1, int64_t x2, int64_t y2, int64_t x3, int64_t y3,
                     int64_t limit, int skip, tuple_sink *s) {
    const int64_t w = x3 - x2, h = y3 - y2;
    if (w * h <= limit) return 0;
    const int64_t reach = (int64_t) sqrt((double) limit);
    const int64_t hx = w / 2 > reach ? reach : w / 2;
    const int64_t hy = h / 2 > reach ? reach : h / 2;
    if (!skip) emit4(s, *x1, *y1, x2 + hx, y2 + hy);
    *x1 = x3 - hx;
    *y1 = y3 - hy;
    return 1;
}

int64_t cpecan_split_points(const int64_t *anchors, int64_t n_anchors, int64_t lX, int64_t lY,
                            int64_t max_matrix_size, int ragged_left, int ragged_right,
                            int64_t *out, int64_t cap) {
    tuple_sink s = { out, 0, out ? cap : 0 };
    int64_t x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    for (int64_t i = 0; i < n_anchors; i++) {
        const int64_t x3 = anchors[2 * i], y3 = anchors[2 * i + 1];
        cut_block(&x1, &y1, x2, y2, x3, y3, max_matrix_size, ragged_left && i == 0, &s);
        x2 = x3 + 1;
        y2 = y3 + 1;
    }
    const int cut = cut_block(&x1, &y1, x2, y2, lX, lY, max_matrix_size,
                              ragged_left && n_anchors == 0, &s);
    if (!cut || !ragged_right) emit4(&s, x1, y1, lX, lY);
    return s.n;
}
