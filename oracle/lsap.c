/*
 * oracle/lsap.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, fp64) of the rectangular linear-sum-assignment
 * solver the reference calls as scipy.optimize.linear_sum_assignment
 * (call sites: /root/reference/models/matcher.py:8,85 and
 * /root/reference/models/mdetr.py:18,100,539).  The algorithm lives in a
 * third-party dependency that is NOT under /root/reference: SciPy (pinned
 * scipy==1.7.3 in /root/reference/requirements.txt:67); it is the
 * "modified Jonker-Volgenant algorithm with no initialization" of
 * D. F. Crouse, "On implementing 2D rectangular assignment algorithms",
 * IEEE Trans. AES 52(4), 2016 -- a shortest-augmenting-path method.
 *
 * Pinning: tests/test_oracle_lsap.py checks this file against the SciPy
 * installed in the build container (1.15.3) on the known-answer tests of
 * SURVEY.md section 4 plus thousands of random / tied / integer / inf cases,
 * and against tests/golden/lsap_kat.json (generated from SciPy by
 * tests/golden/make_lsap_kat.py).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * Behaviour restated (what a caller can observe):
 *   - cost is row-major [nr, nc] float64;
 *   - nr == 0 or nc == 0 -> empty assignment;
 *   - any NaN or -inf entry -> error -2 ("matrix contains invalid numeric entries");
 *   - if nr > nc the transposed problem is solved, so every *column* of the
 *     caller's matrix gets matched and the result is ordered by row index;
 *   - rows are augmented one at a time (0, 1, ...) by a Dijkstra-like scan over
 *     the not-yet-scanned columns; among columns of equal reduced path cost a
 *     column that is still unassigned wins, otherwise the first one met in the
 *     scan order of the "remaining" list (initialised in descending column
 *     order and compacted by moving its last element into the freed slot);
 *   - infeasible (every remaining path cost +inf) -> error -1.
 *   Output: n = min(nr,nc) pairs (row[i], col[i]) with row ascending.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_INFEASIBLE (-1)
#define ORC_INVALID (-2)
#define ORC_NOMEM (-3)

typedef struct {
    int64_t nr, nc;
    const double *c; /* [nr, nc] row-major, already oriented nr <= nc */
    double *u, *v, *spc;
    int64_t *path, *col4row, *row4col, *remaining;
    unsigned char *in_sr, *in_sc;
} lsap_ws;

/* One shortest augmenting path from row `start`; returns the sink column. */
static int64_t grow_path(lsap_ws *w, int64_t start, double *min_out)
{
    const int64_t nc = w->nc;
    int64_t live = nc, sink = -1, i = start;
    double floor_val = 0.0;

    for (int64_t t = 0; t < nc; ++t) {
        w->remaining[t] = nc - 1 - t;
        w->spc[t] = INFINITY;
        w->in_sc[t] = 0;
    }
    memset(w->in_sr, 0, (size_t)w->nr);

    while (sink < 0) {
        int64_t pick = -1;
        double best = INFINITY;
        w->in_sr[i] = 1;
        for (int64_t t = 0; t < live; ++t) {
            const int64_t j = w->remaining[t];
            const double r = floor_val + w->c[i * nc + j] - w->u[i] - w->v[j];
            if (r < w->spc[j]) {
                w->path[j] = i;
                w->spc[j] = r;
            }
            if (w->spc[j] < best || (w->spc[j] == best && w->row4col[j] < 0)) {
                best = w->spc[j];
                pick = t;
            }
        }
        floor_val = best;
        if (best == INFINITY) return -1;
        {
            const int64_t j = w->remaining[pick];
            if (w->row4col[j] < 0) sink = j; else i = w->row4col[j];
            w->in_sc[j] = 1;
            w->remaining[pick] = w->remaining[--live];
        }
    }
    *min_out = floor_val;
    return sink;
}

static int cmp_pair(const void *a, const void *b)
{
    const int64_t x = ((const int64_t *)a)[0], y = ((const int64_t *)b)[0];
    return (x > y) - (x < y);
}

/* Solve; rows_out/cols_out must hold min(nr,nc) entries.  Returns ORC_*. */
int oracle_lsap(const double *cost, int64_t nr, int64_t nc, int64_t *rows_out, int64_t *cols_out)
{
    if (nr == 0 || nc == 0) return ORC_OK;
    const int flip = nc < nr;
    const int64_t R = flip ? nc : nr, C = flip ? nr : nc;
    double *work = (double *)malloc(sizeof(double) * (size_t)(R * C));
    if (!work) return ORC_NOMEM;
    for (int64_t i = 0; i < nr; ++i)
        for (int64_t j = 0; j < nc; ++j) {
            const double x = cost[i * nc + j];
            if (x != x || x == -INFINITY) { free(work); return ORC_INVALID; }
            if (flip) work[j * nr + i] = x; else work[i * nc + j] = x;
        }

    lsap_ws w;
    w.nr = R; w.nc = C; w.c = work;
    w.u = (double *)calloc((size_t)R, sizeof(double));
    w.v = (double *)calloc((size_t)C, sizeof(double));
    w.spc = (double *)malloc(sizeof(double) * (size_t)C);
    w.path = (int64_t *)malloc(sizeof(int64_t) * (size_t)C);
    w.col4row = (int64_t *)malloc(sizeof(int64_t) * (size_t)R);
    w.row4col = (int64_t *)malloc(sizeof(int64_t) * (size_t)C);
    w.remaining = (int64_t *)malloc(sizeof(int64_t) * (size_t)C);
    w.in_sr = (unsigned char *)malloc((size_t)R);
    w.in_sc = (unsigned char *)malloc((size_t)C);
    int rc = ORC_OK;
    if (!w.u || !w.v || !w.spc || !w.path || !w.col4row || !w.row4col || !w.remaining || !w.in_sr || !w.in_sc) {
        rc = ORC_NOMEM;
        goto done;
    }
    for (int64_t j = 0; j < C; ++j) { w.path[j] = -1; w.row4col[j] = -1; }
    for (int64_t i = 0; i < R; ++i) w.col4row[i] = -1;

    for (int64_t cur = 0; cur < R; ++cur) {
        double m = 0.0;
        const int64_t sink = grow_path(&w, cur, &m);
        if (sink < 0) { rc = ORC_INFEASIBLE; goto done; }
        /* dual update */
        w.u[cur] += m;
        for (int64_t i = 0; i < R; ++i)
            if (w.in_sr[i] && i != cur) w.u[i] += m - w.spc[w.col4row[i]];
        for (int64_t j = 0; j < C; ++j)
            if (w.in_sc[j]) w.v[j] -= m - w.spc[j];
        /* flip the matching along the path */
        int64_t j = sink;
        for (;;) {
            const int64_t i = w.path[j];
            w.row4col[j] = i;
            const int64_t prev = w.col4row[i];
            w.col4row[i] = j;
            j = prev;
            if (i == cur) break;
        }
    }

    if (flip) {
        /* pairs (orig_row = col4row[k], orig_col = k), ordered by orig_row */
        int64_t *pairs = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)R);
        if (!pairs) { rc = ORC_NOMEM; goto done; }
        for (int64_t k = 0; k < R; ++k) { pairs[2 * k] = w.col4row[k]; pairs[2 * k + 1] = k; }
        qsort(pairs, (size_t)R, 2 * sizeof(int64_t), cmp_pair);
        for (int64_t k = 0; k < R; ++k) { rows_out[k] = pairs[2 * k]; cols_out[k] = pairs[2 * k + 1]; }
        free(pairs);
    } else {
        for (int64_t i = 0; i < R; ++i) { rows_out[i] = i; cols_out[i] = w.col4row[i]; }
    }
done:
    free(work); free(w.u); free(w.v); free(w.spc); free(w.path); free(w.col4row);
    free(w.row4col); free(w.remaining); free(w.in_sr); free(w.in_sc);
    return rc;
}

/* Batched helper used by the matcher oracle: cost is [Q, Ttot] float64 for one
 * image block already sliced by the caller. */
int oracle_lsap_version(void) { return 1; }
