/* Developer aid (NOT product, NOT the oracle): counts the speculative rounds the branch selection (csrc/skeleton.hip
 * k_sk_select) would need on ONE component under different round schemes, on the CPU, so that a scheme can be judged before
 * it is written as a kernel.  The sequential semantics are those of oracle/skeleton_oracle.c so_sample_tree.
 *   gcc -O2 -shared -fPIC -o /tmp/libsimsel.so tools/sim_select_rounds.c -lm      (driven by tools/sim_select_rounds.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

typedef struct { float lo[3], cell; int dim[3]; i64 *start, *items; } Grid;

static int cell_of(const Grid *g, float v, int a) { return (int)floorf((v - g->lo[a]) / g->cell); }

static void grid_build(Grid *g, const float *pts, i64 n, float cell) {
    float hi[3];
    for (int a = 0; a < 3; a++) { g->lo[a] = 1e30f; hi[a] = -1e30f; }
    for (i64 i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { float v = pts[3 * i + a]; if (v < g->lo[a]) g->lo[a] = v; if (v > hi[a]) hi[a] = v; }
    g->cell = cell;
    for (int a = 0; a < 3; a++) g->dim[a] = (int)floorf((hi[a] - g->lo[a]) / cell) + 1;
    i64 nc = (i64)g->dim[0] * g->dim[1] * g->dim[2];
    g->start = (i64 *)calloc(nc + 1, sizeof(i64));
    g->items = (i64 *)malloc((n > 0 ? n : 1) * sizeof(i64));
    for (i64 i = 0; i < n; i++) {
        i64 c = ((i64)cell_of(g, pts[3 * i], 0) * g->dim[1] + cell_of(g, pts[3 * i + 1], 1)) * g->dim[2] + cell_of(g, pts[3 * i + 2], 2);
        g->start[c + 1]++;
    }
    for (i64 c = 0; c < nc; c++) g->start[c + 1] += g->start[c];
    i64 *cur = (i64 *)malloc((nc + 1) * sizeof(i64));
    memcpy(cur, g->start, (nc + 1) * sizeof(i64));
    for (i64 i = 0; i < n; i++) {
        i64 c = ((i64)cell_of(g, pts[3 * i], 0) * g->dim[1] + cell_of(g, pts[3 * i + 1], 1)) * g->dim[2] + cell_of(g, pts[3 * i + 2], 2);
        g->items[cur[c]++] = i;
    }
    free(cur);
}

static float d2f(const float *a, const float *b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz;
}

/* claims of a path (root side first): points whose nearest path vertex (within rp = max path radius, ties: first) is closer
 * than that vertex's own radius.  Returns the count, fills out[] (capacity n).  bd2 / bpos: scratch, bpos all -1 on entry/exit. */
static i64 claims_of(const Grid *g, const float *pts, const float *rad, const i64 *path, i64 len, float *bd2, i64 *bpos,
                     i64 *out, i64 *ncand) {
    float rp = 0;
    for (i64 q = 0; q < len; q++) if (rad[path[q]] > rp) rp = rad[path[q]];
    float rp2 = rp * rp;
    int reach = (int)ceilf(rp / g->cell);
    if (reach < 1) reach = 1;
    i64 nt = 0, cand = 0;
    for (i64 q = 0; q < len; q++) {
        const float *pv = pts + 3 * path[q];
        int c0[3];
        for (int a = 0; a < 3; a++) c0[a] = cell_of(g, pv[a], a);
        for (int x = c0[0] - reach; x <= c0[0] + reach; x++) {
            if (x < 0 || x >= g->dim[0]) continue;
            for (int y = c0[1] - reach; y <= c0[1] + reach; y++) {
                if (y < 0 || y >= g->dim[1]) continue;
                for (int z = c0[2] - reach; z <= c0[2] + reach; z++) {
                    if (z < 0 || z >= g->dim[2]) continue;
                    i64 c = ((i64)x * g->dim[1] + y) * g->dim[2] + z;
                    for (i64 s = g->start[c]; s < g->start[c + 1]; s++) {
                        i64 p = g->items[s];
                        float d2 = d2f(pts + 3 * p, pv);
                        cand++;
                        if (!(d2 < rp2)) continue;
                        if (bpos[p] < 0) { out[nt++] = p; bd2[p] = d2; bpos[p] = q; }
                        else if (d2 < bd2[p]) { bd2[p] = d2; bpos[p] = q; }
                    }
                }
            }
        }
    }
    i64 n_on = 0;
    for (i64 t = 0; t < nt; t++) {
        i64 p = out[t];
        int on = sqrtf(bd2[p]) < rad[path[bpos[p]]];
        bpos[p] = -1;
        if (on) out[n_on++] = p;
    }
    if (ncand) *ncand = cand;
    return n_on;
}

static const float *g_key;
static int cmp_order(const void *a, const void *b) {
    i64 x = *(const i64 *)a, y = *(const i64 *)b;
    if (g_key[x] > g_key[y]) return -1;
    if (g_key[x] < g_key[y]) return 1;
    return x < y ? -1 : (x > y ? 1 : 0);
}

#define MAXS 256
/* stats: [0] rounds (speculative), [1] slots, [2] commits, [3] big (whole-workgroup) iterations, [4] rounds ended by a conflict,
 * [5] by a wrong guess (a shadowed entry survives), [6] by running out of slots / entries, [7] repairs (mode 1), [8] branches,
 * [9] candidates of speculative slots, [10] candidates of big iterations, [11] rounds cut by a big slot, [12] entries looked at
 * mode: 0 = the round ends at the first conflict; 1 = a conflicting slot is repaired in place (walk cut at the first vertex an
 * accepted earlier slot took) and the replay goes on. */
void sim_rounds(i64 n, const float *pts, const float *rad, const i64 *preds, const float *dist_in, int E, int NS, float prune,
                int mode, int wpath, int NSB, i64 *stats, i64 *branch_len_out) {
    float *dist = (float *)malloc(n * sizeof(float));
    uint8_t *term = (uint8_t *)calloc(n, 1);
    float rmax = 0;
    for (i64 i = 0; i < n; i++) { dist[i] = preds[i] > 0 ? dist_in[i] : -1.0f; if (rad[i] > rmax) rmax = rad[i]; }
    Grid g;
    grid_build(&g, pts, n, rmax > 0 ? rmax * 0.25f : 1.0f);
    i64 *order = (i64 *)malloc(n * sizeof(i64));
    for (i64 i = 0; i < n; i++) order[i] = i;
    g_key = dist;
    float *key0 = (float *)malloc(n * sizeof(float));
    memcpy(key0, dist, n * sizeof(float));
    g_key = key0;
    qsort(order, n, sizeof(i64), cmp_order);
    float *bd2 = (float *)malloc(n * sizeof(float));
    i64 *bpos = (i64 *)malloc(n * sizeof(i64));
    for (i64 i = 0; i < n; i++) bpos[i] = -1;
    uint64_t *mark = (uint64_t *)calloc(n, sizeof(uint64_t)); /* bit s: claimed (or on the path of) slot s this round */
    i64 *ent = (i64 *)malloc(E * sizeof(i64));
    int *slot_of = (int *)malloc(E * sizeof(int));
    i64 *spath[MAXS], *sclaim[MAXS];
    i64 slen[MAXS], snc[MAXS], sterm[MAXS];
    const int nalloc = mode == 4 ? NS : NS + NSB;
    for (int s = 0; s < nalloc; s++) { spath[s] = (i64 *)malloc(n * sizeof(i64)); sclaim[s] = (i64 *)malloc(n * sizeof(i64)); }
    i64 *rpath = (i64 *)malloc(n * sizeof(i64)), *rclaim = (i64 *)malloc(n * sizeof(i64));
    memset(stats, 0, 16 * sizeof(i64));
    i64 cursor = 0, nb = 0;
    for (;;) {
        while (cursor < n && !(key0[order[cursor]] > 0.0f && dist[order[cursor]] > 0.0f)) {
            if (!(key0[order[cursor]] > 0.0f)) { cursor = n; break; }
            cursor++;
        }
        if (cursor >= n) break;
        /* entries: the first E live vertices */
        int ne = 0;
        for (i64 j = cursor; j < n && ne < E; j++) {
            i64 v = order[j];
            if (!(key0[v] > 0.0f)) break;
            if (dist[v] > 0.0f) ent[ne++] = v;
        }
        /* shadow prune */
        int nc = 0;
        {
            uint8_t *sh = (uint8_t *)calloc(ne, 1);
            int cut = ne;
            for (int u = 0; u < ne; u++) {
                slot_of[u] = -1;
                if (sh[u]) continue;
                if (nc == NS) { cut = u; break; }
                slot_of[u] = nc++;
                float ur = rad[ent[u]] * prune;
                for (int v = u + 1; v < ne; v++)
                    if (d2f(pts + 3 * ent[v], pts + 3 * ent[u]) < ur * ur) sh[v] = 1;
            }
            ne = cut;
            free(sh);
        }
        stats[12] += ne;
        /* speculative walks + claims against the state at the start of the round */
        int nslots = 0, big_first = 0;
        for (int u = 0; u < ne; u++) {
            int s = slot_of[u];
            if (s < 0) continue;
            i64 len = 0, idx = ent[u];
            while (idx >= 0 && !term[idx]) { spath[s][len++] = idx; idx = preds[idx]; }
            sterm[s] = idx;
            for (i64 a = 0, b = len - 1; a < b; a++, b--) { i64 t = spath[s][a]; spath[s][a] = spath[s][b]; spath[s][b] = t; }
            slen[s] = len;
            if (len > wpath) {  /* too long for one wavefront */
                if (s == 0) big_first = 1;
                else { ne = u; stats[11]++; }  /* the round ends before this entry */
                break;
            }
            i64 cand;
            snc[s] = claims_of(&g, pts, rad, spath[s], len, bd2, bpos, sclaim[s], &cand);
            stats[9] += cand;
            nslots = s + 1;
        }
        if ((mode == 2 || mode == 3) && !big_first) {
            /* phase B: entries without a slot that no phase-A slot claimed ("survivors") get slots of their own, no pruning */
            for (int s = 0; s < nslots; s++) {
                for (i64 t = 0; t < snc[s]; t++) mark[sclaim[s][t]] |= 1ull << (s & 63);
                for (i64 q = 0; q < slen[s]; q++) mark[spath[s][q]] |= 1ull << (s & 63);
            }
            int nb_slots = nslots;
            /* mode 3: tentative replay over the phase-A slots (tips only): a slot whose tip an earlier live slot claims is dead,
             * and what only dead slots claim is not claimed */
            uint64_t live_slots = ~0ull;
            if (mode == 3) {
                live_slots = 0;
                for (int u = 0; u < ne; u++) {
                    int s = slot_of[u];
                    if (s < 0) continue;
                    if (mark[ent[u]] & live_slots & ~(1ull << (s & 63))) continue;
                    live_slots |= 1ull << (s & 63);
                }
            }
            for (int u = 0; u < ne; u++) {
                if (slot_of[u] >= 0 || (mark[ent[u]] & live_slots)) continue;
                if (nb_slots >= NS + NSB) { ne = u; break; }
                int s = nb_slots;
                i64 len = 0, idx = ent[u];
                while (idx >= 0 && !term[idx]) { spath[s][len++] = idx; idx = preds[idx]; }
                sterm[s] = idx;
                for (i64 a = 0, b = len - 1; a < b; a++, b--) { i64 t = spath[s][a]; spath[s][a] = spath[s][b]; spath[s][b] = t; }
                slen[s] = len;
                if (len > wpath) { ne = u; stats[11]++; break; }
                i64 cand;
                snc[s] = claims_of(&g, pts, rad, spath[s], len, bd2, bpos, sclaim[s], &cand);
                stats[9] += cand;
                stats[13]++;
                slot_of[u] = s;
                nb_slots++;
            }
            for (int s = 0; s < nslots; s++) {
                for (i64 t = 0; t < snc[s]; t++) mark[sclaim[s][t]] = 0;
                for (i64 q = 0; q < slen[s]; q++) mark[spath[s][q]] = 0;
            }
            nslots = nb_slots;
        }
        if (big_first) {  /* whole-workgroup iteration: sequential step for entry 0 */
            i64 cand;
            i64 len = slen[0];
            i64 k = claims_of(&g, pts, rad, spath[0], len, bd2, bpos, sclaim[0], &cand);
            stats[10] += cand;
            for (i64 t = 0; t < k; t++) { dist[sclaim[0][t]] = -1.0f; term[sclaim[0][t]] = 1; }
            for (i64 q = 0; q < len; q++) { dist[spath[0][q]] = -1.0f; term[spath[0][q]] = 1; }
            if (len >= 2) { if (branch_len_out) branch_len_out[nb] = len; nb++; }
            stats[3]++;
            continue;
        }
        stats[0]++;
        stats[1] += nslots;
        /* marks */
        for (int s = 0; s < nslots; s++) {
            for (i64 t = 0; t < snc[s]; t++) mark[sclaim[s][t]] |= 1ull << s;
            for (i64 q = 0; q < slen[s]; q++) mark[spath[s][q]] |= 1ull << (s & 63);
        }
        /* replay */
        uint64_t alive = 0;
        int end_reason = 6, late_left = NSB;
        for (int u = 0; u < ne; u++) {
            i64 tip = ent[u];
            if (dist[tip] < 0.0f) continue;  /* taken by a repaired slot (mode 1) or an accepted one */
            if (mark[tip] & alive) {  /* would never have been selected -- but with repairs the true claims may differ: check state */
                if (mode == 0) continue;
            }
            int s = slot_of[u];
            if (mode == 0) {
                if (s < 0 || s >= nslots) { end_reason = s < 0 ? 5 : 6; break; }
                uint64_t wm = 0;
                for (i64 q = 0; q < slen[s]; q++) wm |= mark[spath[s][q]];
                i64 tv = sterm[s] < 0 ? n - 1 : sterm[s];
                wm |= mark[tv];
                if (wm & alive) { end_reason = 4; break; }
                alive |= 1ull << s;
                for (i64 t = 0; t < snc[s]; t++) { dist[sclaim[s][t]] = -1.0f; term[sclaim[s][t]] = 1; }
                for (i64 q = 0; q < slen[s]; q++) { dist[spath[s][q]] = -1.0f; term[spath[s][q]] = 1; }
                if (slen[s] >= 2) { if (branch_len_out) branch_len_out[nb] = slen[s]; nb++; }
                stats[2]++;
            } else {
                /* mode 1: commits are applied to the state immediately, so "dead" and "conflict" are read off the state */
                if (mode == 4) {  /* late evaluation: a live entry without a slot is evaluated on the spot (budget NSB per round) */
                    if (s < 0 || s >= nslots) {
                        if (late_left == 0) { end_reason = 6; break; }
                        late_left--;
                        i64 len = 0, idx = tip;
                        while (idx >= 0 && !term[idx]) { rpath[len++] = idx; idx = preds[idx]; }
                        if (len > wpath) { end_reason = 6; stats[11]++; break; }  /* a big one: next round's entry 0 */
                        for (i64 a = 0, b = len - 1; a < b; a++, b--) { i64 t = rpath[a]; rpath[a] = rpath[b]; rpath[b] = t; }
                        i64 cand;
                        i64 k = claims_of(&g, pts, rad, rpath, len, bd2, bpos, rclaim, &cand);
                        stats[9] += cand;
                        stats[14]++;
                        for (i64 t = 0; t < k; t++) { dist[rclaim[t]] = -1.0f; term[rclaim[t]] = 1; }
                        for (i64 q = 0; q < len; q++) { dist[rpath[q]] = -1.0f; term[rpath[q]] = 1; }
                        if (len >= 2) { if (branch_len_out) branch_len_out[nb] = len; nb++; }
                        stats[2]++;
                        continue;
                    }
                }
                if (s < 0 || s >= nslots) { end_reason = s < 0 ? 5 : 6; break; }
                /* true walk against the CURRENT state */
                i64 len = 0, idx = tip;
                while (idx >= 0 && !term[idx]) { rpath[len++] = idx; idx = preds[idx]; }
                for (i64 a = 0, b = len - 1; a < b; a++, b--) { i64 t = rpath[a]; rpath[a] = rpath[b]; rpath[b] = t; }
                const i64 *cl = sclaim[s];
                i64 k = snc[s];
                if (len != slen[s]) {  /* repair: claims of the shortened path */
                    if (mode == 4) { if (late_left == 0) { end_reason = 6; break; } late_left--; }
                    i64 cand;
                    k = claims_of(&g, pts, rad, rpath, len, bd2, bpos, rclaim, &cand);
                    cl = rclaim;
                    stats[7]++;
                }
                for (i64 t = 0; t < k; t++) { dist[cl[t]] = -1.0f; term[cl[t]] = 1; }
                for (i64 q = 0; q < len; q++) { dist[rpath[q]] = -1.0f; term[rpath[q]] = 1; }
                if (len >= 2) { if (branch_len_out) branch_len_out[nb] = len; nb++; }
                stats[2]++;
            }
        }
        stats[end_reason]++;
        for (int s = 0; s < nslots; s++) {
            for (i64 t = 0; t < snc[s]; t++) mark[sclaim[s][t]] = 0;
            for (i64 q = 0; q < slen[s]; q++) mark[spath[s][q]] = 0;
        }
    }
    stats[8] = nb;
    free(dist); free(term); free(order); free(key0); free(bd2); free(bpos); free(mark); free(ent); free(slot_of);
    for (int s = 0; s < nalloc; s++) { free(spath[s]); free(sclaim[s]); }
    free(rpath); free(rclaim);
    free(g.start); free(g.items);
}
