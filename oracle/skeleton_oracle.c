/* ORACLE -- test infrastructure only.  Never linked into or called from the product path; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Plain-C restatement of the reference's skeleton stage (all file:line relative to
 * /root/reference/smart_tree):
 *   knn / nn                skeleton/graph.py:12-33      (wrapper around frnn.frnn_grid_points)
 *   outlier_removal         skeleton/filter.py:6-11
 *   nn_graph + make_edges   skeleton/graph.py:36-40,52-60  (incl. the `idx > 0` quirk, :59)
 *   connected components    data_types/graph.py:32-51    (cugraph.connected_components + min size + sort)
 *   process_subgraph        skeleton/skeletonize.py:57-95 (vertex renumber by rank, root = argmin y)
 *   shortest_paths          skeleton/shortest_path.py:12-21 (cugraph.sssp)
 *   pred_graph + 2nd sssp   skeleton/shortest_path.py:46-55, skeletonize.py:80-85
 *   sample_tree             skeleton/path.py:49-140 with trace_route :9-16, select_path_points :19-46
 *
 * Third-party arithmetic restated (sources NOT under /root/reference -- parity UNPINNED):
 *   FRNN (lxxue/FRNN, unpinned git HEAD): for each query the <= K nearest points with
 *     d2 < r*r, ascending, idx -1 / dist -1 padding, SQUARED distances.  Canonical here:
 *     d2 = (dx*dx + dy*dy) + dz*dz in float32 without contraction, ties broken by smaller index.
 *   cugraph 23.02 connected_components / subgraph: weak components of the undirected graph;
 *     canonical component order = size descending, then smallest member vertex ascending.
 *   cugraph 23.02 sssp: distances are the least fixed point of d[v] = min_u fl32(d[u] + w(u,v))
 *     (unique, order independent).  Predecessor ties are unspecified in cugraph; canonical here:
 *     pred[v] = smallest u != v with fl32(d[u]+w) == d[v] and d[u] < d[v]; vertices whose only
 *     tight in-neighbours sit on the same distance plateau are resolved in synchronous rounds
 *     from already-resolved plateau members (smallest u again).  This is always a tree.
 *   torch.norm for the predecessor-tree edge lengths: same float32 formula as above, so the
 *     second SSSP reproduces d exactly; it is still evaluated literally (tree order) below.
 *   select_path_points' K=1 query: ties between path vertices at equal d2 go to the vertex that
 *     comes first on the (root-side-first) path.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

static inline float dist2f(const float *a, const float *b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float s = dx * dx;
    float t = dy * dy;
    s = s + t;
    t = dz * dz;
    return s + t;
}

/* ---------------------------------------------------------------- uniform grid over points */
typedef struct {
    float lo[3];
    float cell;
    int dim[3];
    i64 *start; /* [ncell+1] */
    i64 *items; /* [n] point indices, ascending inside each cell */
} Grid;

static inline int cell_of(const Grid *g, float v, int ax) {
    int c = (int)floorf((v - g->lo[ax]) / g->cell);
    if (c < 0) c = 0;
    if (c >= g->dim[ax]) c = g->dim[ax] - 1;
    return c;
}

static void grid_build(Grid *g, const float *pts, i64 n, float cell) {
    float hi[3];
    for (int a = 0; a < 3; a++) { g->lo[a] = 0; hi[a] = 0; }
    for (i64 i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            float v = pts[3 * i + a];
            if (i == 0 || v < g->lo[a]) g->lo[a] = v;
            if (i == 0 || v > hi[a]) hi[a] = v;
        }
    if (!(cell > 0)) cell = 1.0f;
    for (;;) {
        double total = 1;
        for (int a = 0; a < 3; a++) {
            g->dim[a] = (int)floorf((hi[a] - g->lo[a]) / cell) + 1;
            if (g->dim[a] < 1) g->dim[a] = 1;
            total *= g->dim[a];
        }
        if (total <= 32e6) break;
        cell *= 2;
    }
    g->cell = cell;
    i64 nc = (i64)g->dim[0] * g->dim[1] * g->dim[2];
    g->start = (i64 *)calloc(nc + 1, sizeof(i64));
    g->items = (i64 *)malloc((n > 0 ? n : 1) * sizeof(i64));
    i64 *cid = (i64 *)malloc((n > 0 ? n : 1) * sizeof(i64));
    for (i64 i = 0; i < n; i++) {
        i64 c = ((i64)cell_of(g, pts[3 * i], 0) * g->dim[1] + cell_of(g, pts[3 * i + 1], 1)) * g->dim[2] +
                cell_of(g, pts[3 * i + 2], 2);
        cid[i] = c;
        g->start[c + 1]++;
    }
    for (i64 c = 0; c < nc; c++) g->start[c + 1] += g->start[c];
    i64 *fill = (i64 *)malloc((nc > 0 ? nc : 1) * sizeof(i64));
    memcpy(fill, g->start, nc * sizeof(i64));
    for (i64 i = 0; i < n; i++) g->items[fill[cid[i]]++] = i;
    free(fill);
    free(cid);
}

static void grid_free(Grid *g) {
    free(g->start);
    free(g->items);
}

/* ------------------------------------------------------------------------------- kNN (K9) */
/* idx [n1,K] (-1 pad), d2 [n1,K] (-1 pad); neighbours with d2 < r*r ordered by (d2, index). */
void so_knn(i64 n1, const float *src, i64 n2, const float *dst, int K, float r, i64 *idx, float *d2out) {
    Grid g;
    grid_build(&g, dst, n2, r);
    float r2 = r * r;
    int reach = (int)ceilf(r / g.cell);
    if (reach < 1) reach = 1;
    for (i64 i = 0; i < n1; i++) {
        i64 *bi = idx + i * K;
        float *bd = d2out + i * K;
        int cnt = 0;
        const float *p = src + 3 * i;
        int c0[3];
        for (int a = 0; a < 3; a++) c0[a] = (int)floorf((p[a] - g.lo[a]) / g.cell);
        for (int x = c0[0] - reach; x <= c0[0] + reach; x++) {
            if (x < 0 || x >= g.dim[0]) continue;
            for (int y = c0[1] - reach; y <= c0[1] + reach; y++) {
                if (y < 0 || y >= g.dim[1]) continue;
                for (int z = c0[2] - reach; z <= c0[2] + reach; z++) {
                    if (z < 0 || z >= g.dim[2]) continue;
                    i64 c = ((i64)x * g.dim[1] + y) * g.dim[2] + z;
                    for (i64 s = g.start[c]; s < g.start[c + 1]; s++) {
                        i64 j = g.items[s];
                        float d2 = dist2f(p, dst + 3 * j);
                        if (!(d2 < r2)) continue;
                        /* insertion into the sorted top-K by (d2, j) */
                        int pos = cnt;
                        while (pos > 0 && (bd[pos - 1] > d2 || (bd[pos - 1] == d2 && bi[pos - 1] > j))) pos--;
                        if (pos >= K) continue;
                        int last = cnt < K ? cnt : K - 1;
                        for (int q = last; q > pos; q--) { bd[q] = bd[q - 1]; bi[q] = bi[q - 1]; }
                        bd[pos] = d2;
                        bi[pos] = j;
                        if (cnt < K) cnt++;
                    }
                }
            }
        }
        for (int q = cnt; q < K; q++) { bi[q] = -1; bd[q] = -1.0f; }
    }
    grid_free(&g);
}

/* ------------------------------------------------------------ outlier_removal (filter.py) */
void so_outlier_mask(i64 n, const float *pts, const float *radii, int nb, uint8_t *keep) {
    float rmax = 0;
    for (i64 i = 0; i < n; i++)
        if (i == 0 || radii[i] > rmax) rmax = radii[i];
    i64 *idx = (i64 *)malloc((size_t)(n > 0 ? n : 1) * nb * sizeof(i64));
    float *d2 = (float *)malloc((size_t)(n > 0 ? n : 1) * nb * sizeof(float));
    so_knn(n, pts, n, pts, nb, rmax, idx, d2);
    for (i64 i = 0; i < n; i++) {
        int ok = 0;
        for (int k = 0; k < nb; k++)
            if (idx[i * nb + k] != -1 && sqrtf(d2[i * nb + k]) < radii[i]) ok++;
        keep[i] = ok == nb;
    }
    free(idx);
    free(d2);
}

/* ------------------------------------------------------- nn_graph + make_edges (graph.py) */
/* radii already clamped by the caller (skeletonize.py:39).  Returns E; edges [E,2], w [E]. */
i64 so_nn_graph(i64 n, const float *pts, const float *radii, int K, i64 *edges, float *w) {
    float rmax = 0;
    for (i64 i = 0; i < n; i++)
        if (i == 0 || radii[i] > rmax) rmax = radii[i];
    i64 *idx = (i64 *)malloc((size_t)(n > 0 ? n : 1) * K * sizeof(i64));
    float *d2 = (float *)malloc((size_t)(n > 0 ? n : 1) * K * sizeof(float));
    so_knn(n, pts, n, pts, K, rmax, idx, d2);
    i64 E = 0;
    for (i64 i = 0; i < n; i++)
        for (int k = 0; k < K; k++) {
            i64 j = idx[i * K + k];
            if (j < 0) continue;
            float d = sqrtf(d2[i * K + k]);
            if (d > radii[i]) continue; /* graph.py:38 */
            if (!(j > 0)) continue;     /* graph.py:59 `idx > 0` */
            edges[2 * E] = i;
            edges[2 * E + 1] = j;
            w[E] = d;
            E++;
        }
    free(idx);
    free(d2);
    return E;
}

/* ------------------------------------------------------------------ connected components */
static i64 uf_find(i64 *p, i64 x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}

/* label[v] = smallest vertex id of v's component */
void so_cc_labels(i64 n, i64 E, const i64 *edges, i64 *label) {
    for (i64 i = 0; i < n; i++) label[i] = i;
    for (i64 e = 0; e < E; e++) {
        i64 a = uf_find(label, edges[2 * e]), b = uf_find(label, edges[2 * e + 1]);
        if (a == b) continue;
        if (a < b) label[b] = a; else label[a] = b;
    }
    for (i64 i = 0; i < n; i++) label[i] = uf_find(label, i);
}

/* ------------------------------------------------------------------------------- SSSP ---- */
typedef struct { float d; i64 v; } HeapItem;

static void heap_push(HeapItem *h, i64 *hn, float d, i64 v) {
    i64 i = (*hn)++;
    h[i].d = d; h[i].v = v;
    while (i > 0) {
        i64 p = (i - 1) / 2;
        if (h[p].d < h[i].d || (h[p].d == h[i].d && h[p].v <= h[i].v)) break;
        HeapItem t = h[p]; h[p] = h[i]; h[i] = t;
        i = p;
    }
}

static HeapItem heap_pop(HeapItem *h, i64 *hn) {
    HeapItem top = h[0];
    h[0] = h[--(*hn)];
    i64 i = 0;
    for (;;) {
        i64 l = 2 * i + 1, r = l + 1, m = i;
        if (l < *hn && (h[l].d < h[m].d || (h[l].d == h[m].d && h[l].v < h[m].v))) m = l;
        if (r < *hn && (h[r].d < h[m].d || (h[r].d == h[m].d && h[r].v < h[m].v))) m = r;
        if (m == i) break;
        HeapItem t = h[m]; h[m] = h[i]; h[i] = t;
        i = m;
    }
    return top;
}

/* Undirected CSR over n vertices from an edge list (both directions inserted). */
static void build_csr(i64 n, i64 E, const i64 *edges, const float *w, i64 **off_o, i64 **adj_o, float **aw_o) {
    i64 *off = (i64 *)calloc(n + 1, sizeof(i64));
    for (i64 e = 0; e < E; e++) { off[edges[2 * e] + 1]++; off[edges[2 * e + 1] + 1]++; }
    for (i64 i = 0; i < n; i++) off[i + 1] += off[i];
    i64 *adj = (i64 *)malloc((2 * E + 1) * sizeof(i64));
    float *aw = (float *)malloc((2 * E + 1) * sizeof(float));
    i64 *fill = (i64 *)malloc((n + 1) * sizeof(i64));
    memcpy(fill, off, (n + 1) * sizeof(i64));
    for (i64 e = 0; e < E; e++) {
        i64 a = edges[2 * e], b = edges[2 * e + 1];
        adj[fill[a]] = b; aw[fill[a]++] = w[e];
        adj[fill[b]] = a; aw[fill[b]++] = w[e];
    }
    free(fill);
    *off_o = off; *adj_o = adj; *aw_o = aw;
}

/* dist (least fixed point, float32) + canonical predecessors; unreachable: dist = INF, pred = -1. */
void so_sssp(i64 n, i64 E, const i64 *edges, const float *w, i64 root, float *dist, i64 *pred) {
    i64 *off, *adj;
    float *aw;
    build_csr(n, E, edges, w, &off, &adj, &aw);
    for (i64 i = 0; i < n; i++) { dist[i] = INFINITY; pred[i] = -1; }
    HeapItem *heap = (HeapItem *)malloc((2 * E + n + 2) * sizeof(HeapItem));
    i64 hn = 0;
    dist[root] = 0.0f;
    heap_push(heap, &hn, 0.0f, root);
    while (hn > 0) {
        HeapItem it = heap_pop(heap, &hn);
        if (it.d > dist[it.v]) continue;
        for (i64 s = off[it.v]; s < off[it.v + 1]; s++) {
            float nd = it.d + aw[s];
            if (nd < dist[adj[s]]) { dist[adj[s]] = nd; heap_push(heap, &hn, nd, adj[s]); }
        }
    }
    free(heap);
    /* canonical predecessors */
    uint8_t *resolved = (uint8_t *)calloc(n, 1);
    resolved[root] = 1;
    for (i64 v = 0; v < n; v++) {
        if (v == root || isinf(dist[v])) continue;
        i64 best = -1;
        for (i64 s = off[v]; s < off[v + 1]; s++) {
            i64 u = adj[s];
            if (u == v || !(dist[u] < dist[v])) continue;
            if (dist[u] + aw[s] == dist[v] && (best < 0 || u < best)) best = u;
        }
        if (best >= 0) { pred[v] = best; resolved[v] = 1; }
    }
    for (;;) { /* plateau rounds (synchronous) */
        i64 changed = 0;
        uint8_t *next = (uint8_t *)malloc(n);
        memcpy(next, resolved, n);
        for (i64 v = 0; v < n; v++) {
            if (resolved[v] || isinf(dist[v])) continue;
            i64 best = -1;
            for (i64 s = off[v]; s < off[v + 1]; s++) {
                i64 u = adj[s];
                if (u == v || !resolved[u] || dist[u] != dist[v]) continue;
                if (dist[u] + aw[s] == dist[v] && (best < 0 || u < best)) best = u;
            }
            if (best >= 0) { pred[v] = best; next[v] = 1; changed++; }
        }
        memcpy(resolved, next, n);
        free(next);
        if (!changed) break;
    }
    free(resolved);
    free(off); free(adj); free(aw);
}

/* Second SSSP of the reference (skeletonize.py:80-85) on the predecessor tree with recomputed
 * Euclidean edge lengths (shortest_path.py:46-55): d2[v] = fl32(d2[pred[v]] + |p_v - p_pred|). */
void so_tree_distance(i64 n, const float *pts, const i64 *pred, i64 root, const float *dist_hint, float *out) {
    /* process vertices in an order where parents come first: sort by depth via repeated passes */
    i64 *depth = (i64 *)malloc(n * sizeof(i64));
    for (i64 v = 0; v < n; v++) depth[v] = -1;
    depth[root] = 0;
    out[root] = 0.0f;
    i64 *stack = (i64 *)malloc(n * sizeof(i64));
    for (i64 v = 0; v < n; v++) {
        if (depth[v] >= 0) continue;
        if (pred[v] < 0) { out[v] = INFINITY; continue; } /* unreachable */
        i64 sp = 0, u = v;
        while (depth[u] < 0 && pred[u] >= 0) { stack[sp++] = u; u = pred[u]; }
        if (depth[u] < 0) { /* chain ends in an unreachable vertex */
            while (sp > 0) out[stack[--sp]] = INFINITY;
            continue;
        }
        while (sp > 0) {
            i64 x = stack[--sp];
            depth[x] = depth[pred[x]] + 1;
            out[x] = out[pred[x]] + sqrtf(dist2f(pts + 3 * x, pts + 3 * pred[x]));
        }
    }
    (void)dist_hint;
    free(depth);
    free(stack);
}

/* ------------------------------------------------------------------------- sample_tree --- */
/* Inputs are component-local: medial pts [n,3], raw radii [n], preds [n] (-1 root), distances [n].
 * Outputs: branch table (parent id, path offset, path length) and concatenated path vertex lists
 * (root-side first); branch_of_point [n] = final branch_ids array (path.py:75-80,135-136).
 * Returns the number of branches.  `iters_out` = number of loop iterations (incl. length-1 paths). */
i64 so_sample_tree(i64 n, const float *pts, const float *radii, const i64 *preds, const float *distances_in,
                   i64 *branch_parent, i64 *branch_off, i64 *branch_len, i64 *path_verts, i64 *branch_of_point,
                   i64 *iters_out) {
    float *dist = (float *)malloc((n > 0 ? n : 1) * sizeof(float));
    uint8_t *term = (uint8_t *)calloc(n > 0 ? n : 1, 1);
    float rmax = 0;
    for (i64 i = 0; i < n; i++) {
        dist[i] = preds[i] > 0 ? distances_in[i] : -1.0f; /* path.py:71-72 */
        branch_of_point[i] = -1;
        if (radii[i] > rmax) rmax = radii[i];
    }
    Grid g;
    grid_build(&g, pts, n, rmax > 0 ? rmax * 0.25f : 1.0f);
    i64 *path = (i64 *)malloc((n > 0 ? n : 1) * sizeof(i64));
    /* per-point best (d2, path position) for the current branch; touched list for reset */
    float *bd2 = (float *)malloc((n > 0 ? n : 1) * sizeof(float));
    i64 *bpos = (i64 *)malloc((n > 0 ? n : 1) * sizeof(i64));
    i64 *touched = (i64 *)malloc((n > 0 ? n : 1) * sizeof(i64));
    for (i64 i = 0; i < n; i++) bpos[i] = -1;
    i64 nb = 0, total = 0, iters = 0;
    for (;;) {
        i64 far = -1;
        for (i64 i = 0; i < n; i++)
            if (far < 0 || dist[i] > dist[far]) far = i; /* argmax, first maximum */
        if (far < 0 || dist[far] <= 0) break;
        iters++;
        /* trace_route (path.py:9-16) */
        i64 len = 0, idx = far;
        while (idx >= 0 && !term[idx]) { path[len++] = idx; idx = preds[idx]; }
        i64 termination = idx;
        for (i64 a = 0, b = len - 1; a < b; a++, b--) { i64 t = path[a]; path[a] = path[b]; path[b] = t; }
        /* select_path_points (path.py:19-46): nearest path vertex within r = max path radius */
        float rp = radii[path[0]];
        for (i64 q = 1; q < len; q++)
            if (radii[path[q]] > rp) rp = radii[path[q]];
        float rp2 = rp * rp;
        i64 nt = 0;
        int reach = (int)ceilf(rp / g.cell);
        if (reach < 1) reach = 1;
        for (i64 q = 0; q < len; q++) {
            const float *pv = pts + 3 * path[q];
            int c0[3];
            for (int a = 0; a < 3; a++) c0[a] = (int)floorf((pv[a] - g.lo[a]) / g.cell);
            for (int x = c0[0] - reach; x <= c0[0] + reach; x++) {
                if (x < 0 || x >= g.dim[0]) continue;
                for (int y = c0[1] - reach; y <= c0[1] + reach; y++) {
                    if (y < 0 || y >= g.dim[1]) continue;
                    for (int z = c0[2] - reach; z <= c0[2] + reach; z++) {
                        if (z < 0 || z >= g.dim[2]) continue;
                        i64 c = ((i64)x * g.dim[1] + y) * g.dim[2] + z;
                        for (i64 s = g.start[c]; s < g.start[c + 1]; s++) {
                            i64 p = g.items[s];
                            float d2 = dist2f(pts + 3 * p, pv);
                            if (!(d2 < rp2)) continue;
                            if (bpos[p] < 0) { touched[nt++] = p; bd2[p] = d2; bpos[p] = q; }
                            else if (d2 < bd2[p]) { bd2[p] = d2; bpos[p] = q; } /* q ascending: ties keep first */
                        }
                    }
                }
            }
        }
        /* on-path test (path.py:35-40) */
        i64 n_on = 0;
        for (i64 t = 0; t < nt; t++) {
            i64 p = touched[t];
            int on = sqrtf(bd2[p]) < radii[path[bpos[p]]];
            bpos[p] = -1;
            if (on) touched[n_on++] = p;
        }
        int keep_branch = len >= 2; /* path.py:125-126: short paths still consume their points */
        /* parent id is read BEFORE this branch stamps branch_ids (path.py:128-136);
         * termination -1 reads branch_ids[-1] = the LAST vertex (quirk kept) */
        i64 parent = branch_of_point[termination < 0 ? n - 1 : termination];
        for (i64 t = 0; t < n_on; t++) { /* path.py:112-122,136 */
            i64 p = touched[t];
            dist[p] = -1.0f;
            term[p] = 1;
            if (keep_branch) branch_of_point[p] = nb;
        }
        for (i64 q = 0; q < len; q++) {
            dist[path[q]] = -1.0f;
            term[path[q]] = 1;
            if (keep_branch) branch_of_point[path[q]] = nb;
        }
        if (!keep_branch) continue;
        branch_parent[nb] = parent;
        branch_off[nb] = total;
        branch_len[nb] = len;
        for (i64 q = 0; q < len; q++) path_verts[total + q] = path[q];
        total += len;
        nb++;
    }
    *iters_out = iters;
    free(dist); free(term); free(path); free(bd2); free(bpos); free(touched);
    grid_free(&g);
    return nb;
}
