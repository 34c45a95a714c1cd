/* smarttree_hip.h -- C ABI of libsmarttree_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of the smart-tree inference hot path.  The reference (uc-vision/smart-tree) has no
 * FFI layer of its own: its heavy arithmetic lives in three CUDA-only third-party packages (spconv, FRNN,
 * cugraph/cudf) that it calls from Python.  Each entry point below replaces one of those call sites; the
 * `replaces:` line cites it (paths relative to the reference's smart_tree/ package).  INTEGRATION.md shows
 * the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (torch tensors) unless its name ends in
 *     `_host`; sizes are explicit; the library never frees or keeps caller memory;
 *   - scratch comes from a caller-provided workspace (`ws`, `ws_bytes`) sized by the matching
 *     `*_workspace_bytes` function (or `st_query_workspace`); no hipMalloc inside the library;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).  Functions that
 *     return a count to the host (`*_host` out-parameters) synchronise that stream once before returning;
 *   - return value: 0 = ok, < 0 = error (-1 invalid argument, -2 workspace too small, -3 HIP error);
 *     `st_last_error()` gives the text (thread local).  Nothing throws.
 *   - neighbour tables are OUTPUT-stationary: nbr[k * n_out + o] = input row or -1, k = (kz*3+ky)*3+kx.
 */
#ifndef SMARTTREE_HIP_H
#define SMARTTREE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 101 (round 6): st_abi_entries added; stats_host of the skeleton calls is 16 x int64 (8 before 100) and their tuning array
 * 24 x int64 -- a caller built against an older header must check st_abi_entries before passing shorter arrays. */
int st_version(void);
/* Array lengths this build of the library reads / writes, so that a caller can check them at run time instead of trusting the
 * header it was compiled against: what = 0 -> int64 entries of `stats_host` (st_skeleton_components*, st_sssp, st_tree_distance,
 * st_sample_tree: all ZEROED and written), 1 -> int64 entries of `tuning` (st_skeleton_components_seg: all READ when non-NULL),
 * 2 -> ST_MAX_SEG (clouds per batched call); anything else -> -1 */
#define ST_SKELETON_STATS_ENTRIES 16
#define ST_SKELETON_TUNING_ENTRIES 24
int st_abi_entries(int what);
const char* st_last_error(void);
/* op: 0 scan(n) 1 sort(n) 2 voxelize(n,max_blocks,max_voxels) 3 strided(n) 4 knn(n_dst) 5 make_edges(n)
 *     6 component_layout(n) 7 component_csr(m) 8 skeleton(m,n_comp) */
int64_t st_query_workspace(int op, int64_t a, int64_t b, int64_t c);

/* ---- primitives (exported for tests) ------------------------------------------------------------ */
int64_t st_scan_workspace_bytes(int64_t n);
int st_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total, void* ws, int64_t ws_bytes, void* stream);
int64_t st_sort_workspace_bytes(int64_t n);
int st_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int key_bits, void* ws, int64_t ws_bytes, void* stream);

/* ---- blocks + voxelisation ------------------------------------------------------------------------
 * replaces: dataset/dataset.py:166-226 (SingleTreeInference.compute_blocks / __getitem__, i.e.
 *           spconv.pytorch.utils.PointToVoxel.generate_voxel_with_id on the CPU) and model/sparse.py:40-61
 *           (batch_collate).  feats [max_voxels,6], coords [max_voxels,4] (block,z,y,x), mask [max_voxels],
 *           point_index [max_voxels], block_centres [max_blocks,3]. */
int64_t st_voxelize_workspace_bytes(int64_t n_points, int max_blocks, int64_t max_voxels);
int st_voxelize_blocks(const float* xyz, const float* rgb, int64_t n, double voxel_size, double block_size,
                       double buffer_size, int min_points, int max_blocks, int64_t max_voxels, float* feats,
                       int32_t* coords, uint8_t* mask, int64_t* point_index, float* block_centres,
                       int64_t* n_voxels_host, int64_t* n_blocks_host, void* ws, int64_t ws_bytes, void* stream);

/* ---- rulebooks ------------------------------------------------------------------------------------
 * replaces: the indice-pair generation spconv runs inside every SubMConv3d / SparseConv3d /
 *           SparseInverseConv3d forward (model/model_blocks.py:23-35,57-70,90-101,134-143). */
int64_t st_hash_capacity(int64_t n);
int st_build_coord_hash(const int32_t* coords, int64_t n, unsigned long long* keys, unsigned* vals, int64_t cap, void* stream);
int st_build_subm_rulebook(const int32_t* coords, int64_t n, const unsigned long long* keys, const unsigned* vals,
                           int64_t cap, int32_t* nbr /*[27,n]*/, void* stream);
/* (no reference counterpart: spconv orders its voxels by hash) the permutation that sorts the voxels by (batch index,
 * Morton code); running the network on the permuted set and scattering the outputs back gives the same values faster. */
int64_t st_spatial_order_workspace_bytes(int64_t n);
int st_spatial_order(const int32_t* coords, int64_t n, int32_t* order /*[n]*/, void* ws, int64_t ws_bytes, void* stream);
/* rows of `row_words` 32-bit words moved by a permutation: dst[p] = src[order[p]] (scatter = 0) / dst[order[p]] = src[p] */
int st_move_rows(const void* src, int row_words, const int32_t* order, int64_t n, void* dst, int scatter, void* stream);
int64_t st_strided_workspace_bytes(int64_t n_fine);
int st_build_strided_outputs(const int32_t* coords, int64_t n, int64_t max_out, int32_t* out_coords,
                             unsigned long long* ckeys, unsigned* cvals, int64_t ccap, int64_t* n_out_host,
                             int32_t* extent_host /*[3]*/, void* ws, int64_t ws_bytes, void* stream);
int st_build_strided_rulebook(const int32_t* coords, int64_t n, const unsigned long long* fkeys, const unsigned* fvals,
                              int64_t fcap, const int32_t* out_coords, int64_t n_out, const unsigned long long* ckeys,
                              const unsigned* cvals, int64_t ccap, const int32_t* extent_host,
                              int32_t* nbr_down /*[27,n_out]*/, int32_t* nbr_up /*[27,n]*/,
                              int32_t* up_order /*[n+16] or NULL: fine rows grouped by coordinate parity, entry = row | (8 + class) << 28 (tail = scratch)*/,
                              void* stream);

/* ---- network --------------------------------------------------------------------------------------
 * replaces: spconv's gather-GEMM-scatter conv kernels + torch BatchNorm1d/ReLU/add/cat around them
 *           (model/model_blocks.py:8-243) and, for the heads, SparseFC + F.normalize + exp/argmax
 *           (model_blocks.py:246-285, model/model.py:83-85, model/model_inference.py:87-88).
 * st_sparse_conv_fwd: y = act(bn(sum_k W_k . cat(x0,x1)[nbr[k]]) + residual); w is [K][cin][cout]. */
int st_sparse_conv_fwd(const float* x0, int c0, const float* x1, int cin, const int32_t* nbr, int K, int64_t n_out,
                       const float* w, int cout, const float* scale, const float* shift, const float* residual,
                       int relu, float* y, const int32_t* row_order /*[n_out] or NULL: launch order of the output rows (bits 0-27 row, top 4 bits 0 or the parity tag of st_build_strided_rulebook)*/,
                       void* stream, int64_t nbr_stride /*elements between the rows of nbr; 0 = n_out*/);
/* same contract; weights pre-permuted to wp[K][cin/16][4][cout][4] = W[k][16c+4kg+s][co]; cin, cout, c0 % 16 == 0.
 * The per-offset [16 x cin].[cin x cout] contraction runs on v_mfma_f32_16x16x4_f32 (exact fp32). */
int st_sparse_conv_mfma_fwd(const float* x0, int c0, const float* x1, int cin, const int32_t* nbr, int K, int64_t n_out,
                            const float* wp, int cout, const float* scale, const float* shift, const float* residual,
                            int relu, float* y, const int32_t* row_order /*[n_out] or NULL: launch order of the output rows (bits 0-27 row, top 4 bits 0 or the parity tag of st_build_strided_rulebook)*/,
                            void* stream, int64_t nbr_stride /*0 = n_out*/,
                            int variant /*0 = the library picks the tile shape by size;
 else row tiles per wave | (weights through LDS) << 4 (bench aid)*/);
/* The same contraction on the bf16 matrix pipe at float32 accuracy: every float32 operand is cut into three bf16 pieces that sum
 * to it exactly and six of the nine piece products are issued (csrc/sparse_conv.hip "split-bf16 rule-GEMM").  Features stay float32;
 * wq = the weights as three bf16 planes in operand order [K][Cin/32][3][4][Cout][8] (smart_tree_amd/model/sparse_ops.py b3_weight).
 * Cin % 32 == 0, Cout % 16 == 0, c0 % 8 == 0.  variant: 0 = row tiles per wavefront by size, 1 / 2 = forced.
 * replaces: the same spconv calls as st_sparse_conv_fwd (model_blocks.py:57-70,90-101,134-143) on the 32- and 64-channel levels */
int st_sparse_conv_b3_fwd(const float* x0, int c0, const float* x1, int cin, const int32_t* nbr, int K, int64_t n_out, const void* wq,
                          int cout, const float* scale, const float* shift, const float* residual, int relu, float* y,
                          const int32_t* row_order, void* stream, int64_t nbr_stride, int variant);
/* Half-precision storage (BASELINE.json configs[4]; an extension -- the reference's inference, model/model_inference.py:49-100,
 * is float32): in_half && out_half -> x0 / x1 / residual / y are IEEE half, Cin, Cout and the concat split multiples of 16,
 * matrix cores with float32 accumulation; exactly one of them -> the float32 kernel with a converting load or store
 * (w [K][cin][cout] float32, no residual, no concat).
 * WEIGHT LAYOUT of the in_half && out_half form (half elements; smart_tree_amd/model/sparse_ops.py mfma_weight16_half builds it)
 * depends on the channel counts -- a buffer in another order gives wrong results, the call cannot tell:
 *   Cin % 32 == 0                      wp[K][Cin/32][4][Cout][8]      = W[k][32c + 8g + e][co]   (v_mfma_f32_16x16x32_f16)
 *   Cin == 16 and Cout in {16, 32}     pairs of kernel offsets stacked into 32-channel chunks: W' = W padded with a zero offset to
 *                                      an even K, viewed as [K/2][32][Cout], then the order above (K/2 chunks of 32 channels)
 *   otherwise (Cin % 16 == 0)          wp[K][Cin/16][4][Cout][4]      = W[k][16c + 4g + s][co]   (v_mfma_f32_16x16x16_f16)
 * st_sparse_conv_b3_fwd's wq follows the first two rules with three bf16 planes per chunk ([..][3][4][Cout][8]): Cin == 16 uses
 * the stacked-pair chunks as well (sparse_ops.py b3_weight). */
int st_sparse_conv_f16_fwd(const void* x0, int c0, const void* x1, int cin, const int32_t* nbr, int K, int64_t n_out,
                           const void* w, int cout, const float* scale, const float* shift, const void* residual, int relu,
                           void* y, int in_half, int out_half, const int32_t* row_order, void* stream, int64_t nbr_stride /*0 = n_out*/);

/* ---- rulebooks without hash probes (csrc/brick.hip) ------------------------------------------------
 * replaces, for the network's own forward pass, the per-conv hash build of spconv (model/model_blocks.py:57-70,90-101,134-143)
 * AND the builders above: every level's active set is kept as 8 x 8 x 8 occupancy bricks with popcount ranks, the voxels of a
 * level are ordered (block, brick Morton code, z, y, x), a neighbour look-up is a table entry + one mask word.  One call builds
 * every level (sets, submanifold tables, strided / inverse tables, parity order) with the counts on the device and ONE
 * read-back.  Tables are laid out with row stride caps[level] (pass it as nbr_stride to the convolutions).
 * st_brick_pyramid_workspace_bytes returns 0 when the structure cannot be sized for the arguments (use the builders above). */
int64_t st_brick_pyramid_workspace_bytes(int64_t n0, int n_blocks, int coord_bound, int depth, const int64_t* caps);
int st_brick_pyramid(const int32_t* coords0 /*[n0,4] (batch index, z, y, x), any order*/, int64_t n0, int n_blocks /*batch indices < this*/,
                     int coord_bound /*z, y, x < this*/, int depth, const int32_t* blk_seg /*[n_blocks] cloud of a batch index or NULL*/,
                     int nseg, const int64_t* caps /*[depth+1] row capacity per level, caps[0] >= n0*/,
                     int32_t* order0 /*[n0]: row p of level 0 = input voxel order0[p]*/, int32_t* const* coords_out /*[depth+1] x [caps[l],4]*/,
                     int32_t* const* subm /*[depth+1] x [27][caps[l]]*/, int32_t* const* down /*[depth] x [27][caps[l+1]]*/,
                     int32_t* const* up /*[depth] x [27][caps[l]]*/, int32_t* const* up_order /*[depth] x [caps[l] + 16]*/,
                     int64_t* counts_host /*[depth+1] rows per level*/, void* ws, int64_t ws_bytes, void* stream);
int st_head_param_floats(void);
int st_pointwise_mlp_heads(const float* x, int64_t n, const float* params, float* radius, float* direction,
                           float* class_l, float* medial_vector /*nullable*/, int64_t* class_idx /*nullable*/, void* stream);

/* ---- skeleton stage -------------------------------------------------------------------------------
 * st_medial_points  replaces: Cloud.medial_pts / Cloud.radius, data_types/cloud.py:229-231,254-256
 * st_knn_radius     replaces: frnn.frnn_grid_points via skeleton/graph.py:12-33 (filter.py:7, graph.py:37, path.py:30)
 * st_make_edges     replaces: skeleton/graph.py:52-60
 * st_connected_components / st_component_layout / st_component_csr
 *                   replace: cugraph.connected_components + subgraph + the cudf->pandas->torch renumbering,
 *                            data_types/graph.py:32-66, skeleton/skeletonize.py:60-71, skeleton/graph.py:80-104
 * st_skeleton_components (stages 1|2|4) and its single-stage forms st_sssp / st_tree_distance / st_sample_tree
 *                   replace: cugraph.sssp (skeleton/shortest_path.py:12-21), pred_graph + second sssp
 *                            (shortest_path.py:46-55, skeletonize.py:80-85), sample_tree (skeleton/path.py:9-140)
 * st_assemble_branches replaces: the BranchSkeleton / TreeSkeleton construction of skeleton/path.py:128-140 and
 *                   skeletonize.py:86-95 (flat layout consumed by st_post_process)
 * st_post_process   replaces: pipeline.py:95-106 over data_types/tree.py:73-134,164-176 (+ util/queries.py:89-133).
 *                   Branch tables as st_assemble_branches lays them out: a parent has a smaller id than its children (the
 *                   reference's dict walk relies on the same order), every branch owns at least two rad_out slots (they
 *                   carry a key between the call's launches before the radii are written). */
/* st_centre_cloud replaces: dataset/augmentations.py:38-41 (CentreCloud over Cloud.bbox, data_types/cloud.py:222-227) */
int st_centre_cloud(const float* xyz, int64_t n, float* out, void* ws, int64_t ws_bytes, void* stream);
int st_medial_points(const float* xyz, const float* mv, int64_t n, float* medial, float* radius, void* stream);
int64_t st_knn_workspace_bytes(int64_t n_dst);
/* r < 0 (with a bound array): the search radius is max(bound[0..n1)), reduced on the device -- the callers
 * (skeleton/filter.py, skeleton/graph.py) no longer read it back just to pass it in; cell_hint < 0: cell = max(r / -cell_hint, 1e-4).
 * K = 1, 8, 16 or 32 (the rows come out sorted by (distance, index): any other K <= 32 is the first K columns of the next width).
 * NaN / infinite points are nobody's neighbour and find none; a NaN bound admits nobody, an infinite one every point within r. */
int st_knn_radius(const float* src, int64_t n1, const float* dst, int64_t n2, int K, float r, const float* bound,
                  int bound_mode, float cell_hint, int64_t* idx, float* dist, void* ws, int64_t ws_bytes, void* stream);
int64_t st_make_edges_workspace_bytes(int64_t n);
/* n_edges_host may be NULL: the edge count stays on the device and the tail of edges / w [n*K] is padded with (0, 0)
 * self loops of weight 0, which the component entry points below ignore (real edges have dst > 0) */
int st_make_edges(const int64_t* idx, const float* dist, int64_t n, int K, int64_t* edges, float* w,
                  int64_t* n_edges_host, void* ws, int64_t ws_bytes, void* stream);
int64_t st_connected_components_workspace_bytes(int64_t n);
int st_connected_components(const int64_t* edges, int64_t E, int64_t n, int32_t* labels, void* ws, int64_t ws_bytes, void* stream);
int64_t st_component_layout_workspace_bytes(int64_t n);
int st_component_layout(const int32_t* labels, int64_t n, int min_vertices, int32_t* comp_size, int32_t* comp_off,
                        int32_t* vert_order, int32_t* new_id, int64_t* n_comp_host, int64_t* n_kept_host, void* ws,
                        int64_t ws_bytes, void* stream);
int64_t st_component_csr_workspace_bytes(int64_t m);
int st_component_csr(const int64_t* edges, const float* w, int64_t E, const int32_t* new_id, int64_t m, uint32_t* row_off,
                     uint32_t* col, float* wgt, void* ws, int64_t ws_bytes, void* stream);
int64_t st_skeleton_workspace_bytes(int64_t m, int64_t n_comp);
/* stats_host: optional, 16 x int64 (see csrc/skeleton.hip): [0] SSSP rounds .. [6] = branches of the cloud | path vertices << 32
 * (sample_tree stage), [7] in: time the select launches, [8] helper workgroups were lost (fall-back ran), [9] helper workgroups launched.
 * comp_size_host is unused and may be NULL (the claim grid is laid out on the device from comp_off); grid_cell < 0:
 * cell = max(max(rad) / -grid_cell, 1e-4) with the maximum reduced on the device -- neither costs the caller a read-back */
int st_skeleton_components(int n_comp, const int32_t* comp_off, const int32_t* comp_size_host, int64_t m, const float* pts,
                           const float* rad, const float* ysurf, const uint32_t* row_off, const uint32_t* col,
                           const float* wgt, float grid_cell, int stages, int block_threads, float* dist, int32_t* pred,
                           int32_t* root_local, float* tree_dist, int32_t* branch_parent, int32_t* branch_off,
                           int32_t* branch_len, int32_t* n_branches, int32_t* path_verts, int32_t* branch_of,
                           int64_t* stats_host, void* ws, int64_t ws_bytes, void* stream);
#define ST_SKELETON_STAGE_ARGS                                                                                          \
    int n_comp, const int32_t *comp_off, const int32_t *comp_size_host, int64_t m, const float *pts, const float *rad, \
        const float *ysurf, const uint32_t *row_off, const uint32_t *col, const float *wgt, float grid_cell,           \
        int block_threads, float *dist, int32_t *pred, int32_t *root_local, float *tree_dist, int32_t *branch_parent,  \
        int32_t *branch_off, int32_t *branch_len, int32_t *n_branches, int32_t *path_verts, int32_t *branch_of,        \
        int64_t *stats_host, void *ws, int64_t ws_bytes, void *stream
int st_sssp(ST_SKELETON_STAGE_ARGS);
int st_tree_distance(ST_SKELETON_STAGE_ARGS);
int st_sample_tree(ST_SKELETON_STAGE_ARGS);
int64_t st_assemble_workspace_bytes(int64_t cap_b);
/* counts_host may be NULL: no read-back; the caller takes branches B = stats_host[6] & 0xffffffff and geometry slots
 * P = (stats_host[6] >> 32) + B from st_skeleton_components (totals carried by its last progress read-back) */
int st_assemble_branches(int n_comp, const int32_t* comp_off, const int32_t* n_branches, const int32_t* branch_parent,
                         const int32_t* branch_off, const int32_t* branch_len, const int32_t* path_verts,
                         const int32_t* vert_order, const float* medial, const float* radius, int32_t* tree_off,
                         int32_t* parent, int32_t* start, int32_t* length, float* xyz, float* rad, int64_t cap_b,
                         int64_t cap_p, int64_t* counts_host, void* ws, int64_t ws_bytes, void* stream);
int st_post_process(int n_trees, const int32_t* tree_off, const int32_t* parent, const int32_t* start, const int32_t* len,
                    float* xyz, const float* rad_in, float* rad_out, uint8_t* keep, uint8_t* repaired, uint8_t* smoothed,
                    int32_t* depth_scratch, int do_prune, float min_radius, float min_length, int do_repair, int do_smooth, int kernel_size,
                    void* stream);

/* st_points_to_nearest_tube replaces: pts_to_nearest_tube_gpu (util/queries.py:107-133) and the chunk loop of
 * skeleton_to_points (util/queries.py:139-166): per point the tube minimising |distance - interpolated radius|. */
int st_points_to_nearest_tube(const float* pts, int64_t n, const float* a, const float* b, const float* r1, const float* r2,
                              int64_t m, float* vec, int64_t* idx, float* rad, void* stream);

/* ---- batched forms: B independent clouds in ONE launch set -------------------------------------------
 * replaces: the batch dimension of the reference's data path -- model/sparse.py:40-61 (batch_collate writes the
 *           sample index into coords[:,0]) and model/model_inference.py:62-78 (one forward per collated batch) --
 *           extended over the whole of Pipeline.process_cloud (pipeline.py:55-93), which the reference runs one
 *           cloud at a time.  Cloud s of a batch owns the index range [seg_off[s], seg_off[s+1]) of the batched
 *           arrays (seg_off: device int32 [nseg+1], nseg <= 64).  Every function returns, for every cloud, exactly
 *           what its one-cloud form returns for that cloud alone (indices are positions in the batched arrays):
 *           clouds never share a neighbourhood, a grid slab, a spatial extent or a component. */
int st_centre_cloud_seg(const float* xyz, int64_t n, const int32_t* seg_off, int nseg, float* out, void* ws,
                        int64_t ws_bytes /* >= 24 * nseg + 256 */, void* stream);
int64_t st_voxelize_workspace_bytes_seg(int64_t n_points, int max_blocks, int64_t max_voxels, int nseg);
int st_voxelize_blocks_seg(const float* xyz, const float* rgb, int64_t n, const int32_t* seg_off, int nseg,
                           double voxel_size, double block_size, double buffer_size, int min_points, int max_blocks,
                           int64_t max_voxels, float* feats, int32_t* coords, uint8_t* mask, int64_t* point_index,
                           float* block_centres, int32_t* blk_seg /*[max_blocks] cloud of every block*/,
                           int32_t* seg_vox_off /*[nseg+1]*/, int32_t* seg_blk_off /*[nseg+1]*/, int64_t* n_voxels_host,
                           int64_t* n_blocks_host, void* ws, int64_t ws_bytes, void* stream);
int st_build_strided_outputs_seg(const int32_t* coords, int64_t n, int64_t max_out, int32_t* out_coords,
                                 unsigned long long* ckeys, unsigned* cvals, int64_t ccap, int64_t* n_out_host,
                                 int32_t* extent_host, const int32_t* blk_seg, int nseg, int32_t* ext_dev /*[nseg*3] out*/,
                                 void* ws, int64_t ws_bytes, void* stream);
int st_build_strided_rulebook_seg(const int32_t* coords, int64_t n, const unsigned long long* fkeys, const unsigned* fvals,
                                  int64_t fcap, const int32_t* out_coords, int64_t n_out, const unsigned long long* ckeys,
                                  const unsigned* cvals, int64_t ccap, const int32_t* extent_host, int32_t* nbr_down,
                                  int32_t* nbr_up, int32_t* up_order, const int32_t* blk_seg, const int32_t* ext_dev,
                                  void* stream);
/* Whole-cloud voxelisation of `nseg` clouds (the training / evaluation data path).
 * replaces: TreeDataset.process_cloud's PointToVoxel call (smart_tree/dataset/dataset.py:103-131: range = the cloud's own
 *           bounding box, one point per voxel) + batch_collate's batch column (model/sparse.py:40-61).
 * coords [M,4] = (cloud, z, y, x); feats [M,6] = xyz, rgb of the representative point; point_index [M] = its position in the
 * input (gather any other per-point feature with it); mask [M] = 1 (loss_mask); seg_vox_off [nseg+1] device, may be null for
 * one cloud.  Voxels come out cloud by cloud in order of first appearance. */
int64_t st_voxelize_cloud_workspace_bytes(int64_t n_points, int64_t max_voxels, int nseg);
int st_voxelize_cloud_seg(const float* xyz, const float* rgb, int64_t n, const int32_t* seg_off, int nseg, double voxel_size,
                          int64_t max_voxels, float* feats, int32_t* coords, uint8_t* mask, int64_t* point_index,
                          int32_t* seg_vox_off, int64_t* n_voxels_host, void* ws, int64_t ws_bytes, void* stream);
int64_t st_knn_workspace_bytes_seg(int64_t n_dst, int nseg);
/* cell_mean_mult (with cell_hint < 0): the grid cell is at most this multiple of the MEAN per-query bound (< 0: library
 * default, 0: no cap); it changes the speed of a search, never its result (test hook, tests/test_skeleton.py) */
int st_knn_radius_seg(const float* src, int64_t n1, const float* dst, int64_t n2, int K, float r, const float* bound,
                      int bound_mode, float cell_hint, int64_t* idx, float* dist, const int32_t* src_seg_off,
                      const int32_t* dst_seg_off, int nseg, void* ws, int64_t ws_bytes, void* stream, float cell_mean_mult);
/* replaces: skeleton/filter.py:6-11 (outlier_removal): mask[i] = the query has >= K neighbours inside its bound, i.e.
 *           st_knn_radius_seg(...).idx[:, K-1] != -1 without the neighbour lists (K = 8; workspace as st_knn_radius_seg). */
int st_radius_count_seg(const float* src, int64_t n1, const float* dst, int64_t n2, int K, float r, const float* bound,
                        int bound_mode, float cell_hint, uint8_t* mask, const int32_t* src_seg_off,
                        const int32_t* dst_seg_off, int nseg, void* ws, int64_t ws_bytes, void* stream, float cell_mean_mult,
                        const uint8_t* valid /* optional [n1], needs src == dst: the search runs over the points with valid[i] != 0 only
                        (neighbours and queries alike; mask = 0 for the others) -- the class filter folded into outlier_removal
                        without compacting the cloud in between */);
/* components / adjacency straight from the neighbour search (idx / dist [n,K] of st_knn_radius_seg after the caller's radius
 * filter): the edge set is make_edges' (graph.py:52-60: (i, idx) for idx > vertex 0 of i's cloud) without materialising
 * the int64 edge list.  Same labels as st_connected_components on st_make_edges_seg's output; the CSR holds the same
 * (neighbour, weight) pairs per row as st_component_csr's, but every pair ONCE: a mutual pair (v in kNN(u), u in kNN(v)) is two
 * edges of the list and two entries per row there, one here (the reference's cugraph graph is undirected, one edge per pair).
 * Row layout: valid forward neighbours in table order, then the reverse-only ones.  With a workspace smaller than
 * st_component_csr_knn_workspace_bytes (st_component_csr_workspace_bytes is the minimum) or a K that is not a power of two
 * <= 64 the call builds st_component_csr's rows. */
int64_t st_component_csr_knn_workspace_bytes(int64_t m, int64_t n, int K);
int st_connected_components_knn(const int64_t* idx, int64_t n, int K, const int32_t* first_of /*[n] first vertex of the
                                vertex's cloud; NULL = one cloud*/, int32_t* labels, void* ws, int64_t ws_bytes, void* stream);
int st_component_csr_knn(const int64_t* idx, const float* dist, int64_t n, int K, const int32_t* first_of,
                         const int32_t* new_id, int64_t m, uint32_t* row_off, uint32_t* col, float* wgt, void* ws,
                         int64_t ws_bytes, void* stream);
int st_make_edges_seg(const int64_t* idx, const float* dist, int64_t n, int K, int64_t* edges, float* w,
                      int64_t* n_edges_host, const int32_t* seg_off, int nseg, void* ws, int64_t ws_bytes, void* stream);
int st_component_layout_seg(const int32_t* labels, int64_t n, int min_vertices, const int32_t* seg_off, int nseg,
                            int32_t* comp_size, int32_t* comp_off, int32_t* vert_order, int32_t* new_id,
                            int32_t* comp_seg /*[C]*/, int32_t* comp_seg_off /*[nseg+1]*/, int32_t* vert_seg_off /*[nseg+1]*/,
                            int64_t* n_comp_host, int64_t* n_kept_host, void* ws, int64_t ws_bytes, void* stream,
                            int64_t* max_comp_host /*optional: vertices of the largest kept component (same read-back)*/);
int64_t st_skeleton_workspace_bytes_seg(int64_t m, int64_t n_comp, int nseg);
int st_skeleton_components_seg(int n_comp, const int32_t* comp_off, const int32_t* comp_seg, const int32_t* vert_seg_off,
                               int nseg, int64_t m, const float* pts, const float* rad, const float* ysurf,
                               const uint32_t* row_off, const uint32_t* col, const float* wgt, float grid_cell, int stages,
                               int block_threads, float* dist, int32_t* pred, int32_t* root_local, float* tree_dist,
                               int32_t* branch_parent, int32_t* branch_off, int32_t* branch_len, int32_t* n_branches,
                               int32_t* path_verts, int32_t* branch_of, int64_t* stats_host, void* ws, int64_t ws_bytes,
                               void* stream, const int64_t* tuning /*NULL = defaults; 24 entries, see csrc/skeleton.hip "Tuning of
                               one call": per-call strategy / sweep knobs (no process-global state)*/);
int st_post_process_seg(int n_trees, const int32_t* tree_off, const int32_t* parent, const int32_t* start, const int32_t* len,
                        float* xyz, const float* rad_in, float* rad_out, uint8_t* keep, uint8_t* repaired, uint8_t* smoothed,
                        int32_t* depth_scratch, int do_prune, float min_radius, float min_length, int do_repair,
                        int do_smooth, int kernel_size, const int32_t* first_tree /*[n_first] first tree of every cloud*/,
                        int n_first, void* stream);

/* ---- evaluation-side losses, forward only ----------------------------------
 * replaces: compute_loss + L1Loss + cosine_similarity_loss + focal_loss + dice_loss (smart_tree/model/loss.py:7-97) as ONE
 *           pass over the voxels.  radius [n], direction [n,3], class_l [n,C] are the network's outputs; targets [n,5] =
 *           (radius, direction xyz, class id); mask [n] uint8 or NULL; vector_class < 0 = None.
 * out_host[8] (the call waits for the stream): radius, direction, focal, dice loss; vector rows; class rows; 0; 0.
 * An empty selection gives NaN (mean of an empty tensor).  A target class id outside [0, C) is an error (torch raises). */
int64_t st_loss_workspace_bytes(void);
int st_loss_forward(const float* radius, const float* direction, const float* class_l, int n_classes, const float* targets,
                    int target_cols, const uint8_t* mask, int64_t n, int vector_class, int target_radius_log,
                    double* out_host, void* ws, int64_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMARTTREE_HIP_H */
