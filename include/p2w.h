/*
 * p2w.h - C ABI of libp2w_gfx950.so: the MI355X (gfx950) kernels behind the
 * PointsToWood inference forward.
 *
 * Boundary rules:
 *   - extern "C", plain pointers + sizes, no C++/torch types, no exceptions;
 *   - every entry point is re-entrant, keeps no global mutable state, reads no
 *     environment variables, never allocates and never synchronises: the caller
 *     owns all device buffers (worst-case sized) and passes the HIP stream to
 *     launch on; behaviour switches are explicit `flags` / `prec` arguments;
 *   - every entry point returns an int32 status: 0 = OK, > 0 = hipError_t from
 *     the launch, < 0 = a P2W_E* argument error; p2w_strerror() names it;
 *   - all index arrays are int32, all features fp32 row-major, positions are
 *     float4 records "xyzr" = (x, y, z, reflectance), 16-byte aligned;
 *   - ragged batches are CSR: ptr[B+1] (int32, DEVICE memory) over the
 *     concatenated voxels; element counts that are only known on the device
 *     are read from ptr[B] by the kernels, the host passes an upper bound.
 *
 * Each function names the reference call site it replaces (paths relative to
 * the reference repo, harryjfowen/PointsToWood @ 2025-09-12).  The third-party
 * operators themselves (torch-cluster / torch-scatter / torch-geometric) are not
 * vendored by the reference; their semantics are fixed by oracle/ops.py.
 */
#ifndef P2W_H
#define P2W_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* p2w_stream_t; /* hipStream_t */

#define P2W_OK 0
#define P2W_EINVAL (-1)    /* bad scalar argument (k, sizes, strides)        */
#define P2W_ENULL (-2)     /* required pointer is NULL                        */
#define P2W_EALIGN (-3)    /* pointer / stride not 16-byte aligned            */
#define P2W_EWORKSPACE (-4)/* workspace too small                             */
#define P2W_EUNSUPPORTED (-5)

#define P2W_MAX_K 64       /* neighbours per query of the wave-wide searches (wave64: one lane per slot): the grid searches' limit */
#define P2W_MAX_K_WIDE 100 /* p2w_knn / p2w_ball_query also take 65 .. 100 (torch-cluster's limit) on a one-thread-per-query path */
#define P2W_MAX_K_CONV 32  /* neighbour slots per target of p2w_sa_conv*: one 32-row MFMA tile (Net(k=...) <= 32) */

int32_t p2w_version(void);
const char* p2w_strerror(int32_t code);

/* ---- geometry ---------------------------------------------------------- */

/* xyzr[i] = (pos[i*pos_stride + 0..2], refl[i]); batch[i] = voxel of point i (from ptr).
 * Feeds SAModule.forward's cat(pos[:, :3], reflectance) - pointstowood/src/model.py:109. */
int32_t p2w_pack_xyzr(const float* pos, int32_t pos_stride, const float* refl, const int32_t* ptr, int32_t B,
                      int32_t n, float* xyzr, int32_t* batch, p2w_stream_t stream);

/* Grid sub-sampling of one level: SAModule.voxelsample (model.py:103-106) =
 * PyG voxel_grid -> torch-cluster grid_cluster + PyG consecutive_cluster.
 * In : xyzr[n_bound] (records >= ptr[B] ignored), ptr[B+1], cell size `res`.
 * Out: idx_out[<= n_bound] (index into xyzr of the representative = LARGEST point index of each
 *      occupied cell, ascending cell id i.e. voxel-major), ptr_out[B+1] (CSR of the sampled level;
 *      ptr_out[B] = M), batch_out[M]; order_out[n] (optional, may be NULL) = the input point indices in
 *      ascending (voxel, cell id) order - a spatially coherent visiting order for the searches below.
 * ws : p2w_voxel_sample_ws_bytes(n_bound) bytes of scratch. */
/* Geometry of the cell grid a sampling call used (device-resident, written by p2w_voxel_sample): cell (cx, cy, cz)
 * of voxel b has key ((( (b - b_lo) * dims[2] + cz) * dims[1] + cy) * dims[0] + cx. */
typedef struct p2w_grid {
    float lo[3];      /* batch-global minimum of the sampled coordinates = grid origin */
    float res;        /* cell size */
    float hi[3];      /* batch-global maximum */
    int32_t b_lo;     /* first non-empty voxel */
    int64_t dims[3];  /* cells per axis */
} p2w_grid;

size_t p2w_voxel_sample_ws_bytes(int32_t n_bound);
/* Optional outputs (NULL to skip) that make the level searchable through p2w_knn_grid / p2w_ball_query_grid:
 * sorted_keys_out[n] = the cell keys of all input points in ascending order (the order of order_out),
 * cell_keys_out[M] = the key of every representative (ascending = the output order), grid_out = the grid,
 * inv_out[n] = for every input point the index (in the output level) of its cell's representative; rank_sorted_out[n]
 * = the same for the i-th point of the sorted order (order_out[i]). */
int32_t p2w_voxel_sample(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n_bound, float res,
                         int32_t* idx_out, int32_t* ptr_out, int32_t* batch_out, int32_t* order_out,
                         uint64_t* sorted_keys_out, uint64_t* cell_keys_out, p2w_grid* grid_out, int32_t* inv_out,
                         int32_t* rank_sorted_out, void* ws, size_t ws_bytes, p2w_stream_t stream);

/* The same sub-sampling WITHOUT a sort, for batches whose cell grid fits a direct table of `table_cells` entries (the
 * per-voxel forward: a few voxels of a few metres): a cell's key is its table index; representatives by atomic max, ranks by
 * a scan of the occupancy flags, the optional sorted order by a counting-sort scatter (the order of points INSIDE a cell is
 * unspecified - every consumer orders candidates by the index they carry).  Outputs as p2w_voxel_sample.  Whether the grid
 * fits is only known on the device: if it does not, *status_out (device int32) is set to 1 and an EMPTY level is returned
 * (ptr_out = 0, so work already queued behind the call finds nothing to do) - the caller checks it (with the level sizes it reads back anyway) and repeats the level with p2w_voxel_sample.
 * ws: p2w_voxel_sample_table_ws_bytes(n_bound, table_cells) bytes (20 B per table entry). */
size_t p2w_voxel_sample_table_ws_bytes(int32_t n_bound, int64_t table_cells);
/* cell_start_out / cell_start_sorted_out (optional, table_cells + 1 int32 each; the second needs order_out): for every cell
 * key t <= the grid's cell count the position of the first element with key >= t in the sampled level / in the cell-sorted
 * order of the input points (the last used entry = the totals): p2w_knn_grid_indexed / p2w_ball_query_grid_indexed find
 * their candidate runs with one load from such a table instead of bisecting the keys. */
int32_t p2w_voxel_sample_table(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n_bound, float res,
                               int32_t* idx_out, int32_t* ptr_out, int32_t* batch_out, int32_t* order_out,
                               uint64_t* sorted_keys_out, uint64_t* cell_keys_out, p2w_grid* grid_out, int32_t* inv_out,
                               int32_t* rank_sorted_out, int32_t* cell_start_out, int32_t* cell_start_sorted_out,
                               int32_t* status_out, int64_t table_cells, void* ws, size_t ws_bytes, p2w_stream_t stream);

/* out[i] = (x, y, z, bit pattern of order[i]) of xyzr[order[i]] for i < ptr[B]: the records of a level in another
 * (e.g. cell-sorted) order, each carrying its own index - input for the P2W_SEARCH_*_IN_W modes below. */
int32_t p2w_index_records(const float* xyzr, const int32_t* order, const int32_t* ptr, int32_t B, int32_t n_bound,
                          float* out, p2w_stream_t stream);

/* The two halves of p2w_voxel_sample as separate operators (the reference calls them separately,
 * model.py:104-105): cell ids exactly as PyG voxel_grid(pos, size, batch) returns them (int64), and
 * consecutive_cluster(cell) -> (inv[n] = rank of each point's cell, perm[count] = largest point index
 * per cell, ascending cell id).  inv_out may be NULL.  ws: p2w_voxel_sample_ws_bytes(n). */
int32_t p2w_voxel_grid(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n, float res, int64_t* cell_out,
                       void* ws, size_t ws_bytes, p2w_stream_t stream);
int32_t p2w_consecutive_cluster(const int64_t* cell, int32_t n, int32_t* inv_out, int32_t* perm_out,
                                int32_t* count_out, void* ws, size_t ws_bytes, p2w_stream_t stream);

/* Next level's records: out[i] = ((p / sf_b) * sf_b, refl) of src[idx[i]] - the in-place scale /
 * un-scale round trip of model.py:122,124 followed by the slices of :126. */
int32_t p2w_level_gather(const float* xyzr_src, const int32_t* idx, const int32_t* batch_dst, const int32_t* ptr_dst,
                         int32_t B, int32_t m_bound, const float* sf, float* xyzr_dst, p2w_stream_t stream);

/* Search modes (flags of p2w_ball_query / p2w_knn).  Both exist so that a level can be searched in a spatially
 * coherent storage order (p2w_index_records) while results keep referring to the level's own numbering:
 *   X_INDEX_IN_W: candidate c is reported as (and ties / "first cap" are ordered by) the int32 in xyzr_x[c].w, not c.
 *   Q_ROW_IN_W  : the results of query q are written to row (int32 in its record's .w) of nbr / deg, not row q.
 *   BOX (grid searches only): bound the gathered region in x as well - one run per grid row instead of one per z
 *   layer; for grids whose rows are much longer than a workgroup's reach (plot-scale searches). */
#define P2W_SEARCH_X_INDEX_IN_W 1
#define P2W_SEARCH_Q_ROW_IN_W 2
#define P2W_SEARCH_BOX 4

/* Ball query: torch-cluster radius(x, y, r, batch_x, batch_y, max_num_neighbors) - model.py:118.
 * Queries are x[qidx[q]] (qidx NULL = identity).  For query q of voxel b the `cap` candidates of lowest index among
 * those of ptr_x[b]..ptr_x[b+1] with d2 < (float)(r*r) (r is a double, as torch-cluster's host code takes it) are
 * written in ascending index order to nbr[q*cap + 0..deg[q]-1]; remaining slots are -1.  tile_bbox: optional, see
 * p2w_tile_bbox. */
int32_t p2w_ball_query(const float* xyzr_x, const int32_t* ptr_x, const float* xyzr_q, const int32_t* qidx,
                       const int32_t* ptr_q, int32_t B, int32_t m_bound, double r, int32_t cap, int32_t* nbr,
                       int32_t* deg, const float* tile_bbox, int32_t flags, p2w_stream_t stream);

/* Exact brute-force kNN: torch-cluster knn(x, y, k, batch_x, batch_y) - model.py:120 and inside
 * PyG knn_interpolate (model.py:149).  Ascending (d2, index); deg[q] = min(k, #candidates). */
int32_t p2w_knn(const float* xyzr_x, const int32_t* ptr_x, const float* xyzr_q, const int32_t* qidx,
                const int32_t* ptr_q, int32_t B, int32_t m_bound, int32_t k, int32_t* nbr, int32_t* deg,
                const float* tile_bbox, int32_t flags, p2w_stream_t stream);

/* The same two searches over candidates stored in ascending cell-key order of `grid` (keys_x[c] = key of candidate c;
 * both come from the p2w_voxel_sample call that produced or ordered the level).  Only the grid rows within reach of a
 * workgroup's queries are gathered (two binary searches on the keys per z layer) instead of the whole voxel; the
 * results are identical to p2w_knn / p2w_ball_query.  Queries should be spatially coherent (any level in its own
 * storage order is). */
int32_t p2w_knn_grid(const float* xyzr_x, const uint64_t* keys_x, const int32_t* ptr_x, const p2w_grid* grid,
                     const float* xyzr_q, const int32_t* qidx, const int32_t* ptr_q, int32_t B, int32_t m_bound,
                     int32_t k, int32_t* nbr, int32_t* deg, const float* hint, int32_t flags, p2w_stream_t stream);
/* hint (optional, NULL = none): hint[q] = an upper bound of query q's k-th squared distance (+inf = none for q).  A
 * workgroup whose queries all have one skips the density probe and gathers exactly the box its bounds require; the
 * bound is verified (k candidates must turn up inside it), a wrong one only costs a rescan.  p2w_knn_hint2 derives
 * bounds for k <= 2 when the candidates are the sampled level of the queries' level (the interpolation searches):
 * rank[i] = index of query i's cell representative among the candidates (p2w_voxel_sample's inv_out, or
 * rank_sorted_out when the queries are given in sorted order). */
int32_t p2w_knn_hint2(const float* xyzr_q, const int32_t* rank, const int32_t* ptr_q, int32_t B, int32_t m_bound,
                      const float* xyzr_c, float* hint, p2w_stream_t stream);
int32_t p2w_ball_query_grid(const float* xyzr_x, const uint64_t* keys_x, const int32_t* ptr_x, const p2w_grid* grid,
                            const float* xyzr_q, const int32_t* qidx, const int32_t* ptr_q, int32_t B, int32_t m_bound,
                            double r, int32_t cap, int32_t* nbr, int32_t* deg, int32_t flags, p2w_stream_t stream);
/* p2w_voxel_sample_table on a workspace whose between-calls state (24 bytes of bounding-box words at their atomics' identities, one
 * counter at zero) is already in place: p2w_voxel_sample_table_prepare(ws) establishes it once on a fresh (or foreign-written)
 * workspace, every p2w_voxel_sample_table[_prepared] call leaves it in place again - so the three sub-samplings of a forward
 * (model.py:103-106, once per SA level) take 5 launches each instead of 7, none of them a memset.  Same arguments, same results. */
int32_t p2w_voxel_sample_table_prepare(void* ws, size_t ws_bytes, p2w_stream_t stream);
int32_t p2w_voxel_sample_table_prepared(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n_bound, float res,
                                        int32_t* idx_out, int32_t* ptr_out, int32_t* batch_out, int32_t* order_out,
                                        uint64_t* sorted_keys_out, uint64_t* cell_keys_out, p2w_grid* grid_out, int32_t* inv_out,
                                        int32_t* rank_sorted_out, int32_t* cell_start_out, int32_t* cell_start_sorted_out,
                                        int32_t* status_out, int64_t table_cells, void* ws, size_t ws_bytes, p2w_stream_t stream);

/* The same two searches with the sampler's cell -> position table of the candidates (p2w_voxel_sample_table's
 * cell_start_out when the candidates are the level it produced, cell_start_sorted_out when they are its input points in
 * cell-sorted order; NULL = bisect keys_x as p2w_knn_grid does): identical results, the run tables cost one load per run end. */
int32_t p2w_knn_grid_indexed(const float* xyzr_x, const uint64_t* keys_x, const int32_t* ptr_x, const p2w_grid* grid,
                             const int32_t* cell_start, const float* xyzr_q, const int32_t* qidx, const int32_t* ptr_q, int32_t B,
                             int32_t m_bound, int32_t k, int32_t* nbr, int32_t* deg, const float* hint, int32_t flags,
                             p2w_stream_t stream);
int32_t p2w_ball_query_grid_indexed(const float* xyzr_x, const uint64_t* keys_x, const int32_t* ptr_x, const p2w_grid* grid,
                                    const int32_t* cell_start, const float* xyzr_q, const int32_t* qidx, const int32_t* ptr_q,
                                    int32_t B, int32_t m_bound, double r, int32_t cap, int32_t* nbr, int32_t* deg, int32_t flags,
                                    p2w_stream_t stream);

/* ---- plot scale: back-projection of the classification onto the original points (predicter.py:107-142) ---- */

/* order_out[i] = index of the i-th record in ascending Morton (Z-curve) order of its cell in `grid` (21 bits per
 * axis, cells clamped to the grid): consecutive records are close in all three dimensions - the query order for
 * P2W_SEARCH_BOX searches over a plot.  ws: p2w_morton_order_ws_bytes(n). */
size_t p2w_morton_order_ws_bytes(int32_t n);
int32_t p2w_morton_order(const float* xyzr, int32_t n, const p2w_grid* grid, int32_t* order_out, void* ws, size_t ws_bytes,
                         p2w_stream_t stream);

/* ---- plot -> voxels: the grid step of the reference's voxeliser (pointstowood/src/preprocessing.py:55-64, Voxelise.grid) -------
 * cell_out[i] = PyG voxel_grid(P, size) with batch = None over ALL D <= 16 columns of the row-major table P[n, ld] (the
 * reference hands the whole point table - x, y, z, reflectance, ..., n_z - to voxel_grid): sum_d trunc((P_id - lo_d) / size) *
 * stride_d with the column minima lo_d and the running products of the per-column cell counts as strides (fp32 subtract /
 * divide / truncate, int64 result).  A row with a non-finite value takes no part in the minima / maxima and gets the key
 * P2W_CELL_NONFINITE (sorts last; the reference's own result on such input is the cast of a NaN, i.e. undefined).
 * ws: >= 256 bytes. */
#define P2W_CELL_NONFINITE INT64_MAX
int32_t p2w_cells_nd(const float* P, int32_t n, int32_t D, int32_t ld, float size, int64_t* cell_out, void* ws, size_t ws_bytes,
                     p2w_stream_t stream);
/* STABLE ascending sort of n (64-bit key, int32 value) pairs - hand-written LSD radix sort, 8-bit digits, as many passes as the
 * largest key has bytes (decided on the device).  vals_in NULL: values = 0..n-1, i.e. vals_out = the stable argsort the
 * voxeliser needs (points of a voxel keep their order, as the reference's nonzero() scan gives them).  In and out buffers must
 * differ.  ws: 16-byte aligned, p2w_sort_pairs_u64_ws_bytes(n) bytes. */
size_t p2w_sort_pairs_u64_ws_bytes(int32_t n);
int32_t p2w_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int32_t n, void* ws,
                           size_t ws_bytes, p2w_stream_t stream);
/* Runs of equal keys in a SORTED key array that hold at least min_count elements (the voxeliser's min_pts filter,
 * preprocessing.py:60-62): starts_out / counts_out[0 .. *n_out) in ascending order (buffers of n entries), *n_out on the device.
 * ws: 16-byte aligned, p2w_key_runs_ws_bytes(n) bytes. */
size_t p2w_key_runs_ws_bytes(int32_t n);
int32_t p2w_key_runs(const uint64_t* keys_sorted, int32_t n, int32_t min_count, int32_t* starts_out, int32_t* counts_out, int32_t* n_out,
                     void* ws, size_t ws_bytes, p2w_stream_t stream);

/* Cell -> first-candidate table of a plot-level grid: table_out[c] = number of keys_sorted[0..n) below c, for c = 0 .. n_cells
 * (n_cells + 1 entries; n_cells = dims[0] * dims[1] * dims[2] of the p2w_grid the keys were made on, times the voxel count).  It is
 * the cell_start argument of p2w_knn_grid_indexed / p2w_ball_query_grid_indexed for levels that did not come from the table sampler -
 * the back-projection's search over all classified points of a plot (predicter.py:129-142: the reference's KD-tree): one load per
 * search run instead of a bisection of the keys.  n_cells < 2^31 - 1.  ws: 16-byte aligned, p2w_cell_starts_ws_bytes(n_cells) bytes. */
size_t p2w_cell_starts_ws_bytes(int64_t n_cells);
int32_t p2w_cell_starts(const uint64_t* keys_sorted, int32_t n, int64_t n_cells, int32_t* table_out, void* ws, size_t ws_bytes,
                        p2w_stream_t stream);

/* Exact fp64 re-ranking of a grid kNN result - the neighbour sets of the reference's KD-tree (predicter.py:136-137: pykdtree over
 * the float64 `classified_pc` of :205, float64 queries), which the fp32 searches above can miss where two candidates' distances
 * differ by less than an fp32 rounding.  cand_sorted[nc][3]: the candidates' float64 coordinates in the cell-sorted order of
 * `grid` (p2w_voxel_sample's order_out); cand_index[p] = the index the result reports for sorted position p, cand_pos = its
 * inverse; keys_sorted / cell_start (optional) / grid: as p2w_knn_grid_indexed; (ox, oy, oz): the offset that was subtracted
 * from the float64 coordinates to make the fp32 ones the grid was built on; q[m][3]: float64 queries.  nbr[m, k] / deg[m]: in
 * = any k candidates per query (the fp32 result; deg < k only where fewer than k candidates exist), out = the k nearest in
 * float64, ascending (squared distance = ((dx^2 + dy^2) + dz^2), candidate index).  k <= P2W_MAX_K. */
int32_t p2w_knn_refine_f64(const double* cand_sorted, const int32_t* cand_index, const int32_t* cand_pos,
                           const uint64_t* keys_sorted, const int32_t* cell_start, const p2w_grid* grid, double ox, double oy,
                           double oz, const double* q, int32_t m, int32_t nc, int32_t k, int32_t* nbr, int32_t* deg,
                           p2w_stream_t stream);

/* PointCloudClassifier.compute_labels (predicter.py:112-127) over a neighbour table nbr[n,k] (indices into pred /
 * prob, deg[i] valid entries): pwood_out = median of the neighbours' probabilities (np.median: mean of the two middle
 * values for an even count); label_out: any_wood != 1 -> 1 if any neighbour's prediction > any_wood else 0;
 * any_wood == 1 -> argmax_j sum_{pred == j} prob over j in {0, 1} (first maximum, sums in fp64 like numba's). */
int32_t p2w_vote(const int32_t* nbr, const int32_t* deg, int32_t k, const float* pred, const float* prob, int32_t n,
                 float any_wood, float* label_out, float* pwood_out, p2w_stream_t stream);

/* Optional accelerator for the searches (results are identical with or without it): bounding boxes (lo xyz, hi xyz)
 * of the candidate tiles of xyzr_x (1024 consecutive records of one voxel).  bbox holds p2w_tile_bbox_count(B,
 * n_bound) x 6 floats.  With it, p2w_knn skips tiles whose box is farther than a query's current k-th distance and
 * p2w_ball_query tiles farther than r; it pays when the candidates are stored in a spatially coherent order. */
int32_t p2w_tile_bbox(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n_bound, float* bbox, p2w_stream_t stream);
int32_t p2w_tile_bbox_count(int32_t B, int32_t n_bound);

/* ---- features ---------------------------------------------------------- */

/* stem_mlp: out[n,C] = relu(W[C,3] * xyz + b) - model.py:208,228. */
int32_t p2w_stem(const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out,
                 p2w_stream_t stream);

/* Epilogue description for p2w_gemm: v = acc + bias; relu0; v = v*sc0+sh0; relu1; v = v*sc1+sh1; relu2;
 * v += residual; relu_final.  NULL scale pointers skip their stage. */
typedef struct p2w_epilogue {
    const float* bias;   /* [N] or NULL */
    const float* sc0;    /* [N] */
    const float* sh0;    /* [N] */
    const float* sc1;    /* [N] */
    const float* sh1;    /* [N] */
    const float* residual; /* [M, ldr] or NULL */
    int32_t ldr;
    int32_t relu0, relu1, relu2, relu_final;
    uint32_t* range;     /* NULL, or the launch's range report (the caller zeroes it; p2w_gemm ignores it): P2W_RANGE_SLOTS word
                            pairs, 64 words apart (P2W_RANGE_WORDS words in all; a workgroup writes the pair of its number modulo
                            the slot count - one hot address would queue every store in one L2 channel).  Word 0 of a pair is set
                            to 1 if an epilogue value lies beyond +-P2W_RANGE_HI (or is NaN), word 1 if one lies beyond
                            +-P2W_RANGE_LO; the report is the OR over the slots.  It is the range watch of the split-fp16
                            arithmetic: past 65504 an H value has lost its low part, and a tensor WITHOUT a value of ordinary
                            size has lost it to fp16's subnormal floor (2^-24 absolute) */
    const void* interp;  /* NULL, or [M] records {int32 n0, int32 n1, float a0, float a1} (p2w_interp_weights; 16-byte aligned):
                            p2w_gemm_h2 / _sk add to output row r not residual[r] but a0 * residual[n0] + a1 * residual[n1] - the
                            rows of a COARSER level's fp32 matrix `residual` [interp_rows, ldr] interpolated on the fly (an FP
                            module's layer 0 by linearity: relu(W [interp(y) | skip] + b) = relu(W_s skip + b + interp(W_i y)),
                            model.py:149-153).  128 x 128 tile only (P2W_GEMM_TILE_256 is ignored); with P2W_GEMM_RESIDUAL_H: P2W_EUNSUPPORTED;
                            p2w_gemm and p2w_gemm_h2_rowdot refuse it (P2W_EUNSUPPORTED) */
    int32_t interp_rows; /* rows of `residual` when `interp` is given: every n0, n1 lies below it */
} p2w_epilogue;

#define P2W_RANGE_HI 6.0e4f
#define P2W_RANGE_LO 0.03125f
#define P2W_RANGE_SLOTS 16
#define P2W_RANGE_WORDS (64 * P2W_RANGE_SLOTS)

/* Packed weight: Wp[N_pad][K_pad] fp32, row n = output channel, k contiguous, zero padded
 * (K_pad = round_up(K, 32), N_pad = round_up(N, 256)); see p2w_packed_dims. */
void p2w_packed_dims(int32_t N, int32_t K, int32_t* N_pad, int32_t* K_pad);

/* out[M,N] = epilogue(A[M,K] * W^T): Linear / 1x1 Conv1d + folded BatchNorm / depthwise affines +
 * ReLU + residual - model.py:75-85 (InvertedResidualBlock), :198-202 (MLP), :241-242 (head).
 * fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32 products, fp32 accumulate. */
int32_t p2w_gemm(const float* A, int32_t lda, const float* Wp, int32_t M, int32_t N, int32_t K,
                 const p2w_epilogue* epi, float* out, int32_t ldo, p2w_stream_t stream);

/* Fused PointNetConv (pointnet.py:86-132) for one SA level:
 *   per target i and neighbour slot s (j = nbr[i,s]):
 *     rel  = xyz_s[j] - xyz_s[idx[i]]            on sf-scaled coordinates (model.py:122)
 *     dmax = max_s ||rel||                        (scatter_max, pointnet.py:122)
 *     h1   = relu(P[j] + (rel/(dmax+1e-8)) * W1r + refl_j * W1f)       layer 1, hoisted:
 *            P = x_src * W1x^T + b1 is computed once per source point by p2w_gemm
 *     h2   = relu(h1 * W2^T + b2) * bn_s + bn_t                       layer 2 + BN (model.py:198-202)
 *   out[i] = max over valid slots of h2 (rows without neighbours = 0). */
int32_t p2w_sa_conv(const float* P, int32_t ldp, const float* xyzr_src, const int32_t* idx, const int32_t* batch_dst,
                    const float* sf, const int32_t* nbr, const int32_t* deg, int32_t kw, int32_t M,
                    const float* w1r4 /* [4][C1_pad] */, const float* W2p, int32_t C1, int32_t C2,
                    const float* b2, const float* bn_s, const float* bn_t, float* out, int32_t ldo,
                    p2w_stream_t stream);

/* ---- H family: the same GEMM / PointNetConv on 16-bit MFMA operands ------------------------------------------
 * Precision of an H tensor and of the arithmetic on it (`prec` argument of every entry point below):
 *   P2W_PREC_F16X3  fp16 hi/lo planes, a*w = a_lo*w_hi + a_hi*w_lo + a_hi*w_hi on v_mfma_f32_32x32x16_f16 with fp32
 *                   accumulation (~22-bit operands): the parity mode, meets the 1e-4 probability bar of the fp32 path;
 *   P2W_PREC_F16    one fp16 plane (round to nearest, saturating at +-65504), one MFMA per product;
 *   P2W_PREC_BF16   one bf16 plane, v_mfma_f32_32x32x16_bf16.
 *   The single-plane modes are what the reference's own GPU path computes in (torch.cuda.amp.autocast,
 *   pointstowood/src/predicter.py:197); they do NOT meet the 1e-4 bar (tests report the measured error).
 * H tensor [M, F]: row m holds planes x ldh 16-bit values, ldh >= F, pad columns zero.  Single plane: [v(0..ldh)],
 *   ldh % 64 == 0 when the tensor feeds p2w_gemm_h2.  F16X3: ldh % 32 == 0 and the row is a sequence of blocks of 32
 *   columns, each stored as [hi(32) | lo(32)] (column c: hi at 64*(c/32) + c%32, lo 32 values further; value = hi + lo).
 *   Either way the 64 values a GEMM K-slab takes from a row are one 128-byte cache line, staged by 16-byte direct-to-LDS
 *   copies (8 whole lines per wave-instruction).  F16X3 has the bytes of fp32, the others half.
 * H weights: Wh = [N_pad][planes * K_pad] of W * 2^e, rows (= output channels) laid out like H tensor rows (e chosen at
 *   pack time so that the F16X3 lo part stays a normal fp16; wscale = 2^-e is applied in the epilogue), dims from
 *   p2w_packed_dims_h. */
#define P2W_PREC_F16X3 0
#define P2W_PREC_F16 1
#define P2W_PREC_BF16 2
int32_t p2w_packed_dims_h(int32_t prec, int32_t N, int32_t K, int32_t* N_pad, int32_t* K_pad);

/* flags of p2w_gemm_h2 (0 = let the library choose).  There are no environment switches behind this ABI. */
#define P2W_GEMM_TILE_128 1      /* force the 128 x 128 workgroup tile */
#define P2W_GEMM_TILE_256 2      /* force the 256 x 256 workgroup tile */
#define P2W_GEMM_GENERIC_EPI 4   /* run the runtime-flag epilogue instead of the specialised one */
#define P2W_GEMM_ORDER_ROWS 8    /* tile order: an XCD owns whole row tiles (W re-read from its L2) */
#define P2W_GEMM_ORDER_COLS 16   /* tile order: an XCD owns a slice of column tiles (A streamed per slice) */
#define P2W_GEMM_RESIDUAL_H 32   /* epi->residual is an H tensor of the launch's precision (epi->ldr = its row pitch ldh), not fp32 */
#define P2W_GEMM_STREAMK 64      /* p2w_gemm_h2_sk: run the rows behind the whole chip rounds as a split-K tail even where the cost model says no */
#define P2W_GEMM_NO_STREAMK 128  /* p2w_gemm_h2_sk: never (= p2w_gemm_h2) */
#define P2W_GEMM_TILE_64 (1 << 24)  /* force the 64 x 128 workgroup tile (three workgroups per CU; the library takes it for short-K launches
                                       with one or two column tiles or a mostly empty last round of 128 x 128 tiles) */
#define P2W_GEMM_NO_TILE_64 (1 << 25)  /* never take it */
/* flags of p2w_sa_conv_h (0 = let the library choose the work-item shape by C2) */
#define P2W_SA_ITEM_256 1        /* 4 targets x 256 output columns per work item */
#define P2W_SA_ITEM_128 2        /* 8 targets x 128 output columns per work item */
#define P2W_SA_PACK8 4           /* targets with <= 8 neighbours share a 32-row MFMA tile four at a time (sparse ball-query levels) */
/* bits 16..23 of `flags` of p2w_gemm_h2 / p2w_sa_conv_h: profiling ablations, honoured only by diagnostic builds
 * (-DP2W_GEMM_ABLATE / -DP2W_SA_ABLATE); production builds ignore them.
 * bits 8..15 of `flags` of p2w_gemm_h2: scheduling experiments (tools/gemm_desync.py; results unchanged): bits 8..13 = start
 * stagger of every other workgroup of an XCD in units of 2 us, bit 14 / 15 = persistent grid on 1/2 / 1/4 of the CUs. */

/* p2w_gemm with an H A operand, H weights and fp32 and/or H outputs (either pointer may be NULL):
 * Linear / 1x1 Conv1d + folded BatchNorm / depthwise affines + ReLU + residual - model.py:75-85, :198-202, :241-242.
 * ldh_a / ldh_o are row PITCHES: A_h / out_h may point into wider rows (e.g. the skip columns of a concatenated [interp | skip]
 * row, model.py:151: the producer of the skip features writes them in place).  The launch writes its N output columns and the
 * zero pad columns up to min(ldh_o, round_up(N, K granularity)); it never touches columns beyond that. */
int32_t p2w_gemm_h2(int32_t prec, const void* A_h, int32_t ldh_a, const void* Wh, float wscale, int32_t M, int32_t N,
                    int32_t K, const p2w_epilogue* epi, float* out_f32, int32_t ldo, void* out_h, int32_t ldh_o,
                    int32_t flags, p2w_stream_t stream);
/* p2w_gemm_h2 with a caller-owned workspace (the ABI never allocates): the library may then run the rows that do not fill a whole
 * round of the chip's workgroup slots as a second launch - plainly, or as a SPLIT-K tail: each 128 x 128 tile's K range cut into S
 * pieces that run side by side (raw fp32 partial tiles in `ws`) + a fix-up pass that adds a tile's pieces in ascending K order
 * (deterministic) and runs the same epilogue.  What a launch takes is decided from (M, N, K) alone, so equal calls give equal
 * bits; results differ from p2w_gemm_h2's in the last fp32 bits only (the K range is summed in pieces).  ws: 16-byte aligned,
 * p2w_gemm_h2_sk_ws_bytes() bytes (32 MiB on a 256-CU chip) make every plan possible (a smaller one narrows the choice, ws = NULL
 * is p2w_gemm_h2); it must not be shared by launches that may run concurrently (one per stream). */
size_t p2w_gemm_h2_sk_ws_bytes(void);
int32_t p2w_gemm_h2_sk(int32_t prec, const void* A_h, int32_t ldh_a, const void* Wh, float wscale, int32_t M, int32_t N,
                       int32_t K, const p2w_epilogue* epi, float* out_f32, int32_t ldo, void* out_h, int32_t ldh_o,
                       void* ws, size_t ws_bytes, int32_t flags, p2w_stream_t stream);
/* conv1 + BN + ReLU + conv2 for ONE output channel (model.py:241-243) as one operator: out[i] = dot(epilogue(A_h[i,:] * W^T), dot_w) +
 * dot_b.  The [M, N] intermediate never reaches HBM: the GEMM's epilogue leaves one partial sum per row and 64-column slice in
 * ws, a finishing pass adds the slices in a fixed order (deterministic).  epi->residual is not supported (P2W_EUNSUPPORTED).
 * ws: 16-byte aligned, p2w_gemm_h2_rowdot_ws_bytes(M, N) bytes. */
size_t p2w_gemm_h2_rowdot_ws_bytes(int32_t M, int32_t N);
int32_t p2w_gemm_h2_rowdot(int32_t prec, const void* A_h, int32_t ldh_a, const void* Wh, float wscale, int32_t M, int32_t N,
                           int32_t K, const p2w_epilogue* epi, const float* dot_w, float dot_b, float* out, void* ws,
                           size_t ws_bytes, int32_t flags, p2w_stream_t stream);
/* p2w_sa_conv with H weights W2h and fp32 and/or H outputs.  P (the hoisted layer-1 product, fp32) has n_src + 1 rows of
 * ldp >= round_up(C1, K granularity) floats: rows 0..n_src-1 = x_src * W1x^T + b1 with ZERO pad columns; row n_src is the row
 * empty neighbour slots read (the kernel's loads are unconditional): it must EXIST, the call fills it with zeros itself (the only
 * write through P).
 * ws: 16-byte aligned scratch of >= p2w_sa_conv_h_ws_bytes(M, flags) bytes for the per-edge metadata (P2W_EWORKSPACE otherwise);
 * kw <= 32 (one 32-row MFMA tile per target); round_up(C1, K granularity) <= 512, C2 <= 1024 (LDS tables), M < 2^25 and
 * (n_src + 1) * ldp < 2^33 (32-bit offsets) - P2W_EUNSUPPORTED otherwise. */
size_t p2w_sa_conv_h_ws_bytes(int32_t M, int32_t flags);
int32_t p2w_sa_conv_h(int32_t prec, const float* P, int32_t ldp, int32_t n_src, const float* xyzr_src, const int32_t* idx,
                      const int32_t* batch_dst, const float* sf, const int32_t* nbr, const int32_t* deg, int32_t kw,
                      int32_t M, const float* w1r4, const void* W2h, float wscale, int32_t C1, int32_t C2,
                      const float* b2, const float* bn_s, const float* bn_t, float* out, int32_t ldo, void* out_h,
                      int32_t ldh, void* ws, size_t ws_bytes, int32_t flags, p2w_stream_t stream);
/* p2w_sa_conv_h for a P whose rows are NOT in the source points' own order: src_row[j] = the P row of source point j (NULL =
 * identity = p2w_sa_conv_h).  The engine keeps the level-0 features in the sampler's cell order (p2w_stem_h2_indexed): neighbours
 * in space are neighbours in memory for the P gather and for the last interpolation. */
int32_t p2w_sa_conv_h_rows(int32_t prec, const float* P, int32_t ldp, int32_t n_src, const float* xyzr_src, const int32_t* idx,
                           const int32_t* batch_dst, const float* sf, const int32_t* nbr, const int32_t* deg, int32_t kw,
                           int32_t M, const float* w1r4, const void* W2h, float wscale, int32_t C1, int32_t C2,
                           const float* b2, const float* bn_s, const float* bn_t, float* out, int32_t ldo, void* out_h,
                           int32_t ldh, void* ws, size_t ws_bytes, int32_t flags, const int32_t* src_row, uint32_t* range,
                           p2w_stream_t stream);
/* The small kernels writing H (and fp32 where given): stem (model.py:208,228), knn_interpolate + cat (:149-151),
 * cat(x, pos) (:135).  For the stem and the interpolation `ldh` is the row pitch as in p2w_gemm_h2: they write their columns
 * (C, resp. Fc + Fs) plus the zero pad to the next K-slab boundary and leave the rest of a wider row alone (p2w_interp_concat_h2
 * with skip = NULL, Fs = 0 writes only the interpolated part of a row whose skip columns another producer has written). */
/* range (p2w_stem_h2, p2w_stem_h2_indexed, p2w_sa_conv_h_rows; NULL = off): the range watch of p2w_epilogue.range - a device
 * block of P2W_RANGE_WORDS words (the caller zeroes them). */
int32_t p2w_stem_h2(int32_t prec, const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out,
                    void* out_h, int32_t ldh, uint32_t* range, p2w_stream_t stream);
/* ... for records that come in another order (p2w_index_records: e.g. the sampler's cell order) and carry their own row as the
 * int32 in .w: the fp32 row of record i is out[.w], its H row is out_h[i] (model.py:228 stores the stem features on the batch in
 * input order; the H copy feeds the GEMMs in the records' order). */
int32_t p2w_stem_h2_indexed(int32_t prec, const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out,
                            void* out_h, int32_t ldh, uint32_t* range, p2w_stream_t stream);
int32_t p2w_interp_concat_h2(int32_t prec, const float* xc, int32_t Fc, const float* xyzr_c, const float* xyzr_f,
                             const int32_t* nbr, const int32_t* deg, int32_t kw, const float* skip, int32_t Fs, int32_t m,
                             void* out_h, int32_t ldh, p2w_stream_t stream);
/* knn_interpolate's weights alone (model.py:149, kw <= 2), for p2w_epilogue.interp: records[q] = {n0, n1, a0, a1}, a_s = w_s /
 * (w_0 + w_1), w_s = 1 / max(d2, 1e-16); a row with one neighbour gets {n0, n0, 1, 0}.  kw > 2: P2W_EUNSUPPORTED. */
int32_t p2w_interp_weights(const float* xyzr_c, const float* xyzr_f, const int32_t* nbr, const int32_t* deg, int32_t kw, int32_t m,
                           void* records, p2w_stream_t stream);
int32_t p2w_concat_xyz_h2(int32_t prec, const float* x, int32_t F, const float* xyzr, int32_t m, void* out_h, int32_t ldh,
                          p2w_stream_t stream);

/* knn_interpolate (k<=2) + concat with the skip features - model.py:149-151:
 *   out[q, 0:Fc] = (sum_s w_s * xc[nbr[q,s]]) / (sum_s w_s),  w_s = 1/max(d2, 1e-16)
 *   out[q, Fc:Fc+Fs] = skip[q].  Rows padded to ldo are zero-filled. */
int32_t p2w_interp_concat(const float* xc, int32_t Fc, const float* xyzr_c, const float* xyzr_f, const int32_t* nbr,
                          const int32_t* deg, int32_t kw, const float* skip, int32_t Fs, int32_t m, float* out,
                          int32_t ldo, p2w_stream_t stream);

/* [x | xyz] concat feeding GlobalSAModule.NN - model.py:135. */
int32_t p2w_concat_xyz(const float* x, int32_t F, const float* xyzr, int32_t m, float* out, int32_t ldo,
                       p2w_stream_t stream);

/* global_max_pool - model.py:136: out[b,:] = max over rows ptr[b]..ptr[b+1] (empty = 0). */
int32_t p2w_segment_max(const float* x, int32_t ldx, int32_t F, const int32_t* ptr, int32_t B, float* out,
                        p2w_stream_t stream);

/* conv2 with one output channel - model.py:243: out[i] = dot(x[i,:], w) + b. */
int32_t p2w_rowdot(const float* x, int32_t ldx, int32_t F, const float* w, float b, int32_t m, float* out,
                   p2w_stream_t stream);

/* FPModule 4 (model.py:236): the coarse level has ONE point per voxel: nbr[q] = batch[q], deg = 1. */
int32_t p2w_fill_batch_nbr(const int32_t* batch, int32_t m, int32_t* nbr, int32_t* deg, p2w_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* P2W_H */
