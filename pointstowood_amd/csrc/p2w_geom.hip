// Geometry kernels: record packing, grid sub-sampling, level gather, ball query, exact kNN.
// gfx950 only (wave64, LDS-staged candidate tiles, wave-level ballot/popcount selection).
#include "p2w_common.h"
#include "p2w_sort.h"
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------------
// pack: xyzr + batch ids
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_xyzr_kernel(const float* __restrict__ pos, int pos_stride,
                                                        const float* __restrict__ refl, const int* __restrict__ ptr,
                                                        int B, int n, float4* __restrict__ xyzr, int* __restrict__ batch) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = pos + (size_t)i * pos_stride;
    xyzr[i] = make_float4(p[0], p[1], p[2], refl ? refl[i] : 0.0f);
    batch[i] = p2w_find_segment(ptr, B, i);
}

extern "C" int32_t p2w_pack_xyzr(const float* pos, int32_t pos_stride, const float* refl, const int32_t* ptr, int32_t B,
                                 int32_t n, float* xyzr, int32_t* batch, p2w_stream_t stream) {
    if (n == 0) return P2W_OK;
    P2W_CHECK_PTR(pos); P2W_CHECK_PTR(ptr); P2W_CHECK_PTR(xyzr); P2W_CHECK_PTR(batch);
    P2W_CHECK_ALIGN16(xyzr);
    if (n < 0 || B <= 0 || pos_stride < 3) return P2W_EINVAL;
    pack_xyzr_kernel<<<p2w_cdiv(n, 256), 256, 0, p2w_s(stream)>>>(pos, pos_stride, refl, ptr, B, n,
                                                                   reinterpret_cast<float4*>(xyzr), batch);
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// voxel sample = voxel_grid + consecutive_cluster
// ------------------------------------------------------------------------------------------------
// float <-> order-preserving uint (for atomic min/max)
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

struct VsHeader {          // lives at the start of the workspace
    unsigned lo[3], hi[3]; // order-encoded min / max of x, y, z over the whole batch
    int pad[2];
};

__global__ void vs_init_kernel(VsHeader* h) {
    if (threadIdx.x < 3) { h->lo[threadIdx.x] = 0xffffffffu; h->hi[threadIdx.x] = 0u; }
}

__global__ __launch_bounds__(256) void vs_minmax_kernel(const float4* __restrict__ xyzr, const int* __restrict__ ptr, int B,
                                                        VsHeader* h, int* __restrict__ status_zero = nullptr) {
    __shared__ float red[4][6];
    const int n = ptr[B];
    if (status_zero && blockIdx.x == 0 && threadIdx.x == 0) *status_zero = 0;   // (the table sampler's status word: set by tk_insert_kernel, behind this launch)
    float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};  // lo xyz, hi xyz
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 p = xyzr[i];
        v[0] = fminf(v[0], p.x); v[3] = fmaxf(v[3], p.x);
        v[1] = fminf(v[1], p.y); v[4] = fmaxf(v[4], p.y);
        v[2] = fminf(v[2], p.z); v[5] = fmaxf(v[5], p.z);
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float o = __shfl_xor(v[d], off);
            v[d] = d < 3 ? fminf(v[d], o) : fmaxf(v[d], o);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 6; ++d) red[wave][d] = v[d];
    }
    __syncthreads();
    if (threadIdx.x < 6 && n > 0) {  // one atomic per block and component
        const int d = threadIdx.x;
        float r = red[0][d];
        for (int w = 1; w < 4; ++w) r = d < 3 ? fminf(r, red[w][d]) : fmaxf(r, red[w][d]);
        if (d < 3) atomicMin(&h->lo[d], f2ord(r)); else atomicMax(&h->hi[d - 3], f2ord(r));
    }
}

// key_i = sum_d trunc((P_id - lo_d) / S_d) * stride_d over d = x, y, z, batch   (oracle/ops.py voxel_grid)
__global__ __launch_bounds__(256) void vs_keys_kernel(const float4* __restrict__ xyzr, const int* __restrict__ ptr, int B,
                                                      int n_bound, float res, const VsHeader* __restrict__ h,
                                                      unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_bound) return;
    const int n = ptr[B];
    if (vals) vals[i] = i;
    if (i >= n) { keys[i] = ~0ull; return; }
    // batch column: min of the batch ids that occur
    int b_lo = 0;
    while (b_lo < B - 1 && ptr[b_lo + 1] == ptr[b_lo]) ++b_lo;
    const float lo0 = ord2f(h->lo[0]), lo1 = ord2f(h->lo[1]), lo2 = ord2f(h->lo[2]);
    const long long c0 = (long long)((ord2f(h->hi[0]) - lo0) / res) + 1;
    const long long c1 = (long long)((ord2f(h->hi[1]) - lo1) / res) + 1;
    const long long c2 = (long long)((ord2f(h->hi[2]) - lo2) / res) + 1;
    const float4 p = xyzr[i];
    const int b = p2w_find_segment(ptr, B, i);
    const long long k0 = (long long)((p.x - lo0) / res);
    const long long k1 = (long long)((p.y - lo1) / res);
    const long long k2 = (long long)((p.z - lo2) / res);
    const long long kb = (long long)(((float)b - (float)b_lo) / 1.0f);
    const long long key = k0 + k1 * c0 + k2 * (c0 * c1) + kb * (c0 * c1 * c2);
    keys[i] = (unsigned long long)key;
}

__global__ __launch_bounds__(256) void vs_iota_kernel(int n, int* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) vals[i] = i;
}

// n_dev: device count of valid (non-padding) keys, or nullptr when all n_bound keys are valid
__global__ __launch_bounds__(256) void vs_flags_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ n_dev,
                                                       int n_bound, int* __restrict__ flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_bound) return;
    const int n = n_dev ? *n_dev : n_bound;
    flags[i] = (i < n && (i == n - 1 || keys[i] != keys[i + 1])) ? 1 : 0;  // last of each run = largest point index
}

// perm (one representative per run, ascending key), optional inv (rank of every point's key), optional CSR of the result,
// optional search index of the result (sorted keys of all points, key of each representative, grid geometry)
__global__ __launch_bounds__(256) void vs_scatter_kernel(const int* __restrict__ flags, const int* __restrict__ scan,
                                                         const int* __restrict__ vals, const int* __restrict__ ptr, int B,
                                                         const int* __restrict__ n_dev, int n_bound, int* __restrict__ idx_out,
                                                         int* __restrict__ ptr_out, int* __restrict__ batch_out,
                                                         int* __restrict__ inv_out, int* __restrict__ count_out,
                                                         int* __restrict__ order_out, int* __restrict__ rank_sorted_out,
                                                         const unsigned long long* __restrict__ keys_sorted,
                                                         unsigned long long* __restrict__ sorted_keys_out,
                                                         unsigned long long* __restrict__ cell_keys_out,
                                                         const VsHeader* __restrict__ h, float res, p2w_grid* __restrict__ grid_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_dev ? *n_dev : n_bound;
    const int total = (n == 0) ? 0 : scan[n - 1] + flags[n - 1];
    if (i == 0 && count_out) *count_out = total;
    if (i == 0 && grid_out && h) {   // the geometry vs_keys_kernel used
        int b_lo = 0;
        while (b_lo < B - 1 && ptr[b_lo + 1] == ptr[b_lo]) ++b_lo;
        p2w_grid g;
        for (int d = 0; d < 3; ++d) {
            g.lo[d] = ord2f(h->lo[d]); g.hi[d] = ord2f(h->hi[d]);
            g.dims[d] = (n == 0) ? 1 : (long long)((g.hi[d] - g.lo[d]) / res) + 1;
        }
        g.res = res; g.b_lo = b_lo;
        *grid_out = g;
    }
    if (ptr_out && i <= B) {  // sorted position p belongs to voxel b iff ptr[b] <= p < ptr[b+1] (keys are voxel-major)
        const int p = ptr[i];
        ptr_out[i] = (p < n) ? scan[p] : total;
    }
    if (i >= n || i >= n_bound) return;
    if (inv_out) inv_out[vals[i]] = scan[i];
    if (order_out) order_out[i] = vals[i];
    if (rank_sorted_out) rank_sorted_out[i] = scan[i];
    if (sorted_keys_out) sorted_keys_out[i] = keys_sorted[i];
    if (flags[i]) {
        const int o = scan[i];
        idx_out[o] = vals[i];
        if (batch_out) batch_out[o] = p2w_find_segment(ptr, B, i);
        if (cell_keys_out) cell_keys_out[o] = keys_sorted[i];
    }
}

struct VsLayout { size_t hdr, keys_in, keys_out, vals_in, vals_out, flags, scan, temp, temp_bytes, total; };

static hipError_t vs_layout(int n_bound, VsLayout* L) {
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    RsLayout R;
    rs_layout(n_bound, &R);
    const size_t sort_bytes = R.bytes, scan_bytes = xs_ws_bytes(n_bound);   // the hand-written sort / scan of p2w_sort.h
    size_t off = 0;
    L->hdr = off; off += up(sizeof(VsHeader));
    L->keys_in = off; off += up(sizeof(unsigned long long) * n_bound);
    L->keys_out = off; off += up(sizeof(unsigned long long) * (n_bound + 1));
    L->vals_in = off; off += up(sizeof(int) * n_bound);
    L->vals_out = off; off += up(sizeof(int) * n_bound);
    L->flags = off; off += up(sizeof(int) * n_bound);
    L->scan = off; off += up(sizeof(int) * n_bound);
    L->temp = off; L->temp_bytes = up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes); off += L->temp_bytes;
    L->total = off;
    return hipSuccess;
}

extern "C" size_t p2w_voxel_sample_ws_bytes(int32_t n_bound) {
    if (n_bound <= 0) return 256;
    VsLayout L;
    if (vs_layout(n_bound, &L) != hipSuccess) return 0;
    return L.total;
}

// min/max + keys into keys_out[n_bound] (padding keys = ~0)
static int32_t vs_compute_keys(const float4* x4, const int* ptr, int B, int n_bound, float res, VsHeader* hdr,
                               unsigned long long* keys, int* vals, hipStream_t s) {
    const int nblk = p2w_cdiv(n_bound, 256);
    vs_init_kernel<<<1, 64, 0, s>>>(hdr);
    vs_minmax_kernel<<<nblk < 256 ? nblk : 256, 256, 0, s>>>(x4, ptr, B, hdr);
    vs_keys_kernel<<<nblk, 256, 0, s>>>(x4, ptr, B, n_bound, res, hdr, keys, vals);
    return P2W_LAUNCH_STATUS();
}

// sort (key, point) pairs, flag the last element of each run, compact
static int32_t vs_cluster(char* w, const VsLayout& L, const unsigned long long* keys_in, const int* ptr, int B,
                          const int* n_dev, int n_bound, int* idx_out, int* ptr_out, int* batch_out, int* inv_out,
                          int* count_out, int* order_out, int* rank_sorted_out, unsigned long long* sorted_keys_out,
                          unsigned long long* cell_keys_out, float res, p2w_grid* grid_out, hipStream_t s) {
    auto* keys_out = reinterpret_cast<unsigned long long*>(w + L.keys_out);
    int* vals_in = reinterpret_cast<int*>(w + L.vals_in);
    int* vals_out = reinterpret_cast<int*>(w + L.vals_out);
    int* flags = reinterpret_cast<int*>(w + L.flags);
    int* scan = reinterpret_cast<int*>(w + L.scan);
    const int nblk = p2w_cdiv(n_bound, 256);
    // stable radix sort of the (key, point) pairs: only the valid ones (the padding keys beyond *n_dev stay where they are and
    // would make every one of the eight digit passes run), then flags -> ranks
    hipError_t e = rs_sort_pairs(w + L.temp, keys_in, keys_out, vals_in, vals_out, n_dev, n_bound, s);
    if (e != hipSuccess) return (int32_t)e;
    vs_flags_kernel<<<nblk, 256, 0, s>>>(keys_out, n_dev, n_bound, flags);
    e = xs_exclusive_scan(w + L.temp, flags, scan, n_bound, s);
    if (e != hipSuccess) return (int32_t)e;
    const int nblk2 = p2w_cdiv((n_bound > B + 1 ? n_bound : B + 1), 256);
    vs_scatter_kernel<<<nblk2, 256, 0, s>>>(flags, scan, vals_out, ptr, B, n_dev, n_bound, idx_out, ptr_out, batch_out,
                                            inv_out, count_out, order_out, rank_sorted_out, keys_out, sorted_keys_out, cell_keys_out,
                                            reinterpret_cast<const VsHeader*>(w + L.hdr), res, grid_out);
    return P2W_LAUNCH_STATUS();
}

extern "C" int32_t p2w_voxel_sample(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n_bound, float res,
                                    int32_t* idx_out, int32_t* ptr_out, int32_t* batch_out, int32_t* order_out,
                                    uint64_t* sorted_keys_out, uint64_t* cell_keys_out, p2w_grid* grid_out, int32_t* inv_out,
                                    int32_t* rank_sorted_out, void* ws, size_t ws_bytes, p2w_stream_t stream) {
    P2W_CHECK_PTR(ptr); P2W_CHECK_PTR(ptr_out);
    if (B <= 0 || n_bound < 0 || !(res > 0.0f)) return P2W_EINVAL;
    hipStream_t s = p2w_s(stream);
    if (n_bound == 0) return (int32_t)hipMemsetAsync(ptr_out, 0, sizeof(int) * (B + 1), s);
    P2W_CHECK_PTR(xyzr); P2W_CHECK_PTR(idx_out); P2W_CHECK_PTR(batch_out); P2W_CHECK_PTR(ws);
    P2W_CHECK_ALIGN16(xyzr); P2W_CHECK_ALIGN16(ws);
    VsLayout L;
    hipError_t e = vs_layout(n_bound, &L);
    if (e != hipSuccess) return (int32_t)e;
    if (ws_bytes < L.total) return P2W_EWORKSPACE;
    char* w = static_cast<char*>(ws);
    auto* keys_in = reinterpret_cast<unsigned long long*>(w + L.keys_in);
    int32_t st = vs_compute_keys(reinterpret_cast<const float4*>(xyzr), ptr, B, n_bound, res,
                                 reinterpret_cast<VsHeader*>(w + L.hdr), keys_in, reinterpret_cast<int*>(w + L.vals_in), s);
    if (st != P2W_OK) return st;
    return vs_cluster(w, L, keys_in, ptr, B, ptr + B, n_bound, idx_out, ptr_out, batch_out, inv_out, nullptr, order_out,
                      rank_sorted_out, reinterpret_cast<unsigned long long*>(sorted_keys_out),
                      reinterpret_cast<unsigned long long*>(cell_keys_out), res, grid_out, s);
}

extern "C" int32_t p2w_voxel_grid(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n, float res, int64_t* cell_out,
                                  void* ws, size_t ws_bytes, p2w_stream_t stream) {
    if (n == 0) return P2W_OK;
    P2W_CHECK_PTR(xyzr); P2W_CHECK_PTR(ptr); P2W_CHECK_PTR(cell_out); P2W_CHECK_PTR(ws);
    P2W_CHECK_ALIGN16(xyzr); P2W_CHECK_ALIGN16(ws);
    if (B <= 0 || n < 0 || !(res > 0.0f)) return P2W_EINVAL;
    if (ws_bytes < 256) return P2W_EWORKSPACE;
    return vs_compute_keys(reinterpret_cast<const float4*>(xyzr), ptr, B, n, res, static_cast<VsHeader*>(ws),
                           reinterpret_cast<unsigned long long*>(cell_out), nullptr, p2w_s(stream));
}

extern "C" int32_t p2w_consecutive_cluster(const int64_t* cell, int32_t n, int32_t* inv_out, int32_t* perm_out,
                                           int32_t* count_out, void* ws, size_t ws_bytes, p2w_stream_t stream) {
    P2W_CHECK_PTR(count_out);
    hipStream_t s = p2w_s(stream);
    if (n == 0) return (int32_t)hipMemsetAsync(count_out, 0, sizeof(int), s);
    P2W_CHECK_PTR(cell); P2W_CHECK_PTR(perm_out); P2W_CHECK_PTR(ws); P2W_CHECK_ALIGN16(ws);
    if (n < 0) return P2W_EINVAL;
    VsLayout L;
    hipError_t e = vs_layout(n, &L);
    if (e != hipSuccess) return (int32_t)e;
    if (ws_bytes < L.total) return P2W_EWORKSPACE;
    char* w = static_cast<char*>(ws);
    vs_iota_kernel<<<p2w_cdiv(n, 256), 256, 0, s>>>(n, reinterpret_cast<int*>(w + L.vals_in));
    // non-negative int64 cell ids order like their unsigned bit patterns
    return vs_cluster(w, L, reinterpret_cast<const unsigned long long*>(cell), nullptr, 0, nullptr, n, perm_out, nullptr,
                      nullptr, inv_out, count_out, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, s);
}

// ------------------------------------------------------------------------------------------------
// Table sampler: the same outputs as p2w_voxel_sample WITHOUT a sort, for batches whose cell grid is small enough to be
// held as a direct table (the per-voxel forward: B voxels of a few metres at 4 / 8 / 16 cm cells).  A cell's key
// ((b - b_lo) * c2 + z) * c1 + y) * c0 + x IS its table index, so
//   insert : tab_max[key] = max point index (atomicMax: the representative the reference's scatter leaves), cnt[key]++
//   scan   : exclusive scan of the occupancy flags (rank of every cell = its position in the output level) and of the
//            counts (start of every cell in the cell-sorted order), hand-written two-level scan
//   compact: representatives, batch, keys, CSR in ascending key order;  points: rank of every point's cell, and - when
//            the sorted order is asked for - a counting-sort scatter (order inside a cell is unspecified; every consumer
//            orders candidates by their carried index, never by storage position)
// Whether the table fits is only known on the device (the grid's extent is data): the kernels then set *status_out = 1
// and hand out an EMPTY level (ptr_out = 0: work already queued behind the call finds nothing to do); the caller reads the
// status with the level sizes it fetches anyway and repeats the level with p2w_voxel_sample.
// ------------------------------------------------------------------------------------------------
constexpr int TK_ITEMS = 4, TK_BLOCK = 1024, TK_TILE = TK_ITEMS * TK_BLOCK;

struct TkGeom { long long c0, c1, c2, cells, T; int b_lo; float lo0, lo1, lo2; };
__device__ __forceinline__ TkGeom tk_geom(const VsHeader* h, const int* ptr, int B, float res) {
    TkGeom g;
    g.b_lo = 0;
    while (g.b_lo < B - 1 && ptr[g.b_lo + 1] == ptr[g.b_lo]) ++g.b_lo;
    int b_hi = B - 1;
    while (b_hi > g.b_lo && ptr[b_hi + 1] == ptr[b_hi]) --b_hi;
    g.lo0 = ord2f(h->lo[0]); g.lo1 = ord2f(h->lo[1]); g.lo2 = ord2f(h->lo[2]);
    g.c0 = (long long)((ord2f(h->hi[0]) - g.lo0) / res) + 1;
    g.c1 = (long long)((ord2f(h->hi[1]) - g.lo1) / res) + 1;
    g.c2 = (long long)((ord2f(h->hi[2]) - g.lo2) / res) + 1;
    g.cells = g.c0 * g.c1 * g.c2;
    g.T = (g.c0 > (1 << 20) || g.c1 > (1 << 20) || g.c2 > (1 << 20)) ? (1ll << 62) : g.cells * (long long)(b_hi - g.b_lo + 1);
    return g;
}

// The table's cost follows the batch's ACTUAL grid, not the capacity the caller provisioned: only the cells the grid uses
// (rounded up to whole scan tiles) are cleared, scanned and compacted; blocks beyond them leave at once.
__device__ __forceinline__ long long tk_used(const TkGeom& g, long long T_cap) {
    if (g.T > T_cap) return 0;                                  // does not fit: nothing is inserted (status), nothing to prepare
    const long long up = (g.T + TK_TILE - 1) / TK_TILE * TK_TILE;
    return up < T_cap ? up : T_cap;
}
__global__ __launch_bounds__(256) void tk_clear_kernel(const int* __restrict__ ptr, int B, float res, const VsHeader* __restrict__ h,
                                                       long long T_cap, int* __restrict__ tab_max, int* __restrict__ cnt,
                                                       int* __restrict__ fill) {
    const long long t0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (ptr[B] == 0) return;
    const long long lim = tk_used(tk_geom(h, ptr, B, res), T_cap);
    if (t0 >= lim) return;
    if (t0 + 4 <= lim) {   // T_cap is padded to 4 entries by the layout; lim is a multiple of TK_TILE or T_cap itself
        *reinterpret_cast<int4*>(tab_max + t0) = make_int4(-1, -1, -1, -1);
        if (cnt) { *reinterpret_cast<int4*>(cnt + t0) = make_int4(0, 0, 0, 0); *reinterpret_cast<int4*>(fill + t0) = make_int4(0, 0, 0, 0); }
    } else {
        for (long long t = t0; t < lim; ++t) { tab_max[t] = -1; if (cnt) { cnt[t] = 0; fill[t] = 0; } }
    }
}

__global__ __launch_bounds__(256) void tk_insert_kernel(const float4* __restrict__ xyzr, const int* __restrict__ ptr, int B,
                                                        int n_bound, float res, const VsHeader* __restrict__ h, long long T_cap,
                                                        int* __restrict__ tab_max, int* __restrict__ cnt, int* __restrict__ key32,
                                                        int* __restrict__ status) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = ptr[B];
    if (n == 0) return;
    const TkGeom g = tk_geom(h, ptr, B, res);
    if (g.T > T_cap) {
        if (i == 0) *status = 1;
        return;
    }
    if (i >= n || i >= n_bound) return;
    const float4 p = xyzr[i];
    const int b = p2w_find_segment(ptr, B, i);
    const long long k0 = (long long)((p.x - g.lo0) / res);     // the arithmetic of vs_keys_kernel (oracle/ops.py voxel_grid)
    const long long k1 = (long long)((p.y - g.lo1) / res);
    const long long k2 = (long long)((p.z - g.lo2) / res);
    const long long kb = (long long)(((float)b - (float)g.b_lo) / 1.0f);
    const long long tl = k0 + k1 * g.c0 + k2 * (g.c0 * g.c1) + kb * g.cells;
    // a non-finite coordinate (fminf / fmaxf of the bounding-box pass ignore NaN; the float -> int64 cast of one is undefined)
    // must not become a table index: report it like a grid that does not fit - the caller repeats the level with the sort
    // sampler, which only ever forms a (garbage) key from it
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) || tl < 0 || tl >= g.T || k0 < 0 || k0 >= g.c0 || k1 < 0 || k1 >= g.c1 || k2 < 0 || k2 >= g.c2) {
        *status = 1;
        key32[i] = 0;
        return;
    }
    const int t = (int)tl;
    key32[i] = t;
    atomicMax(&tab_max[t], i);
    if (cnt) atomicAdd(&cnt[t], 1);
}

// exclusive scan of `nblk` block totals (occupancy and counts), in place, by ONE workgroup of 1024 threads; *total_out = the
// occupancy total.  The totals were written by other workgroups of the same launch: read at agent scope.
__device__ __forceinline__ void tk_scan_totals(int* __restrict__ bs_occ, int* __restrict__ bs_cnt, int nblk, int* __restrict__ total_out) {
    __shared__ int wsum2[2][16];
    __shared__ int carry[2];
    if (threadIdx.x == 0) { carry[0] = 0; carry[1] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        const int vf = i < nblk ? __hip_atomic_load(&bs_occ[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int vc = i < nblk ? __hip_atomic_load(&bs_cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int xf = vf, xc = vc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int of = __shfl_up(xf, d), oc = __shfl_up(xc, d);
            if (lane >= d) { xf += of; xc += oc; }
        }
        if (lane == 63) { wsum2[0][wave] = xf; wsum2[1][wave] = xc; }
        __syncthreads();
        int bf = carry[0], bc = carry[1], tf = 0, tc = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) { bf += wsum2[0][w]; bc += wsum2[1][w]; }
            tf += wsum2[0][w]; tc += wsum2[1][w];
        }
        if (i < nblk) { bs_occ[i] = bf + xf - vf; bs_cnt[i] = bc + xc - vc; }
        __syncthreads();
        if (threadIdx.x == 0) { carry[0] += tf; carry[1] += tc; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry[0];
}

// block-local exclusive scans of the occupancy flags (-> rank) and, optionally, of the counts (-> off); block totals.  The LAST
// workgroup to finish (a counter in the workspace: zero between calls, reset by that workgroup) scans the block totals in place -
// what a second, one-workgroup launch used to do.
__global__ __launch_bounds__(TK_BLOCK) void tk_scan1_kernel(const int* __restrict__ tab_max, const int* __restrict__ cnt, long long T_cap,
                                                           int* __restrict__ rank, int* __restrict__ off, int* __restrict__ bs_occ,
                                                           int* __restrict__ bs_cnt, const int* __restrict__ status,
                                                           const int* __restrict__ ptr, int B, float res, VsHeader* __restrict__ h,
                                                           int* __restrict__ done, int* __restrict__ total_out, VsHeader* __restrict__ h2) {
    __shared__ int wsum[2][TK_BLOCK / 64];
    __shared__ int is_last;
    if (*status || ptr[B] == 0) return;
    const long long used = tk_used(tk_geom(h, ptr, B, res), T_cap);
    if ((long long)blockIdx.x * TK_TILE >= used) return;   // tiles the grid does not reach
    const long long t0 = (long long)blockIdx.x * TK_TILE + (long long)threadIdx.x * TK_ITEMS;
    int f[TK_ITEMS], c[TK_ITEMS], sf = 0, sc = 0;
#pragma unroll
    for (int e = 0; e < TK_ITEMS; ++e) {
        const long long t = t0 + e;
        f[e] = (t < T_cap && tab_max[t] >= 0) ? 1 : 0;
        c[e] = (cnt && t < T_cap) ? cnt[t] : 0;
        sf += f[e]; sc += c[e];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int xf = sf, xc = sc;   // inclusive wave scans
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int of = __shfl_up(xf, d), oc = __shfl_up(xc, d);
        if (lane >= d) { xf += of; xc += oc; }
    }
    if (lane == 63) { wsum[0][wave] = xf; wsum[1][wave] = xc; }
    __syncthreads();
    int bf = 0, bc = 0, tf = 0, tc = 0;
    for (int w = 0; w < TK_BLOCK / 64; ++w) {
        if (w < wave) { bf += wsum[0][w]; bc += wsum[1][w]; }
        tf += wsum[0][w]; tc += wsum[1][w];
    }
    int rf = bf + xf - sf, rc = bc + xc - sc;   // exclusive prefix of this thread's first item within the block
#pragma unroll
    for (int e = 0; e < TK_ITEMS; ++e) {
        const long long t = t0 + e;
        if (t < T_cap) { rank[t] = rf; if (cnt) off[t] = rc; }
        rf += f[e]; rc += c[e];
    }
    const int nb = (int)((used + TK_TILE - 1) / TK_TILE);   // workgroups that take part
    if (threadIdx.x == 0) {
        __hip_atomic_store(&bs_occ[blockIdx.x], tf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&bs_cnt[blockIdx.x], tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        is_last = (atomicAdd(done, 1) == nb - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    tk_scan_totals(bs_occ, bs_cnt, nb, total_out);
    if (threadIdx.x == 0) {
        *done = 0;   // zero again for the next call on this workspace
        // every workgroup of this launch has read the bounding box: hand the call's last kernel a copy and give the next call's
        // bounding-box pass its atomics' identities back (the workspace's between-calls state)
        *h2 = *h;
        for (int d = 0; d < 3; ++d) { h->lo[d] = 0xffffffffu; h->hi[d] = 0u; }
    }
}

__device__ __forceinline__ void tk_compact_body(long long t, const int* __restrict__ tab_max, const int* __restrict__ rank,
                                                const int* __restrict__ bs_occ, const int* __restrict__ total, long long T_cap,
                                                const int* __restrict__ ptr, int B, float res, const VsHeader* __restrict__ h,
                                                int* __restrict__ idx_out, int* __restrict__ ptr_out, int* __restrict__ batch_out,
                                                unsigned long long* __restrict__ cell_keys_out, p2w_grid* __restrict__ grid_out,
                                                const int* __restrict__ status, const int* __restrict__ off,
                                                const int* __restrict__ bs_cnt, int* __restrict__ cell_start_out,
                                                int* __restrict__ cell_start_sorted_out) {
    const int n = ptr[B];
    if (*status || n == 0) {   // overflow: hand out an EMPTY level, so that whatever the caller has already queued behind this
        if (t <= B) ptr_out[t] = 0;   // call (searches, the next level's sampling) sees zero points instead of undefined sizes
        if (t == 0 && grid_out) {     // ... and a well-formed one-cell grid (searches of a valid level INTO this one read it)
            p2w_grid gg;
            for (int d = 0; d < 3; ++d) { gg.lo[d] = 0.f; gg.hi[d] = 0.f; gg.dims[d] = 1; }
            gg.res = res; gg.b_lo = 0;
            *grid_out = gg;
        }
        return;
    }
    const TkGeom g = tk_geom(h, ptr, B, res);
    if (t == 0 && grid_out) {
        p2w_grid gg;
        gg.lo[0] = g.lo0; gg.lo[1] = g.lo1; gg.lo[2] = g.lo2;
        for (int d = 0; d < 3; ++d) gg.hi[d] = ord2f(h->hi[d]);
        gg.dims[0] = g.c0; gg.dims[1] = g.c1; gg.dims[2] = g.c2;
        gg.res = res; gg.b_lo = g.b_lo;
        *grid_out = gg;
    }
    if (t <= B) {   // CSR of the sampled level: cells of the voxels before b
        const long long first = ((long long)t - g.b_lo) * g.cells;
        ptr_out[t] = (t <= g.b_lo) ? 0 : (first < g.T ? rank[first] + bs_occ[first / TK_TILE] : *total);
    }
    // cell -> first element at or after it: in the sampled level (one point per occupied cell, ascending key) and in the
    // cell-sorted order of the input points; entry g.T = the totals.  The grid searches look runs up here instead of
    // bisecting the keys (p2w_knn_grid_indexed).
    if (t == g.T && t <= T_cap) {
        if (cell_start_out) cell_start_out[t] = *total;
        if (cell_start_sorted_out) cell_start_sorted_out[t] = n;
    }
    if (t >= g.T || t >= T_cap) return;
    const int o = rank[t] + bs_occ[t / TK_TILE];
    if (cell_start_out) cell_start_out[t] = o;
    if (cell_start_sorted_out) cell_start_sorted_out[t] = off[t] + bs_cnt[t / TK_TILE];
    const int rep = tab_max[t];
    if (rep < 0) return;
    idx_out[o] = rep;
    if (batch_out) batch_out[o] = g.b_lo + (int)(t / g.cells);
    if (cell_keys_out) cell_keys_out[o] = (unsigned long long)t;
}

__device__ __forceinline__ void tk_points_body(int i, const int* __restrict__ key32, const int* __restrict__ rank, const int* __restrict__ bs_occ,
                                               const int* __restrict__ off, const int* __restrict__ bs_cnt, int* __restrict__ fill,
                                               const int* __restrict__ ptr, int B, int n_bound, int* __restrict__ inv_out,
                                               int* __restrict__ order_out, unsigned long long* __restrict__ sorted_keys_out,
                                               int* __restrict__ rank_sorted_out, const int* __restrict__ status) {
    if (i >= ptr[B] || i >= n_bound) return;
    if (*status) {   // overflow: harmless per-point outputs (rank 0, identity order) for whatever is already queued behind us
        if (inv_out) inv_out[i] = 0;
        if (order_out) order_out[i] = i;
        if (sorted_keys_out) sorted_keys_out[i] = 0ull;
        if (rank_sorted_out) rank_sorted_out[i] = 0;
        return;
    }
    const int t = key32[i];
    const int r = rank[t] + bs_occ[t / TK_TILE];
    if (inv_out) inv_out[i] = r;
    if (order_out) {
        const int pos = off[t] + bs_cnt[t / TK_TILE] + atomicAdd(&fill[t], 1);
        order_out[pos] = i;
        if (sorted_keys_out) sorted_keys_out[pos] = (unsigned long long)t;
        if (rank_sorted_out) rank_sorted_out[pos] = r;
    }
}

// compaction (one thread per table cell) and the per-point outputs (one thread per input point) only READ the scan's results:
// one launch does both.  The grid geometry comes from the COPY of the bounding box the scan's last workgroup left (`h2`): the
// primary one already carries the next call's identities.  Where the scan did not run (status set, empty batch) nothing here
// reads a bounding box, and thread 0 restores the primary one.
__global__ __launch_bounds__(256) void tk_finish_kernel(const int* __restrict__ tab_max, const int* __restrict__ rank,
                                                        const int* __restrict__ bs_occ, const int* __restrict__ total, long long T_cap,
                                                        const int* __restrict__ ptr, int B, float res, const VsHeader* __restrict__ h2,
                                                        int* __restrict__ idx_out, int* __restrict__ ptr_out, int* __restrict__ batch_out,
                                                        unsigned long long* __restrict__ cell_keys_out, p2w_grid* __restrict__ grid_out,
                                                        const int* __restrict__ status, const int* __restrict__ off,
                                                        const int* __restrict__ bs_cnt, int* __restrict__ cell_start_out,
                                                        int* __restrict__ cell_start_sorted_out, const int* __restrict__ key32,
                                                        int* __restrict__ fill, int n_bound, int* __restrict__ inv_out,
                                                        int* __restrict__ order_out, unsigned long long* __restrict__ sorted_keys_out,
                                                        int* __restrict__ rank_sorted_out, long long cgrid, VsHeader* __restrict__ h) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t == 0 && (*status || ptr[B] == 0)) {
        for (int d = 0; d < 3; ++d) { h->lo[d] = 0xffffffffu; h->hi[d] = 0u; }
    }
    if (t < cgrid)
        tk_compact_body(t, tab_max, rank, bs_occ, total, T_cap, ptr, B, res, h2, idx_out, ptr_out, batch_out, cell_keys_out, grid_out, status, off,
                        bs_cnt, cell_start_out, cell_start_sorted_out);
    if (t < n_bound && (inv_out || order_out))
        tk_points_body((int)t, key32, rank, bs_occ, off, bs_cnt, fill, ptr, B, n_bound, inv_out, order_out, sorted_keys_out, rank_sorted_out, status);
}

struct TkLayout { size_t hdr, key32, bs_occ, bs_cnt, total, tab_max, rank, cnt, off, fill, bytes; long long T_cap; int nblk; };
static void tk_layout(int n_bound, long long T_cap, TkLayout* L) {
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    L->T_cap = T_cap;
    L->nblk = (int)((T_cap + TK_TILE - 1) / TK_TILE);
    size_t o = 0;
    L->hdr = o; o += up(sizeof(VsHeader));
    L->total = o; o += 256;
    L->key32 = o; o += up(sizeof(int) * (size_t)n_bound);
    L->bs_occ = o; o += up(sizeof(int) * (size_t)L->nblk);
    L->bs_cnt = o; o += up(sizeof(int) * (size_t)L->nblk);
    L->tab_max = o; o += up(sizeof(int) * (size_t)T_cap);
    L->cnt = o; o += up(sizeof(int) * (size_t)T_cap);
    L->fill = o; o += up(sizeof(int) * (size_t)T_cap);
    L->rank = o; o += up(sizeof(int) * (size_t)T_cap);
    L->off = o; o += up(sizeof(int) * (size_t)T_cap);
    L->bytes = o;
}

extern "C" size_t p2w_voxel_sample_table_ws_bytes(int32_t n_bound, int64_t table_cells) {
    if (n_bound < 0 || table_cells <= 0 || table_cells > ((int64_t)1 << 30)) return 0;
    TkLayout L;
    tk_layout(n_bound > 0 ? n_bound : 1, table_cells, &L);
    return L.bytes;
}

// Workspace state between calls (p2w_voxel_sample_table_prepared): bounding-box words at their atomics' identities, the scan's
// completion counter at zero.  p2w_voxel_sample_table_prepare() establishes it on a fresh workspace, every call restores it.
extern "C" int32_t p2w_voxel_sample_table_prepare(void* ws, size_t ws_bytes, p2w_stream_t stream) {
    P2W_CHECK_PTR(ws); P2W_CHECK_ALIGN16(ws);
    TkLayout L;
    tk_layout(1, 1, &L);
    if (ws_bytes < L.total + 256) return P2W_EWORKSPACE;
    char* w = static_cast<char*>(ws);
    hipError_t e = hipMemsetAsync(w + L.total, 0, 256, p2w_s(stream));
    if (e != hipSuccess) return (int32_t)e;
    vs_init_kernel<<<1, 64, 0, p2w_s(stream)>>>(reinterpret_cast<VsHeader*>(w + L.hdr));
    return P2W_LAUNCH_STATUS();
}

static int32_t voxel_sample_table_impl(bool prepared, const float* xyzr, const int32_t* ptr, int32_t B, int32_t n_bound, float res,
                                       int32_t* idx_out, int32_t* ptr_out, int32_t* batch_out, int32_t* order_out,
                                       uint64_t* sorted_keys_out, uint64_t* cell_keys_out, p2w_grid* grid_out,
                                       int32_t* inv_out, int32_t* rank_sorted_out, int32_t* cell_start_out,
                                       int32_t* cell_start_sorted_out, int32_t* status_out, int64_t table_cells,
                                       void* ws, size_t ws_bytes, p2w_stream_t stream) {
    P2W_CHECK_PTR(ptr); P2W_CHECK_PTR(ptr_out); P2W_CHECK_PTR(status_out);
    if (cell_start_sorted_out && !order_out) return P2W_ENULL;   // the sorted order's counts are only kept when it is asked for
    if (B <= 0 || n_bound < 0 || !(res > 0.0f) || table_cells <= 0 || table_cells > ((int64_t)1 << 30)) return P2W_EINVAL;
    hipStream_t s = p2w_s(stream);
    if (n_bound == 0) {
        hipError_t e = hipMemsetAsync(status_out, 0, sizeof(int), s);
        if (e != hipSuccess) return (int32_t)e;
        return (int32_t)hipMemsetAsync(ptr_out, 0, sizeof(int) * (B + 1), s);
    }
    P2W_CHECK_PTR(xyzr); P2W_CHECK_PTR(idx_out); P2W_CHECK_PTR(batch_out); P2W_CHECK_PTR(ws);
    P2W_CHECK_ALIGN16(xyzr); P2W_CHECK_ALIGN16(ws);
    if ((sorted_keys_out || rank_sorted_out) && !order_out) return P2W_ENULL;
    TkLayout L;
    tk_layout(n_bound, table_cells, &L);
    if (ws_bytes < L.bytes) return P2W_EWORKSPACE;
    if (!prepared) {
        const int32_t rc = p2w_voxel_sample_table_prepare(ws, ws_bytes, stream);
        if (rc != P2W_OK) return rc;
    }
    char* w = static_cast<char*>(ws);
    auto* hdr = reinterpret_cast<VsHeader*>(w + L.hdr);
    int* tab_max = reinterpret_cast<int*>(w + L.tab_max);
    int* cnt = order_out ? reinterpret_cast<int*>(w + L.cnt) : nullptr;
    int* fill = reinterpret_cast<int*>(w + L.fill);
    int* rank = reinterpret_cast<int*>(w + L.rank);
    int* off = reinterpret_cast<int*>(w + L.off);
    int* key32 = reinterpret_cast<int*>(w + L.key32);
    int* bs_occ = reinterpret_cast<int*>(w + L.bs_occ);
    int* bs_cnt = reinterpret_cast<int*>(w + L.bs_cnt);
    int* total = reinterpret_cast<int*>(w + L.total);
    int* done = total + 16;                                  // the scan's completion counter (zero between calls)
    const auto* x4 = reinterpret_cast<const float4*>(xyzr);
    const int nblk_pts = p2w_cdiv(n_bound, 256);
    // 5 launches: bounding box (+ status = 0) -> clear -> insert -> scan (its last workgroup scans the block totals and restores the
    // workspace's between-calls state) -> compact + points
    vs_minmax_kernel<<<nblk_pts < 256 ? nblk_pts : 256, 256, 0, s>>>(x4, ptr, B, hdr, status_out);
    // -1 = empty cell, zero counts: only over the part of the table this batch's grid uses (known on the device)
    tk_clear_kernel<<<p2w_cdiv(table_cells, 1024), 256, 0, s>>>(ptr, B, res, hdr, (long long)table_cells, tab_max, cnt, fill);
    tk_insert_kernel<<<nblk_pts, 256, 0, s>>>(x4, ptr, B, n_bound, res, hdr, (long long)table_cells, tab_max, cnt, key32, status_out);
    auto* hdr2 = reinterpret_cast<VsHeader*>(total + 32);    // the bounding box as this call measured it (the primary is reset by the scan)
    tk_scan1_kernel<<<L.nblk, TK_BLOCK, 0, s>>>(tab_max, cnt, (long long)table_cells, rank, off, bs_occ, bs_cnt, status_out, ptr, B, res, hdr,
                                                 done, total, hdr2);
    const long long cgrid = (table_cells + 1 > B + 1 ? table_cells + 1 : B + 1);
    const long long fgrid = cgrid > n_bound ? cgrid : n_bound;
    tk_finish_kernel<<<p2w_cdiv(fgrid, 256), 256, 0, s>>>(tab_max, rank, bs_occ, total, (long long)table_cells, ptr, B, res, hdr2, idx_out, ptr_out,
                                                           batch_out, reinterpret_cast<unsigned long long*>(cell_keys_out), grid_out, status_out, off,
                                                           bs_cnt, cell_start_out, cell_start_sorted_out, key32, fill, n_bound, inv_out, order_out,
                                                           reinterpret_cast<unsigned long long*>(sorted_keys_out), rank_sorted_out, cgrid, hdr);
    return P2W_LAUNCH_STATUS();
}

extern "C" int32_t p2w_voxel_sample_table(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n_bound, float res,
                                          int32_t* idx_out, int32_t* ptr_out, int32_t* batch_out, int32_t* order_out,
                                          uint64_t* sorted_keys_out, uint64_t* cell_keys_out, p2w_grid* grid_out,
                                          int32_t* inv_out, int32_t* rank_sorted_out, int32_t* cell_start_out,
                                          int32_t* cell_start_sorted_out, int32_t* status_out, int64_t table_cells,
                                          void* ws, size_t ws_bytes, p2w_stream_t stream) {
    return voxel_sample_table_impl(false, xyzr, ptr, B, n_bound, res, idx_out, ptr_out, batch_out, order_out, sorted_keys_out, cell_keys_out,
                                   grid_out, inv_out, rank_sorted_out, cell_start_out, cell_start_sorted_out, status_out, table_cells, ws,
                                   ws_bytes, stream);
}

extern "C" int32_t p2w_voxel_sample_table_prepared(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n_bound, float res,
                                                   int32_t* idx_out, int32_t* ptr_out, int32_t* batch_out, int32_t* order_out,
                                                   uint64_t* sorted_keys_out, uint64_t* cell_keys_out, p2w_grid* grid_out,
                                                   int32_t* inv_out, int32_t* rank_sorted_out, int32_t* cell_start_out,
                                                   int32_t* cell_start_sorted_out, int32_t* status_out, int64_t table_cells,
                                                   void* ws, size_t ws_bytes, p2w_stream_t stream) {
    return voxel_sample_table_impl(true, xyzr, ptr, B, n_bound, res, idx_out, ptr_out, batch_out, order_out, sorted_keys_out, cell_keys_out,
                                   grid_out, inv_out, rank_sorted_out, cell_start_out, cell_start_sorted_out, status_out, table_cells, ws,
                                   ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// level gather with the (p / sf) * sf round trip
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void level_gather_kernel(const float4* __restrict__ src, const int* __restrict__ idx,
                                                           const int* __restrict__ batch_dst, const int* __restrict__ ptr_dst,
                                                           int B, const float* __restrict__ sf, float4* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ptr_dst[B]) return;
    const float4 p = src[idx[i]];
    const float s = sf[batch_dst[i]];
    dst[i] = make_float4((p.x / s) * s, (p.y / s) * s, (p.z / s) * s, p.w);
}

// records in a given order, each carrying its source index in the 4th component (bit pattern of the int32)
__global__ __launch_bounds__(256) void index_records_kernel(const float4* __restrict__ src, const int* __restrict__ order,
                                                            const int* __restrict__ n_dev, float4* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= *n_dev) return;
    const int o = order[i];
    const float4 p = src[o];
    dst[i] = make_float4(p.x, p.y, p.z, __int_as_float(o));
}

extern "C" int32_t p2w_index_records(const float* xyzr, const int32_t* order, const int32_t* ptr, int32_t B, int32_t n_bound,
                                     float* out, p2w_stream_t stream) {
    if (n_bound == 0) return P2W_OK;
    P2W_CHECK_PTR(xyzr); P2W_CHECK_PTR(order); P2W_CHECK_PTR(ptr); P2W_CHECK_PTR(out);
    P2W_CHECK_ALIGN16(xyzr); P2W_CHECK_ALIGN16(out);
    if (n_bound < 0 || B <= 0) return P2W_EINVAL;
    index_records_kernel<<<p2w_cdiv(n_bound, 256), 256, 0, p2w_s(stream)>>>(reinterpret_cast<const float4*>(xyzr), order, ptr + B,
                                                                            reinterpret_cast<float4*>(out));
    return P2W_LAUNCH_STATUS();
}

extern "C" int32_t p2w_level_gather(const float* xyzr_src, const int32_t* idx, const int32_t* batch_dst, const int32_t* ptr_dst,
                                    int32_t B, int32_t m_bound, const float* sf, float* xyzr_dst, p2w_stream_t stream) {
    if (m_bound == 0) return P2W_OK;
    P2W_CHECK_PTR(xyzr_src); P2W_CHECK_PTR(idx); P2W_CHECK_PTR(batch_dst); P2W_CHECK_PTR(ptr_dst); P2W_CHECK_PTR(sf);
    P2W_CHECK_PTR(xyzr_dst); P2W_CHECK_ALIGN16(xyzr_src); P2W_CHECK_ALIGN16(xyzr_dst);
    if (m_bound < 0 || B <= 0) return P2W_EINVAL;
    level_gather_kernel<<<p2w_cdiv(m_bound, 256), 256, 0, p2w_s(stream)>>>(
        reinterpret_cast<const float4*>(xyzr_src), idx, batch_dst, ptr_dst, B, sf, reinterpret_cast<float4*>(xyzr_dst));
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// brute-force neighbour search
//
// Work decomposition: a workgroup (4 waves) owns a tile of QT = 4*QPW queries of ONE voxel; it streams
// that voxel's candidates HBM -> LDS in coalesced 16-byte records (TILE per stage); each wave keeps
// QPW queries' selection state in registers and visits 64 candidates per step (one per lane), so a
// candidate record is read from LDS once per QPW queries.  Query coordinates and thresholds are
// wave-uniform (SGPRs); admission is a ballot, so the common "nothing admitted" case costs
// 8 VALU + 1 compare + 1 scalar branch per 64 pairs and nothing diverges.
// ------------------------------------------------------------------------------------------------
constexpr int S_QPW = 8;             // queries per wave
constexpr int S_QT = 4 * S_QPW;      // queries per workgroup
constexpr int S_TILE = 1024;         // candidates per LDS stage (16 KiB)

// blockIdx -> (voxel b, first query q0, end q1).  Tiles never straddle voxels.
// Query tile of a workgroup.  Voxel b owns the virtual tiles [V(b), V(b + 1)), V(b) = ptr_q[b] / S_QT + b: at least
// ceil(m_b / S_QT) of them (floor(x + y) >= floor(x) + floor(y)), at most one more, V(B) <= the launch's
// ceil(m_bound / S_QT) + B workgroups, and V is strictly increasing and computable from ptr_q alone - so a workgroup finds its
// voxel by bisection.  (A running sum of the voxels' tile counts, the previous form, is a walk over all B voxels in every
// workgroup: invisible at B = 8, 4 x the search time at B = 1500.)
__device__ __forceinline__ bool search_tile(const int* __restrict__ ptr_q, int B, int tile, int* b_out, int* q0, int* q1) {
    if (tile >= ptr_q[B] / S_QT + B) return false;
    int lo = 0, hi = B;                         // V(lo) <= tile < V(hi)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr_q[mid] / S_QT + mid <= tile) lo = mid; else hi = mid;
    }
    const int s = ptr_q[lo], e = ptr_q[lo + 1];
    const int first = s + (tile - (s / S_QT + lo)) * S_QT;
    if (first >= e) return false;               // the voxel's spare virtual tile (or an empty voxel)
    *b_out = lo;
    *q0 = first;
    *q1 = min(first + S_QT, e);
    return true;
}

__device__ __forceinline__ float rdlane(float v, int l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }
// value of lane-1 (lane 0 keeps its own): v_mov_b32_dpp wave_shr:1
__device__ __forceinline__ float shr1(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }

// wave-uniform query coordinates: the index is built from SGPR values only, so these are scalar loads
struct UQuery { float x, y, z; bool valid; };
__device__ __forceinline__ UQuery load_query(const float4* __restrict__ xq, const int* __restrict__ qidx, int q, int q1) {
    UQuery u;
    u.valid = q < q1;
    const int src = u.valid ? (qidx ? qidx[q] : q) : 0;
    const float4 v = xq[src];
    u.x = v.x; u.y = v.y; u.z = v.z;
    return u;
}

__device__ __forceinline__ void stage_candidates(float4* cand, const float4* __restrict__ x, int base, int c1, int tid) {
#pragma unroll
    for (int r = 0; r < S_TILE / 256; ++r) {
        const int c = base + tid + 256 * r;
        cand[tid + 256 * r] = (c < c1) ? x[c] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
    }
}

// Staging permutes the records of a tile (slot s holds candidate base + ((s * 389) & 1023)) so that every
// 64-slot chunk samples the whole tile: levels >= 1 are stored in grid-cell order, and a scan in storage order
// approaches each query monotonically, which makes almost every candidate a new admission.  The candidate's
// index travels in the record's 4th component; admission is order-independent (lexicographic (d2, index)).
__device__ __forceinline__ void stage_shuffled(float4* cand, const float4* __restrict__ x, int base, int c1, int tid,
                                               bool index_in_w) {
#pragma unroll
    for (int r = 0; r < S_TILE / 256; ++r) {
        const int s = tid + 256 * r;
        const int c = base + ((s * 389) & (S_TILE - 1));
        float4 v = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(0x7fffffff));
        if (c < c1) { v = x[c]; if (!index_in_w) v.w = __int_as_float(c); }
        cand[s] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// bounding boxes of the candidate tiles (S_TILE consecutive records of one voxel): lo xyz, hi xyz per tile.
// Tile numbering: voxel b's tiles follow those of voxels < b (sum of ceil(n_b / S_TILE)).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool bbox_tile(const int* __restrict__ ptr, int B, int tile, int* c_lo, int* c_hi) {
    int acc = 0;
    for (int b = 0; b < B; ++b) {
        const int s = ptr[b], e = ptr[b + 1];
        const int t = (e - s + S_TILE - 1) / S_TILE;
        if (tile < acc + t) {
            *c_lo = s + (tile - acc) * S_TILE;
            *c_hi = min(*c_lo + S_TILE, e);
            return true;
        }
        acc += t;
    }
    return false;
}
__device__ __forceinline__ int bbox_tile_base(const int* __restrict__ ptr, int b) {
    int acc = 0;
    for (int i = 0; i < b; ++i) acc += (ptr[i + 1] - ptr[i] + S_TILE - 1) / S_TILE;
    return acc;
}

__global__ __launch_bounds__(256) void tile_bbox_kernel(const float4* __restrict__ x, const int* __restrict__ ptr, int B,
                                                        float* __restrict__ bbox) {
    __shared__ float red[4][6];
    int c_lo, c_hi;
    if (!bbox_tile(ptr, B, blockIdx.x, &c_lo, &c_hi)) return;
    float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int c = c_lo + threadIdx.x; c < c_hi; c += 256) {
        const float4 p = x[c];
        v[0] = fminf(v[0], p.x); v[3] = fmaxf(v[3], p.x);
        v[1] = fminf(v[1], p.y); v[4] = fmaxf(v[4], p.y);
        v[2] = fminf(v[2], p.z); v[5] = fmaxf(v[5], p.z);
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float o = __shfl_xor(v[d], off);
            v[d] = d < 3 ? fminf(v[d], o) : fmaxf(v[d], o);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 6; ++d) red[wave][d] = v[d];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        float r = red[0][d];
        for (int w = 1; w < 4; ++w) r = d < 3 ? fminf(r, red[w][d]) : fmaxf(r, red[w][d]);
        bbox[(size_t)blockIdx.x * 6 + d] = r;
    }
}

extern "C" int32_t p2w_tile_bbox(const float* xyzr, const int32_t* ptr, int32_t B, int32_t n_bound, float* bbox,
                                 p2w_stream_t stream) {
    if (n_bound == 0) return P2W_OK;
    P2W_CHECK_PTR(xyzr); P2W_CHECK_PTR(ptr); P2W_CHECK_PTR(bbox); P2W_CHECK_ALIGN16(xyzr);
    if (B <= 0 || n_bound < 0) return P2W_EINVAL;
    tile_bbox_kernel<<<p2w_cdiv(n_bound, S_TILE) + B, 256, 0, p2w_s(stream)>>>(reinterpret_cast<const float4*>(xyzr), ptr, B, bbox);
    return P2W_LAUNCH_STATUS();
}
extern "C" int32_t p2w_tile_bbox_count(int32_t B, int32_t n_bound) { return p2w_cdiv(n_bound, S_TILE) + B; }

// start tile of a workgroup whose queries are not candidates: the tile whose box is nearest to the first query (box gap
// first, then distance to the box centre).  Only the visiting order depends on this, never a result.
__device__ __forceinline__ int nearest_tile(const float* __restrict__ vb, int ntiles, const float4 f) {
    float best_lb = INFINITY, best_c = INFINITY;
    int t0 = 0;
    for (int t = 0; t < ntiles; ++t) {
        const float lx = vb[t * 6 + 0], ly = vb[t * 6 + 1], lz = vb[t * 6 + 2];
        const float hx = vb[t * 6 + 3], hy = vb[t * 6 + 4], hz = vb[t * 6 + 5];
        const float ex = fmaxf(fmaxf(lx - f.x, f.x - hx), 0.f), ey = fmaxf(fmaxf(ly - f.y, f.y - hy), 0.f);
        const float ez = fmaxf(fmaxf(lz - f.z, f.z - hz), 0.f);
        const float lb = ex * ex + ey * ey + ez * ez;
        const float cx = 0.5f * (lx + hx) - f.x, cy = 0.5f * (ly + hy) - f.y, cz = 0.5f * (lz + hz) - f.z;
        const float cd = cx * cx + cy * cy + cz * cz;
        if (lb < best_lb || (lb == best_lb && cd < best_c)) { best_lb = lb; best_c = cd; t0 = t; }
    }
    return t0;
}

__global__ __launch_bounds__(256) void knn_kernel(const float4* __restrict__ x, const int* __restrict__ ptr_x,
                                                  const float4* __restrict__ xq, const int* __restrict__ qidx,
                                                  const int* __restrict__ ptr_q, int B, int k, int* __restrict__ nbr,
                                                  int* __restrict__ deg, const float* __restrict__ bbox,
                                                  int qidx_is_candidate, int flags) {
    __shared__ float4 cand[S_TILE];
    __shared__ int need[2][4];
    int b, q0, q1;
    if (!search_tile(ptr_q, B, blockIdx.x, &b, &q0, &q1)) return;
    const int c0 = ptr_x[b], c1 = ptr_x[b + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = q0 + wave * S_QPW;
    UQuery uq[S_QPW];
    // selection state: lane l of (best_d, best_i)[j] = the l-th smallest (d2, index) so far of query j;
    // empty slots and lanes >= k hold (+inf, INT_MAX); thr = wave-uniform copy of slot k-1's distance
    float best_d[S_QPW], thr[S_QPW];
    int best_i[S_QPW];
#pragma unroll
    for (int j = 0; j < S_QPW; ++j) {
        uq[j] = load_query(xq, qidx, qw + j, q1);
        best_d[j] = INFINITY; best_i[j] = 0x7fffffff;
        thr[j] = uq[j].valid ? INFINITY : -INFINITY;
    }
    const bool in_k = lane < k;
    // tile visiting order: start where the block's first query most likely has its neighbours and move outwards
    const int ntiles = (c1 - c0 + S_TILE - 1) / S_TILE;
    int t0;
    if (qidx) t0 = (qidx[q0] - c0) / S_TILE;                                    // the query is itself a candidate
    else t0 = (int)(((long long)(q0 - ptr_q[b]) * ntiles) / max(1, ptr_q[b + 1] - ptr_q[b]));  // same storage order
    t0 = min(max(t0, 0), max(ntiles - 1, 0));
    const float* vb = bbox ? bbox + (size_t)bbox_tile_base(ptr_x, b) * 6 : nullptr;
    if (vb && !qidx_is_candidate) t0 = nearest_tile(vb, ntiles, xq[qidx ? qidx[q0] : q0]);   // queries of another level
    int visit = 0;
    for (int step = 0; step < 2 * ntiles - 1; ++step) {
        const int off = (step + 1) >> 1;
        const int tile = (step & 1) ? t0 + off : t0 - off;
        if (tile < 0 || tile >= ntiles) continue;
        const int base = c0 + tile * S_TILE;
        // exact pruning: per dimension e = gap between the query and the tile's box; ((ex^2+ey^2)+ez^2) in fp32 is a
        // lower bound of p2w_d2(query, c) for every c in the tile (fp32 subtract / multiply / add are monotone), so a
        // tile whose bound exceeds a query's current k-th distance cannot change that query's result.
        unsigned qmask = (1u << S_QPW) - 1u;
        if (vb) {
            const float lx = vb[tile * 6 + 0], ly = vb[tile * 6 + 1], lz = vb[tile * 6 + 2];
            const float hx = vb[tile * 6 + 3], hy = vb[tile * 6 + 4], hz = vb[tile * 6 + 5];
            qmask = 0u;
#pragma unroll
            for (int j = 0; j < S_QPW; ++j) {
                const float ex = fmaxf(fmaxf(lx - uq[j].x, uq[j].x - hx), 0.f);
                const float ey = fmaxf(fmaxf(ly - uq[j].y, uq[j].y - hy), 0.f);
                const float ez = fmaxf(fmaxf(lz - uq[j].z, uq[j].z - hz), 0.f);
                const float lb = ((ex * ex) + (ey * ey)) + (ez * ez);
                if (lb <= thr[j]) qmask |= 1u << j;
            }
            if (lane == 0) need[visit & 1][wave] = qmask != 0u;
        }
        __syncthreads();
        if (vb) {
            const int any = need[visit & 1][0] | need[visit & 1][1] | need[visit & 1][2] | need[visit & 1][3];
            ++visit;
            if (!any) continue;   // no query of this workgroup can gain from the tile: skip staging it
        }
        stage_shuffled(cand, x, base, c1, tid, (flags & P2W_SEARCH_X_INDEX_IN_W) != 0);
        __syncthreads();
        // groups of 4 chunks (256 candidates) are held in registers while the wave walks its queries, so the
        // per-query state is in scalars inside the admission loop
        const int ngr = (min(S_TILE, c1 - base) + 255) >> 8;
        for (int gr = 0; gr < S_TILE / 256; ++gr) {   // all groups: the shuffle spreads a partial tile over every group
            float4 c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = cand[gr * 256 + u * 64 + lane];
#pragma unroll
            for (int j = 0; j < S_QPW; ++j) {
                if (!((qmask >> j) & 1u)) continue;   // wave-uniform
                float bd = best_d[j], t = thr[j];
                int bi = best_i[j];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d = p2w_d2(uq[j].x, uq[j].y, uq[j].z, c[u].x, c[u].y, c[u].z);
                    const int ci = __float_as_int(c[u].w);
                    unsigned long long m = __ballot(d <= t);
                    while (m) {  // wave-uniform loop over the candidates that may enter
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const float dn = rdlane(d, src);
                        const int in = __builtin_amdgcn_readlane(ci, src);
                        // (d2, index) pairs order like the 64-bit integers (bits(d2) << 32 | index): d2 >= +0, index >= 0.
                        // pos = number of kept pairs below the new one; it enters iff pos < k
                        const unsigned long long kn = ((unsigned long long)__float_as_uint(dn) << 32) | (unsigned)in;
                        const unsigned long long kb = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bi;
                        const int pos = __popcll(__ballot(kb < kn));
                        if (pos < k) {  // scalar branch
                            const float up_d = shr1(bd);
                            const int up_i = shr1(bi);
                            const bool here = lane == pos, sh = (lane > pos) & in_k;
                            bd = here ? dn : (sh ? up_d : bd);
                            bi = here ? in : (sh ? up_i : bi);
                            t = rdlane(bd, k - 1);
                        }
                    }
                }
                best_d[j] = bd; best_i[j] = bi; thr[j] = t;
            }
        }
        (void)ngr;
    }
    const int cnt = min(k, c1 - c0);
#pragma unroll
    for (int j = 0; j < S_QPW; ++j) {
        const int q = qw + j;
        if (q < q1) {
            const int row = (flags & P2W_SEARCH_Q_ROW_IN_W) ? __float_as_int(xq[qidx ? qidx[q] : q].w) : q;
            const int kept = min(cnt, __popcll(__ballot(best_i[j] != 0x7fffffff)));   // NaN query: nothing was admitted
            if (lane < k) nbr[(size_t)row * k + lane] = (lane < kept) ? best_i[j] : -1;
            if (lane == 0) deg[row] = kept;
        }
    }
}

// Ball query: the `cap` in-ball candidates (d2 < r2) with the smallest reported indices, ascending - which is
// torch-cluster's "first cap hits in index order" whatever order the candidates are stored or visited in.  Lane l of
// best_i[j] = the l-th smallest in-ball index so far of query j (INT_MAX = empty); thi = wave-uniform copy of slot
// cap-1.  With tile boxes, tiles farther than r from every query of the workgroup are never staged (exact: the box
// gap is a lower bound of p2w_d2 under fp32 rounding).
__global__ __launch_bounds__(256) void ball_kernel(const float4* __restrict__ x, const int* __restrict__ ptr_x,
                                                   const float4* __restrict__ xq, const int* __restrict__ qidx,
                                                   const int* __restrict__ ptr_q, int B, float r2, int cap,
                                                   int* __restrict__ nbr, int* __restrict__ deg,
                                                   const float* __restrict__ bbox, int flags) {
    __shared__ float4 cand[S_TILE];
    __shared__ int need[2][4];
    int b, q0, q1;
    if (!search_tile(ptr_q, B, blockIdx.x, &b, &q0, &q1)) return;
    const int c0 = ptr_x[b], c1 = ptr_x[b + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = q0 + wave * S_QPW;
    const bool index_in_w = (flags & P2W_SEARCH_X_INDEX_IN_W) != 0;
    UQuery uq[S_QPW];
    int best_i[S_QPW], thi[S_QPW], cnt[S_QPW];
#pragma unroll
    for (int j = 0; j < S_QPW; ++j) {
        uq[j] = load_query(xq, qidx, qw + j, q1);
        best_i[j] = 0x7fffffff;
        thi[j] = uq[j].valid ? 0x7fffffff : -1;   // -1: nothing is ever admitted for an unused slot
        cnt[j] = 0;
    }
    const bool in_cap = lane < cap;
    const int ntiles = (c1 - c0 + S_TILE - 1) / S_TILE;
    const float* vb = bbox ? bbox + (size_t)bbox_tile_base(ptr_x, b) * 6 : nullptr;
    const int t0 = vb ? nearest_tile(vb, ntiles, xq[qidx ? qidx[q0] : q0]) : 0;
    int visit = 0;
    for (int step = 0; step < 2 * ntiles - 1; ++step) {
        const int off = (step + 1) >> 1;
        const int tile = (step & 1) ? t0 + off : t0 - off;
        if (tile < 0 || tile >= ntiles) continue;
        const int base = c0 + tile * S_TILE;
        unsigned qmask = 0u;
#pragma unroll
        for (int j = 0; j < S_QPW; ++j) qmask |= uq[j].valid ? (1u << j) : 0u;
        if (vb) {
            const float lx = vb[tile * 6 + 0], ly = vb[tile * 6 + 1], lz = vb[tile * 6 + 2];
            const float hx = vb[tile * 6 + 3], hy = vb[tile * 6 + 4], hz = vb[tile * 6 + 5];
            unsigned near = 0u;
#pragma unroll
            for (int j = 0; j < S_QPW; ++j) {
                const float ex = fmaxf(fmaxf(lx - uq[j].x, uq[j].x - hx), 0.f);
                const float ey = fmaxf(fmaxf(ly - uq[j].y, uq[j].y - hy), 0.f);
                const float ez = fmaxf(fmaxf(lz - uq[j].z, uq[j].z - hz), 0.f);
                const float lb = ((ex * ex) + (ey * ey)) + (ez * ez);
                if (lb < r2) near |= 1u << j;
            }
            qmask &= near;
            if (lane == 0) need[visit & 1][wave] = qmask != 0u;
        }
        __syncthreads();
        if (vb) {
            const int any = need[visit & 1][0] | need[visit & 1][1] | need[visit & 1][2] | need[visit & 1][3];
            ++visit;
            if (!any) continue;
        }
#pragma unroll
        for (int r = 0; r < S_TILE / 256; ++r) {
            const int c = base + tid + 256 * r;
            float4 v = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(0x7fffffff));
            if (c < c1) { v = x[c]; if (!index_in_w) v.w = __int_as_float(c); }
            cand[tid + 256 * r] = v;
        }
        __syncthreads();
        const int nch = (min(S_TILE, c1 - base) + 63) >> 6;
        for (int ch = 0; ch < nch; ++ch) {
            const float4 c = cand[ch * 64 + lane];
            const int ci = __float_as_int(c.w);
#pragma unroll
            for (int j = 0; j < S_QPW; ++j) {
                if (!((qmask >> j) & 1u)) continue;   // wave-uniform
                const float d = p2w_d2(uq[j].x, uq[j].y, uq[j].z, c.x, c.y, c.z);
                const bool hit = d < r2;
                const unsigned long long mh = __ballot(hit);
                if (mh == 0ull) continue;
                cnt[j] += __popcll(mh);
                unsigned long long m = __ballot(hit && ci < thi[j]);
                int bi = best_i[j], ti = thi[j];
                while (m) {
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int in = __builtin_amdgcn_readlane(ci, src);
                    if (in < ti) {
                        const int pos = __popcll(__ballot(bi < in));
                        const int up_i = shr1(bi);
                        bi = (lane == pos) ? in : (((lane > pos) & in_cap) ? up_i : bi);
                        ti = __builtin_amdgcn_readlane(bi, cap - 1);
                    }
                }
                best_i[j] = bi; thi[j] = ti;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < S_QPW; ++j) {
        const int q = qw + j;
        if (q < q1) {
            const int row = (flags & P2W_SEARCH_Q_ROW_IN_W) ? __float_as_int(xq[qidx ? qidx[q] : q].w) : q;
            const int kept = min(cnt[j], cap);
            if (lane < cap) nbr[(size_t)row * cap + lane] = (lane < kept) ? best_i[j] : -1;
            if (lane == 0) deg[row] = kept;
        }
    }
}




// ------------------------------------------------------------------------------------------------
// grid-indexed neighbour search
//
// The candidates are stored in ascending cell-key order of a p2w_grid (the order p2w_voxel_sample produces), `keys`
// holds their keys.  A workgroup still owns 32 consecutive queries of one voxel, but instead of streaming the whole
// voxel it gathers only the rows of the grid within a radius rho of its queries' bounding box: for every z layer the
// rows [Ylo, Yhi] are one contiguous run of the storage order, found with two binary searches on the keys.  The runs
// are concatenated into LDS tiles and scanned exactly like the brute-force kernels do.
//   ball query: rho = r, one pass.
//   kNN       : rho starts from a local density estimate; after a pass every query checks that its k-th distance is
//               not larger than its distance to the nearest face of the gathered region (rows outside it can only hold
//               farther candidates); otherwise the region grows and only the NEW rows are scanned.
// Exactness: a candidate's key cell comes from fp32 arithmetic on (possibly one level older) coordinates, so faces
// are pulled in by eps = res/32 + 1e-5*max|coordinate| >> every rounding involved; a candidate that is not gathered
// is provably farther than the accepted k-th distance (or than r), so the result equals the brute-force one bit for bit.
// ------------------------------------------------------------------------------------------------
constexpr int G_MAXRUN = 256;        // runs per pass (2 per z layer)
#ifndef P2W_KNN_TILE
#define P2W_KNN_TILE 1792            // candidates per LDS stage of the k >= 8 searches (a multiple of 256; the typical k = 32 region holds ~1200).
                                     // 1792 and not 2048: 28 KiB + run tables fit BESIDE a 256 x 256 GEMM workgroup's 128 KiB on a CU (2048: 162 KiB
                                     // of 160) - searches run next to feature kernels; same search time, lone forward -0.6 %, pipelined -0.4 %
#endif

__device__ __forceinline__ int lower_bound_key(const unsigned long long* __restrict__ keys, int lo, int hi,
                                               unsigned long long key) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// cell index of coordinate v on one axis, clamped to [0, dim-1]
__device__ __forceinline__ int grid_cell(float v, float lo, float res, long long dim) {
    const float f = floorf((v - lo) / res);
    const float top = (float)(dim - 1 < (1ll << 30) ? dim - 1 : (1ll << 30));
    return (int)fminf(fmaxf(f, 0.f), top);   // NaN -> 0
}

#ifdef P2W_SLAB_PROFILE
__device__ unsigned long long g_slab_prof[16];
#define SLAB_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (tid == 0) atomicAdd(&g_slab_prof[i], now_ - stamp_); stamp_ = now_; } while (0)
#define SLAB_COUNT(i, v) do { if (tid == 0) atomicAdd(&g_slab_prof[i], (unsigned long long)(v)); } while (0)
extern "C" int32_t p2w_debug_slab_prof(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slab_prof), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) { unsigned long long z[16] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_slab_prof), z, sizeof(z)); }
    return (int32_t)e;
}
#else
#define SLAB_STAMP(i) do {} while (0)
#define SLAB_COUNT(i, v) do {} while (0)
#endif

// MODE: 0 = kNN, 1 = ball query; TILE: candidates per LDS stage; SEL (kNN): 0 = sorted insertion from a cold start (small k),
// 1 = counted threshold ladder + sorted insertion (k >= 8).  (Round 4 also carried SEL = 2 - candidates collected per query and
// merged by 64-lane sorting networks: exact, 15 % slower, removed in round 5; docs/LAB_NOTES.md.)
// BOX: the gathered region is also bounded in x (one run per grid row instead of one per z layer) - for grids whose
// rows are much longer than a workgroup's reach (plot-scale searches); per-voxel searches gather whole rows.
template <int MODE, int TILE, int SEL, bool BOX>
__global__ __launch_bounds__(256, 1) void slab_search_kernel(const float4* __restrict__ x, const unsigned long long* __restrict__ keys,
                                                          const int* __restrict__ ptr_x, const p2w_grid* __restrict__ grid,
                                                          const float4* __restrict__ xq, const int* __restrict__ qidx,
                                                          const int* __restrict__ ptr_q, int B, int k, float r, float r2,
                                                          int* __restrict__ nbr, int* __restrict__ deg, int flags,
                                                          const float* __restrict__ hint, const int* __restrict__ cell_start) {
    __shared__ float4 cand[TILE];
    __shared__ int run_start[G_MAXRUN];
    __shared__ int run_pre[G_MAXRUN + 1];
    __shared__ int wsum[4];
    __shared__ float wred[4][8];
    constexpr bool LADDER = SEL >= 1;
    int b, q0, q1;
    if (!search_tile(ptr_q, B, blockIdx.x, &b, &q0, &q1)) return;
    const int c0 = ptr_x[b], c1 = ptr_x[b + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = q0 + wave * S_QPW;
    const bool index_in_w = (flags & P2W_SEARCH_X_INDEX_IN_W) != 0;
#ifdef P2W_SLAB_PROFILE
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
    // grid geometry (wave-uniform)
    const float lo_x = grid->lo[0], lo_y = grid->lo[1], lo_z = grid->lo[2], res = grid->res;
    const long long g0 = grid->dims[0], g1 = grid->dims[1], g2 = grid->dims[2];
    const long long kb = (long long)((float)b - (float)grid->b_lo);
    float amax = 1.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) amax = fmaxf(amax, fmaxf(fabsf(grid->lo[d]), fabsf(grid->hi[d])));
    const float eps = res * (1.f / 32.f) + 1e-5f * amax;

    UQuery uq[S_QPW];
    float best_d[S_QPW], thr[S_QPW];
    int best_i[S_QPW], cnt[S_QPW];
    unsigned active = 0u;   // queries of this wave whose result is not final yet
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
    float hmax = 0.f, hinted = (MODE == 0 && hint) ? 1.f : 0.f;   // largest hint / every query of the workgroup has one
#pragma unroll
    for (int j = 0; j < S_QPW; ++j) {
        uq[j] = load_query(xq, qidx, qw + j, q1);
        best_d[j] = INFINITY; best_i[j] = 0x7fffffff; cnt[j] = 0;
        thr[j] = MODE == 0 ? INFINITY : __int_as_float(0x7fffffff);   // ball: thr holds the index threshold's bits
        if (uq[j].valid) {
            active |= 1u << j;
            if (MODE == 0 && hint) {   // caller's upper bound of this query's k-th distance (checked after the first pass)
                const float hq = hint[qw + j];
                if (hq < INFINITY) { thr[j] = hq; hmax = fmaxf(hmax, hq); } else hinted = 0.f;
            }
            if (BOX) { xmin = fminf(xmin, uq[j].x); xmax = fmaxf(xmax, uq[j].x); }
            ymin = fminf(ymin, uq[j].y); ymax = fmaxf(ymax, uq[j].y);
            zmin = fminf(zmin, uq[j].z); zmax = fmaxf(zmax, uq[j].z);
        }
    }
    if (lane == 0) {
        wred[wave][0] = ymin; wred[wave][1] = ymax; wred[wave][2] = zmin; wred[wave][3] = zmax;
        wred[wave][5] = xmin; wred[wave][6] = xmax;
        wred[wave][4] = hmax; wred[wave][7] = hinted;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        ymin = fminf(ymin, wred[w][0]); ymax = fmaxf(ymax, wred[w][1]);
        zmin = fminf(zmin, wred[w][2]); zmax = fmaxf(zmax, wred[w][3]);
        if (BOX) { xmin = fminf(xmin, wred[w][5]); xmax = fmaxf(xmax, wred[w][6]); }
        hmax = fmaxf(hmax, wred[w][4]); hinted = fminf(hinted, wred[w][7]);
    }
    __syncthreads();   // wred[.][4] is reused by the pass loop
    const bool in_k = lane < k;
    const int total = c1 - c0;
    SLAB_STAMP(0);   // setup
    // The region is the cell box "bounding box of the queries grown by rho" (all of x unless BOX).  region() moves the
    // target box n* (never shrinking below the scanned box o*) and returns the number of runs that are NEW relative to
    // the scanned box; build(rb) fills the LDS run table with runs rb .. rb+G_MAXRUN-1 of them and returns how many
    // candidates they hold.  Runs: !BOX: 2 per z layer (rows below / above the scanned rows, or all rows of a new
    // layer); BOX: 2 per (z, y) row (cells left / right of the scanned cells, or the whole x range of a new row).
    int oXlo = 0, oXhi = -1, oYlo = 0, oYhi = -1, oZlo = 0, oZhi = -1;   // scanned box (empty)
    int nXlo = 0, nXhi = -1, nYlo = 0, nYhi = -1, nZlo = 0, nZhi = -1;
    bool whole = false;                                                // the scanned region is the whole voxel
    auto region = [&](float rho, bool force_whole) -> int {
        nYlo = grid_cell(ymin - rho - eps, lo_y, res, g1); nYhi = grid_cell(ymax + rho + eps, lo_y, res, g1);
        nZlo = grid_cell(zmin - rho - eps, lo_z, res, g2); nZhi = grid_cell(zmax + rho + eps, lo_z, res, g2);
        if (BOX) { nXlo = grid_cell(xmin - rho - eps, lo_x, res, g0); nXhi = grid_cell(xmax + rho + eps, lo_x, res, g0); }
        else { nXlo = 0; nXhi = (int)(g0 - 1 < (1ll << 30) ? g0 - 1 : (1ll << 30)); }
        if (oYhi >= oYlo) {
            nXlo = min(nXlo, oXlo); nXhi = max(nXhi, oXhi); nYlo = min(nYlo, oYlo); nYhi = max(nYhi, oYhi);
            nZlo = min(nZlo, oZlo); nZhi = max(nZhi, oZhi);
        }
        const long long rows = BOX ? (long long)(nZhi - nZlo + 1) * (nYhi - nYlo + 1) : (long long)(nZhi - nZlo + 1);
        if (force_whole || rows > (BOX ? 32768 : G_MAXRUN / 2) || !(rho == rho)) {   // degenerate: everything, one run
            whole = true;
            return 1;
        }
        return (int)(2 * rows);
    };
    auto build = [&](int rb) -> int {
        __syncthreads();   // run table free
        if (whole) {
            if (tid == 0) { run_start[0] = c0; run_pre[0] = 0; }
            run_pre[tid + 1] = total;
            __syncthreads();
            return total;
        }
        const int ny = nYhi - nYlo + 1;
        const long long nrt = 2ll * (BOX ? (long long)(nZhi - nZlo + 1) * ny : (long long)(nZhi - nZlo + 1));
        const long long t = (long long)rb + tid;
        int len = 0;
        run_start[tid] = c0;
        if (t < nrt) {
            const int side = (int)(t & 1);
            const int row = (int)(t >> 1);
            const int z = nZlo + (BOX ? row / ny : row);
            const bool seen_z = oYhi >= oYlo && z >= oZlo && z <= oZhi;
            int ya, yb, xa = nXlo, xb = nXhi;
            if (BOX) {
                const int y = nYlo + row % ny;
                ya = yb = y;
                if (seen_z && y >= oYlo && y <= oYhi) { xa = side ? oXhi + 1 : nXlo; xb = side ? nXhi : oXlo - 1; }
                else if (side) xb = xa - 1;
            } else {
                if (seen_z) { ya = side ? oYhi + 1 : nYlo; yb = side ? nYhi : oYlo - 1; }
                else { ya = nYlo; yb = side ? nYlo - 1 : nYhi; }
            }
            if (ya <= yb && xa <= xb) {
                const long long rowbase = (kb * g2 + z) * g1;
                const unsigned long long ka = (unsigned long long)((rowbase + ya) * g0 + xa);
                const unsigned long long kz = (unsigned long long)((rowbase + yb) * g0 + xb + 1);
                // first candidate at or after a key: one load from the sampler's cell -> position table when the caller has
                // it (every key formed here is <= the grid's cell count, the table's last entry), else a bisection of the
                // keys (14 dependent loads for a 16 k-point voxel: the run tables were most of a k = 2 search's time)
                // (clamped to the voxel's range: a sampler call whose grid did not fit its table leaves the table unwritten -
                // the level is empty then, c0 == c1, and the caller repeats the geometry; nothing may be read through garbage)
                const int s0 = cell_start ? min(max(cell_start[ka], c0), c1) : lower_bound_key(keys, c0, c1, ka);
                const int s1 = cell_start ? min(max(cell_start[kz], s0), c1) : lower_bound_key(keys, s0, c1, kz);
                run_start[tid] = s0;
                len = s1 - s0;
            }
        }
        // exclusive prefix sum of len over the workgroup
        int inc = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        run_pre[tid + 1] = base + inc;
        if (tid == 0) run_pre[0] = 0;
        __syncthreads();
        return run_pre[G_MAXRUN];
    };

    float rho, rk = 0.f;
    // every query starts from a caller-supplied bound and the largest of them is within a few cells: no density probe,
    // the region is the box of the queries grown by that bound.  (One loose bound would inflate the region of all 32
    // queries - on sparse levels the probe's density estimate is the better radius; the bounds still seed thr.)
    const bool use_hint = MODE == 0 && hinted > 0.f && hmax <= 9.f * res * res;
    if (MODE == 1) {
        rho = r;
    } else if (use_hint) {
        rho = sqrtf(hmax) * 1.0001f + 3.f * eps;
        rk = rho;
    } else {
        // local density probe: candidates in the rows (BOX: cells) that hold the queries themselves
        const int nr0 = region(0.f, false);
        long long n0 = 0;
        for (int rb = 0; rb < nr0 && !whole; rb += G_MAXRUN) n0 += build(rb);
        float dens = 0.f;
        if (!whole && n0 >= 8) {
            const float vol = ((float)(nXhi - nXlo + 1) * res) * ((float)(nYhi - nYlo + 1) * res) * ((float)(nZhi - nZlo + 1) * res);
            dens = (float)n0 / vol;
        } else {
            dens = (float)total / fmaxf(((float)g0 * res) * ((float)g1 * res) * ((float)g2 * res), 1e-30f);
        }
        rk = cbrtf(0.75f * (float)k / (3.14159265f * fmaxf(dens, 1e-30f)));
        SLAB_STAMP(1);   // probe
        rho = fminf(1.1f * rk, 1e30f) + 0.5f * res;
        whole = false;
    }

    for (int pass = 0; pass < 12; ++pass) {
        const bool was_whole = whole;
        const int nrt = was_whole ? 0 : region(rho, pass >= 8);   // give up on the geometry after 8 growth passes
        if (whole && !was_whole && oYhi >= oYlo) {
            // the region outgrew the run budget (or the pass budget): the whole voxel is scanned as one run, including
            // what was scanned before, so the unfinished queries start again from scratch
#pragma unroll
            for (int j = 0; j < S_QPW; ++j) {
                if ((active >> j) & 1u) {
                    best_d[j] = INFINITY; best_i[j] = 0x7fffffff; cnt[j] = 0;
                    thr[j] = MODE == 0 ? INFINITY : __int_as_float(0x7fffffff);
                }
            }
        }
        for (int rb = 0; rb < nrt; rb += G_MAXRUN) {
        const int n_c = build(rb);
        SLAB_STAMP(2);   // plan
        SLAB_COUNT(8, 1); SLAB_COUNT(9, n_c); SLAB_COUNT(10, __popc(active)); SLAB_COUNT(13, (unsigned long long)n_c * __popc(active));   // 13: distance evaluations
        for (int tbase = 0; tbase < n_c; tbase += TILE) {
            __syncthreads();   // previous tile consumed
            // slots used by this tile: a power of two >= the candidates left (>= 256), so that the shuffle stays a
            // permutation and a short gather does not pay for 1024 slots
            const int left = n_c - tbase;
            // kNN tiles are staged shuffled (storage order approaches a query monotonically: every candidate would be a new
            // admission - even k = 2 runs 40 % slower unshuffled).  slot -> (slot * 389) mod tsz is a permutation for
            // every tsz that is not a multiple of the prime 389; tsz = the candidates left rounded up to whole 256-candidate
            // groups (round 4: the k >= 8 kernels used to round up to a power of two for a cheap mask - a 1190-candidate
            // gather, the typical k = 32 region, then ran its threshold ladder and its scan over 2048 slots, 42 % of them
            // padding; the modulo is now one division per tile and thread, the slots of a thread follow by addition)
            const bool pow2 = MODE != 0;
            const int tsz = pow2 ? (left > 512 ? 1024 : (left > 256 ? 512 : 256)) : min(TILE, (left + 255) & ~255);
            const int g_step = pow2 ? 0 : (256 * 389) % tsz;
            int g_run = pow2 ? 0 : (tid * 389) % tsz;
            // three sweeps over the thread's slots instead of one (the search kernels wait, they do not compute: VALU active
            // 21-26 % of the wave cycles, profiles/r3_f16x3_valu.csv): all run lookups, then all record loads in flight
            // together, then the LDS stores - not a bisection, a load and a store in a dependent chain per slot
            int cidx[TILE / 256];
#pragma unroll
            for (int rr = 0; rr < TILE / 256; ++rr) {
                const int s = tid + 256 * rr;
                const int g = tbase + (MODE != 0 ? s : g_run);      // (s * 389) mod tsz
                g_run += g_step;
                g_run -= g_run >= tsz ? tsz : 0;
                cidx[rr] = -1;
                if (s < tsz && g < n_c) {
                    int lo = 0, hi = G_MAXRUN;   // largest run with run_pre[run] <= g
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (run_pre[mid] <= g) lo = mid; else hi = mid; }
                    cidx[rr] = run_start[lo] + (g - run_pre[lo]);
                }
            }
            float4 cv[TILE / 256];
#pragma unroll
            for (int rr = 0; rr < TILE / 256; ++rr) {
                cv[rr] = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(0x7fffffff));
                if (cidx[rr] >= 0) {
                    cv[rr] = x[cidx[rr]];
                    if (!index_in_w) cv[rr].w = __int_as_float(cidx[rr]);
                }
            }
#pragma unroll
            for (int rr = 0; rr < TILE / 256; ++rr) {
                const int s = tid + 256 * rr;
                if (s < tsz) cand[s] = cv[rr];
            }
            __syncthreads();
            SLAB_STAMP(3);   // staging
            if (MODE == 0 && LADDER && pass == 0 && rb == 0 && tbase == 0) {
                // Threshold ladder: before any insertion, count this tile's candidates inside a few trial radii around
                // the density estimate and start from the smallest one that already holds k of them - a valid upper
                // bound of the final k-th distance, so nothing that can end up in the result is refused, while the
                // ~k*ln(n/k) warm-up insertions of a cold start are not done.
                const float scale = n_c > tsz ? cbrtf((float)n_c / (float)tsz) : 1.f;   // first tile of several: sparser
                float u[6];
                {
                    const float f[6] = {0.7f, 0.85f, 1.0f, 1.2f, 1.45f, 1.8f};
#pragma unroll
                    for (int i = 0; i < 6; ++i) { const float rr_ = f[i] * rk * scale; u[i] = rr_ * rr_; }
                }
#pragma unroll
                for (int j = 0; j < S_QPW; ++j) {
                    if (!((active >> j) & 1u)) continue;
                    int c_[6] = {0, 0, 0, 0, 0, 0};
                    for (int ch = 0; ch < (tsz >> 6); ++ch) {
                        const float4 c = cand[ch * 64 + lane];
                        const float d = p2w_d2(uq[j].x, uq[j].y, uq[j].z, c.x, c.y, c.z);
#pragma unroll
                        for (int i = 0; i < 6; ++i) c_[i] += __popcll(__ballot(d <= u[i]));
                    }
                    float t = INFINITY;
#pragma unroll
                    for (int i = 5; i >= 0; --i) t = c_[i] >= k ? u[i] : t;
                    thr[j] = t;
                }
            }
            SLAB_STAMP(6);   // (ladder, when taken; else ~0)
            if (MODE == 0) {
                for (int gr = 0; gr < (tsz >> 8); ++gr) {
                    float4 c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) c[u] = cand[gr * 256 + u * 64 + lane];
#pragma unroll
                    for (int j = 0; j < S_QPW; ++j) {
                        if (!((active >> j) & 1u)) continue;   // wave-uniform
                        float bd = best_d[j], t = thr[j];
                        int bi = best_i[j];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float d = p2w_d2(uq[j].x, uq[j].y, uq[j].z, c[u].x, c[u].y, c[u].z);
                            const int ci = __float_as_int(c[u].w);
                            unsigned long long m = __ballot(d <= t);
                            while (m) {
                                const int src = __ffsll((long long)m) - 1;
                                m &= m - 1;
                                const float dn = rdlane(d, src);
                                const int in = __builtin_amdgcn_readlane(ci, src);
                                const unsigned long long kn = ((unsigned long long)__float_as_uint(dn) << 32) | (unsigned)in;
                                const unsigned long long kbst = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bi;
                                const int pos = __popcll(__ballot(kbst < kn));
                                if (pos < k) {
                                    const float up_d = shr1(bd);
                                    const int up_i = shr1(bi);
                                    const bool here = lane == pos, sh = (lane > pos) & in_k;
                                    bd = here ? dn : (sh ? up_d : bd);
                                    bi = here ? in : (sh ? up_i : bi);
                                    t = fminf(t, rdlane(bd, k - 1));   // slot k-1 is +inf until k pairs are kept
                                }
                            }
                        }
                        best_d[j] = bd; best_i[j] = bi; thr[j] = t;
                    }
                }
            } else {
                const int nch = (min(TILE, n_c - tbase) + 63) >> 6;
                for (int ch = 0; ch < nch; ++ch) {
                    const float4 c = cand[ch * 64 + lane];
                    const int ci = __float_as_int(c.w);
#pragma unroll
                    for (int j = 0; j < S_QPW; ++j) {
                        if (!((active >> j) & 1u)) continue;
                        const float d = p2w_d2(uq[j].x, uq[j].y, uq[j].z, c.x, c.y, c.z);
                        const bool hit = d < r2;
                        const unsigned long long mh = __ballot(hit);
                        if (mh == 0ull) continue;
                        cnt[j] += __popcll(mh);
                        int bi = best_i[j], ti = __float_as_int(thr[j]);
                        unsigned long long m = __ballot(hit && ci < ti);
                        while (m) {
                            const int src = __ffsll((long long)m) - 1;
                            m &= m - 1;
                            const int in = __builtin_amdgcn_readlane(ci, src);
                            if (in < ti) {
                                const int pos = __popcll(__ballot(bi < in));
                                const int up_i = shr1(bi);
                                bi = (lane == pos) ? in : (((lane > pos) & in_k) ? up_i : bi);
                                ti = __builtin_amdgcn_readlane(bi, k - 1);
                            }
                        }
                        best_i[j] = bi; thr[j] = __int_as_float(ti);
                    }
                }
            }
        }
        }
        if (!was_whole) { oXlo = nXlo; oXhi = nXhi; oYlo = nYlo; oYhi = nYhi; oZlo = nZlo; oZhi = nZhi; }
        SLAB_STAMP(4);   // scan
        if (MODE == 1) break;
        if (MODE == 0 && hint && pass == 0) {
            // a hint is only trusted after the fact: the ball it describes must have yielded k candidates (or all there
            // are).  Otherwise the scanned region is declared empty, so that the growth loop below rescans it without the
            // bogus bound - and EVERY query of the workgroup forgets what it holds: the rescan presents the region's
            // candidates again, and a query that kept its list would keep each of them twice.
            bool bogus = false;
#pragma unroll
            for (int j = 0; j < S_QPW; ++j) {
                if (!((active >> j) & 1u)) continue;
                const int found = __popcll(__ballot(best_i[j] != 0x7fffffff));
                if (found < min(k, total)) bogus = true;
            }
            __syncthreads();
            if (lane == 0) wred[wave][4] = bogus ? 1.f : 0.f;
            __syncthreads();
            if (wred[0][4] + wred[1][4] + wred[2][4] + wred[3][4] > 0.f) {
                oXlo = 0; oXhi = -1; oYlo = 0; oYhi = -1; oZlo = 0; oZhi = -1;
#pragma unroll
                for (int j = 0; j < S_QPW; ++j)
                    if ((active >> j) & 1u) { best_d[j] = INFINITY; best_i[j] = 0x7fffffff; thr[j] = INFINITY; }
            }
        }
        // which queries are final?  k-th distance <= distance to the nearest face of the scanned region
        float need = 0.f;
        if (!whole) {
            const float fx0 = (!BOX || oXlo <= 0) ? -INFINITY : lo_x + (float)oXlo * res;
            const float fx1 = (!BOX || (long long)oXhi >= g0 - 1) ? INFINITY : lo_x + (float)(oXhi + 1) * res;
            const float fy0 = oYlo <= 0 ? -INFINITY : lo_y + (float)oYlo * res;
            const float fy1 = (long long)oYhi >= g1 - 1 ? INFINITY : lo_y + (float)(oYhi + 1) * res;
            const float fz0 = oZlo <= 0 ? -INFINITY : lo_z + (float)oZlo * res;
            const float fz1 = (long long)oZhi >= g2 - 1 ? INFINITY : lo_z + (float)(oZhi + 1) * res;
#pragma unroll
            for (int j = 0; j < S_QPW; ++j) {
                if (!((active >> j) & 1u)) continue;
                float gap = fminf(fminf(uq[j].y - fy0, fy1 - uq[j].y), fminf(uq[j].z - fz0, fz1 - uq[j].z));
                if (BOX) gap = fminf(gap, fminf(uq[j].x - fx0, fx1 - uq[j].x));
                gap -= eps;
                const bool ok = gap > 0.f && thr[j] <= gap * gap;      // thr = +inf until k candidates were found
                if (ok) active &= ~(1u << j);
                else need = fmaxf(need, thr[j] < INFINITY ? sqrtf(thr[j]) * 1.0001f + 3.f * eps : 2.f * rho + res);
            }
        } else {
            active = 0u;
        }
        __syncthreads();
        if (lane == 0) wred[wave][4] = need;
        __syncthreads();
        need = fmaxf(fmaxf(wred[0][4], wred[1][4]), fmaxf(wred[2][4], wred[3][4]));
        SLAB_STAMP(5);   // check
        if (!(need > 0.f)) break;          // every query of the workgroup is final
        rho = fmaxf(need, rho);
    }
    SLAB_COUNT(11, 1);
    const int kept_max = min(k, total);
#pragma unroll
    for (int j = 0; j < S_QPW; ++j) {
        const int q = qw + j;
        if (q < q1) {
            const int row = (flags & P2W_SEARCH_Q_ROW_IN_W) ? __float_as_int(xq[qidx ? qidx[q] : q].w) : q;
            int kept = MODE == 0 ? kept_max : min(cnt[j], k);
            // a query with NaN coordinates admits nothing: report what was really found, never an unset slot
            kept = min(kept, __popcll(__ballot(best_i[j] != 0x7fffffff)));
            if (lane < k) nbr[(size_t)row * k + lane] = (lane < kept) ? best_i[j] : -1;
            if (lane == 0) deg[row] = kept;
        }
    }
}

static int32_t search_args(const float* xyzr_x, const int32_t* ptr_x, const float* xyzr_q, const int32_t* ptr_q, int32_t B,
                           int32_t m_bound, int32_t k, const int32_t* nbr, const int32_t* deg, int32_t k_max = P2W_MAX_K) {
    P2W_CHECK_PTR(xyzr_x); P2W_CHECK_PTR(ptr_x); P2W_CHECK_PTR(xyzr_q); P2W_CHECK_PTR(ptr_q); P2W_CHECK_PTR(nbr);
    P2W_CHECK_PTR(deg); P2W_CHECK_ALIGN16(xyzr_x); P2W_CHECK_ALIGN16(xyzr_q);
    if (B <= 0 || m_bound < 0 || k <= 0 || k > k_max) return P2W_EINVAL;
    return P2W_OK;
}

// ------------------------------------------------------------------------------------------------
// k in 65 .. 100 (torch-cluster's limit is 100; the wave-wide kernels keep one list slot per lane, i.e. 64): ONE THREAD per
// query, upstream's CUDA algorithm itself - the candidates of the query's voxel in storage order, a list of the k smallest
// (d2, index) keys per thread (in scratch: nothing on the inference path uses k > 64 - model.py:210-212: 32, predicter.py:137: 64 -
// so this path is for the operators' completeness, not for speed).  Same arithmetic, tie rule and flags as knn_kernel / ball_kernel.
// ------------------------------------------------------------------------------------------------
template <int MODE>   // 0 = kNN (k smallest (d2, index) keys, ascending), 1 = ball query (the `cap` lowest indices with d2 < r2, ascending)
__global__ __launch_bounds__(64) void search_wide_kernel(const float4* __restrict__ x, const int* __restrict__ ptr_x,
                                                         const float4* __restrict__ xq, const int* __restrict__ qidx,
                                                         const int* __restrict__ ptr_q, int B, int k, float r2, int* __restrict__ nbr,
                                                         int* __restrict__ deg, int flags) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= ptr_q[B]) return;
    const int b = p2w_find_segment(ptr_q, B, q);
    const int c0 = ptr_x[b], c1 = ptr_x[b + 1];
    const float4 pq = xq[qidx ? qidx[q] : q];
    const bool index_in_w = (flags & P2W_SEARCH_X_INDEX_IN_W) != 0;
    unsigned long long key[P2W_MAX_K_WIDE];   // sorted ascending; kNN: (d2 bits << 32) | index, ball: index
    int kept = 0, hits = 0;
    for (int c = c0; c < c1; ++c) {
        const float4 pc = x[c];
        const float d = p2w_d2(pq.x, pq.y, pq.z, pc.x, pc.y, pc.z);
        const unsigned idx = (unsigned)(index_in_w ? __float_as_int(pc.w) : c);
        unsigned long long kn;
        if (MODE == 0) {
            if (!(d <= INFINITY)) continue;   // a NaN distance admits nothing
            kn = ((unsigned long long)__float_as_uint(d) << 32) | idx;
        } else {
            if (!(d < r2)) continue;
            ++hits;
            kn = idx;
        }
        if (kept == k && kn >= key[k - 1]) continue;
        int pos = kept < k ? kept : k - 1;     // the slot that falls off (or the free one) ...
        while (pos > 0 && key[pos - 1] > kn) { key[pos] = key[pos - 1]; --pos; }   // ... moves down to the insertion point
        key[pos] = kn;
        if (kept < k) ++kept;
    }
    const int row = (flags & P2W_SEARCH_Q_ROW_IN_W) ? __float_as_int(pq.w) : q;
    const int n_out = MODE == 0 ? kept : min(hits, k);
    for (int s = 0; s < k; ++s) nbr[(size_t)row * k + s] = s < n_out ? (int)(unsigned)key[s] : -1;
    deg[row] = n_out;
}

extern "C" int32_t p2w_knn(const float* xyzr_x, const int32_t* ptr_x, const float* xyzr_q, const int32_t* qidx,
                           const int32_t* ptr_q, int32_t B, int32_t m_bound, int32_t k, int32_t* nbr, int32_t* deg,
                           const float* tile_bbox, int32_t flags, p2w_stream_t stream) {
    if (m_bound == 0) return P2W_OK;
    const int32_t st = search_args(xyzr_x, ptr_x, xyzr_q, ptr_q, B, m_bound, k, nbr, deg, P2W_MAX_K_WIDE);
    if (st != P2W_OK) return st;
    if (flags & ~(P2W_SEARCH_X_INDEX_IN_W | P2W_SEARCH_Q_ROW_IN_W)) return P2W_EINVAL;
    if (k > P2W_MAX_K) {   // 65 .. 100: one thread per query (see search_wide_kernel)
        search_wide_kernel<0><<<p2w_cdiv(m_bound, 64), 64, 0, p2w_s(stream)>>>(reinterpret_cast<const float4*>(xyzr_x), ptr_x,
                                                                               reinterpret_cast<const float4*>(xyzr_q), qidx, ptr_q, B, k, 0.f, nbr, deg, flags);
        return P2W_LAUNCH_STATUS();
    }
    const int grid = p2w_cdiv(m_bound, S_QT) + B;  // upper bound on sum_b ceil(m_b / QT)
    knn_kernel<<<grid, 256, 0, p2w_s(stream)>>>(reinterpret_cast<const float4*>(xyzr_x), ptr_x,
                                                reinterpret_cast<const float4*>(xyzr_q), qidx, ptr_q, B, k, nbr, deg,
                                                tile_bbox, xyzr_q == xyzr_x, flags);
    return P2W_LAUNCH_STATUS();
}

extern "C" int32_t p2w_ball_query(const float* xyzr_x, const int32_t* ptr_x, const float* xyzr_q, const int32_t* qidx,
                                  const int32_t* ptr_q, int32_t B, int32_t m_bound, double r, int32_t cap, int32_t* nbr,
                                  int32_t* deg, const float* tile_bbox, int32_t flags, p2w_stream_t stream) {
    if (m_bound == 0) return P2W_OK;
    const int32_t st = search_args(xyzr_x, ptr_x, xyzr_q, ptr_q, B, m_bound, cap, nbr, deg, P2W_MAX_K_WIDE);
    if (st != P2W_OK) return st;
    const float r2 = (float)(r * r);
    if (!(r > 0.0)) return P2W_EINVAL;
    if (flags & ~(P2W_SEARCH_X_INDEX_IN_W | P2W_SEARCH_Q_ROW_IN_W)) return P2W_EINVAL;
    if (cap > P2W_MAX_K) {
        search_wide_kernel<1><<<p2w_cdiv(m_bound, 64), 64, 0, p2w_s(stream)>>>(reinterpret_cast<const float4*>(xyzr_x), ptr_x,
                                                                               reinterpret_cast<const float4*>(xyzr_q), qidx, ptr_q, B, cap, r2, nbr, deg, flags);
        return P2W_LAUNCH_STATUS();
    }
    const int grid = p2w_cdiv(m_bound, S_QT) + B;
    ball_kernel<<<grid, 256, 0, p2w_s(stream)>>>(reinterpret_cast<const float4*>(xyzr_x), ptr_x,
                                                 reinterpret_cast<const float4*>(xyzr_q), qidx, ptr_q, B, r2, cap, nbr, deg,
                                                 tile_bbox, flags);
    return P2W_LAUNCH_STATUS();
}


static int32_t grid_args(const uint64_t* keys, const p2w_grid* grid, int32_t flags) {
    P2W_CHECK_PTR(keys); P2W_CHECK_PTR(grid);
    if ((reinterpret_cast<uintptr_t>(keys) & 7u) || (reinterpret_cast<uintptr_t>(grid) & 7u)) return P2W_EALIGN;
    if (flags & ~(P2W_SEARCH_X_INDEX_IN_W | P2W_SEARCH_Q_ROW_IN_W | P2W_SEARCH_BOX)) return P2W_EINVAL;
    return P2W_OK;
}

extern "C" int32_t p2w_knn_grid_indexed(const float* xyzr_x, const uint64_t* keys_x, const int32_t* ptr_x, const p2w_grid* grid,
                                        const int32_t* cell_start, const float* xyzr_q, const int32_t* qidx, const int32_t* ptr_q,
                                        int32_t B, int32_t m_bound, int32_t k, int32_t* nbr, int32_t* deg, const float* hint,
                                        int32_t flags, p2w_stream_t stream) {
    if (m_bound == 0) return P2W_OK;
    int32_t st = search_args(xyzr_x, ptr_x, xyzr_q, ptr_q, B, m_bound, k, nbr, deg);
    if (st != P2W_OK) return st;
    if ((st = grid_args(keys_x, grid, flags)) != P2W_OK) return st;
    const int grid_dim = p2w_cdiv(m_bound, S_QT) + B;
    auto* kern = (flags & P2W_SEARCH_BOX) ? ((k >= 8) ? slab_search_kernel<0, P2W_KNN_TILE, 1, true> : slab_search_kernel<0, 1024, 0, true>)
                                          : ((k >= 8) ? slab_search_kernel<0, P2W_KNN_TILE, 1, false> : slab_search_kernel<0, 1024, 0, false>);
    kern<<<grid_dim, 256, 0, p2w_s(stream)>>>(
        reinterpret_cast<const float4*>(xyzr_x), reinterpret_cast<const unsigned long long*>(keys_x), ptr_x, grid,
        reinterpret_cast<const float4*>(xyzr_q), qidx, ptr_q, B, k, 0.f, 0.f, nbr, deg, flags, hint, cell_start);
    return P2W_LAUNCH_STATUS();
}
extern "C" int32_t p2w_knn_grid(const float* xyzr_x, const uint64_t* keys_x, const int32_t* ptr_x, const p2w_grid* grid,
                                const float* xyzr_q, const int32_t* qidx, const int32_t* ptr_q, int32_t B, int32_t m_bound,
                                int32_t k, int32_t* nbr, int32_t* deg, const float* hint, int32_t flags, p2w_stream_t stream) {
    return p2w_knn_grid_indexed(xyzr_x, keys_x, ptr_x, grid, nullptr, xyzr_q, qidx, ptr_q, B, m_bound, k, nbr, deg, hint, flags, stream);
}

extern "C" int32_t p2w_ball_query_grid_indexed(const float* xyzr_x, const uint64_t* keys_x, const int32_t* ptr_x, const p2w_grid* grid,
                                               const int32_t* cell_start, const float* xyzr_q, const int32_t* qidx,
                                               const int32_t* ptr_q, int32_t B, int32_t m_bound, double r, int32_t cap, int32_t* nbr,
                                               int32_t* deg, int32_t flags, p2w_stream_t stream) {
    if (m_bound == 0) return P2W_OK;
    int32_t st = search_args(xyzr_x, ptr_x, xyzr_q, ptr_q, B, m_bound, cap, nbr, deg);
    if (st != P2W_OK) return st;
    if ((st = grid_args(keys_x, grid, flags)) != P2W_OK) return st;
    if (!(r > 0.0)) return P2W_EINVAL;
    const int grid_dim = p2w_cdiv(m_bound, S_QT) + B;
    auto* kern = (flags & P2W_SEARCH_BOX) ? slab_search_kernel<1, 1024, 0, true> : slab_search_kernel<1, 1024, 0, false>;
    kern<<<grid_dim, 256, 0, p2w_s(stream)>>>(
        reinterpret_cast<const float4*>(xyzr_x), reinterpret_cast<const unsigned long long*>(keys_x), ptr_x, grid,
        reinterpret_cast<const float4*>(xyzr_q), qidx, ptr_q, B, cap, (float)r, (float)(r * r), nbr, deg, flags, nullptr, cell_start);
    return P2W_LAUNCH_STATUS();
}
extern "C" int32_t p2w_ball_query_grid(const float* xyzr_x, const uint64_t* keys_x, const int32_t* ptr_x, const p2w_grid* grid,
                                       const float* xyzr_q, const int32_t* qidx, const int32_t* ptr_q, int32_t B,
                                       int32_t m_bound, double r, int32_t cap, int32_t* nbr, int32_t* deg, int32_t flags,
                                       p2w_stream_t stream) {
    return p2w_ball_query_grid_indexed(xyzr_x, keys_x, ptr_x, grid, nullptr, xyzr_q, qidx, ptr_q, B, m_bound, r, cap, nbr, deg, flags, stream);
}



// Upper bound of the 2nd-nearest-candidate distance of every fine point from the sampler's own bookkeeping: the point's
// cell representative (rank[i]) is one coarse point, the representative of the nearest point in storage order (same
// voxel) that lies in another cell is a second one; hint = the larger of the two squared distances, +inf when no second
// representative turns up within 8 storage neighbours.  Valid for k <= 2 (the interpolation searches).
__global__ __launch_bounds__(256) void knn_hint2_kernel(const float4* __restrict__ xq, const int* __restrict__ rank,
                                                        const int* __restrict__ ptr_q, int B, const float4* __restrict__ xc,
                                                        float* __restrict__ hint) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ptr_q[B]) return;
    const int b = p2w_find_segment(ptr_q, B, i);
    const int lo = ptr_q[b], hi = ptr_q[b + 1];
    const float4 q = xq[i];
    const int c1 = rank[i];
    const float4 p1 = xc[c1];
    const float d1 = p2w_d2(q.x, q.y, q.z, p1.x, p1.y, p1.z);
    float d2nd = INFINITY;
    for (int s = 1; s <= 8; ++s) {
        const int ja = i + s, jb = i - s;
        if (ja < hi) {
            const int c = rank[ja];
            if (c != c1) { const float4 p = xc[c]; d2nd = fminf(d2nd, p2w_d2(q.x, q.y, q.z, p.x, p.y, p.z)); }
        }
        if (jb >= lo) {
            const int c = rank[jb];
            if (c != c1) { const float4 p = xc[c]; d2nd = fminf(d2nd, p2w_d2(q.x, q.y, q.z, p.x, p.y, p.z)); }
        }
        if (d2nd < INFINITY && s >= 2) break;
    }
    hint[i] = fmaxf(d1, d2nd);
}

extern "C" int32_t p2w_knn_hint2(const float* xyzr_q, const int32_t* rank, const int32_t* ptr_q, int32_t B, int32_t m_bound,
                                 const float* xyzr_c, float* hint, p2w_stream_t stream) {
    if (m_bound == 0) return P2W_OK;
    P2W_CHECK_PTR(xyzr_q); P2W_CHECK_PTR(rank); P2W_CHECK_PTR(ptr_q); P2W_CHECK_PTR(xyzr_c); P2W_CHECK_PTR(hint);
    P2W_CHECK_ALIGN16(xyzr_q); P2W_CHECK_ALIGN16(xyzr_c);
    if (B <= 0 || m_bound < 0) return P2W_EINVAL;
    knn_hint2_kernel<<<p2w_cdiv(m_bound, 256), 256, 0, p2w_s(stream)>>>(reinterpret_cast<const float4*>(xyzr_q), rank, ptr_q, B,
                                                                       reinterpret_cast<const float4*>(xyzr_c), hint);
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// plot-scale helpers: Morton order of records on a grid, neighbourhood vote
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long spread3(unsigned v) {   // 21 bits -> every third bit
    unsigned long long x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void morton_keys_kernel(const float4* __restrict__ xyzr, int n, const p2w_grid* __restrict__ grid,
                                                          unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = xyzr[i];
    const float res = grid->res;
    const unsigned cx = (unsigned)grid_cell(p.x, grid->lo[0], res, 1ll << 21);
    const unsigned cy = (unsigned)grid_cell(p.y, grid->lo[1], res, 1ll << 21);
    const unsigned cz = (unsigned)grid_cell(p.z, grid->lo[2], res, 1ll << 21);
    keys[i] = spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2);
    if (vals) vals[i] = i;
}

struct MoLayout { size_t keys_in, keys_out, temp, temp_bytes, total; };
static hipError_t mo_layout(int n, MoLayout* L) {
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    RsLayout R;
    rs_layout(n, &R);
    size_t off = 0;
    L->keys_in = off; off += up(sizeof(unsigned long long) * n);
    L->keys_out = off; off += up(sizeof(unsigned long long) * n);
    L->temp = off; L->temp_bytes = up(R.bytes); off += L->temp_bytes;
    L->total = off;
    return hipSuccess;
}

extern "C" size_t p2w_morton_order_ws_bytes(int32_t n) {
    if (n <= 0) return 256;
    MoLayout L;
    if (mo_layout(n, &L) != hipSuccess) return 0;
    return L.total;
}

extern "C" int32_t p2w_morton_order(const float* xyzr, int32_t n, const p2w_grid* grid, int32_t* order_out, void* ws,
                                    size_t ws_bytes, p2w_stream_t stream) {
    if (n == 0) return P2W_OK;
    P2W_CHECK_PTR(xyzr); P2W_CHECK_PTR(grid); P2W_CHECK_PTR(order_out); P2W_CHECK_PTR(ws);
    P2W_CHECK_ALIGN16(xyzr); P2W_CHECK_ALIGN16(ws);
    if (n < 0) return P2W_EINVAL;
    MoLayout L;
    hipError_t e = mo_layout(n, &L);
    if (e != hipSuccess) return (int32_t)e;
    if (ws_bytes < L.total) return P2W_EWORKSPACE;
    char* w = static_cast<char*>(ws);
    hipStream_t s = p2w_s(stream);
    auto* keys_in = reinterpret_cast<unsigned long long*>(w + L.keys_in);
    morton_keys_kernel<<<p2w_cdiv(n, 256), 256, 0, s>>>(reinterpret_cast<const float4*>(xyzr), n, grid, keys_in, nullptr);
    // argsort (values = 0..n-1) by the hand-written radix sort: as many digit passes as the largest key has bytes
    e = rs_sort_pairs(w + L.temp, keys_in, reinterpret_cast<unsigned long long*>(w + L.keys_out), nullptr, order_out, nullptr, n, s);
    if (e != hipSuccess) return (int32_t)e;
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// plot -> voxels (the reference's Voxelise.grid, pointstowood/src/preprocessing.py:55-64): cell ids over ALL columns of the
// point table, a stable argsort of them, and the runs of equal cells that hold at least min_pts points
// ------------------------------------------------------------------------------------------------
constexpr int VC_MAXD = 16;
struct VcHeader { unsigned lo[VC_MAXD], hi[VC_MAXD]; };
__global__ void vc_init_kernel(VcHeader* h) {
    if (threadIdx.x < VC_MAXD) { h->lo[threadIdx.x] = 0xffffffffu; h->hi[threadIdx.x] = 0u; }
}
__global__ __launch_bounds__(256) void vc_minmax_kernel(const float* __restrict__ P, int n, int D, int ld, VcHeader* h) {
    __shared__ float red[4][2 * VC_MAXD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lo[VC_MAXD], hi[VC_MAXD];
#pragma unroll
    for (int d = 0; d < VC_MAXD; ++d) { lo[d] = INFINITY; hi[d] = -INFINITY; }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v[VC_MAXD];
        bool ok = true;
#pragma unroll
        for (int d = 0; d < VC_MAXD; ++d) {
            v[d] = d < D ? P[i * ld + d] : 0.f;
            ok = ok && isfinite(v[d]);
        }
        if (ok) {   // a ROW with a non-finite value takes no part in the grid (see vc_cells_kernel): none of its columns does
#pragma unroll
            for (int d = 0; d < VC_MAXD; ++d) { lo[d] = fminf(lo[d], v[d]); hi[d] = fmaxf(hi[d], v[d]); }
        }
    }
#pragma unroll
    for (int d = 0; d < VC_MAXD; ++d) {
        float a = lo[d], b = hi[d];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a = fminf(a, __shfl_xor(a, off)); b = fmaxf(b, __shfl_xor(b, off)); }
        if (lane == 0) { red[wave][2 * d] = a; red[wave][2 * d + 1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < D) {
        const int d = threadIdx.x;
        float a = red[0][2 * d], b = red[0][2 * d + 1];
        for (int w = 1; w < 4; ++w) { a = fminf(a, red[w][2 * d]); b = fmaxf(b, red[w][2 * d + 1]); }
        atomicMin(&h->lo[d], f2ord(a)); atomicMax(&h->hi[d], f2ord(b));
    }
}
// PyG voxel_grid(P, size) with batch = None: sum_d trunc((P_d - lo_d) / size) * stride_d, strides = running products of
// the per-column cell counts trunc((hi_d - lo_d) / size) + 1 (fp32 subtract / divide / truncate as oracle/ops.py voxel_grid).
// Rows with a non-finite value: the reference's min / max propagate a NaN into the grid origin and every cell id of the plot
// becomes the cast of a NaN (undefined); here such rows stay out of the minima / maxima and get the dedicated key
// P2W_CELL_NONFINITE (INT64_MAX: they sort last as a run of their own, which the voxeliser drops) - defined behaviour on both
// the HIP path and its tensor restatement (oracle/preprocess.py cells_nd), identical to the reference on finite input.
__global__ __launch_bounds__(256) void vc_cells_kernel(const float* __restrict__ P, int n, int D, int ld, float size,
                                                       const VcHeader* __restrict__ h, long long* __restrict__ cell) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    long long key = 0, stride = 1;
    bool finite = true;
    for (int d = 0; d < D; ++d) {
        const float lo = ord2f(h->lo[d]), hi = ord2f(h->hi[d]);
        const float v = P[i * ld + d];
        finite = finite && isfinite(v);
        const long long cnt = (long long)((hi - lo) / size) + 1;
        key += (long long)(((finite ? v : lo) - lo) / size) * stride;
        stride *= cnt;
    }
    cell[i] = finite ? key : 0x7fffffffffffffffll;
}
extern "C" int32_t p2w_cells_nd(const float* P, int32_t n, int32_t D, int32_t ld, float size, int64_t* cell_out, void* ws,
                                size_t ws_bytes, p2w_stream_t stream) {
    if (n == 0) return P2W_OK;
    P2W_CHECK_PTR(P); P2W_CHECK_PTR(cell_out); P2W_CHECK_PTR(ws);
    if (n < 0 || D <= 0 || D > VC_MAXD || ld < D || !(size > 0.f)) return P2W_EINVAL;
    if (ws_bytes < sizeof(VcHeader)) return P2W_EWORKSPACE;
    hipStream_t s = p2w_s(stream);
    auto* h = static_cast<VcHeader*>(ws);
    const int nblk = p2w_cdiv(n, 256);
    vc_init_kernel<<<1, 64, 0, s>>>(h);
    vc_minmax_kernel<<<nblk < 512 ? nblk : 512, 256, 0, s>>>(P, n, D, ld, h);
    vc_cells_kernel<<<nblk, 256, 0, s>>>(P, n, D, ld, size, h, reinterpret_cast<long long*>(cell_out));
    return P2W_LAUNCH_STATUS();
}

extern "C" size_t p2w_sort_pairs_u64_ws_bytes(int32_t n) {
    RsLayout R;
    rs_layout(n > 0 ? n : 1, &R);
    return R.bytes;
}
extern "C" int32_t p2w_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int32_t n,
                                      void* ws, size_t ws_bytes, p2w_stream_t stream) {
    if (n == 0) return P2W_OK;
    P2W_CHECK_PTR(keys_in); P2W_CHECK_PTR(keys_out); P2W_CHECK_PTR(vals_out); P2W_CHECK_PTR(ws); P2W_CHECK_ALIGN16(ws);
    if (n < 0 || keys_in == keys_out || (vals_in && vals_in == vals_out)) return P2W_EINVAL;
    if (ws_bytes < p2w_sort_pairs_u64_ws_bytes(n)) return P2W_EWORKSPACE;
    return (int32_t)rs_sort_pairs(ws, reinterpret_cast<const unsigned long long*>(keys_in), reinterpret_cast<unsigned long long*>(keys_out),
                                  vals_in, vals_out, nullptr, n, p2w_s(stream));
}

// runs of equal keys in a sorted array that hold at least min_count elements: (start, count) of each, in order, + how many
__global__ __launch_bounds__(256) void vr_flag_kernel(const unsigned long long* __restrict__ keys, int n, int* __restrict__ flag) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;      // first element of a run
}
__global__ __launch_bounds__(256) void vr_starts_kernel(const int* __restrict__ flag, const int* __restrict__ rank, int n,
                                                        int* __restrict__ run_start, int* __restrict__ n_runs) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) run_start[rank[i]] = (int)i;
    if (i == n - 1) { const int r = rank[i] + flag[i]; run_start[r] = n; *n_runs = r; }
}
__global__ __launch_bounds__(256) void vr_keep_kernel(const int* __restrict__ run_start, const int* __restrict__ n_runs, int min_count,
                                                      int n, int* __restrict__ keep) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    keep[r] = (r < *n_runs && run_start[r + 1] - run_start[r] >= min_count) ? 1 : 0;
}
__global__ __launch_bounds__(256) void vr_compact_kernel(const int* __restrict__ run_start, const int* __restrict__ n_runs,
                                                         const int* __restrict__ keep, const int* __restrict__ pos, int n,
                                                         int* __restrict__ starts_out, int* __restrict__ counts_out,
                                                         int* __restrict__ n_out) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nr = *n_runs;
    if (r == 0) *n_out = nr > 0 ? pos[nr - 1] + keep[nr - 1] : 0;
    if (r >= nr || !keep[r]) return;
    starts_out[pos[r]] = run_start[r];
    counts_out[pos[r]] = run_start[r + 1] - run_start[r];
}
extern "C" size_t p2w_key_runs_ws_bytes(int32_t n) {
    const size_t m = (size_t)(n > 0 ? n : 1) + 1;
    return 4 * ((m * sizeof(int) + 255) & ~size_t(255)) + 256 + xs_ws_bytes(n);
}
extern "C" int32_t p2w_key_runs(const uint64_t* keys_sorted, int32_t n, int32_t min_count, int32_t* starts_out, int32_t* counts_out,
                                int32_t* n_out, void* ws, size_t ws_bytes, p2w_stream_t stream) {
    P2W_CHECK_PTR(n_out);
    hipStream_t s = p2w_s(stream);
    if (n == 0) return (int32_t)hipMemsetAsync(n_out, 0, sizeof(int), s);
    P2W_CHECK_PTR(keys_sorted); P2W_CHECK_PTR(starts_out); P2W_CHECK_PTR(counts_out); P2W_CHECK_PTR(ws); P2W_CHECK_ALIGN16(ws);
    if (n < 0) return P2W_EINVAL;
    if (ws_bytes < p2w_key_runs_ws_bytes(n)) return P2W_EWORKSPACE;
    const size_t seg = (((size_t)n + 1) * sizeof(int) + 255) & ~size_t(255);
    char* w = static_cast<char*>(ws);
    int* a = reinterpret_cast<int*>(w);             // run-start flags, then keep flags
    int* b = reinterpret_cast<int*>(w + seg);       // their exclusive scans
    int* run_start = reinterpret_cast<int*>(w + 2 * seg);
    int* n_runs = reinterpret_cast<int*>(w + 3 * seg);
    void* xs = w + 3 * seg + 256;
    const int nblk = p2w_cdiv(n, 256);
    const auto* k = reinterpret_cast<const unsigned long long*>(keys_sorted);
    vr_flag_kernel<<<nblk, 256, 0, s>>>(k, n, a);
    hipError_t e = xs_exclusive_scan(xs, a, b, n, s);
    if (e != hipSuccess) return (int32_t)e;
    vr_starts_kernel<<<nblk, 256, 0, s>>>(a, b, n, run_start, n_runs);
    vr_keep_kernel<<<nblk, 256, 0, s>>>(run_start, n_runs, min_count, n, a);
    e = xs_exclusive_scan(xs, a, b, n, s);
    if (e != hipSuccess) return (int32_t)e;
    vr_compact_kernel<<<nblk, 256, 0, s>>>(run_start, n_runs, a, b, n, starts_out, counts_out, n_out);
    return P2W_LAUNCH_STATUS();
}

// cell -> first-candidate table of a sorted key array (the plot-level grid of the back-projection): table[c] = number of
// keys < c = the position a search run that starts at cell c begins at, for c = 0 .. n_cells (n_cells + 1 entries).  With it a
// run lookup of the grid searches is one load; without it, a bisection of the key array (25 dependent loads at 19 M keys: two
// thirds of the back-projection's search time).  One count per key (the keys are sorted: a wave's atomics fall on a few
// neighbouring counters), then the exclusive scan in place.
__global__ __launch_bounds__(256) void cs_count_kernel(const unsigned long long* __restrict__ keys, int n, long long n_cells, int* __restrict__ table) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const bool first = i == 0 || keys[i - 1] != k;
    if (!first || k >= (unsigned long long)n_cells) return;      // (keys beyond the table belong to no cell a search asks for)
    long long step = 1, lo = i, hi;                              // run end: gallop, then bisect (a cell may hold any number of points)
    while ((hi = lo + step) < n && keys[hi] == k) { lo = hi; step <<= 1; }
    hi = hi < n ? hi : n;                                        // keys[lo] == k, keys[hi] != k (or hi == n)
    while (hi - lo > 1) { const long long mid = lo + ((hi - lo) >> 1); if (keys[mid] == k) lo = mid; else hi = mid; }
    table[k] = (int)(hi - i);
}
extern "C" size_t p2w_cell_starts_ws_bytes(int64_t n_cells) { return xs_ws_bytes(n_cells + 1); }
extern "C" int32_t p2w_cell_starts(const uint64_t* keys_sorted, int32_t n, int64_t n_cells, int32_t* table_out, void* ws, size_t ws_bytes,
                                   p2w_stream_t stream) {
    P2W_CHECK_PTR(table_out); P2W_CHECK_PTR(ws); P2W_CHECK_ALIGN16(ws);
    if (n < 0 || n_cells < 0 || n_cells >= (int64_t)0x7fffffff) return P2W_EINVAL;     // the scan counts in int32
    if (ws_bytes < p2w_cell_starts_ws_bytes(n_cells)) return P2W_EWORKSPACE;
    hipStream_t s = p2w_s(stream);
    hipError_t e = hipMemsetAsync(table_out, 0, sizeof(int) * (size_t)(n_cells + 1), s);
    if (e != hipSuccess) return (int32_t)e;
    if (n > 0) {
        P2W_CHECK_PTR(keys_sorted);
        cs_count_kernel<<<p2w_cdiv(n, 256), 256, 0, s>>>(reinterpret_cast<const unsigned long long*>(keys_sorted), n, (long long)n_cells, table_out);
    }
    e = xs_exclusive_scan(ws, table_out, table_out, (int)(n_cells + 1), s);
    if (e != hipSuccess) return (int32_t)e;
    return P2W_LAUNCH_STATUS();
}

// one wave per point: lane l holds neighbour l's (prediction, probability)
__global__ __launch_bounds__(256) void vote_kernel(const int* __restrict__ nbr, const int* __restrict__ deg, int k,
                                                   const float* __restrict__ pred, const float* __restrict__ prob, int n,
                                                   float any_wood, float* __restrict__ label_out, float* __restrict__ pwood_out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int m = min(deg[i], k);
    int j = lane < m ? nbr[(size_t)i * k + lane] : -1;
    const bool on = j >= 0;
    const float pr = on ? prob[j] : 0.f;
    const float pd = on ? pred[j] : 0.f;
    // label
    float label;
    if (any_wood != 1.f) {
        label = __ballot(on && pd > any_wood) != 0ull ? 1.f : 0.f;
    } else {
        double v0 = (on && pd == 0.f) ? (double)pr : 0.0, v1 = (on && pd == 1.f) ? (double)pr : 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off); }
        label = v1 > v0 ? 1.f : 0.f;   // argmax keeps the first maximum
    }
    // median: ascending bitonic sort of the probabilities across the wave, missing slots = +inf
    float v = on ? pr : INFINITY;
    const int cnt = __popcll(__ballot(on));
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            const float o = __shfl_xor(v, stride);
            const bool up = (lane & size) == 0, lower = (lane & stride) == 0;
            v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
        }
    }
    const float a = rdlane(v, cnt > 0 ? (cnt - 1) >> 1 : 0), b2 = rdlane(v, cnt >> 1);
    if (lane == 0) {
        label_out[i] = label;
        pwood_out[i] = cnt == 0 ? 0.f : ((cnt & 1) ? a : (a + b2) * 0.5f);
    }
}

extern "C" int32_t p2w_vote(const int32_t* nbr, const int32_t* deg, int32_t k, const float* pred, const float* prob, int32_t n,
                            float any_wood, float* label_out, float* pwood_out, p2w_stream_t stream) {
    if (n == 0) return P2W_OK;
    P2W_CHECK_PTR(nbr); P2W_CHECK_PTR(deg); P2W_CHECK_PTR(pred); P2W_CHECK_PTR(prob); P2W_CHECK_PTR(label_out);
    P2W_CHECK_PTR(pwood_out);
    if (n < 0 || k <= 0 || k > P2W_MAX_K) return P2W_EINVAL;
    vote_kernel<<<p2w_cdiv(n, 4), 256, 0, p2w_s(stream)>>>(nbr, deg, k, pred, prob, n, any_wood, label_out, pwood_out);
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Exact fp64 re-ranking of a kNN result: the back-projection's neighbour sets as the reference's KD-tree finds them
// (predicter.py:136-137: KDTree over the float64 `classified_pc`, queried with float64 originals - distances are fp64).
// The grid search above measures in fp32 on plot-local coordinates, so candidates whose fp64 distances differ by less than an
// fp32 rounding can swap at the k-th place.  Any k candidates bound the true k-th distance from above: with R2 = the largest
// fp64 distance of the fp32 result's members, the true k nearest all lie within R2.  One wave per query gathers every
// candidate within R2 (fp64 distance, ((dx^2 + dy^2) + dz^2) rounded like the CPU KD-trees' loops) from the grid rows the
// ball touches - normally k + a few - and ranks them by (distance, candidate index): the first k, in that order, replace
// the result.  More than RF_CAP candidates within R2 (piles of coincident points): the k smallest are extracted one by
// one by repeated scans instead (slow, exact).
// ------------------------------------------------------------------------------------------------
constexpr int RF_CAP = 128;
constexpr int RF_FLAT = 256;   // candidate positions of the flattened runs staged per step (per wave)
__global__ __launch_bounds__(256) void knn_refine_kernel(const double* __restrict__ cs, const int* __restrict__ cidx,
                                                         const int* __restrict__ inv, const unsigned long long* __restrict__ keys,
                                                         const int* __restrict__ cell_start, const p2w_grid* __restrict__ gridp,
                                                         double ox, double oy, double oz, const double* __restrict__ q, int m, int nc,
                                                         int k, int* __restrict__ nbr, int* __restrict__ deg) {
    __shared__ double s_d[4][RF_CAP];
    __shared__ int s_i[4][RF_CAP];
    __shared__ int s_f[4][RF_FLAT];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int qi = blockIdx.x * 4 + w;
    if (qi >= m) return;                                  // (no workgroup barrier below: waves are independent)
    const int n_in = min(deg[qi], k);
    if (n_in <= 0) return;
    const double qx = q[3 * (size_t)qi], qy = q[3 * (size_t)qi + 1], qz = q[3 * (size_t)qi + 2];
    auto d2 = [&](const double* c) {
        const double dx = qx - c[0], dy = qy - c[1], dz = qz - c[2];
        return (dx * dx + dy * dy) + dz * dz;
    };
    auto wmax = [](double v) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
        return v;
    };
    double dl = -1.0;
    int jl = -1;
    if (lane < n_in) {
        jl = nbr[(size_t)qi * k + lane];
        dl = d2(cs + 3 * (size_t)inv[jl]);
    }
    double* sd = s_d[w];
    int* si = s_i[w];
    int T = 0;
    if (n_in < k) {   // fewer candidates than k exist: the result holds all of them, only their order is at stake
        if (lane < n_in) { sd[lane] = dl; si[lane] = jl; }
        T = n_in;
    }
    const double R2 = wmax(dl);
    const p2w_grid g = *gridp;
    int lo_c[3], hi_c[3];
    {
        const double R = sqrt(R2) * (1.0 + 1e-12) + 1e-300;
        const double ql[3] = {qx - ox - (double)g.lo[0], qy - oy - (double)g.lo[1], qz - oz - (double)g.lo[2]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // cells were taken from fp32 local coordinates by fp32 arithmetic: allow for both roundings (in cells)
            const double mrg = 1e-6 + 2.4e-7 * (double)g.dims[a];
            const double top = (double)(g.dims[a] - 1);
            lo_c[a] = (int)fmin(fmax(floor((ql[a] - R) / (double)g.res - mrg), 0.0), top);
            hi_c[a] = (int)fmin(fmax(floor((ql[a] + R) / (double)g.res + mrg), 0.0), top);
        }
    }
    const int ny = hi_c[1] - lo_c[1] + 1, nz = hi_c[2] - lo_c[2] + 1;
    const long long rows = (long long)ny * nz;
    auto start = [&](unsigned long long key) {
        return cell_start ? cell_start[key] : lower_bound_key(keys, 0, nc, key);
    };
    // visit(p, d, act): every candidate position p of the rows the ball touches, lanes in lockstep.  A lane owns a grid row (its run
    // [a, b) of the cell-sorted candidates); the runs are FLATTENED through a per-wave LDS list (each lane writes its run's
    // positions behind the wave's running total), so that consecutive lanes then measure consecutive candidates of a run:
    // 24-byte records of one run share cache lines, a lane per run fetched every line ~5 times from L2.
    int* fl = s_f[w];
    auto scan = [&](auto&& visit) {
        for (long long t0 = 0; t0 < rows; t0 += 64) {
            const long long t = t0 + lane;
            int a = 0, b = 0;
            if (t < rows) {
                const long long cy = lo_c[1] + t % ny, cz = lo_c[2] + t / ny;
                const unsigned long long base = (unsigned long long)((cz * g.dims[1] + cy) * g.dims[0]);
                a = start(base + lo_c[0]);
                b = start(base + hi_c[0] + 1);
            }
            const int len = b - a;
            int pre = len;                                   // inclusive scan of the run lengths over the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(pre, off);
                if (lane >= off) pre += o;
            }
            const int total = __shfl(pre, 63);
            pre -= len;                                      // exclusive: this lane's first flat index
            for (int base_f = 0; base_f < total; base_f += RF_FLAT) {
                // the part of every lane's run that falls into [base_f, base_f + RF_FLAT)
                const int s_lo = max(0, base_f - pre), s_hi = min(len, base_f + RF_FLAT - pre);
                for (int s2 = s_lo; s2 < s_hi; ++s2) fl[pre + s2 - base_f] = a + s2;
                __builtin_amdgcn_wave_barrier();
                const int n_f = min(RF_FLAT, total - base_f);
                for (int f0 = 0; f0 < n_f; f0 += 64) {
                    const bool act = f0 + lane < n_f;
                    const int p = act ? fl[f0 + lane] : 0;
                    const double d = act ? d2(cs + 3 * (size_t)p) : INFINITY;
                    visit(p, d, act);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    if (n_in >= k) {
        scan([&](int p, double d, bool act) {
            const bool hit = act && d <= R2;
            const unsigned long long bal = __ballot(hit);
            if (hit) {
                const int slot = T + __popcll(bal & ((1ull << lane) - 1ull));
                if (slot < RF_CAP) { sd[slot] = d; si[slot] = cidx[p]; }
            }
            T += __popcll(bal);
        });
    }
    __builtin_amdgcn_wave_barrier();
    if (T <= RF_CAP) {
        for (int e = lane; e < T; e += 64) {
            const double de = sd[e];
            const int ie = si[e];
            int rank = 0;
            for (int f = 0; f < T; ++f) {
                const double df = sd[f];
                const int jf = si[f];
                rank += (df < de || (df == de && jf < ie)) ? 1 : 0;
            }
            if (rank < k) nbr[(size_t)qi * k + rank] = ie;
        }
        if (lane == 0) deg[qi] = min(T, k);
        return;
    }
    // overflow: the k smallest (distance, index) pairs one after the other
    double pd = -1.0;
    int pi = -1;
    for (int r = 0; r < k; ++r) {
        double bd = INFINITY;
        int bi = 0x7fffffff;
        scan([&](int p, double d, bool act) {
            if (!act || d > R2) return;
            const int i = cidx[p];
            if (!(d > pd || (d == pd && i > pi))) return;       // not behind the previous pick
            if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
        });
        double md = bd;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) md = fmin(md, __shfl_xor(md, off));
        int mi = bd == md ? bi : 0x7fffffff;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mi = min(mi, __shfl_xor(mi, off));
        if (lane == 0) nbr[(size_t)qi * k + r] = mi;
        pd = md; pi = mi;
    }
    if (lane == 0) deg[qi] = k;
}

extern "C" int32_t p2w_knn_refine_f64(const double* cand_sorted, const int32_t* cand_index, const int32_t* cand_pos,
                                      const uint64_t* keys_sorted, const int32_t* cell_start, const p2w_grid* grid, double ox, double oy,
                                      double oz, const double* q, int32_t m, int32_t nc, int32_t k, int32_t* nbr, int32_t* deg,
                                      p2w_stream_t stream) {
    if (m == 0 || nc == 0) return P2W_OK;
    P2W_CHECK_PTR(cand_sorted); P2W_CHECK_PTR(cand_index); P2W_CHECK_PTR(cand_pos); P2W_CHECK_PTR(keys_sorted); P2W_CHECK_PTR(grid);
    P2W_CHECK_PTR(q); P2W_CHECK_PTR(nbr); P2W_CHECK_PTR(deg);
    if (m < 0 || nc < 0 || k <= 0 || k > P2W_MAX_K) return P2W_EINVAL;
    knn_refine_kernel<<<p2w_cdiv(m, 4), 256, 0, p2w_s(stream)>>>(cand_sorted, cand_index, cand_pos,
                                                                reinterpret_cast<const unsigned long long*>(keys_sorted), cell_start, grid,
                                                                ox, oy, oz, q, m, nc, k, nbr, deg);
    return P2W_LAUNCH_STATUS();
}

__global__ __launch_bounds__(256) void fill_batch_nbr_kernel(const int* __restrict__ batch, int m, int* __restrict__ nbr,
                                                             int* __restrict__ deg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    nbr[i] = batch[i];
    deg[i] = 1;
}

extern "C" int32_t p2w_fill_batch_nbr(const int32_t* batch, int32_t m, int32_t* nbr, int32_t* deg, p2w_stream_t stream) {
    if (m == 0) return P2W_OK;
    P2W_CHECK_PTR(batch); P2W_CHECK_PTR(nbr); P2W_CHECK_PTR(deg);
    if (m < 0) return P2W_EINVAL;
    fill_batch_nbr_kernel<<<p2w_cdiv(m, 256), 256, 0, p2w_s(stream)>>>(batch, m, nbr, deg);
    return P2W_LAUNCH_STATUS();
}
