// Device-wide primitives of the geometry / plot-level kernels, hand-written for gfx950 (wave64): a STABLE least-significant-
// digit radix sort of (64-bit key, 32-bit value) pairs and an exclusive scan of int32.  They replace rocPRIM in the sort
// sampler (p2w_voxel_sample / p2w_consecutive_cluster), the Morton ordering of the back-projection (p2w_morton_order) and
// the voxeliser (p2w_sort_pairs_u64, p2w_voxel_runs).
//
// Radix sort: 8-bit digits, one pass = histogram (per 4096-key tile) -> scan of the [digit][tile] table -> stable scatter.
//   * the number of passes follows the DATA: an OR over the keys gives the highest set bit, passes above it return at once
//     (cell keys of a plot need 3-4 passes, Morton keys 6-8); every pass is enqueued, nothing is read back;
//   * stability inside a tile without sorting it: a wave owns 1024 consecutive keys and walks them 64 at a time; the lanes
//     that hold the same digit find each other with eight ballots (one per digit bit), a lane's rank among them is a
//     popcount, the running per-digit offsets sit in LDS (one row per wave, updated by the group's first lane);
//   * buffers ping-pong between `out` and a temporary: pass 0 reads the input, the last pass may end in the temporary, a
//     final kernel copies it over (or copies the input when no pass ran: all keys zero).
// The element count may live on the device (n_dev): only the first min(*n_dev, n_bound) pairs are sorted / written.
#pragma once
#include "p2w_common.h"

constexpr int RS_BLOCK = 256, RS_TILE = 4096, RS_RADIX = 256;
constexpr int RS_SCAN_ONE = 8192;    // histogram tables up to this many entries (131 k keys) are scanned by one workgroup in one round; above, two levels (a lone
                                     // workgroup walks 65 536 entries in 169 us - per digit pass: 0.2 s of the 10 M-point plot's sampler sorts)

struct RsCtl { unsigned long long orv; int pad[2]; };   // OR of all keys (-> number of passes), zeroed by the host call

__device__ __forceinline__ int rs_passes(const RsCtl* c) {
    const unsigned long long v = c->orv;
    return v ? (64 - __clzll((long long)v) + 7) >> 3 : 0;
}
__device__ __forceinline__ int rs_count(const int* n_dev, int n_bound) {
    if (!n_dev) return n_bound;
    const int n = *n_dev;
    return n < n_bound ? (n < 0 ? 0 : n) : n_bound;
}

__global__ __launch_bounds__(RS_BLOCK) void rs_or_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ n_dev,
                                                         int n_bound, RsCtl* __restrict__ ctl) {
    const int n = rs_count(n_dev, n_bound);
    unsigned long long v = 0;
    for (long long i = (long long)blockIdx.x * RS_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * RS_BLOCK) v |= keys[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicOr(&ctl->orv, v);
}

// src of pass p: the input (p == 0), else the buffer pass p-1 wrote: tmp after even passes, out after odd ones
template <typename T>
__device__ __forceinline__ const T* rs_src(int p, const T* in, const T* out, const T* tmp) { return p == 0 ? in : ((p & 1) ? tmp : out); }
template <typename T>
__device__ __forceinline__ T* rs_dst(int p, T* out, T* tmp) { return (p & 1) ? out : tmp; }

__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(const unsigned long long* __restrict__ in, const unsigned long long* __restrict__ out,
                                                           const unsigned long long* __restrict__ tmp, const int* __restrict__ n_dev,
                                                           int n_bound, int pass, const RsCtl* __restrict__ ctl, int nblk,
                                                           int* __restrict__ hist) {
    __shared__ int h[RS_RADIX];
    if (pass >= rs_passes(ctl)) return;
    const int n = rs_count(n_dev, n_bound);
    const long long t0 = (long long)blockIdx.x * RS_TILE;
    if (t0 >= n) { hist[threadIdx.x * nblk + blockIdx.x] = 0; return; }   // (RS_BLOCK == RS_RADIX: one digit per thread)
    const unsigned long long* src = rs_src(pass, in, out, tmp);
    h[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 8 * pass;
#pragma unroll 4
    for (int e = 0; e < RS_TILE / RS_BLOCK; ++e) {
        const long long i = t0 + e * RS_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&h[(int)((src[i] >> shift) & 255ull)], 1);
    }
    __syncthreads();
    hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `len` ints in place by ONE workgroup (8 per thread per round), optional total
__device__ __forceinline__ void rs_block_scan_inplace(int* __restrict__ a, long long len, int* __restrict__ total_out) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (long long base = 0; base < len; base += 8192) {
        const long long i0 = base + (long long)threadIdx.x * 8;
        int v[8], s = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = (i0 + e < len) ? a[i0 + e] : 0; s += v[e]; }
        int inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int before = carry_s, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) before += wsum[w]; tot += wsum[w]; }
        int run = before + inc - s;
#pragma unroll
        for (int e = 0; e < 8; ++e) { if (i0 + e < len) a[i0 + e] = run; run += v[e]; }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
    if (total_out && threadIdx.x == 0) *total_out = carry_s;
}

__global__ __launch_bounds__(1024) void rs_scan_kernel(int* __restrict__ hist, int nblk, int pass, const RsCtl* __restrict__ ctl) {
    if (pass >= rs_passes(ctl)) return;
    rs_block_scan_inplace(hist, (long long)RS_RADIX * nblk, nullptr);
}
// Above RS_SCAN_ONE entries the [digit][tile] table is scanned on two levels like xs_exclusive_scan (4096-entry tiles by as many workgroups, one
// workgroup over the tile sums, add): at plot scale (19 M keys: 1.19 M table entries) a single workgroup would walk 145
// rounds of 8192 entries per digit pass on one CU while the chip idles.  In place: a thread loads its 4 entries before it
// stores them (no __restrict__ on the table).
__global__ __launch_bounds__(1024) void rs_scan_tile_kernel(int* hist, int len, int pass, const RsCtl* __restrict__ ctl,
                                                            int* __restrict__ tsum) {
    __shared__ int wsum[16];
    if (pass >= rs_passes(ctl)) return;
    const long long i0 = (long long)blockIdx.x * 4096 + (long long)threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (i0 + e < len) ? hist[i0 + e] : 0; s += v[e]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) before += wsum[w]; tot += wsum[w]; }
    int run = before + inc - s;
#pragma unroll
    for (int e = 0; e < 4; ++e) { if (i0 + e < len) hist[i0 + e] = run; run += v[e]; }
    if (threadIdx.x == 0) tsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void rs_scan_sums_kernel(int* __restrict__ tsum, int nt, int pass, const RsCtl* __restrict__ ctl) {
    if (pass >= rs_passes(ctl)) return;
    rs_block_scan_inplace(tsum, nt, nullptr);
}
__global__ __launch_bounds__(256) void rs_scan_add_kernel(int* __restrict__ hist, int len, int pass, const RsCtl* __restrict__ ctl,
                                                          const int* __restrict__ tsum) {
    if (pass >= rs_passes(ctl)) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < len) hist[i] += tsum[i >> 12];
}

// One digit pass over a 4096-key tile.  The tile's pairs are read ONCE into registers (16 per thread, coalesced), ranked with the
// ballot scheme of the header comment into their position in the tile's digit-sorted order, parked there in LDS, and written out in
// that order: the keys of one digit leave the workgroup as one contiguous run of the destination (16 keys = 128 bytes on average)
// instead of ~50 partial segments per wave instruction (round 4: the direct scatter moved 32 B per pair at 0.84 TB/s and was
// 1.3 - 2.8 x slower than rocPRIM at plot scale, tools/sort_bench.py).
__global__ __launch_bounds__(RS_BLOCK) void rs_scatter_kernel(const unsigned long long* __restrict__ kin, const unsigned long long* __restrict__ kout,
                                                              const unsigned long long* __restrict__ ktmp, unsigned long long* __restrict__ kout_w,
                                                              unsigned long long* __restrict__ ktmp_w, const int* __restrict__ vin,
                                                              const int* __restrict__ vout, const int* __restrict__ vtmp,
                                                              int* __restrict__ vout_w, int* __restrict__ vtmp_w,
                                                              const int* __restrict__ n_dev, int n_bound, int pass,
                                                              const RsCtl* __restrict__ ctl, int nblk, const int* __restrict__ hist) {
    constexpr int PER = RS_TILE / 4 / 64;               // 64-key chunks per wave (a wave owns 1024 consecutive keys)
    __shared__ int cnt[4][RS_RADIX];                    // digits per wave quarter of the tile, then the running tile-local offsets
    __shared__ int gbase[RS_RADIX];                     // destination of the tile's first key of digit d, minus its tile-local position
    __shared__ int wsum[4];
    __shared__ unsigned long long skey[RS_TILE];        // the tile in digit order
    __shared__ int sval[RS_TILE];
    if (pass >= rs_passes(ctl)) return;
    const int n = rs_count(n_dev, n_bound);
    const long long t0 = (long long)blockIdx.x * RS_TILE;
    if (t0 >= n) return;
    const unsigned long long* ks = rs_src(pass, kin, kout, ktmp);
    unsigned long long* kd = rs_dst(pass, kout_w, ktmp_w);
    const int* vs = rs_src(pass, vin, vout, vtmp);          // vin == nullptr: the values are the indices 0..n-1 (pass 0 only)
    int* vd = rs_dst(pass, vout_w, vtmp_w);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int shift = 8 * pass;
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const long long w0 = t0 + (long long)wave * (RS_TILE / 4);
    unsigned long long key[PER];
    int val[PER];
#pragma unroll
    for (int c = 0; c < PER; ++c) {
        const long long i = w0 + c * 64 + lane;
        const bool valid = i < n;
        key[c] = valid ? ks[i] : ~0ull;
        val[c] = valid ? (vs ? vs[i] : (int)i) : 0;
        if (valid) atomicAdd(&cnt[wave][(int)((key[c] >> shift) & 255ull)], 1);
    }
    __syncthreads();
    {   // thread d: the tile's keys of digit d start at tile position (exclusive scan of the tile's digit counts) and at `hist` globally
        int c[4], tc = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { c[w] = cnt[w][threadIdx.x]; tc += c[w]; }
        int inc = tc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        int run = before + inc - tc;                    // tile-local start of digit threadIdx.x
        gbase[threadIdx.x] = hist[threadIdx.x * nblk + blockIdx.x] - run;
#pragma unroll
        for (int w = 0; w < 4; ++w) { cnt[w][threadIdx.x] = run; run += c[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < PER; ++c) {
        const bool valid = w0 + c * 64 + lane < n;
        const int d = (int)((key[c] >> shift) & 255ull);
        unsigned long long peers = __ballot(valid);
        if (!peers) break;                                   // (whole chunks past the end: wave-uniform)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot(valid && ((d >> b) & 1));
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        if (valid) {
            const int rank = __popcll(peers & ((1ull << lane) - 1ull));
            const int base = cnt[wave][d];                   // read by every lane of the group before its first lane advances it:
            const int pos = base + rank;                     // LDS operations of one wave execute in program order
            skey[pos] = key[c];
            sval[pos] = val[c];
            if (rank == 0) cnt[wave][d] = base + __popcll(peers);
        }
    }
    __syncthreads();
    const int tile_n = (int)(n - t0 < RS_TILE ? n - t0 : RS_TILE);
    for (int p = threadIdx.x; p < tile_n; p += RS_BLOCK) {
        const unsigned long long k = skey[p];
        const int dst = gbase[(int)((k >> shift) & 255ull)] + p;
        kd[dst] = k;
        vd[dst] = sval[p];
    }
}

__global__ __launch_bounds__(RS_BLOCK) void rs_finish_kernel(const unsigned long long* __restrict__ kin, unsigned long long* __restrict__ kout,
                                                             const unsigned long long* __restrict__ ktmp, const int* __restrict__ vin,
                                                             int* __restrict__ vout, const int* __restrict__ vtmp,
                                                             const int* __restrict__ n_dev, int n_bound, const RsCtl* __restrict__ ctl) {
    const int P = rs_passes(ctl);
    if (P > 0 && (P & 1) == 0) return;                       // an even number of passes ended in `out`
    const int n = rs_count(n_dev, n_bound);
    for (long long i = (long long)blockIdx.x * RS_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * RS_BLOCK) {
        kout[i] = P ? ktmp[i] : kin[i];
        vout[i] = P ? vtmp[i] : (vin ? vin[i] : (int)i);
    }
}

struct RsLayout { size_t ctl, hist, tsum, ktmp, vtmp, bytes; int nblk; };
static inline void rs_layout(long long n_bound, RsLayout* L) {
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const long long n = n_bound > 0 ? n_bound : 1;
    L->nblk = (int)((n + RS_TILE - 1) / RS_TILE);
    size_t o = 0;
    L->ctl = o; o += 256;
    L->hist = o; o += up(sizeof(int) * (size_t)RS_RADIX * L->nblk);
    L->tsum = o; o += up(sizeof(int) * (((size_t)RS_RADIX * L->nblk + 4095) / 4096));
    L->ktmp = o; o += up(sizeof(unsigned long long) * (size_t)n);
    L->vtmp = o; o += up(sizeof(int) * (size_t)n);
    L->bytes = o;
}

// Stable ascending sort of the first min(*n_dev, n_bound) (keys, values) pairs; vals_in == nullptr: values = 0..n-1 (an
// argsort).  keys_in / vals_in are not modified; `ws` holds rs_layout(n_bound).bytes bytes.
static inline hipError_t rs_sort_pairs(void* ws, const unsigned long long* keys_in, unsigned long long* keys_out, const int* vals_in,
                                       int* vals_out, const int* n_dev, int n_bound, hipStream_t s) {
    if (n_bound <= 0) return hipSuccess;
    RsLayout L;
    rs_layout(n_bound, &L);
    char* w = static_cast<char*>(ws);
    auto* ctl = reinterpret_cast<RsCtl*>(w + L.ctl);
    int* hist = reinterpret_cast<int*>(w + L.hist);
    int* tsum = reinterpret_cast<int*>(w + L.tsum);
    const int hlen = RS_RADIX * L.nblk, ht = (hlen + 4095) / 4096;
    auto* ktmp = reinterpret_cast<unsigned long long*>(w + L.ktmp);
    int* vtmp = reinterpret_cast<int*>(w + L.vtmp);
    hipError_t e = hipMemsetAsync(ctl, 0, sizeof(RsCtl), s);
    if (e != hipSuccess) return e;
    const int g1 = L.nblk < 1024 ? L.nblk : 1024;
    rs_or_kernel<<<g1, RS_BLOCK, 0, s>>>(keys_in, n_dev, n_bound, ctl);
    for (int p = 0; p < 8; ++p) {
        rs_hist_kernel<<<L.nblk, RS_BLOCK, 0, s>>>(keys_in, keys_out, ktmp, n_dev, n_bound, p, ctl, L.nblk, hist);
        if (hlen <= RS_SCAN_ONE) {
            rs_scan_kernel<<<1, 1024, 0, s>>>(hist, L.nblk, p, ctl);
        } else {
            rs_scan_tile_kernel<<<ht, 1024, 0, s>>>(hist, hlen, p, ctl, tsum);
            rs_scan_sums_kernel<<<1, 1024, 0, s>>>(tsum, ht, p, ctl);
            rs_scan_add_kernel<<<(hlen + 255) / 256, 256, 0, s>>>(hist, hlen, p, ctl, tsum);
        }
        rs_scatter_kernel<<<L.nblk, RS_BLOCK, 0, s>>>(keys_in, keys_out, ktmp, keys_out, ktmp, vals_in, vals_out, vtmp, vals_out, vtmp,
                                                       n_dev, n_bound, p, ctl, L.nblk, hist);
    }
    rs_finish_kernel<<<g1, RS_BLOCK, 0, s>>>(keys_in, keys_out, ktmp, vals_in, vals_out, vtmp, n_dev, n_bound, ctl);
    return hipGetLastError();
}

// ---- exclusive scan of int32 (two levels: 4096-element tiles, one workgroup over the tile sums, add) ----------------------
// (`in` may be `out`: a thread loads its 4 values before it stores them, hence no __restrict__ on the two)
__global__ __launch_bounds__(1024) void xs_tile_kernel(const int* in, int* out, int n, int* __restrict__ tsum) {
    __shared__ int wsum[16];
    const long long i0 = (long long)blockIdx.x * 4096 + (long long)threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (i0 + e < n) ? in[i0 + e] : 0; s += v[e]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) before += wsum[w]; tot += wsum[w]; }
    int run = before + inc - s;
#pragma unroll
    for (int e = 0; e < 4; ++e) { if (i0 + e < n) out[i0 + e] = run; run += v[e]; }
    if (threadIdx.x == 0) tsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void xs_sums_kernel(int* __restrict__ tsum, int nt) { rs_block_scan_inplace(tsum, nt, nullptr); }
__global__ __launch_bounds__(256) void xs_add_kernel(int* __restrict__ out, int n, const int* __restrict__ tsum) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += tsum[i >> 12];
}
static inline size_t xs_ws_bytes(long long n) { return (size_t)(((n > 0 ? n : 1) + 4095) / 4096) * sizeof(int) + 256; }
// out[i] = sum of in[0..i-1] for i < n (in place allowed: in == out); `ws`: xs_ws_bytes(n)
static inline hipError_t xs_exclusive_scan(void* ws, const int* in, int* out, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int* tsum = static_cast<int*>(ws);
    const int nt = (int)(((long long)n + 4095) / 4096);
    xs_tile_kernel<<<nt, 1024, 0, s>>>(in, out, n, tsum);
    if (nt > 1) {
        xs_sums_kernel<<<1, 1024, 0, s>>>(tsum, nt);
        xs_add_kernel<<<(int)(((long long)n + 255) / 256), 256, 0, s>>>(out, n, tsum);
    }
    return hipGetLastError();
}
