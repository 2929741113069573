// Shared helpers for the gfx950 kernels of libp2w_gfx950.so.
// Built with -ffp-contract=off: every fp32 product/sum below is individually rounded unless it is
// written as an explicit fmaf(); the geometry kernels depend on that for bit-exact neighbour sets.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/p2w.h"

#define P2W_WAVE 64

#define P2W_CHECK_PTR(p) do { if ((p) == nullptr) return P2W_ENULL; } while (0)
#define P2W_CHECK_ALIGN16(p) do { if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) return P2W_EALIGN; } while (0)
#define P2W_LAUNCH_STATUS() static_cast<int32_t>(hipGetLastError())

static inline hipStream_t p2w_s(p2w_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int p2w_cdiv(long a, long b) { return static_cast<int>((a + b - 1) / b); }

// squared distance in the normative form ((dx*dx)+(dy*dy))+(dz*dz)
__device__ __forceinline__ float p2w_d2(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return ((dx * dx) + (dy * dy)) + (dz * dz);
}

// largest b with ptr[b] <= i  (ptr is a non-decreasing CSR offset array of B+1 entries, i < ptr[B])
__device__ __forceinline__ int p2w_find_segment(const int* __restrict__ ptr, int B, int i) {
    int lo = 0, hi = B;  // answer in [lo, hi)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}
