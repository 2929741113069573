// Micro-benchmark (tools/store_granule.py): how fast does the chip take an epilogue-shaped store stream?  A [rows x 2048 B] matrix
// (the 4 F-wide H rows of level 1) is written once by 512 workgroups of 256 threads, each wave owning 64-row x 256-byte blocks like a
// GEMM wave tile, in four store shapes:
//   0: per instruction 4 rows x 64 B (16 lanes x 4 B), the other 64 B of the same lines by the next instruction   (the H epilogue today)
//   1: per instruction 4 rows x 128 B (16 lanes x 8 B): whole lines                                                (the fp32 epilogue)
//   2: per instruction 4 rows x 256 B (16 lanes x 16 B)
//   3: per instruction 1 row x 1 KiB (64 lanes x 16 B): a streaming store
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(char* __restrict__ out, int rows, int row_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, kg = lane >> 4;
    // a workgroup tile = 128 rows x 512 B (wave = 64 rows x 256 B, 2 x 2 waves); 4 column tiles per row tile
    for (int t = blockIdx.x; t < row_tiles * 4; t += gridDim.x) {
        const int rt = t >> 2, ct = t & 3;
        const int row0 = rt * 128 + (wave >> 1) * 64;
        const size_t col0 = (size_t)ct * 512 + (wave & 1) * 256;
        if (MODE == 3) {
            for (int r = 0; r < 64; r += 4) {           // 4 rows x 256 B per instruction in 1 KiB: rows r..r+3, lane -> (row r + lane / 16, 16 B)
                const int row = row0 + r + kg;
                if (row < rows) *reinterpret_cast<uint4*>(out + (size_t)row * 2048 + col0 + 16 * c16) = make_uint4(lane, r, t, 1);
            }
        } else {
            for (int st = 0; st < 16; ++st) {
                const int row = row0 + 16 * (st >> 2) + 4 * kg + (st & 3);
                if (row >= rows) continue;
                char* p = out + (size_t)row * 2048 + col0;
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint32_t*>(p + 64 * q + 4 * c16) = (uint32_t)(lane + st + q);
                } else if (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2*>(p + 128 * q + 8 * c16) = make_uint2(lane, st + q);
                } else {
                    *reinterpret_cast<uint4*>(p + 16 * c16) = make_uint4(lane, st, t, 2);
                }
            }
        }
    }
}

extern "C" int store_granule(void* out, int rows, int mode, int grid, hipStream_t s) {
    const int row_tiles = (rows + 127) / 128;
    char* o = static_cast<char*>(out);
    if (mode == 0) store_kernel<0><<<grid, 256, 0, s>>>(o, rows, row_tiles);
    else if (mode == 1) store_kernel<1><<<grid, 256, 0, s>>>(o, rows, row_tiles);
    else if (mode == 2) store_kernel<2><<<grid, 256, 0, s>>>(o, rows, row_tiles);
    else store_kernel<3><<<grid, 256, 0, s>>>(o, rows, row_tiles);
    return (int)hipGetLastError();
}
