// Feature kernels: stem, MFMA GEMM + fused epilogue, fused PointNetConv (gather + edge MLP + segmented max),
// kNN-interpolation + concat, segment max, row dot.  gfx950 only.
//
// MFMA core: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate - the parity mode).
//   A operand: lane l supplies A[row = l&31][k = l>>5];  B operand: B[k = l>>5][col = l&31];
//   C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5), reg in [0,16).
// Both operands are staged k-contiguous in LDS ([row][k] and [col][k], row stride 36 floats = one
// ds_read_b128 of padding -> conflict-free) so one 16-byte LDS read feeds 4 MFMA steps: within an
// 8-wide k group, lane half h takes k = 4h..4h+3 and step s uses element s (the k order inside the
// group is permuted identically for A and B, which leaves the sum unchanged).
#include "p2w_common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G_BM = 128, G_BN = 128, G_BK = 32, G_LD = 36;

extern "C" void p2w_packed_dims(int32_t N, int32_t K, int32_t* N_pad, int32_t* K_pad) {
    if (N_pad) *N_pad = (N + 255) / 256 * 256;  // widest column tile of any kernel
    if (K_pad) *K_pad = (K + G_BK - 1) / G_BK * G_BK;
}

// XCD-aware tile order: blocks L, L+8, L+16.. share an XCD (round-robin dispatch), i.e. one 4 MiB L2.
//  mode 0 (W fits in L2): an XCD owns whole row tiles and walks their column tiles back to back -> the A row tile is
//          fetched once, W is always an L2 hit.
//  mode 1 (W larger than L2, few rows): an XCD owns a slice of column tiles (its W slice stays L2-resident) and sweeps
//          ALL row tiles; A is streamed once per XCD slice instead of W once per row tile.
__device__ __forceinline__ bool tile_coords(int nMt, int nNt, int* mt, int* nt, int mode = 0) {
    const int L = blockIdx.x;
    const int xcd = L & 7, w = L >> 3;
    if (mode == 0) {
        *mt = xcd + 8 * (w / nNt);
        *nt = w % nNt;
        return *mt < nMt;
    }
    if (nNt >= 8) {
        const int cpx = (nNt + 7) >> 3;     // column tiles per XCD
        *mt = w / cpx;
        *nt = xcd + 8 * (w % cpx);
        return *mt < nMt && *nt < nNt;
    }
    const int r = 8 / nNt;                  // XCDs sharing one column tile (nNt in {1, 2, 4})
    *nt = xcd % nNt;
    *mt = xcd / nNt + r * w;
    return *mt < nMt;
}
static inline int tile_grid(int nMt, int nNt, int mode = 0) {
    if (mode == 0) return 8 * ((nMt + 7) / 8) * nNt;
    if (nNt >= 8) return 8 * nMt * ((nNt + 7) / 8);
    const int r = 8 / nNt;
    return 8 * ((nMt + r - 1) / r);
}

// one BK-slab of MFMAs for a 64x64 wave tile
__device__ __forceinline__ void mma_slab(const float* __restrict__ As, const float* __restrict__ Bs, int wr, int wc, int lane,
                                         f32x16 (&acc)[2][2]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < G_BK; kk += 8) {
        const float4 a0 = *reinterpret_cast<const float4*>(&As[(wr * 64 + r) * G_LD + kk + 4 * h]);
        const float4 a1 = *reinterpret_cast<const float4*>(&As[(wr * 64 + 32 + r) * G_LD + kk + 4 * h]);
        const float4 b0 = *reinterpret_cast<const float4*>(&Bs[(wc * 64 + r) * G_LD + kk + 4 * h]);
        const float4 b1 = *reinterpret_cast<const float4*>(&Bs[(wc * 64 + 32 + r) * G_LD + kk + 4 * h]);
        const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
        const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void load_w_tile(const float* __restrict__ Wp, int Kpad, int n0, int k0, int lrow, int lkq,
                                            float4 (&rb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        rb[i] = *reinterpret_cast<const float4*>(&Wp[(size_t)(n0 + lrow + 32 * i) * Kpad + k0 + 4 * lkq]);
}

__device__ __forceinline__ void store_tile(float* __restrict__ S, int lrow, int lkq, const float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&S[(lrow + 32 * i) * G_LD + 4 * lkq]) = r[i];
}

// ------------------------------------------------------------------------------------------------
// GEMM + epilogue
// ------------------------------------------------------------------------------------------------
struct EpiArgs {
    const float *bias, *sc0, *sh0, *sc1, *sh1, *residual;
    int ldr, relu0, relu1, relu2, relu_final;
};

// v = acc * wscale + bias; relu0; v*sc0+sh0; relu1; v*sc1+sh1; relu2; + residual; relu_final   (p2w_epilogue in p2w.h)
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[2][2], const EpiArgs& ep, float wscale, int row0, int col0,
                                              int lane, int M, int N, float* __restrict__ out, int ldo) {
    const int h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + j * 32 + (lane & 31);
        if (col >= N) continue;
        const float bias = ep.bias ? ep.bias[col] : 0.f;
        const float s0 = ep.sc0 ? ep.sc0[col] : 1.f, t0 = ep.sc0 ? ep.sh0[col] : 0.f;
        const float s1 = ep.sc1 ? ep.sc1[col] : 1.f, t1 = ep.sc1 ? ep.sh1[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M) continue;
                float v = fmaf(acc[i][j][r], wscale, bias);
                if (ep.relu0) v = fmaxf(v, 0.f);
                if (ep.sc0) { v = fmaf(v, s0, t0); }
                if (ep.relu1) v = fmaxf(v, 0.f);
                if (ep.sc1) { v = fmaf(v, s1, t1); }
                if (ep.relu2) v = fmaxf(v, 0.f);
                if (ep.residual) v += ep.residual[(size_t)row * ep.ldr + col];
                if (ep.relu_final) v = fmaxf(v, 0.f);
                out[(size_t)row * ldo + col] = v;
            }
        }
    }
}

__device__ __forceinline__ void load_a_tile(const float* __restrict__ A, int lda, int M, int K, int m0, int k0, int lrow,
                                            int lkq, float4 (&ra)[4]) {
    const int k = k0 + 4 * lkq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + lrow + 32 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < M && k < K) {
            const float* p = A + (size_t)row * lda + k;
            if (k + 3 < K) v = *reinterpret_cast<const float4*>(p);
            else { v.x = p[0]; if (k + 1 < K) v.y = p[1]; if (k + 2 < K) v.z = p[2]; }
        }
        ra[i] = v;
    }
}

__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Wp,
                                                   int M, int N, int K, int Kpad, int nMt, int nNt, EpiArgs ep,
                                                   float* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) float As[G_BM * G_LD];
    __shared__ __attribute__((aligned(16))) float Bs[G_BN * G_LD];
    int mt, nt;
    if (!tile_coords(nMt, nNt, &mt, &nt)) return;
    const int m0 = mt * G_BM, n0 = nt * G_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 3, lkq = tid & 7;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4], rb[4];
    load_a_tile(A, lda, M, K, m0, 0, lrow, lkq, ra);
    load_w_tile(Wp, Kpad, n0, 0, lrow, lkq, rb);
    for (int k0 = 0; k0 < Kpad; k0 += G_BK) {
        __syncthreads();
        store_tile(As, lrow, lkq, ra);
        store_tile(Bs, lrow, lkq, rb);
        __syncthreads();
        if (k0 + G_BK < Kpad) {  // next slab's global loads fly under this slab's MFMAs
            load_a_tile(A, lda, M, K, m0, k0 + G_BK, lrow, lkq, ra);
            load_w_tile(Wp, Kpad, n0, k0 + G_BK, lrow, lkq, rb);
        }
        mma_slab(As, Bs, wr, wc, lane, acc);
    }
    gemm_epilogue(acc, ep, 1.0f, m0 + wr * 64, n0 + wc * 64, lane, M, N, out, ldo);
}

extern "C" int32_t p2w_gemm(const float* A, int32_t lda, const float* Wp, int32_t M, int32_t N, int32_t K,
                            const p2w_epilogue* epi, float* out, int32_t ldo, p2w_stream_t stream) {
    if (M == 0) return P2W_OK;
    P2W_CHECK_PTR(A); P2W_CHECK_PTR(Wp); P2W_CHECK_PTR(out);
    P2W_CHECK_ALIGN16(A); P2W_CHECK_ALIGN16(Wp);
    if (M < 0 || N <= 0 || K <= 0 || lda < K || ldo < N || (lda & 3) != 0) return P2W_EINVAL;
    EpiArgs ep = {};
    if (epi) {
        if ((epi->sc0 && !epi->sh0) || (epi->sc1 && !epi->sh1)) return P2W_ENULL;
        if (epi->residual && epi->ldr < N) return P2W_EINVAL;
        ep = {epi->bias, epi->sc0, epi->sh0, epi->sc1, epi->sh1, epi->residual,
              epi->ldr, epi->relu0, epi->relu1, epi->relu2, epi->relu_final};
    }
    int Npad, Kpad;
    p2w_packed_dims(N, K, &Npad, &Kpad);
    const int nMt = p2w_cdiv(M, G_BM), nNt = p2w_cdiv(N, G_BN);
    gemm_kernel<<<tile_grid(nMt, nNt), 256, 0, p2w_s(stream)>>>(A, lda, Wp, M, N, K, Kpad, nMt, nNt, ep, out, ldo);
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// fused PointNetConv: rows of the GEMM = (target, neighbour slot); a 32-row MFMA tile = one target,
// so the max over neighbours is a max over the accumulator tile's rows (16 registers + one lane^32
// exchange) and the [E, C] edge tensors of the reference never exist in HBM.
// ------------------------------------------------------------------------------------------------
// fp16 hi/lo split of two values at once: hi = round-toward-zero fp16 of v (v_cvt_pkrtz_f16_f32 converts a PAIR per
// instruction and, rounding toward zero, saturates at +-65504 instead of overflowing to inf), lo = round-to-nearest
// fp16 of the exact fp32 remainder v - hi (v_cvt_pk_f16_f32, also a pair per instruction; nearest keeps the split
// unbiased).  hi + lo reproduces v to <= 2^-22 relative; |v| up to ~1.3e5 still splits exactly enough.
// 6 VALU per pair (2 packed conversions, 2 conversions back, 2 subtractions) instead of 14 with clamps and single
// conversions.
typedef __fp16 hpair __attribute__((ext_vector_type(2)));
typedef _Float16 hpairn __attribute__((ext_vector_type(2)));
typedef float fpair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const hpair h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const fpair rem = {a - (float)h[0], b - (float)h[1]};
#ifdef P2W_SPLIT_RTZ   // A/B (P2W_EXTRA_CFLAGS): remainder toward zero as well: -1.2 % feature time, but biased (worst
                      // golden logit error 1.1e-4 instead of 9.6e-5; the clamped round-to-nearest split it replaces: +0.8 %)
    const hpair l = __builtin_amdgcn_cvt_pkrtz(rem[0], rem[1]);
#else
    const hpairn l = __builtin_convertvector(rem, hpairn);
#endif
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void sa_split(float v, _Float16& hi, _Float16& lo) {
    unsigned h, l;
    split_pair(v, 0.f, h, l);
    hi = __builtin_bit_cast(_Float16, (unsigned short)(h & 0xffffu));
    lo = __builtin_bit_cast(_Float16, (unsigned short)(l & 0xffffu));
}
__device__ __forceinline__ unsigned sa_pack(_Float16 a, _Float16 b) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    h2v p = {a, b};
    return __builtin_bit_cast(unsigned, p);
}

// layer-2 bias + ReLU + BN affine, then max over the target's valid neighbour slots (rows of the 32-row MFMA tile)
__device__ __forceinline__ void sa_epilogue(const f32x16 (&acc)[2][2], float wscale, int t0, int n0, int wr, int wc, int lane,
                                            int M, int kw, const int* __restrict__ deg, int C2, const float* __restrict__ b2,
                                            const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                            float* __restrict__ out, int ldo, _Float16* __restrict__ out_h2 = nullptr,
                                            int ldh = 0) {
    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tgt = t0 + wr * 2 + i;
        if (tgt >= M) continue;
        const int d = min(deg[tgt], kw);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + (lane & 31);
            const bool cv = col < C2;
            const float bias = cv ? b2[col] : 0.f, s = cv ? bn_s[col] : 0.f, t = cv ? bn_t[col] : 0.f;
            float vmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int slot = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fmaf(fmaxf(fmaf(acc[i][j][r], wscale, bias), 0.f), s, t);
                if (slot < d) vmax = fmaxf(vmax, v);
            }
            vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
            if (d == 0) vmax = 0.f;
            if (cv && h == 0 && out) out[(size_t)tgt * ldo + col] = vmax;
            if (out_h2) {  // lanes (2p, 2p+1) hold adjacent columns: the even lane stores both as one word per plane
                const float nb = __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(vmax), 0xB1, 0xf, 0xf, false));
                if (h == 0 && (lane & 1) == 0 && col < ldh) {
                    unsigned hw, lw;
                    split_pair(vmax, nb, hw, lw);
                    _Float16* p = out_h2 + (size_t)tgt * (2 * ldh) + col;
                    *reinterpret_cast<unsigned*>(p) = hw;
                    *reinterpret_cast<unsigned*>(p + ldh) = lw;
                }
            }
        }
    }
}

// same as sa_epilogue with the per-column parameters and the two target degrees already in registers
template <int RT> struct SaEpiRegs { float bias[2], s[2], t[2]; int d[RT]; };
template <int RT>   // RT 32-row tiles (= targets) per wave
__device__ __forceinline__ void sa_epilogue_regs(const f32x16 (&acc)[RT][2], float wscale, int t0, int n0, int wr, int wc,
                                                 int lane, int M, const SaEpiRegs<RT>& e, int C2, float* __restrict__ out, int ldo,
                                                 _Float16* __restrict__ out_h2, int ldh) {
    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int tgt = t0 + wr * RT + i;
        if (tgt >= M) continue;
        const int d = e.d[i];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + (lane & 31);
            const bool cv = col < C2;
            float vmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int slot = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fmaf(fmaxf(fmaf(acc[i][j][r], wscale, e.bias[j]), 0.f), e.s[j], e.t[j]);
                if (slot < d) vmax = fmaxf(vmax, v);
            }
            vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
            if (d == 0) vmax = 0.f;
            if (cv && h == 0 && out) out[(size_t)tgt * ldo + col] = vmax;
            if (out_h2) {
                const float nb = __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(vmax), 0xB1, 0xf, 0xf, false));
                if (h == 0 && (lane & 1) == 0 && col < ldh) {
                    unsigned hw, lw;
                    split_pair(vmax, nb, hw, lw);
                    _Float16* p = out_h2 + (size_t)tgt * (2 * ldh) + col;
                    *reinterpret_cast<unsigned*>(p) = hw;
                    *reinterpret_cast<unsigned*>(p + ldh) = lw;
                }
            }
        }
    }
}

// per-row geometry of a 4-target row tile (pointnet.py:119-129): rows tid<128 = (target t0 + tid/32, slot tid%32);
// writes the source index and (normalised relative position, source reflectance) of every row to LDS
__device__ __forceinline__ void sa_row_geometry(int tid, int t0, int M, int kw, const float4* __restrict__ xyzr,
                                                const int* __restrict__ idx, const int* __restrict__ batch_dst,
                                                const float* __restrict__ sf, const int* __restrict__ nbr,
                                                const int* __restrict__ deg, int* m_j, float (*m_g)[4]) {
    if (tid < G_BM) {
        const int tgt = t0 + (tid >> 5), slot = tid & 31;
        int j = 0;
        float rx = 0.f, ry = 0.f, rz = 0.f, rf = 0.f, nrm = 0.f;
        if (tgt < M) {
            const int d = deg[tgt];
            const float s = sf[batch_dst[tgt]];
            const float4 pi = xyzr[idx[tgt]];
            const bool valid = slot < d && slot < kw;
            j = nbr[(size_t)tgt * kw + (valid ? slot : 0)];
            if (j < 0) j = idx[tgt];
            const float4 pj = xyzr[j];
            if (valid) {
                rx = pj.x / s - pi.x / s; ry = pj.y / s - pi.y / s; rz = pj.z / s - pi.z / s;
                nrm = sqrtf(((rx * rx) + (ry * ry)) + (rz * rz));
                rf = pj.w;
            }
        }
        float dmax = nrm;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));  // stays inside the 32-lane half
        const float den = dmax + 1e-8f;
        m_j[tid] = j;
        m_g[tid][0] = rx / den; m_g[tid][1] = ry / den; m_g[tid][2] = rz / den; m_g[tid][3] = rf;
    }
}

// A producer of the fused kernel: h1[row][k..k+3] = relu(P[j][k] + g . W1r[:, k])
__device__ __forceinline__ void sa_load_h1(const float* __restrict__ P, int ldp, const float* __restrict__ w1r4, int C1,
                                           int C1pad, int k, const int (&rj)[4], const float4 (&rg)[4], float4 (&ra)[4]) {
    if (k < C1) {
        const float4 wx = *reinterpret_cast<const float4*>(&w1r4[0 * C1pad + k]);
        const float4 wy = *reinterpret_cast<const float4*>(&w1r4[1 * C1pad + k]);
        const float4 wz = *reinterpret_cast<const float4*>(&w1r4[2 * C1pad + k]);
        const float4 wf = *reinterpret_cast<const float4*>(&w1r4[3 * C1pad + k]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 p = *reinterpret_cast<const float4*>(&P[(size_t)rj[i] * ldp + k]);
            const float4 g = rg[i];
            float4 v;
            v.x = fmaxf(fmaf(g.w, wf.x, fmaf(g.z, wz.x, fmaf(g.y, wy.x, fmaf(g.x, wx.x, p.x)))), 0.f);
            v.y = fmaxf(fmaf(g.w, wf.y, fmaf(g.z, wz.y, fmaf(g.y, wy.y, fmaf(g.x, wx.y, p.y)))), 0.f);
            v.z = fmaxf(fmaf(g.w, wf.z, fmaf(g.z, wz.z, fmaf(g.y, wy.z, fmaf(g.x, wx.z, p.z)))), 0.f);
            v.w = fmaxf(fmaf(g.w, wf.w, fmaf(g.z, wz.w, fmaf(g.y, wy.w, fmaf(g.x, wx.w, p.w)))), 0.f);
            ra[i] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__global__ __launch_bounds__(256) void sa_conv_kernel(const float* __restrict__ P, int ldp, const float4* __restrict__ xyzr,
                                                      const int* __restrict__ idx, const int* __restrict__ batch_dst,
                                                      const float* __restrict__ sf, const int* __restrict__ nbr,
                                                      const int* __restrict__ deg, int kw, int M,
                                                      const float* __restrict__ w1r4, int C1, int C1pad,
                                                      const float* __restrict__ W2p, int C2, int nMt, int nNt,
                                                      const float* __restrict__ b2, const float* __restrict__ bn_s,
                                                      const float* __restrict__ bn_t, float* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) float As[G_BM * G_LD];
    __shared__ __attribute__((aligned(16))) float Bs[G_BN * G_LD];
    __shared__ int m_j[G_BM];
    __shared__ float m_g[G_BM][4];  // normalised relative position (3) + reflectance of the source point
    int mt, nt;
    if (!tile_coords(nMt, nNt, &mt, &nt)) return;
    const int t0 = mt * 4, n0 = nt * G_BN;  // 4 targets per row tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 3, lkq = tid & 7;

    sa_row_geometry(tid, t0, M, kw, xyzr, idx, batch_dst, sf, nbr, deg, m_j, m_g);
    __syncthreads();
    int rj[4];
    float4 rg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rj[i] = m_j[lrow + 32 * i];
        rg[i] = *reinterpret_cast<const float4*>(&m_g[lrow + 32 * i][0]);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    sa_load_h1(P, ldp, w1r4, C1, C1pad, 4 * lkq, rj, rg, ra);
    load_w_tile(W2p, C1pad, n0, 0, lrow, lkq, rb);
    for (int k0 = 0; k0 < C1pad; k0 += G_BK) {
        __syncthreads();
        store_tile(As, lrow, lkq, ra);
        store_tile(Bs, lrow, lkq, rb);
        __syncthreads();
        if (k0 + G_BK < C1pad) {
            sa_load_h1(P, ldp, w1r4, C1, C1pad, k0 + G_BK + 4 * lkq, rj, rg, ra);
            load_w_tile(W2p, C1pad, n0, k0 + G_BK, lrow, lkq, rb);
        }
        mma_slab(As, Bs, wr, wc, lane, acc);
    }

    sa_epilogue(acc, 1.0f, t0, n0, wr, wc, lane, M, kw, deg, C2, b2, bn_s, bn_t, out, ldo);
}

extern "C" int32_t p2w_sa_conv(const float* P, int32_t ldp, const float* xyzr_src, const int32_t* idx, const int32_t* batch_dst,
                               const float* sf, const int32_t* nbr, const int32_t* deg, int32_t kw, int32_t M,
                               const float* w1r4, const float* W2p, int32_t C1, int32_t C2, const float* b2,
                               const float* bn_s, const float* bn_t, float* out, int32_t ldo, p2w_stream_t stream) {
    if (M == 0) return P2W_OK;
    P2W_CHECK_PTR(P); P2W_CHECK_PTR(xyzr_src); P2W_CHECK_PTR(idx); P2W_CHECK_PTR(batch_dst); P2W_CHECK_PTR(sf);
    P2W_CHECK_PTR(nbr); P2W_CHECK_PTR(deg); P2W_CHECK_PTR(w1r4); P2W_CHECK_PTR(W2p); P2W_CHECK_PTR(b2);
    P2W_CHECK_PTR(bn_s); P2W_CHECK_PTR(bn_t); P2W_CHECK_PTR(out);
    P2W_CHECK_ALIGN16(P); P2W_CHECK_ALIGN16(xyzr_src); P2W_CHECK_ALIGN16(w1r4); P2W_CHECK_ALIGN16(W2p);
    if (M < 0 || kw <= 0 || kw > 32 || C1 <= 0 || C2 <= 0 || (C1 & 3) || (ldp & 3) || ldp < C1 || ldo < C2) return P2W_EINVAL;
    int C2pad, C1pad;
    p2w_packed_dims(C2, C1, &C2pad, &C1pad);
    const int nMt = p2w_cdiv(M, 4), nNt = p2w_cdiv(C2, G_BN);
    sa_conv_kernel<<<tile_grid(nMt, nNt), 256, 0, p2w_s(stream)>>>(
        P, ldp, reinterpret_cast<const float4*>(xyzr_src), idx, batch_dst, sf, nbr, deg, kw, M, w1r4, C1, C1pad, W2p, C2,
        nMt, nNt, b2, bn_s, bn_t, out, ldo);
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// split-fp16 MFMA path ("f16x3"): a = a_hi + a_lo, w*2^e = w_hi + w_lo (fp16 pairs, ~22 mantissa bits);
//   a*w ~= (a_lo*w_hi + a_hi*w_lo + a_hi*w_hi) * 2^-e      three v_mfma_f32_32x32x16_f16, fp32 accumulate
// = fp32-class accuracy at 3/16 of the fp32-MFMA cycles.  Activations stay fp32 in HBM and are split while
// they are staged into LDS; weights are split (and scaled by a power of two so that w_lo stays in the normal
// fp16 range) once, when the checkpoint is packed: Wh[2][N_pad][K_pad] halfs, plane 0 = hi, plane 1 = lo.
//   A operand of 32x32x16: lane l supplies A[row l&31][k = 8*(l>>5) + 0..7] (16 contiguous bytes), B alike.
// LDS rows are 32 halfs + 8 pad = 80 bytes: 16-byte aligned fragments, conflict-free ds_read_b128.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr int H_LD = 40;

__device__ __forceinline__ void split_store(_Float16* __restrict__ Sh, _Float16* __restrict__ Sl, int lrow, int lkq,
                                            const float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        uint2 hi, lo;
        split_pair(v[0], v[1], hi.x, lo.x);
        split_pair(v[2], v[3], hi.y, lo.y);
        *reinterpret_cast<uint2*>(&Sh[(lrow + 32 * i) * H_LD + 4 * lkq]) = hi;
        *reinterpret_cast<uint2*>(&Sl[(lrow + 32 * i) * H_LD + 4 * lkq]) = lo;
    }
}

// W tile: 128 rows x 32 halfs per plane; thread -> rows (tid>>2) + 64*i, 8 halfs at k = 8*(tid&3)
__device__ __forceinline__ void load_w16_tile(const _Float16* __restrict__ Wh, size_t plane, int Kpad, int n0, int k0, int tid,
                                              h8 (&rb)[2][2]) {
    const int row = tid >> 2, kq = tid & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const size_t o = (size_t)(n0 + row + 64 * i) * Kpad + k0 + 8 * kq;
        rb[0][i] = *reinterpret_cast<const h8*>(&Wh[o]);
        rb[1][i] = *reinterpret_cast<const h8*>(&Wh[plane + o]);
    }
}
__device__ __forceinline__ void store_w16_tile(_Float16* __restrict__ Bh, _Float16* __restrict__ Bl, int tid,
                                               const h8 (&rb)[2][2]) {
    const int row = tid >> 2, kq = tid & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<h8*>(&Bh[(row + 64 * i) * H_LD + 8 * kq]) = rb[0][i];
        *reinterpret_cast<h8*>(&Bl[(row + 64 * i) * H_LD + 8 * kq]) = rb[1][i];
    }
}

__device__ __forceinline__ void mma16_slab(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                           const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl, int wr, int wc,
                                           int lane, f32x16 (&acc)[2][2]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < G_BK; kk += 16) {
        h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const h8*>(&Ah[(wr * 64 + 32 * i + r) * H_LD + kk + 8 * h]);
            al[i] = *reinterpret_cast<const h8*>(&Al[(wr * 64 + 32 * i + r) * H_LD + kk + 8 * h]);
            bh[i] = *reinterpret_cast<const h8*>(&Bh[(wc * 64 + 32 * i + r) * H_LD + kk + 8 * h]);
            bl[i] = *reinterpret_cast<const h8*>(&Bl[(wc * 64 + 32 * i + r) * H_LD + kk + 8 * h]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
    }
}

__global__ __launch_bounds__(256) void gemm16_kernel(const float* __restrict__ A, int lda, const _Float16* __restrict__ Wh,
                                                     size_t plane, float wscale, int M, int N, int K, int Kpad, int nMt,
                                                     int nNt, EpiArgs ep, float* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) _Float16 S[4 * G_BM * H_LD];  // Ah | Al | Bh | Bl
    _Float16 *Ah = S, *Al = S + G_BM * H_LD, *Bh = S + 2 * G_BM * H_LD, *Bl = S + 3 * G_BM * H_LD;
    int mt, nt;
    if (!tile_coords(nMt, nNt, &mt, &nt)) return;
    const int m0 = mt * G_BM, n0 = nt * G_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 3, lkq = tid & 7;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4];
    h8 rb[2][2];
    load_a_tile(A, lda, M, K, m0, 0, lrow, lkq, ra);
    load_w16_tile(Wh, plane, Kpad, n0, 0, tid, rb);
    for (int k0 = 0; k0 < Kpad; k0 += G_BK) {
        __syncthreads();
        split_store(Ah, Al, lrow, lkq, ra);
        store_w16_tile(Bh, Bl, tid, rb);
        __syncthreads();
        if (k0 + G_BK < Kpad) {
            load_a_tile(A, lda, M, K, m0, k0 + G_BK, lrow, lkq, ra);
            load_w16_tile(Wh, plane, Kpad, n0, k0 + G_BK, tid, rb);
        }
        mma16_slab(Ah, Al, Bh, Bl, wr, wc, lane, acc);
    }
    gemm_epilogue(acc, ep, wscale, m0 + wr * 64, n0 + wc * 64, lane, M, N, out, ldo);
}

extern "C" int32_t p2w_gemm_f16x3(const float* A, int32_t lda, const void* Wh, float wscale, int32_t M, int32_t N, int32_t K,
                                  const p2w_epilogue* epi, float* out, int32_t ldo, p2w_stream_t stream) {
    if (M == 0) return P2W_OK;
    P2W_CHECK_PTR(A); P2W_CHECK_PTR(Wh); P2W_CHECK_PTR(out);
    P2W_CHECK_ALIGN16(A); P2W_CHECK_ALIGN16(Wh);
    if (M < 0 || N <= 0 || K <= 0 || lda < K || ldo < N || (lda & 3) != 0 || !(wscale > 0.f)) return P2W_EINVAL;
    EpiArgs ep = {};
    if (epi) {
        if ((epi->sc0 && !epi->sh0) || (epi->sc1 && !epi->sh1)) return P2W_ENULL;
        if (epi->residual && epi->ldr < N) return P2W_EINVAL;
        ep = {epi->bias, epi->sc0, epi->sh0, epi->sc1, epi->sh1, epi->residual,
              epi->ldr, epi->relu0, epi->relu1, epi->relu2, epi->relu_final};
    }
    int Npad, Kpad;
    p2w_packed_dims(N, K, &Npad, &Kpad);
    const int nMt = p2w_cdiv(M, G_BM), nNt = p2w_cdiv(N, G_BN);
    gemm16_kernel<<<tile_grid(nMt, nNt), 256, 0, p2w_s(stream)>>>(A, lda, static_cast<const _Float16*>(Wh),
                                                                   (size_t)Npad * Kpad, wscale, M, N, K, Kpad, nMt, nNt, ep,
                                                                   out, ldo);
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// "H2" activations: a tensor [M, F] stored as fp16 hi/lo planes, row-interleaved: row m = [hi(0..ldh) | lo(0..ldh)],
// ldh = round_up(F, 8) halfs, pad columns zero.  Same bytes as fp32, but a consumer GEMM stages it with plain
// 16-byte copies (no conversion), exactly like the packed weights.  Producers split once, in their epilogue.
// ------------------------------------------------------------------------------------------------
// hi saturates at +-65504 instead of overflowing to inf; the remainder goes to lo (usable range ~1.3e5)
__device__ __forceinline__ void h2_split(float v, _Float16& hi, _Float16& lo) { sa_split(v, hi, lo); }
__device__ __forceinline__ unsigned h2_pack(_Float16 a, _Float16 b) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    h2v p = {a, b};
    return __builtin_bit_cast(unsigned, p);
}
// store 4 consecutive columns of one row (col % 4 == 0)
__device__ __forceinline__ void h2_store4(_Float16* __restrict__ base, int ldh, size_t row, int col, const float (&v)[4]) {
    uint2 hi, lo;
    split_pair(v[0], v[1], hi.x, lo.x);
    split_pair(v[2], v[3], hi.y, lo.y);
    _Float16* p = base + row * (size_t)(2 * ldh) + col;
    *reinterpret_cast<uint2*>(p) = hi;
    *reinterpret_cast<uint2*>(p + ldh) = lo;
}

// A tile from an H2 tensor: 128 rows x 32 halfs per plane; thread -> rows (tid>>2) + 64*i, 8 halfs at k = 8*(tid&3)
__device__ __forceinline__ void load_a16_tile(const _Float16* __restrict__ A, int ldh, int M, int m0, int k0, int tid,
                                              h8 (&ra)[2][2]) {
    const int row = tid >> 2, k = k0 + 8 * (tid & 3);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = m0 + row + 64 * i;
        h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        ra[0][i] = z; ra[1][i] = z;
        if (r < M && k < ldh) {
            const _Float16* p = A + (size_t)r * (2 * ldh) + k;
            ra[0][i] = *reinterpret_cast<const h8*>(p);
            ra[1][i] = *reinterpret_cast<const h8*>(p + ldh);
        }
    }
}

// epilogue value -> optional fp32 store + optional H2 store.  H2: lanes (2p, 2p+1) own adjacent columns of the same
// rows; they swap one register of each (r, r+1) pair so that every lane stores two adjacent columns as one 32-bit
// word per plane (even lane: row(r), odd lane: row(r+1)).
struct OutArgs { float* f32; int ldo; _Float16* h2; int ldh; };

__device__ __forceinline__ float epi_value(float a, float wscale, float bias, const EpiArgs& ep, float s0, float t0, float s1,
                                           float t1, size_t row, int col, bool ok) {
    float v = fmaf(a, wscale, bias);
    if (ep.relu0) v = fmaxf(v, 0.f);
    if (ep.sc0) { v = fmaf(v, s0, t0); }
    if (ep.relu1) v = fmaxf(v, 0.f);
    if (ep.sc1) { v = fmaf(v, s1, t1); }
    if (ep.relu2) v = fmaxf(v, 0.f);
    if (ep.residual && ok) v += ep.residual[row * ep.ldr + col];
    if (ep.relu_final) v = fmaxf(v, 0.f);
    return v;
}

template <int RT, int CT>
__device__ __forceinline__ void gemm_epilogue2(const f32x16 (&acc)[RT][CT], const EpiArgs& ep, float wscale, int row0, int col0,
                                               int lane, int M, int N, const OutArgs& o) {
    const int h = lane >> 5, odd = lane & 1;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int col = col0 + j * 32 + (lane & 31);
        const bool cv = col < N;
        const float bias = (cv && ep.bias) ? ep.bias[col] : 0.f;
        const float s0 = (cv && ep.sc0) ? ep.sc0[col] : 1.f, t0 = (cv && ep.sc0) ? ep.sh0[col] : 0.f;
        const float s1 = (cv && ep.sc1) ? ep.sc1[col] : 1.f, t1 = (cv && ep.sc1) ? ep.sh1[col] : 0.f;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int rowa = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;  // row of register r; r+1 is rowa + 1
                float va = epi_value(acc[i][j][r], wscale, bias, ep, s0, t0, s1, t1, (size_t)rowa, col, cv && rowa < M);
                float vb = epi_value(acc[i][j][r + 1], wscale, bias, ep, s0, t0, s1, t1, (size_t)rowa + 1, col,
                                     cv && rowa + 1 < M);
                if (!cv) { va = 0.f; vb = 0.f; }  // pad columns of an H2 row must be zero
                if (o.f32 && cv) {
                    if (rowa < M) o.f32[(size_t)rowa * o.ldo + col] = va;
                    if (rowa + 1 < M) o.f32[(size_t)(rowa + 1) * o.ldo + col] = vb;
                }
                if (o.h2) {
                    const float send = odd ? va : vb;
                    const float recv = __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(send), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false));
                    const float c0v = odd ? recv : va, c1v = odd ? vb : recv;  // columns (col & ~1), (col | 1)
                    const int roww = rowa + odd, colw = col & ~1;
                    if (roww < M && colw < o.ldh) {
                        unsigned hw, lw;
                        split_pair(c0v, c1v, hw, lw);
                        _Float16* p = o.h2 + (size_t)roww * (2 * o.ldh) + colw;
                        *reinterpret_cast<unsigned*>(p) = hw;
                        *reinterpret_cast<unsigned*>(p + o.ldh) = lw;
                    }
                }
            }
        }
    }
}

// Compile-time specialised epilogue for interior tiles (every row < M, every column < N): no per-element guards,
// no flag selects, 32-bit offsets.  EF bits: 1 relu0, 2 sc0, 4 relu1, 8 sc1, 16 relu2, 32 residual, 64 relu_final,
// 128 fp32 out, 256 H2 out.  Edge tiles and unlisted combinations use gemm_epilogue2 (runtime flags).
template <int RT, int CT, int EF>
__device__ __forceinline__ void gemm_epilogue3(const f32x16 (&acc)[RT][CT], const EpiArgs& ep, float wscale, int row0, int col0,
                                               int lane, const OutArgs& o) {
    constexpr bool R0 = EF & 1, S0 = EF & 2, R1 = EF & 4, S1 = EF & 8, R2 = EF & 16, RES = EF & 32, RF = EF & 64,
                   OF = EF & 128, OH = EF & 256;
    const int h = lane >> 5, odd = lane & 1;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int col = col0 + j * 32 + (lane & 31);
        const float bias = ep.bias ? ep.bias[col] : 0.f;
        float s0 = 1.f, t0 = 0.f, s1 = 1.f, t1 = 0.f;
        if (S0) { s0 = ep.sc0[col]; t0 = ep.sh0[col]; }
        if (S1) { s1 = ep.sc1[col]; t1 = ep.sh1[col]; }
        auto f = [&](float a, unsigned roff) {
            float v = fmaf(a, wscale, bias);
            if (R0) v = fmaxf(v, 0.f);
            if (S0) v = fmaf(v, s0, t0);
            if (R1) v = fmaxf(v, 0.f);
            if (S1) v = fmaf(v, s1, t1);
            if (R2) v = fmaxf(v, 0.f);
            if (RES) v += ep.residual[roff];
            if (RF) v = fmaxf(v, 0.f);
            return v;
        };
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const unsigned rbase = (unsigned)(row0 + i * 32 + 4 * h);
            __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting every tile's loads at once (spills)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if ((r & 7) == 0) __builtin_amdgcn_sched_barrier(0);
                const unsigned rowa = rbase + (r & 3) + 8 * (r >> 2);
                const float va = f(acc[i][j][r], RES ? rowa * (unsigned)ep.ldr + col : 0u);
                const float vb = f(acc[i][j][r + 1], RES ? (rowa + 1) * (unsigned)ep.ldr + col : 0u);
                if (OF) {
                    o.f32[rowa * (unsigned)o.ldo + col] = va;
                    o.f32[(rowa + 1) * (unsigned)o.ldo + col] = vb;
                }
                if (OH) {
                    const float send = odd ? va : vb;
                    const float recv = __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(send), 0xB1, 0xf, 0xf, false));
                    const float c0v = odd ? recv : va, c1v = odd ? vb : recv;
                    unsigned hw, lw;
                    split_pair(c0v, c1v, hw, lw);
                    _Float16* p = o.h2 + (rowa + odd) * (unsigned)(2 * o.ldh) + (col & ~1);
                    *reinterpret_cast<unsigned*>(p) = hw;
                    *reinterpret_cast<unsigned*>(p + o.ldh) = lw;
                }
            }
        }
    }
}

template <int RT, int CT>
__device__ __forceinline__ void gemm_epilogue_dispatch(const f32x16 (&acc)[RT][CT], const EpiArgs& ep, float wscale, int row0,
                                                       int col0, int lane, int M, int N, const OutArgs& o, int ef) {
    const bool full = (row0 + 32 * RT <= M) && (col0 + 32 * CT <= N) && ef != 0;
    if (full) {
        switch (ef) {
#define P2W_EPI_CASE(E) case E: gemm_epilogue3<RT, CT, E>(acc, ep, wscale, row0, col0, lane, o); return;
            P2W_EPI_CASE(128) P2W_EPI_CASE(257) P2W_EPI_CASE(263) P2W_EPI_CASE(287) P2W_EPI_CASE(480) P2W_EPI_CASE(224)
            P2W_EPI_CASE(131) P2W_EPI_CASE(259) P2W_EPI_CASE(387) P2W_EPI_CASE(129)
#undef P2W_EPI_CASE
            default: break;
        }
    }
    gemm_epilogue2<RT, CT>(acc, ep, wscale, row0, col0, lane, M, N, o);
}

__global__ __launch_bounds__(256) void gemm_h2_kernel(const _Float16* __restrict__ A, int ldh_a, const _Float16* __restrict__ Wh,
                                                      size_t plane, float wscale, int M, int N, int Kpad, int nMt, int nNt,
                                                      EpiArgs ep, OutArgs o) {
    __shared__ __attribute__((aligned(16))) _Float16 S[4 * G_BM * H_LD];  // Ah | Al | Bh | Bl
    _Float16 *Ah = S, *Al = S + G_BM * H_LD, *Bh = S + 2 * G_BM * H_LD, *Bl = S + 3 * G_BM * H_LD;
    int mt, nt;
    if (!tile_coords(nMt, nNt, &mt, &nt)) return;
    const int m0 = mt * G_BM, n0 = nt * G_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    h8 ra[2][2], rb[2][2];
    load_a16_tile(A, ldh_a, M, m0, 0, tid, ra);
    load_w16_tile(Wh, plane, Kpad, n0, 0, tid, rb);
    for (int k0 = 0; k0 < Kpad; k0 += G_BK) {
        __syncthreads();
        store_w16_tile(Ah, Al, tid, ra);
        store_w16_tile(Bh, Bl, tid, rb);
        __syncthreads();
        if (k0 + G_BK < Kpad) {
            load_a16_tile(A, ldh_a, M, m0, k0 + G_BK, tid, ra);
            load_w16_tile(Wh, plane, Kpad, n0, k0 + G_BK, tid, rb);
        }
        mma16_slab(Ah, Al, Bh, Bl, wr, wc, lane, acc);
    }
    gemm_epilogue2<2, 2>(acc, ep, wscale, m0 + wr * 64, n0 + wc * 64, lane, M, N, o);
}

// ------------------------------------------------------------------------------------------------
// gemm_h2 v2: both operands are fp16 hi/lo planes in HBM, so a K-slab is staged with direct-to-LDS loads
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write) into a 2-stage ring; one barrier per slab, the next
// slab's DMA is in flight during the whole MFMA phase of the current one.
// LDS image of a stage (16-byte chunks): A plane p, row r, chunk q -> ((p*BM + r)*4 + (q ^ ((r>>2)&3)));  B after A.
// Rows are 64 B unpadded (the DMA writes 1 KiB linearly per wave-instruction: lane L -> chunk base+L), so the XOR
// swizzle is applied on the per-lane SOURCE address and again on the ds_read address: conflict-free ds_read_b128.
// Out-of-range A rows are clamped to M-1 (their results are never stored); K padding is zero in both operands
// (H2 tensors have ldh % 32 == 0 with zero pad columns when they feed this kernel).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

template <int WR, int WC, int RT, int CT>   // waves WR x WC, wave tile (32*RT) x (32*CT)
__global__ __launch_bounds__(64 * WR * WC, 2) void gemm_h2g_kernel(const _Float16* __restrict__ A, int ldh_a,
                                                                const _Float16* __restrict__ Wh, size_t plane, float wscale,
                                                                int M, int N, int Kpad, int nMt, int nNt, EpiArgs ep,
                                                                OutArgs o, int dbg_, int ef, int tmode) {
    // dbg (profiling ablations): 1 = skip the epilogue, 2 = issue only the first slab's DMA, 4 = skip the MFMAs,
    // 8 = fragments loaded once, 16 = no barrier, 32 = every workgroup reads row tile 0.  Compiled in only by diagnostic
    // builds (P2W_EXTRA_CFLAGS=-DP2W_GEMM_ABLATE): this kernel sits at 256 VGPRs and every extra path costs scratch.
#ifdef P2W_GEMM_ABLATE
    const int dbg = dbg_;
#else
    constexpr int dbg = 0;
    (void)dbg_;
#endif
    constexpr int BM = 32 * RT * WR, BN = 32 * CT * WC, NW = WR * WC;
    constexpr int A_CH = 8 * BM, STAGE_CH = A_CH + 8 * BN;   // 16-byte chunks per stage (2 planes x rows x 4)
    constexpr int NI = STAGE_CH / 64 / NW;                   // DMA instructions per wave per stage
    static_assert(STAGE_CH % (64 * NW) == 0, "stage must split evenly over the waves");
    __shared__ __attribute__((aligned(16))) char S[2 * STAGE_CH * 16];
    int mt, nt;
    if (!tile_coords(nMt, nNt, &mt, &nt, tmode)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave / WC, wc = wave % WC;

    // per-lane DMA sources (advance 64 B per slab) and wave-uniform LDS chunk bases
    const _Float16* src[NI];
    int dstc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g = wave + NW * i;
        const int rloc = lane >> 2;
        if (g < BM / 8) {
            const int p = g / (BM / 16), rb = g % (BM / 16);
            const int row = 16 * rb + rloc, q = (lane & 3) ^ ((row >> 2) & 3);
            const int grow = (dbg & 32) ? row : min(m0 + row, M - 1);   // dbg 32: every workgroup reads row tile 0 (no A traffic)
            src[i] = A + (size_t)grow * (2 * ldh_a) + (size_t)p * ldh_a + 8 * q;
            dstc[i] = g * 64;
        } else {
            const int g2 = g - BM / 8, p = g2 / (BN / 16), rb = g2 % (BN / 16);
            const int row = 16 * rb + rloc, q = (lane & 3) ^ ((row >> 2) & 3);
            src[i] = Wh + (size_t)p * plane + (size_t)(n0 + row) * Kpad + 8 * q;
            dstc[i] = A_CH + g2 * 64;
        }
    }
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds((glb_vp)(src[i] + k0), (lds_vp)(S + ((size_t)stage * STAGE_CH + dstc[i]) * 16), 16, 0, 0);
    };

    // fragment read offsets (bytes within a stage) for kk = 0; kk = 16 flips chunk bit 1 (q ^= 2)
    const int r = lane & 31, h = lane >> 5;
    int offA[2][RT], offB[2][CT];  // [plane][tile]
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int ra = wr * 32 * RT + 32 * t + r;
            offA[p][t] = ((p * BM + ra) * 4 + (h ^ ((ra >> 2) & 3))) * 16;
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int rb = wc * 32 * CT + 32 * t + r;
            offB[p][t] = (A_CH + (p * BN + rb) * 4 + (h ^ ((rb >> 2) & 3))) * 16;
        }
    }

    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nslab = Kpad / G_BK;
    issue(0, 0);
    h8 ah[RT], al[RT], bh[CT], bl[CT];
    if (dbg & 8) {   // diagnostic: fragments loaded once, the loop below is MFMA (+ optional barrier) only
        const char* st0 = S;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RT; ++t) { ah[t] = *reinterpret_cast<const h8*>(st0 + offA[0][t]); al[t] = *reinterpret_cast<const h8*>(st0 + offA[1][t]); }
#pragma unroll
        for (int t = 0; t < CT; ++t) { bh[t] = *reinterpret_cast<const h8*>(st0 + offB[0][t]); bl[t] = *reinterpret_cast<const h8*>(st0 + offB[1][t]); }
    }
    for (int s = 0; s < nslab; ++s) {
        if (!(dbg & 16)) __syncthreads();  // = s_waitcnt vmcnt(0) + barrier: slab s has landed for every wave, slab s-1's buffer is free
        if (s + 1 < nslab && !(dbg & 2)) issue((s + 1) & 1, (s + 1) * G_BK);
        const char* st = S + (size_t)(s & 1) * STAGE_CH * 16;
        if (dbg & 4) continue;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (!(dbg & 8)) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                ah[t] = *reinterpret_cast<const h8*>(st + (offA[0][t] ^ (kk << 5)));
                al[t] = *reinterpret_cast<const h8*>(st + (offA[1][t] ^ (kk << 5)));
            }
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                bh[t] = *reinterpret_cast<const h8*>(st + (offB[0][t] ^ (kk << 5)));
                bl[t] = *reinterpret_cast<const h8*>(st + (offB[1][t] ^ (kk << 5)));
            }
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    if (dbg & 1) {
        if (acc[0][0][0] + acc[0][CT - 1][1] + acc[RT - 1][0][2] + acc[RT - 1][CT - 1][3] == 12345.678f && o.f32) o.f32[0] = 1.f;
        return;
    }
    gemm_epilogue_dispatch<RT, CT>(acc, ep, wscale, m0 + wr * 32 * RT, n0 + wc * 32 * CT, lane, M, N, o, ef);
}

extern "C" int32_t p2w_gemm_h2(const void* A_h2, int32_t ldh_a, const void* Wh, float wscale, int32_t M, int32_t N, int32_t K,
                               const p2w_epilogue* epi, float* out_f32, int32_t ldo, void* out_h2, int32_t ldh_o,
                               p2w_stream_t stream) {
    if (M == 0) return P2W_OK;
    P2W_CHECK_PTR(A_h2); P2W_CHECK_PTR(Wh);
    if (!out_f32 && !out_h2) return P2W_ENULL;
    P2W_CHECK_ALIGN16(A_h2); P2W_CHECK_ALIGN16(Wh);
    if (out_h2) P2W_CHECK_ALIGN16(out_h2);
    if (M < 0 || N <= 0 || K <= 0 || ldh_a < K || (ldh_a & 7) || !(wscale > 0.f)) return P2W_EINVAL;
    if (out_f32 && ldo < N) return P2W_EINVAL;
    if (out_h2 && (ldh_o < N || (ldh_o & 7))) return P2W_EINVAL;
    EpiArgs ep = {};
    if (epi) {
        if ((epi->sc0 && !epi->sh0) || (epi->sc1 && !epi->sh1)) return P2W_ENULL;
        if (epi->residual && epi->ldr < N) return P2W_EINVAL;
        ep = {epi->bias, epi->sc0, epi->sh0, epi->sc1, epi->sh1, epi->residual,
              epi->ldr, epi->relu0, epi->relu1, epi->relu2, epi->relu_final};
    }
    int Npad, Kpad;
    p2w_packed_dims(N, K, &Npad, &Kpad);
    if (out_h2 && ldh_o > Npad) return P2W_EINVAL;
    const int nNt = p2w_cdiv(N, G_BN);
    OutArgs o = {out_f32, ldo, static_cast<_Float16*>(out_h2), ldh_o};
    const _Float16* Ah = static_cast<const _Float16*>(A_h2);
    const _Float16* Wp = static_cast<const _Float16*>(Wh);
    if ((ldh_a & 31) == 0 && ldh_a >= Kpad) {  // direct-to-LDS path: K padding must exist (and be zero) in A as well
        static const int force = []() { const char* e = getenv("P2W_GEMM_TILE"); return e ? atoi(e) : 0; }();
        const char* de = getenv("P2W_GEMM_DBG");
        const int dbg = de ? atoi(de) : 0;
        // 256x256 tiles halve the L2->LDS bytes per MFMA; they need enough tiles to fill 256 CUs and a wide N
        const long tiles256 = (long)p2w_cdiv(M, 256) * (Npad / 256);
        static const int n_cu_g = []() {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            return n > 0 ? n : 256;
        }();
        // one 256x256 workgroup per CU: worth it when N has no column padding at that width and the tiles fill >= 78 % of
        // whole rounds of the chip, from 3/4 of one round up (per-launch A/B over the network's 54 GEMMs: 207 tiles on
        // 256 CUs still win by 8 %, 340 of 512 or N = 640 padded to 768 lose by 10-25 %)
        const long rounds = (tiles256 + n_cu_g - 1) / n_cu_g;
        const bool fills = tiles256 * 100 >= rounds * n_cu_g * 78;
        const bool big = force ? (force == 256) : (N >= 256 && (N % 256) == 0 && tiles256 * 4 >= 3 * n_cu_g && fills);
        // epilogue class for the specialised interior-tile path (0 = generic); needs 32-bit element offsets
        int ef = (ep.relu0 ? 1 : 0) | (ep.sc0 ? 2 : 0) | (ep.relu1 ? 4 : 0) | (ep.sc1 ? 8 : 0) | (ep.relu2 ? 16 : 0) |
                 (ep.residual ? 32 : 0) | (ep.relu_final ? 64 : 0) | (out_f32 ? 128 : 0) | (out_h2 ? 256 : 0);
        const size_t lim = (size_t)1 << 31;
        if ((size_t)M * (size_t)(ldo > 2 * ldh_o ? ldo : 2 * ldh_o) >= lim || (ep.residual && (size_t)M * ep.ldr >= lim) ||
            (N & 1) || getenv("P2W_GEMM_GENERIC_EPI"))
            ef = 0;
        // tile order: keep W L2-resident per XCD when it does not fit an XCD's L2 (see tile_coords)
        static const int force_mode = []() { const char* e = getenv("P2W_GEMM_TMODE"); return e ? atoi(e) : -1; }();
        const size_t w_bytes = (size_t)N * Kpad * 4;
        auto pick_mode = [&](int nNtx) {
            const bool ok = nNtx >= 8 || (nNtx > 0 && 8 % nNtx == 0);
            if (!ok) return 0;
            if (force_mode >= 0) return force_mode;
            return w_bytes > (size_t)3 * 1024 * 1024 ? 1 : 0;
        };
        if (big) {
            const int nMt = p2w_cdiv(M, 256), nNt2 = Npad / 256;
            const int tm = pick_mode(nNt2);
            gemm_h2g_kernel<2, 4, 4, 2><<<tile_grid(nMt, nNt2, tm), 512, 0, p2w_s(stream)>>>(
                Ah, ldh_a, Wp, (size_t)Npad * Kpad, wscale, M, N, Kpad, nMt, nNt2, ep, o, dbg, ef, tm);
        } else {
            const int nMt = p2w_cdiv(M, 128), nNt1 = p2w_cdiv(N, 128);
            const int tm = pick_mode(nNt1);
            gemm_h2g_kernel<2, 2, 2, 2><<<tile_grid(nMt, nNt1, tm), 256, 0, p2w_s(stream)>>>(
                Ah, ldh_a, Wp, (size_t)Npad * Kpad, wscale, M, N, Kpad, nMt, nNt1, ep, o, dbg, ef, tm);
        }
        return P2W_LAUNCH_STATUS();
    }
    const int nMt = p2w_cdiv(M, G_BM);
    gemm_h2_kernel<<<tile_grid(nMt, nNt), 256, 0, p2w_s(stream)>>>(Ah, ldh_a, Wp, (size_t)Npad * Kpad, wscale, M, N, Kpad, nMt,
                                                                    nNt, ep, o);
    return P2W_LAUNCH_STATUS();
}

__global__ __launch_bounds__(256) void sa_conv16_kernel(const float* __restrict__ P, int ldp, const float4* __restrict__ xyzr,
                                                        const int* __restrict__ idx, const int* __restrict__ batch_dst,
                                                        const float* __restrict__ sf, const int* __restrict__ nbr,
                                                        const int* __restrict__ deg, int kw, int M,
                                                        const float* __restrict__ w1r4, int C1, int C1pad,
                                                        const _Float16* __restrict__ W2h, size_t plane, float wscale, int C2,
                                                        int nMt, int nNt, const float* __restrict__ b2,
                                                        const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                                        float* __restrict__ out, int ldo, _Float16* __restrict__ out_h2,
                                                        int ldh) {
    __shared__ __attribute__((aligned(16))) _Float16 S[4 * G_BM * H_LD];
    _Float16 *Ah = S, *Al = S + G_BM * H_LD, *Bh = S + 2 * G_BM * H_LD, *Bl = S + 3 * G_BM * H_LD;
    __shared__ int m_j[G_BM];
    __shared__ float m_g[G_BM][4];
    int mt, nt;
    if (!tile_coords(nMt, nNt, &mt, &nt)) return;
    const int t0 = mt * 4, n0 = nt * G_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 3, lkq = tid & 7;
    sa_row_geometry(tid, t0, M, kw, xyzr, idx, batch_dst, sf, nbr, deg, m_j, m_g);
    __syncthreads();
    int rj[4];
    float4 rg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rj[i] = m_j[lrow + 32 * i];
        rg[i] = *reinterpret_cast<const float4*>(&m_g[lrow + 32 * i][0]);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4];
    h8 rb[2][2];
    sa_load_h1(P, ldp, w1r4, C1, C1pad, 4 * lkq, rj, rg, ra);
    load_w16_tile(W2h, plane, C1pad, n0, 0, tid, rb);
    for (int k0 = 0; k0 < C1pad; k0 += G_BK) {
        __syncthreads();
        split_store(Ah, Al, lrow, lkq, ra);
        store_w16_tile(Bh, Bl, tid, rb);
        __syncthreads();
        if (k0 + G_BK < C1pad) {
            sa_load_h1(P, ldp, w1r4, C1, C1pad, k0 + G_BK + 4 * lkq, rj, rg, ra);
            load_w16_tile(W2h, plane, C1pad, n0, k0 + G_BK, tid, rb);
        }
        mma16_slab(Ah, Al, Bh, Bl, wr, wc, lane, acc);
    }
    sa_epilogue(acc, wscale, t0, n0, wr, wc, lane, M, kw, deg, C2, b2, bn_s, bn_t, out, ldo, out_h2, ldh);
}

// ------------------------------------------------------------------------------------------------
// fused PointNetConv v2 (f16x3): 4 targets (128 edge rows) x 256 output columns per workgroup, 8 waves (2 x 4, each
// 64 x 64).  B (W2 hi/lo) arrives by direct-to-LDS DMA into a 2-stage ring; A is PRODUCED on the VALU (gather P[j],
// add the relative-position term, ReLU, split hi/lo) one slab ahead into the other stage, so gather latency and
// the producer's VALU work overlap with the MFMAs of the current slab.  One barrier per slab.  LDS image and XOR
// swizzle are those of gemm_h2g_kernel (A: 2 planes x 128 rows, B: 2 planes x 256 rows, 64-byte rows).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void sa_conv16g_kernel(const float* __restrict__ P, int ldp, const float4* __restrict__ xyzr,
                                                            const int* __restrict__ idx, const int* __restrict__ batch_dst,
                                                            const float* __restrict__ sf, const int* __restrict__ nbr,
                                                            const int* __restrict__ deg, int kw, int M,
                                                            const float* __restrict__ w1r4, int C1, int C1pad,
                                                            const _Float16* __restrict__ W2h, size_t plane, float wscale, int C2,
                                                            int nMt, int nNt, const float* __restrict__ b2,
                                                            const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                                            float* __restrict__ out, int ldo, _Float16* __restrict__ out_h2,
                                                            int ldh, int dbg) {
    // dbg (profiling ablations, 0 in production): 1 = no epilogue, 2 = no B DMA after slab 0, 4 = no MFMA,
    // 8 = no A production after slab 0, 16 = produce A without gathering P
    constexpr int BM = 128, BN = 256, NW = 8;
    constexpr int A_CH = 8 * BM, STAGE_CH = A_CH + 8 * BN;   // 1024 + 2048 chunks = 48 KiB per stage
    constexpr int NI = (8 * BN) / 64 / NW;                   // 4 DMA instructions per wave per stage (B only)
    __shared__ __attribute__((aligned(16))) char S[2 * STAGE_CH * 16];
    __shared__ int m_j[BM];
    __shared__ float m_g[BM][4];
    int mt, nt;
    if (!tile_coords(nMt, nNt, &mt, &nt)) return;
    const int t0 = mt * 4, n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 2, wc = wave & 3;

    sa_row_geometry(tid, t0, M, kw, xyzr, idx, batch_dst, sf, nbr, deg, m_j, m_g);  // threads 0..127
    // B DMA sources
    const _Float16* src[NI];
    int dstc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g2 = wave + NW * i, p = g2 / (BN / 16), rb = g2 % (BN / 16);
        const int row = 16 * rb + (lane >> 2), q = (lane & 3) ^ ((row >> 2) & 3);
        src[i] = W2h + (size_t)p * plane + (size_t)(n0 + row) * C1pad + 8 * q;
        dstc[i] = A_CH + g2 * 64;
    }
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds((glb_vp)(src[i] + k0), (lds_vp)(S + ((size_t)stage * STAGE_CH + dstc[i]) * 16), 16, 0, 0);
    };
    __syncthreads();
    // A producer: thread -> edge row (tid>>2), 8 consecutive k (one 16-byte chunk per plane)
    const int prow = tid >> 2, pq = tid & 3;
    const int rj = m_j[prow];
    const float4 rg = *reinterpret_cast<const float4*>(&m_g[prow][0]);
    const int a_dst = (prow * 4 + (pq ^ ((prow >> 2) & 3))) * 16;           // hi plane; lo plane = + BM*64 bytes
    const float* prow_ptr = P + (size_t)rj * ldp + 8 * pq;
    float4 pv[2];
    auto gather = [&](int k0) {
        const int k = k0 + 8 * pq;
        pv[0] = (k < C1) ? *reinterpret_cast<const float4*>(prow_ptr + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        pv[1] = (k + 4 < C1) ? *reinterpret_cast<const float4*>(prow_ptr + k0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto produce = [&](int stage, int k0) {
        const int k = k0 + 8 * pq;
        unsigned hiw[4], low[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kk = k + 4 * half;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (kk < C1) {
                const float4 wx = *reinterpret_cast<const float4*>(&w1r4[0 * C1pad + kk]);
                const float4 wy = *reinterpret_cast<const float4*>(&w1r4[1 * C1pad + kk]);
                const float4 wz = *reinterpret_cast<const float4*>(&w1r4[2 * C1pad + kk]);
                const float4 wf = *reinterpret_cast<const float4*>(&w1r4[3 * C1pad + kk]);
                const float4 p = pv[half];
                v[0] = fmaxf(fmaf(rg.w, wf.x, fmaf(rg.z, wz.x, fmaf(rg.y, wy.x, fmaf(rg.x, wx.x, p.x)))), 0.f);
                v[1] = fmaxf(fmaf(rg.w, wf.y, fmaf(rg.z, wz.y, fmaf(rg.y, wy.y, fmaf(rg.x, wx.y, p.y)))), 0.f);
                v[2] = fmaxf(fmaf(rg.w, wf.z, fmaf(rg.z, wz.z, fmaf(rg.y, wy.z, fmaf(rg.x, wx.z, p.z)))), 0.f);
                v[3] = fmaxf(fmaf(rg.w, wf.w, fmaf(rg.z, wz.w, fmaf(rg.y, wy.w, fmaf(rg.x, wx.w, p.w)))), 0.f);
            }
            {
                unsigned h01, l01, h23, l23;
                split_pair(v[0], v[1], h01, l01);
                split_pair(v[2], v[3], h23, l23);
                hiw[2 * half] = h01; hiw[2 * half + 1] = h23;
                low[2 * half] = l01; low[2 * half + 1] = l23;
            }
        }
        char* st = S + (size_t)stage * STAGE_CH * 16;
        *reinterpret_cast<uint4*>(st + a_dst) = make_uint4(hiw[0], hiw[1], hiw[2], hiw[3]);
        *reinterpret_cast<uint4*>(st + BM * 64 + a_dst) = make_uint4(low[0], low[1], low[2], low[3]);
    };

    const int r = lane & 31, h = lane >> 5;
    int offA[2][2], offB[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ra = wr * 64 + 32 * t + r, rb = wc * 64 + 32 * t + r;
            offA[p][t] = ((p * BM + ra) * 4 + (h ^ ((ra >> 2) & 3))) * 16;
            offB[p][t] = (A_CH + (p * BN + rb) * 4 + (h ^ ((rb >> 2) & 3))) * 16;
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nslab = C1pad / G_BK;
    issue(0, 0);
    gather(0);
    produce(0, 0);
    for (int s = 0; s < nslab; ++s) {
        __syncthreads();  // B(s) landed, A(s) written by every thread, stage (s+1)&1 no longer read
        const bool more = s + 1 < nslab;
        if (more && !(dbg & 2)) issue((s + 1) & 1, (s + 1) * G_BK);
        if (more && !(dbg & 24)) gather((s + 1) * G_BK);
        const char* st = S + (size_t)(s & 1) * STAGE_CH * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (dbg & 4) break;
            h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const h8*>(st + (offA[0][t] ^ (kk << 5)));
                al[t] = *reinterpret_cast<const h8*>(st + (offA[1][t] ^ (kk << 5)));
                bh[t] = *reinterpret_cast<const h8*>(st + (offB[0][t] ^ (kk << 5)));
                bl[t] = *reinterpret_cast<const h8*>(st + (offB[1][t] ^ (kk << 5)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more && !(dbg & 8)) produce((s + 1) & 1, (s + 1) * G_BK);
    }
    if (dbg & 1) {
        if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 12345.678f && out) out[0] = 1.f;
        return;
    }
    sa_epilogue(acc, wscale, t0, n0, wr, wc, lane, M, kw, deg, C2, b2, bn_s, bn_t, out, ldo, out_h2, ldh);
}

// ------------------------------------------------------------------------------------------------
// fused PointNetConv v3 (f16x3), two kernels:
//  1. sa_edge_meta_kernel: one thread per (target, slot): source index j and g = (rel/(dmax+1e-8), refl_j)
//     (pointnet.py:119-129) -> meta_j[M*32], meta_g[M*32] (20 B per slot).  The chain of dependent loads
//     (deg -> nbr -> xyzr) is hidden by plain occupancy here instead of stalling a GEMM-shaped workgroup.
//  2. sa_conv16p_kernel: PERSISTENT workgroups (one per CU, 8 waves) walk (row tile, column tile) work items; the
//     K slabs of consecutive items form ONE software pipeline: while slab g runs on the MFMAs, slab g+1's W2 DMA,
//     P-row gather and A production (possibly of the NEXT item) are in flight, and the next item's metadata is
//     prefetched a whole item ahead.  128 edge rows (4 targets) x 256 columns per item.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sa_edge_meta_kernel(const float4* __restrict__ xyzr, const int* __restrict__ idx,
                                                           const int* __restrict__ batch_dst, const float* __restrict__ sf,
                                                           const int* __restrict__ nbr, const int* __restrict__ deg, int kw,
                                                           int M, int* __restrict__ meta_j, float4* __restrict__ meta_g) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const int tgt = (int)(g >> 5), slot = (int)(g & 31);
    int j = 0;
    float rx = 0.f, ry = 0.f, rz = 0.f, rf = 0.f, nrm = 0.f;
    if (tgt < M) {
        const int d = deg[tgt];
        const int self = idx[tgt];
        const float s = sf[batch_dst[tgt]];
        const float4 pi = xyzr[self];
        const bool valid = slot < d && slot < kw;
        j = valid ? nbr[(size_t)tgt * kw + slot] : self;
        if (j < 0) j = self;
        const float4 pj = xyzr[j];
        if (valid) {
            rx = pj.x / s - pi.x / s; ry = pj.y / s - pi.y / s; rz = pj.z / s - pi.z / s;
            nrm = sqrtf(((rx * rx) + (ry * ry)) + (rz * rz));
            rf = pj.w;
        }
    }
    float dmax = nrm;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));  // 32 lanes = one target
    const float den = dmax + 1e-8f;
    if (tgt < M) {
        const bool valid = slot < deg[tgt] && slot < kw;
        meta_j[g] = valid ? j : -1;     // empty slot: the producer writes a zero row, the epilogue masks it
        meta_g[g] = make_float4(rx / den, ry / den, rz / den, rf);
    }
}

// <BN, RT>: <256, 2>: 4 targets x 256 columns (waves 2 x 4, wave tile 64 x 64);  <128, 2>: 8 targets x 128 columns (waves 4 x 2);
// <256, 4>: 8 targets x 256 columns (waves 2 x 4, wave tile 128 x 64): half the W2 DMA, barriers and per-item overhead per
// FLOP, at 2 x the accumulators (one workgroup per CU either way: LDS)
template <int BN, int RT>
__global__ __launch_bounds__(512, RT == 2 ? 2 : 1) void sa_conv16p_kernel(const float* __restrict__ P, int ldp, const int* __restrict__ meta_j,
                                                            const float4* __restrict__ meta_g, const int* __restrict__ deg,
                                                            int kw, int M, const float* __restrict__ w1r4, int C1, int C1pad,
                                                            const _Float16* __restrict__ W2h, size_t plane, float wscale, int C2,
                                                            int nMt, int nNt, const float* __restrict__ b2,
                                                            const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                                            float* __restrict__ out, int ldo, _Float16* __restrict__ out_h2,
                                                            int ldh, int dbg) {
    // dbg (profiling ablations, 0 in production): 1 no epilogue, 2 no W2 DMA after the first, 4 no MFMA, 8 no producer,
    // 16 no P gather
    constexpr int WCn = BN / 64, BM = 32 * RT * (8 / WCn), NW = 8, NR = BM / 128;   // NR producer rows per thread
    constexpr int A_CH = 8 * BM, STAGE_CH = A_CH + 8 * BN;
    constexpr int NI = (8 * BN) / 64 / NW;
    __shared__ __attribute__((aligned(16))) char S[2 * STAGE_CH * 16];
    __shared__ __attribute__((aligned(16))) float Wr[4 * 512];   // layer-1 geometry weights (rx, ry, rz, refl rows), C1pad <= 512
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4 * C1pad; i += 512) Wr[i] = w1r4[i];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave / WCn, wc = wave % WCn;
    const int nitems = nMt * nNt, nslab = C1pad / G_BK;
    // XCD-aware work assignment: workgroups b, b+8, b+16.. share an XCD (round-robin dispatch) and therefore an L2.
    // Each XCD walks ONE contiguous chunk of work items, its workgroups taking consecutive items at every step, so the
    // P rows gathered by an XCD at any time belong to spatially adjacent targets (levels are stored in grid-cell
    // order) and are re-used out of that XCD's L2 instead of being streamed by all eight.
    int first, stride, limit;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int chunk = (nitems + 7) >> 3;
        first = xcd * chunk + slot; stride = per; limit = min((xcd + 1) * chunk, nitems);
    } else {
        first = blockIdx.x; stride = gridDim.x; limit = nitems;
    }
    if (first >= limit) return;
    const int my_items = (limit - first + stride - 1) / stride;
    const int total = my_items * nslab;

    // item -> (row tile, column tile): column tiles of one row tile are adjacent work items
    auto item_mt = [&](int it) { return (first + it * stride) / nNt; };
    auto item_nt = [&](int it) { return (first + it * stride) % nNt; };

    const int prow = tid >> 2, pq = tid & 3;   // rows prow + 128*u, u < NR ((row>>2)&3 is the same for all of them)
    const int a_dst = (prow * 4 + (pq ^ ((prow >> 2) & 3))) * 16;
    // per-lane pieces of the B DMA source that do not depend on the item
    size_t boff[NI];
    int dstc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g2 = wave + NW * i, p = g2 / (BN / 16), rb = g2 % (BN / 16);
        const int row = 16 * rb + (lane >> 2), q = (lane & 3) ^ ((row >> 2) & 3);
        boff[i] = (size_t)p * plane + (size_t)row * C1pad + 8 * q;
        dstc[i] = A_CH + g2 * 64;
    }
    auto issue = [&](int stage, const _Float16* wbase, int k0) {   // wbase = W2h + nt * BN * C1pad (per item)
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds((glb_vp)(wbase + boff[i] + k0),
                                             (lds_vp)(S + ((size_t)stage * STAGE_CH + dstc[i]) * 16), 16, 0, 0);
    };
    // metadata of the producer's edge row (clamped: rows past the last target replay the last valid row)
    struct Meta { int j[NR]; float4 g[NR]; };
    struct Vals { float4 v[NR][2]; };
    auto load_meta = [&](int it, Meta& m) {
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            long row = (long)item_mt(it) * BM + prow + 128 * u;
            const long last = (long)M * 32 - 1;
            row = row < last ? row : last;
            m.j[u] = meta_j[row];      // < 0: empty neighbour slot (its row is masked in the epilogue)
            m.g[u] = meta_g[row];
        }
    };
    Vals pv;    // slab 0 only (prologue)
    auto gather = [&](const Meta& m, int k0, Vals& dst) {
        const int k = k0 + 8 * pq;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const bool on = m.j[u] >= 0;
            const float* p = P + (size_t)(on ? m.j[u] : 0) * ldp + 8 * pq + k0;
            dst.v[u][0] = (on && k < C1) ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
            dst.v[u][1] = (on && k + 4 < C1) ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto produce = [&](int stage, const Meta& m, int k0, const Vals& src) {
        const int k = k0 + 8 * pq;
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        const float4 rg = m.g[u];
        const bool on = m.j[u] >= 0;
        unsigned hiw[4], low[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kk = k + 4 * half;
            // branch-free (so the scheduler can interleave it with MFMAs): Wr is zero-padded to C1pad, P values of
            // empty slots / padded k are zero, and the geometry term is switched off with a select
            const float4 wx = *reinterpret_cast<const float4*>(&Wr[0 * C1pad + kk]);
            const float4 wy = *reinterpret_cast<const float4*>(&Wr[1 * C1pad + kk]);
            const float4 wz = *reinterpret_cast<const float4*>(&Wr[2 * C1pad + kk]);
            const float4 wf = *reinterpret_cast<const float4*>(&Wr[3 * C1pad + kk]);
            const float4 p = src.v[u][half];
            const float gx = on ? rg.x : 0.f, gy = on ? rg.y : 0.f, gz = on ? rg.z : 0.f, gw = on ? rg.w : 0.f;
            float v[4];
            v[0] = fmaxf(fmaf(gw, wf.x, fmaf(gz, wz.x, fmaf(gy, wy.x, fmaf(gx, wx.x, p.x)))), 0.f);
            v[1] = fmaxf(fmaf(gw, wf.y, fmaf(gz, wz.y, fmaf(gy, wy.y, fmaf(gx, wx.y, p.y)))), 0.f);
            v[2] = fmaxf(fmaf(gw, wf.z, fmaf(gz, wz.z, fmaf(gy, wy.z, fmaf(gx, wx.z, p.z)))), 0.f);
            v[3] = fmaxf(fmaf(gw, wf.w, fmaf(gz, wz.w, fmaf(gy, wy.w, fmaf(gx, wx.w, p.w)))), 0.f);
            {
                unsigned h01, l01, h23, l23;
                split_pair(v[0], v[1], h01, l01);
                split_pair(v[2], v[3], h23, l23);
                hiw[2 * half] = h01; hiw[2 * half + 1] = h23;
                low[2 * half] = l01; low[2 * half + 1] = l23;
            }
        }
        char* st = S + (size_t)stage * STAGE_CH * 16;
        *reinterpret_cast<uint4*>(st + a_dst + u * 128 * 64) = make_uint4(hiw[0], hiw[1], hiw[2], hiw[3]);
        *reinterpret_cast<uint4*>(st + BM * 64 + a_dst + u * 128 * 64) = make_uint4(low[0], low[1], low[2], low[3]);
      }
    };

    const int r = lane & 31, h = lane >> 5;
    int offA[2][RT], offB[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int ra = wr * 32 * RT + 32 * t + r;
            offA[p][t] = ((p * BM + ra) * 4 + (h ^ ((ra >> 2) & 3))) * 16;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int rb = wc * 64 + 32 * t + r;
            offB[p][t] = (A_CH + (p * BN + rb) * 4 + (h ^ ((rb >> 2) & 3))) * 16;
        }
    }
    f32x16 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Software pipeline over the flattened slab sequence g = 0..total-1 of this workgroup's items:
    //   MFMA stage    : slab g        (reads LDS stage g&1)
    //   produce stage : slab g+1      (A rows: VALU on P values gathered one slab EARLIER, written to stage (g+1)&1;
    //                                  placed between the two MFMA groups of slab g so it co-issues with MFMAs in flight)
    //   gather stage  : slab g+2      (global loads of P rows + W2 DMA of slab g+1 issued right after the barrier)
    // Metadata (source row, normalised offset) of an item is prefetched one item ahead of the gather stage.
    int it_q = 0, s_q = 0;                   // item / slab of the gather stage
    Meta m_q, m_nxt;
    load_meta(0, m_q);
    m_nxt = m_q;
    if (my_items > 1) load_meta(1, m_nxt);
    auto advance_q = [&]() {                 // move the gather stage to the next slab (possibly the next item)
        if (++s_q == nslab) {
            s_q = 0; ++it_q;
            m_q = m_nxt;
            if (it_q + 1 < my_items) load_meta(it_q + 1, m_nxt);
        }
    };
    // prologue: slab 0 produced synchronously, slab 1 gathered
    auto load_epi = [&](int mt_, int nt_, SaEpiRegs<RT>& e) {   // parameters of an item's epilogue, fetched an item ahead
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = nt_ * BN + wc * 64 + j * 32 + (lane & 31);
            const bool cv = col < C2;
            e.bias[j] = cv ? b2[col] : 0.f; e.s[j] = cv ? bn_s[col] : 0.f; e.t[j] = cv ? bn_t[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int tgt = mt_ * (BM / 32) + wr * RT + i;
            e.d[i] = tgt < M ? min(deg[tgt], kw) : 0;
        }
    };
    SaEpiRegs<RT> e_cur, e_1;
    load_epi(item_mt(0), item_nt(0), e_cur);
    e_1 = e_cur;
    const _Float16* wb1 = W2h + (size_t)item_nt(0) * BN * C1pad;   // W2 panel of the item in the produce stage
    int mt_cur = item_mt(0), nt_cur = item_nt(0);                   // item in the MFMA stage
    int mt_1 = mt_cur, nt_1 = nt_cur;                               // item in the produce stage
    __syncthreads();   // Wr staged
    issue(0, wb1, 0);
    gather(m_q, 0, pv);
    produce(0, m_q, 0, pv);
    Vals pn = pv;          // gathered values of slab g+1
    Meta m_n = m_q;
    int k_n = 0;
    if (total > 1) {
        advance_q();
        gather(m_q, s_q * G_BK, pn);
        m_n = m_q; k_n = s_q * G_BK;
    }
    int it = 0, s = 0;                       // item / slab of the MFMA stage
    int it1 = 0, s1 = 0;                     // item / slab of the produce stage (g+1)
    for (int g = 0; g < total; ++g) {
        __syncthreads();  // B(g) landed, A(g) written, stage (g+1)&1 free  (a counted vmcnt that leaves the gather in
                          // flight across the barrier measured 2 % slower)
        const bool more = g + 1 < total;
        if (more) {
            s1 = s + 1; it1 = it;
            if (s1 == nslab) {
                s1 = 0; it1 = it + 1;
                mt_1 = item_mt(it1); nt_1 = item_nt(it1);
                wb1 = W2h + (size_t)nt_1 * BN * C1pad;
                load_epi(mt_1, nt_1, e_1);
            }
            if (!(dbg & 2)) issue((g + 1) & 1, wb1, s1 * G_BK);
        }
        // values for the produce stage were gathered during the previous iteration
        const Vals pu = pn;
        const Meta m_u = m_n;
        const int k_u = k_n;
        // The gather of slab g+2 is issued here and lands in `pg` while this iteration's MFMAs run; it is only moved
        // into the loop-carried registers at the END of the iteration (a register copy is a use: placed here, it
        // would make the compiler wait for the loads before the first MFMA, which is what the kernel used to do).
        Vals pg = pn;
        const bool fetch = g + 2 < total;
        if (fetch) {
            advance_q();
            if (!(dbg & 16)) gather(m_q, s_q * G_BK, pg);
#ifdef P2W_SA_EARLY_COPY   // A/B: the previous placement of the copy (forces the wait before the MFMAs)
            pn = pg; m_n = m_q; k_n = s_q * G_BK;
#endif
        }
        const char* st = S + (size_t)(g & 1) * STAGE_CH * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8 ah[RT], al[RT], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                ah[t] = *reinterpret_cast<const h8*>(st + (offA[0][t] ^ (kk << 5)));
                al[t] = *reinterpret_cast<const h8*>(st + (offA[1][t] ^ (kk << 5)));
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t] = *reinterpret_cast<const h8*>(st + (offB[0][t] ^ (kk << 5)));
                bl[t] = *reinterpret_cast<const h8*>(st + (offB[1][t] ^ (kk << 5)));
            }
            if (!(dbg & 4)) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
            if (kk == 0 && !(dbg & 8)) {  // producer VALU work is interleaved into the gaps of the 12 MFMAs above (1 MFMA : 8 VALU)
                produce((g + 1) & 1, m_u, k_u, pu);   // unconditional: after the last slab it fills a stage nobody reads
#pragma unroll
                for (int q = 0; q < 6 * RT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 16 * NR / RT, 0);   // VALU
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (s == nslab - 1 && !(dbg & 1)) {  // item finished: reduce over neighbour slots and store, then start the next accumulation
            sa_epilogue_regs<RT>(acc, wscale, mt_cur * (BM / 32), nt_cur * BN, wr, wc, lane, M, e_cur, C2, out, ldo, out_h2, ldh);
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
        if (s == nslab - 1) e_cur = e_1;
        s = s1; it = it1; mt_cur = mt_1; nt_cur = nt_1;
        __builtin_amdgcn_sched_barrier(0);
#ifndef P2W_SA_EARLY_COPY
        if (fetch) { pn = pg; m_n = m_q; k_n = s_q * G_BK; }
#endif
    }
}


// ------------------------------------------------------------------------------------------------
// fused PointNetConv v4 (f16x3): producer / consumer wave specialisation
//
// The ablation of the v3 kernel shows its parts adding up instead of overlapping (level 2: skeleton 254 + epilogue
// 140 + W2 DMA 153 + MFMA 394 + producer/gather 169 = 1110 us): all eight waves run the same phase at the same time,
// re-aligned by the per-slab barrier, so the MFMA pipe idles while everybody produces and vice versa.  Here the two
// waves of every SIMD have different jobs: waves 0-3 (consumers, 2 x 2) only read fragments and issue MFMAs for the
// whole BM x BN item tile and run the epilogue; waves 4-7 (producers) issue the W2 DMA, gather the P rows and build
// the A rows of the NEXT slab.  One workgroup barrier per slab hands a stage of the 2-stage LDS ring over; between two
// barriers the MFMA pipe and the VALU / memory pipes of a SIMD work on different slabs.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>   // (128, 256): 4 targets x 256 columns;  (256, 128): 8 targets x 128 columns
__global__ __launch_bounds__(512, 1) void sa_conv16w_kernel(const float* __restrict__ P, int ldp, const int* __restrict__ meta_j,
                                                            const float4* __restrict__ meta_g, const int* __restrict__ deg,
                                                            int kw, int M, const float* __restrict__ w1r4, int C1, int C1pad,
                                                            const _Float16* __restrict__ W2h, size_t plane, float wscale, int C2,
                                                            int nMt, int nNt, const float* __restrict__ b2,
                                                            const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                                            float* __restrict__ out, int ldo, _Float16* __restrict__ out_h2,
                                                            int ldh, int dbg) {
    // dbg (profiling ablations, 0 in production): 1 no epilogue, 2 no W2 DMA, 4 no MFMA, 8 no A production, 16 no P gather
    constexpr int RT = BM / 64, CT = BN / 64;      // consumer wave tile (32 RT) x (32 CT), consumers arranged 2 x 2
    constexpr int A_CH = 8 * BM, STAGE_CH = A_CH + 8 * BN;
    constexpr int NRP = BM / 64;                    // producer rows per thread: 256 producer threads, 4 per row
    constexpr int NIP = (8 * BN) / 64 / 4;          // W2 DMA pieces per producer wave per slab
    __shared__ __attribute__((aligned(16))) char S[2 * STAGE_CH * 16];
    __shared__ __attribute__((aligned(16))) float Wr[4 * 512];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4 * C1pad; i += 512) Wr[i] = w1r4[i];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nitems = nMt * nNt, nslab = C1pad / G_BK;
    int first, stride, limit;   // XCD-aware work assignment, see sa_conv16p_kernel
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int chunk = (nitems + 7) >> 3;
        first = xcd * chunk + slot; stride = per; limit = min((xcd + 1) * chunk, nitems);
    } else {
        first = blockIdx.x; stride = gridDim.x; limit = nitems;
    }
    if (first >= limit) return;
    const int my_items = (limit - first + stride - 1) / stride;
    const int total = my_items * nslab;
    auto item_mt = [&](int it) { return (first + it * stride) / nNt; };
    auto item_nt = [&](int it) { return (first + it * stride) % nNt; };
    __syncthreads();   // Wr staged

    if (wave >= 4) {
        // ---------------------------------------------------------------- producers
        const int pw = wave - 4, ptid = tid - 256;
        const int prow = ptid >> 2, pq = ptid & 3;          // rows prow + 64 u, u < NRP ((row >> 2) & 3 is the same for all)
        const int a_dst = (prow * 4 + (pq ^ ((prow >> 2) & 3))) * 16;
        size_t boff[NIP];
        int dstc[NIP];
#pragma unroll
        for (int i = 0; i < NIP; ++i) {
            const int g2 = pw + 4 * i, p = g2 / (BN / 16), rb = g2 % (BN / 16);
            const int row = 16 * rb + (lane >> 2), q = (lane & 3) ^ ((row >> 2) & 3);
            boff[i] = (size_t)p * plane + (size_t)row * C1pad + 8 * q;
            dstc[i] = A_CH + g2 * 64;
        }
        auto issue = [&](int stage, const _Float16* wbase, int k0) {
#pragma unroll
            for (int i = 0; i < NIP; ++i)
                __builtin_amdgcn_global_load_lds((glb_vp)(wbase + boff[i] + k0),
                                                 (lds_vp)(S + ((size_t)stage * STAGE_CH + dstc[i]) * 16), 16, 0, 0);
        };
        struct Meta { int j[NRP]; float4 g[NRP]; };
        struct Buf { float4 v[NRP][2]; Meta m; int k0; };
        auto load_meta = [&](int it, Meta& m) {
#pragma unroll
            for (int u = 0; u < NRP; ++u) {
                long row = (long)item_mt(it) * BM + prow + 64 * u;
                const long last = (long)M * 32 - 1;
                row = row < last ? row : last;
                m.j[u] = meta_j[row];
                m.g[u] = meta_g[row];
            }
        };
        auto gather = [&](Buf& b) {   // b.m / b.k0 set by the caller
            const int k = b.k0 + 8 * pq;
#pragma unroll
            for (int u = 0; u < NRP; ++u) {
                const bool on = b.m.j[u] >= 0;
                const float* p = P + (size_t)(on ? b.m.j[u] : 0) * ldp + 8 * pq + b.k0;
                b.v[u][0] = (on && k < C1) ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
                b.v[u][1] = (on && k + 4 < C1) ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto produce = [&](int stage, const Buf& b) {
            const int k = b.k0 + 8 * pq;
#pragma unroll
            for (int u = 0; u < NRP; ++u) {
                const float4 rg = b.m.g[u];
                const bool on = b.m.j[u] >= 0;
                unsigned hiw[4], low[4];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int kk = k + 4 * half;
                    const float4 wx = *reinterpret_cast<const float4*>(&Wr[0 * C1pad + kk]);
                    const float4 wy = *reinterpret_cast<const float4*>(&Wr[1 * C1pad + kk]);
                    const float4 wz = *reinterpret_cast<const float4*>(&Wr[2 * C1pad + kk]);
                    const float4 wf = *reinterpret_cast<const float4*>(&Wr[3 * C1pad + kk]);
                    const float4 p = b.v[u][half];
                    const float gx = on ? rg.x : 0.f, gy = on ? rg.y : 0.f, gz = on ? rg.z : 0.f, gw = on ? rg.w : 0.f;
                    float v[4];
                    v[0] = fmaxf(fmaf(gw, wf.x, fmaf(gz, wz.x, fmaf(gy, wy.x, fmaf(gx, wx.x, p.x)))), 0.f);
                    v[1] = fmaxf(fmaf(gw, wf.y, fmaf(gz, wz.y, fmaf(gy, wy.y, fmaf(gx, wx.y, p.y)))), 0.f);
                    v[2] = fmaxf(fmaf(gw, wf.z, fmaf(gz, wz.z, fmaf(gy, wy.z, fmaf(gx, wx.z, p.z)))), 0.f);
                    v[3] = fmaxf(fmaf(gw, wf.w, fmaf(gz, wz.w, fmaf(gy, wy.w, fmaf(gx, wx.w, p.w)))), 0.f);
                    unsigned h01, l01, h23, l23;
                    split_pair(v[0], v[1], h01, l01);
                    split_pair(v[2], v[3], h23, l23);
                    hiw[2 * half] = h01; hiw[2 * half + 1] = h23;
                    low[2 * half] = l01; low[2 * half + 1] = l23;
                }
                char* st = S + (size_t)stage * STAGE_CH * 16;
                *reinterpret_cast<uint4*>(st + a_dst + u * 64 * 64) = make_uint4(hiw[0], hiw[1], hiw[2], hiw[3]);
                *reinterpret_cast<uint4*>(st + BM * 64 + a_dst + u * 64 * 64) = make_uint4(low[0], low[1], low[2], low[3]);
            }
        };
        // gather stage cursor (item, slab) and its metadata; the next item's metadata is fetched an item ahead
        int it_q = 0, s_q = 0;
        Meta m_q, m_nxt;
        load_meta(0, m_q);
        m_nxt = m_q;
        if (my_items > 1) load_meta(1, m_nxt);
        auto advance_q = [&]() {
            if (++s_q == nslab) {
                s_q = 0; ++it_q;
                m_q = m_nxt;
                if (it_q + 1 < my_items) load_meta(it_q + 1, m_nxt);
            }
        };
        Buf b0, b1, b2_;
        // prologue: slab 0 complete in stage 0, slabs 1 and 2 gathered
        issue(0, W2h + (size_t)item_nt(0) * BN * C1pad, 0);
        b0.m = m_q; b0.k0 = 0;
        gather(b0);
        produce(0, b0);
        b1 = b0; b2_ = b0;
        if (total > 1) {
            advance_q();
            b1.m = m_q; b1.k0 = s_q * G_BK;
            gather(b1);
        }
        if (total > 2) {
            advance_q();
            b2_.m = m_q; b2_.k0 = s_q * G_BK;
            gather(b2_);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // step g: W2 DMA + A rows of slab g+1 into stage (g+1)&1 from buffer `use` (gathered TWO steps earlier, so a
        // whole step of MFMA time covers the latency of the scattered P-row loads), gather of slab g+3 into `fill`
        // (= the buffer whose slab g was produced in the previous step)
        int it1 = 0, s1 = 0;   // item / slab of g+1
        auto pstep = [&](int g, const Buf& use, Buf& fill) {
            const bool fetch = g + 3 < total;
            if (g + 1 < total) {
                if (++s1 == nslab) { s1 = 0; ++it1; }
                if (!(dbg & 2)) issue((g + 1) & 1, W2h + (size_t)item_nt(it1) * BN * C1pad, s1 * G_BK);
                if (!(dbg & 8)) produce((g + 1) & 1, use);
                if (fetch) {
                    advance_q();
                    fill.m = m_q; fill.k0 = s_q * G_BK;
                    if (!(dbg & 16)) gather(fill);
                }
            }
            // the DMA was issued before this step's gather and loads return in order: with at most that gather's
            // 2 NRP loads outstanding the DMA has landed (the gather issued in the previous step is older still)
            if (!fetch) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (NRP == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        for (int g = 0; g < total; g += 3) {     // slab g+1 lives in b1 / b2_ / b0, slab g+3 goes where slab g was
            pstep(g, b1, b0);
            if (g + 1 < total) pstep(g + 1, b2_, b1);
            if (g + 2 < total) pstep(g + 2, b0, b2_);
        }
        return;
    }

    // -------------------------------------------------------------------- consumers
    const int cwr = wave >> 1, cwc = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    int offA[2][RT], offB[2][CT];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int ra = cwr * 32 * RT + 32 * t + r;
            offA[p][t] = ((p * BM + ra) * 4 + (h ^ ((ra >> 2) & 3))) * 16;
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int rb = cwc * 32 * CT + 32 * t + r;
            offB[p][t] = (A_CH + (p * BN + rb) * 4 + (h ^ ((rb >> 2) & 3))) * 16;
        }
    }
    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // prologue done: slab 0 is in stage 0
    int it = 0, s = 0;
    float e_bias[CT], e_s[CT], e_t[CT];
    int e_d[RT];
    for (int g = 0; g < total; ++g) {
        const int mt = item_mt(it), nt = item_nt(it);
        if (s == nslab - 1) {   // epilogue parameters of this item: fetched under its last slab's MFMAs
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int col = nt * BN + cwc * 32 * CT + j * 32 + (lane & 31);
                const bool cv = col < C2;
                e_bias[j] = cv ? b2[col] : 0.f; e_s[j] = cv ? bn_s[col] : 0.f; e_t[j] = cv ? bn_t[col] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int tgt = mt * (BM / 32) + cwr * RT + i;
                e_d[i] = tgt < M ? min(deg[tgt], kw) : 0;
            }
        }
        const char* st = S + (size_t)(g & 1) * STAGE_CH * 16;
        h8 ah[2][RT], al[2][RT], bh[2][CT], bl[2][CT];   // both half-slabs' fragments are in flight before the first MFMA
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                ah[kk][t] = *reinterpret_cast<const h8*>(st + (offA[0][t] ^ (kk << 5)));
                al[kk][t] = *reinterpret_cast<const h8*>(st + (offA[1][t] ^ (kk << 5)));
            }
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                bh[kk][t] = *reinterpret_cast<const h8*>(st + (offB[0][t] ^ (kk << 5)));
                bl[kk][t] = *reinterpret_cast<const h8*>(st + (offB[1][t] ^ (kk << 5)));
            }
        }
        if (!(dbg & 4)) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk][i], bh[kk][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk][i], bl[kk][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk][i], bh[kk][j], acc[i][j], 0, 0, 0);
                    }
        }
        if (s == nslab - 1 && !(dbg & 1)) {   // item finished: max over the neighbour slots of each target, store, restart the accumulation
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int tgt = mt * (BM / 32) + cwr * RT + i;
                const int d = e_d[i];
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const int col = nt * BN + cwc * 32 * CT + j * 32 + (lane & 31);
                    // bias + ReLU + BN affine are monotone in the accumulator (wscale > 0; increasing for s >= 0,
                    // decreasing for s < 0) and fp rounding keeps (weak) monotonicity, so the maximum over the slots
                    // of the transformed values is the transform of the maximum (s >= 0) or minimum (s < 0) of the
                    // raw accumulators, bit for bit: 2 VALU per value instead of 5, the transform once per column
                    const float sgn = e_s[j] < 0.f ? -1.f : 1.f;
                    float ext = -INFINITY;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int slot = (q & 3) + 8 * (q >> 2) + 4 * h;
                        if (slot < d) ext = fmaxf(ext, sgn * acc[i][j][q]);
                        acc[i][j][q] = 0.f;
                    }
                    ext = fmaxf(ext, __shfl_xor(ext, 32));
                    float vmax = fmaf(fmaxf(fmaf(sgn * ext, wscale, e_bias[j]), 0.f), e_s[j], e_t[j]);
                    if (d == 0) vmax = 0.f;
                    if (tgt < M) {
                        if (col < C2 && h == 0 && out) out[(size_t)tgt * ldo + col] = vmax;
                        if (out_h2) {
                            const float nb = __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(vmax), 0xB1, 0xf, 0xf, false));
                            if (h == 0 && (lane & 1) == 0 && col < ldh) {
                                unsigned hw, lw;
                                split_pair(vmax, nb, hw, lw);
                                _Float16* p = out_h2 + (size_t)tgt * (2 * ldh) + col;
                                *reinterpret_cast<unsigned*>(p) = hw;
                                *reinterpret_cast<unsigned*>(p + ldh) = lw;
                            }
                        }
                    }
                }
            }
        }
        if (++s == nslab) { s = 0; ++it; }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage g&1 read; stage (g+1)&1 is complete
    }
}

extern "C" int32_t p2w_sa_conv_f16x3(const float* P, int32_t ldp, const float* xyzr_src, const int32_t* idx,
                                     const int32_t* batch_dst, const float* sf, const int32_t* nbr, const int32_t* deg,
                                     int32_t kw, int32_t M, const float* w1r4, const void* W2h, float wscale, int32_t C1,
                                     int32_t C2, const float* b2, const float* bn_s, const float* bn_t, float* out,
                                     int32_t ldo, void* out_h2, int32_t ldh, void* ws, size_t ws_bytes,
                                     p2w_stream_t stream) {
    if (M == 0) return P2W_OK;
    P2W_CHECK_PTR(P); P2W_CHECK_PTR(xyzr_src); P2W_CHECK_PTR(idx); P2W_CHECK_PTR(batch_dst); P2W_CHECK_PTR(sf);
    P2W_CHECK_PTR(nbr); P2W_CHECK_PTR(deg); P2W_CHECK_PTR(w1r4); P2W_CHECK_PTR(W2h); P2W_CHECK_PTR(b2);
    P2W_CHECK_PTR(bn_s); P2W_CHECK_PTR(bn_t);
    if (!out && !out_h2) return P2W_ENULL;
    P2W_CHECK_ALIGN16(P); P2W_CHECK_ALIGN16(xyzr_src); P2W_CHECK_ALIGN16(w1r4); P2W_CHECK_ALIGN16(W2h);
    if (M < 0 || kw <= 0 || kw > 32 || C1 <= 0 || C2 <= 0 || (C1 & 3) || (ldp & 3) || ldp < C1 || (out && ldo < C2) ||
        (out_h2 && (ldh < C2 || (ldh & 7))) || !(wscale > 0.f))
        return P2W_EINVAL;
    int C2pad, C1pad;
    p2w_packed_dims(C2, C1, &C2pad, &C1pad);
    static const int sa_v1 = []() { const char* e = getenv("P2W_SA_V1"); return e ? atoi(e) : 0; }();
    if (sa_v1 == 0 && ws != nullptr && ws_bytes >= (size_t)M * 32 * 20 && C1pad <= 512) {
        // v3: edge metadata pre-pass + persistent pipelined kernel (one workgroup per CU)
        if (reinterpret_cast<uintptr_t>(ws) & 15u) return P2W_EALIGN;
        float4* meta_g = static_cast<float4*>(ws);
        int* meta_j = reinterpret_cast<int*>(meta_g + (size_t)M * 32);
        sa_edge_meta_kernel<<<p2w_cdiv((long)M * 32, 256), 256, 0, p2w_s(stream)>>>(
            reinterpret_cast<const float4*>(xyzr_src), idx, batch_dst, sf, nbr, deg, kw, M, meta_j, meta_g);
        static const int n_cu = []() {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            return n > 0 ? n : 256;
        }();
        const bool wide = C2 > 128;
        const int sadbg = getenv("P2W_SA_DBG") ? atoi(getenv("P2W_SA_DBG")) : 0;
        // (a <256, 4> instance - 8 targets x 256 columns, wave tile 128 x 64 - halves the DMA / barrier / per-item cost per
        // FLOP but needs 125 spilled VGPRs next to the producer state and measured 2 x SLOWER: not instantiated)
        const int nMt3 = p2w_cdiv(M, wide ? 4 : 8), nNt3 = p2w_cdiv(C2, wide ? 256 : 128);
        const long items = (long)nMt3 * nNt3;
        int grid = (int)(items < n_cu ? items : n_cu);
        if (grid >= 8) grid &= ~7;   // whole XCD rounds (see the kernel's work assignment)
        // P2W_SA_V=4 selects the producer / consumer variant: 4 % less kernel time when it runs alone, but its 232 VGPRs
        // and 104 KB of LDS leave no room for the geometry stream's workgroups next to it, so the pipelined step is 0.7 %
        // SLOWER than with the 128-VGPR kernel below (9.85 vs 9.78 ms, alternating runs on one box): not the default
        static const int sa_ver = []() { const char* e = getenv("P2W_SA_V"); return e ? atoi(e) : 3; }();
        if (sa_ver == 4) {
            if (wide)
                sa_conv16w_kernel<128, 256><<<grid, 512, 0, p2w_s(stream)>>>(
                    P, ldp, meta_j, meta_g, deg, kw, M, w1r4, C1, C1pad, static_cast<const _Float16*>(W2h),
                    (size_t)C2pad * C1pad, wscale, C2, nMt3, nNt3, b2, bn_s, bn_t, out, ldo, static_cast<_Float16*>(out_h2), ldh, sadbg);
            else
                sa_conv16w_kernel<256, 128><<<grid, 512, 0, p2w_s(stream)>>>(
                    P, ldp, meta_j, meta_g, deg, kw, M, w1r4, C1, C1pad, static_cast<const _Float16*>(W2h),
                    (size_t)C2pad * C1pad, wscale, C2, nMt3, nNt3, b2, bn_s, bn_t, out, ldo, static_cast<_Float16*>(out_h2), ldh, sadbg);
            return P2W_LAUNCH_STATUS();
        }
        if (wide)
            sa_conv16p_kernel<256, 2><<<grid, 512, 0, p2w_s(stream)>>>(
                P, ldp, meta_j, meta_g, deg, kw, M, w1r4, C1, C1pad, static_cast<const _Float16*>(W2h), (size_t)C2pad * C1pad,
                wscale, C2, nMt3, nNt3, b2, bn_s, bn_t, out, ldo, static_cast<_Float16*>(out_h2), ldh, sadbg);
        else
            sa_conv16p_kernel<128, 2><<<grid, 512, 0, p2w_s(stream)>>>(
                P, ldp, meta_j, meta_g, deg, kw, M, w1r4, C1, C1pad, static_cast<const _Float16*>(W2h), (size_t)C2pad * C1pad,
                wscale, C2, nMt3, nNt3, b2, bn_s, bn_t, out, ldo, static_cast<_Float16*>(out_h2), ldh, sadbg);
        return P2W_LAUNCH_STATUS();
    }
    if (C2 >= 256 && sa_v1 != 1) {  // wide layers: 128 x 256 tile, W2 on the DMA ring, A produced one slab ahead
        const int nMt2 = p2w_cdiv(M, 4), nNt2 = p2w_cdiv(C2, 256);
        sa_conv16g_kernel<<<tile_grid(nMt2, nNt2), 512, 0, p2w_s(stream)>>>(
            P, ldp, reinterpret_cast<const float4*>(xyzr_src), idx, batch_dst, sf, nbr, deg, kw, M, w1r4, C1, C1pad,
            static_cast<const _Float16*>(W2h), (size_t)C2pad * C1pad, wscale, C2, nMt2, nNt2, b2, bn_s, bn_t, out, ldo,
            static_cast<_Float16*>(out_h2), ldh, getenv("P2W_SA_DBG") ? atoi(getenv("P2W_SA_DBG")) : 0);
        return P2W_LAUNCH_STATUS();
    }
    const int nMt = p2w_cdiv(M, 4), nNt = p2w_cdiv(C2, G_BN);
    sa_conv16_kernel<<<tile_grid(nMt, nNt), 256, 0, p2w_s(stream)>>>(
        P, ldp, reinterpret_cast<const float4*>(xyzr_src), idx, batch_dst, sf, nbr, deg, kw, M, w1r4, C1, C1pad,
        static_cast<const _Float16*>(W2h), (size_t)C2pad * C1pad, wscale, C2, nMt, nNt, b2, bn_s, bn_t, out, ldo,
        static_cast<_Float16*>(out_h2), ldh);
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// small HBM-bound kernels
// ------------------------------------------------------------------------------------------------
// 4 consecutive output columns of one row -> fp32 row (pitch ldo) and/or H2 row (pitch ldh)
__device__ __forceinline__ void store4(const OutArgs& o, size_t row, int c, const float (&v)[4]) {
    if (o.f32 && c < o.ldo) *reinterpret_cast<float4*>(&o.f32[row * o.ldo + c]) = make_float4(v[0], v[1], v[2], v[3]);
    if (o.h2 && c < o.ldh) h2_store4(o.h2, o.ldh, row, c, v);
}

__global__ __launch_bounds__(256) void stem_kernel(const float4* __restrict__ xyzr, int n, const float* __restrict__ w,
                                                   const float* __restrict__ b, int C, int q4, OutArgs o) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per (row, 4 channels incl. zero padding)
    if (g >= (long)n * q4) return;
    const int row = (int)(g / q4), c0 = (int)(g % q4) * 4;
    const float4 p = xyzr[row];
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        v[e] = (c < C) ? fmaxf(fmaf(p.z, w[c * 3 + 2], fmaf(p.y, w[c * 3 + 1], fmaf(p.x, w[c * 3 + 0], b[c]))), 0.f) : 0.f;
    }
    store4(o, (size_t)row, c0, v);
}

static int32_t stem_launch(const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out, void* out_h2,
                           int32_t ldh, p2w_stream_t stream) {
    if (n == 0) return P2W_OK;
    P2W_CHECK_PTR(xyzr); P2W_CHECK_PTR(w); P2W_CHECK_PTR(b); P2W_CHECK_ALIGN16(xyzr);
    if (!out && !out_h2) return P2W_ENULL;
    if (n < 0 || C <= 0 || (C & 3) || (out_h2 && (ldh < C || (ldh & 7)))) return P2W_EINVAL;
    OutArgs o = {out, C, static_cast<_Float16*>(out_h2), out_h2 ? ldh : 0};
    const int q4 = (out_h2 ? ldh : C) >> 2;
    stem_kernel<<<p2w_cdiv((long)n * q4, 256), 256, 0, p2w_s(stream)>>>(reinterpret_cast<const float4*>(xyzr), n, w, b, C, q4, o);
    return P2W_LAUNCH_STATUS();
}
extern "C" int32_t p2w_stem(const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out,
                            p2w_stream_t stream) {
    P2W_CHECK_PTR(out);
    return stem_launch(xyzr, n, w, b, C, out, nullptr, 0, stream);
}
extern "C" int32_t p2w_stem_h2(const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out,
                               void* out_h2, int32_t ldh, p2w_stream_t stream) {
    return stem_launch(xyzr, n, w, b, C, out, out_h2, ldh, stream);
}

// One wave per output row: the row's neighbours, their inverse-square-distance weights and the denominator are
// wave-uniform (scalar loads, computed once per row instead of once per 4-column chunk); the lanes then sweep the row
// 256 columns at a time with coalesced 16-byte loads of the coarse features.  The arithmetic per element is the same
// as torch-scatter's: products and sums in neighbour order from 0, then a literal division.
#ifndef P2W_IC_ROWS
#define P2W_IC_ROWS 1   // swept 1..16 on the bench forward: 447 / 473 / 500 / 582 / 682 us for 1 / 2 / 4 / 8 / 16
#endif
constexpr int IC_ROWS = P2W_IC_ROWS;   // rows per wave (consecutive)
__global__ __launch_bounds__(256) void interp_concat_kernel(const float* __restrict__ xc, int Fc, const float4* __restrict__ xyzr_c,
                                                            const float4* __restrict__ xyzr_f, const int* __restrict__ nbr,
                                                            const int* __restrict__ deg, int kw, const float* __restrict__ skip,
                                                            int Fs, int m, int q4, OutArgs o) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = (blockIdx.x * 4 + wave) * IC_ROWS;
    for (int rr = 0; rr < IC_ROWS; ++rr) {
        const int q = q0 + rr;   // wave-uniform
        if (q >= m) return;
        const int d = min(deg[q], kw);
        const float4 pf = xyzr_f[q];
        // up to 4 neighbours keep (index, weight) in registers; more (not used by the model) are re-derived per chunk
        int js[4] = {0, 0, 0, 0};
        float ws[4] = {0.f, 0.f, 0.f, 0.f};
        float den = 0.f;
        for (int s = 0; s < d; ++s) {
            const int j = nbr[(size_t)q * kw + s];
            const float4 pc = xyzr_c[j];
            const float dx = pc.x - pf.x, dy = pc.y - pf.y, dz = pc.z - pf.z;
            const float d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
            const float w = 1.0f / fmaxf(d2, 1e-16f);
            if (s < 4) { js[s] = j; ws[s] = w; }
            den = den + w;
        }
        for (int c = 4 * lane; c < 4 * q4; c += 256) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (c < Fc) {
                float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int s = 0; s < d; ++s) {
                    int j; float w;
                    if (s < 4) { j = js[s]; w = ws[s]; }
                    else {
                        j = nbr[(size_t)q * kw + s];
                        const float4 pc = xyzr_c[j];
                        const float dx = pc.x - pf.x, dy = pc.y - pf.y, dz = pc.z - pf.z;
                        w = 1.0f / fmaxf(((dx * dx) + (dy * dy)) + (dz * dz), 1e-16f);
                    }
                    const float4 x = *reinterpret_cast<const float4*>(&xc[(size_t)j * Fc + c]);
                    num.x = num.x + x.x * w; num.y = num.y + x.y * w; num.z = num.z + x.z * w; num.w = num.w + x.w * w;
                }
                if (d > 0) { v[0] = num.x / den; v[1] = num.y / den; v[2] = num.z / den; v[3] = num.w / den; }
            } else if (c < Fc + Fs) {
                const float4 t = *reinterpret_cast<const float4*>(&skip[(size_t)q * Fs + (c - Fc)]);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            }
            store4(o, (size_t)q, c, v);
        }
    }
}

static int32_t interp_launch(const float* xc, int32_t Fc, const float* xyzr_c, const float* xyzr_f, const int32_t* nbr,
                             const int32_t* deg, int32_t kw, const float* skip, int32_t Fs, int32_t m, float* out, int32_t ldo,
                             void* out_h2, int32_t ldh, p2w_stream_t stream) {
    if (m == 0) return P2W_OK;
    P2W_CHECK_PTR(xc); P2W_CHECK_PTR(xyzr_c); P2W_CHECK_PTR(xyzr_f); P2W_CHECK_PTR(nbr); P2W_CHECK_PTR(deg);
    if (!out && !out_h2) return P2W_ENULL;
    if (Fs > 0) { P2W_CHECK_PTR(skip); P2W_CHECK_ALIGN16(skip); }
    P2W_CHECK_ALIGN16(xc); P2W_CHECK_ALIGN16(xyzr_c); P2W_CHECK_ALIGN16(xyzr_f);
    if (out) P2W_CHECK_ALIGN16(out);
    if (out_h2) P2W_CHECK_ALIGN16(out_h2);
    if (m < 0 || kw <= 0 || Fc <= 0 || Fs < 0 || (Fc & 3) || (Fs & 3)) return P2W_EINVAL;
    if (out && ((ldo & 3) || ldo < Fc + Fs)) return P2W_EINVAL;
    if (out_h2 && ((ldh & 7) || ldh < Fc + Fs)) return P2W_EINVAL;
    const int width = (out ? ldo : 0) > (out_h2 ? ldh : 0) ? ldo : ldh;
    OutArgs o = {out, out ? ldo : 0, static_cast<_Float16*>(out_h2), out_h2 ? ldh : 0};
    interp_concat_kernel<<<p2w_cdiv(m, 4 * IC_ROWS), 256, 0, p2w_s(stream)>>>(
        xc, Fc, reinterpret_cast<const float4*>(xyzr_c), reinterpret_cast<const float4*>(xyzr_f), nbr, deg, kw, skip, Fs, m,
        width >> 2, o);
    return P2W_LAUNCH_STATUS();
}
extern "C" int32_t p2w_interp_concat(const float* xc, int32_t Fc, const float* xyzr_c, const float* xyzr_f, const int32_t* nbr,
                                     const int32_t* deg, int32_t kw, const float* skip, int32_t Fs, int32_t m, float* out,
                                     int32_t ldo, p2w_stream_t stream) {
    P2W_CHECK_PTR(out);
    return interp_launch(xc, Fc, xyzr_c, xyzr_f, nbr, deg, kw, skip, Fs, m, out, ldo, nullptr, 0, stream);
}
extern "C" int32_t p2w_interp_concat_h2(const float* xc, int32_t Fc, const float* xyzr_c, const float* xyzr_f,
                                        const int32_t* nbr, const int32_t* deg, int32_t kw, const float* skip, int32_t Fs,
                                        int32_t m, void* out_h2, int32_t ldh, p2w_stream_t stream) {
    P2W_CHECK_PTR(out_h2);
    return interp_launch(xc, Fc, xyzr_c, xyzr_f, nbr, deg, kw, skip, Fs, m, nullptr, 0, out_h2, ldh, stream);
}

__global__ __launch_bounds__(256) void concat_xyz_kernel(const float* __restrict__ x, int F, const float4* __restrict__ xyzr,
                                                         int m, int q4, OutArgs o) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long)m * q4) return;
    const int q = (int)(g / q4), c = (int)(g % q4) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < F) {
        const float4 t = *reinterpret_cast<const float4*>(&x[(size_t)q * F + c]);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if (c == F) {
        const float4 p = xyzr[q];
        v[0] = p.x; v[1] = p.y; v[2] = p.z;
    }
    store4(o, (size_t)q, c, v);
}

static int32_t concat_launch(const float* x, int32_t F, const float* xyzr, int32_t m, float* out, int32_t ldo, void* out_h2,
                             int32_t ldh, p2w_stream_t stream) {
    if (m == 0) return P2W_OK;
    P2W_CHECK_PTR(x); P2W_CHECK_PTR(xyzr);
    if (!out && !out_h2) return P2W_ENULL;
    P2W_CHECK_ALIGN16(x); P2W_CHECK_ALIGN16(xyzr);
    if (m < 0 || F <= 0 || (F & 3)) return P2W_EINVAL;
    if (out && ((ldo & 3) || ldo < F + 4)) return P2W_EINVAL;
    if (out_h2 && ((ldh & 7) || ldh < F + 4)) return P2W_EINVAL;
    const int width = (out ? ldo : 0) > (out_h2 ? ldh : 0) ? ldo : ldh;
    OutArgs o = {out, out ? ldo : 0, static_cast<_Float16*>(out_h2), out_h2 ? ldh : 0};
    concat_xyz_kernel<<<p2w_cdiv((long)m * (width >> 2), 256), 256, 0, p2w_s(stream)>>>(
        x, F, reinterpret_cast<const float4*>(xyzr), m, width >> 2, o);
    return P2W_LAUNCH_STATUS();
}
extern "C" int32_t p2w_concat_xyz(const float* x, int32_t F, const float* xyzr, int32_t m, float* out, int32_t ldo,
                                  p2w_stream_t stream) {
    P2W_CHECK_PTR(out);
    return concat_launch(x, F, xyzr, m, out, ldo, nullptr, 0, stream);
}
extern "C" int32_t p2w_concat_xyz_h2(const float* x, int32_t F, const float* xyzr, int32_t m, void* out_h2, int32_t ldh,
                                     p2w_stream_t stream) {
    P2W_CHECK_PTR(out_h2);
    return concat_launch(x, F, xyzr, m, nullptr, 0, out_h2, ldh, stream);
}

// global max pool: (voxel, 64-column group, row split) blocks; 4 row lanes x 64 columns per block, LDS combine, then
// one order-preserving-uint atomicMax per column into `out` (pre-set to 0 = below every float); a second tiny kernel
// decodes in place.  Empty voxels decode to 0 like the reference's scatter.
__device__ __forceinline__ unsigned sm_f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(256) void segment_max_kernel(const float* __restrict__ x, int ldx, int F, const int* __restrict__ ptr,
                                                          unsigned* __restrict__ out) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int s = ptr[b], e = ptr[b + 1];
    const int per = (e - s + gridDim.z - 1) / gridDim.z;
    const int r0 = s + blockIdx.z * per, r1 = min(e, r0 + per);
    float v = -INFINITY;
    if (c < F)
        for (int r = r0 + rl; r < r1; r += 4) v = fmaxf(v, x[(size_t)r * ldx + c]);
    red[rl][threadIdx.x & 63] = v;
    __syncthreads();
    if (rl == 0 && c < F && r1 > r0) {
        v = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
        atomicMax(&out[(size_t)b * F + c], sm_f2ord(v));
    }
}
__global__ __launch_bounds__(256) void segment_max_decode_kernel(unsigned* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned o = out[i];
    const unsigned u = (o == 0u) ? 0u : ((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
    out[i] = u;  // same bits reinterpreted as float by the caller
}

extern "C" int32_t p2w_segment_max(const float* x, int32_t ldx, int32_t F, const int32_t* ptr, int32_t B, float* out,
                                   p2w_stream_t stream) {
    P2W_CHECK_PTR(x); P2W_CHECK_PTR(ptr); P2W_CHECK_PTR(out);
    if (B <= 0 || F <= 0 || ldx < F) return P2W_EINVAL;
    hipStream_t s = p2w_s(stream);
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * F, s);
    if (e != hipSuccess) return (int32_t)e;
    segment_max_kernel<<<dim3(p2w_cdiv(F, 64), B, 16), 256, 0, s>>>(x, ldx, F, ptr, reinterpret_cast<unsigned*>(out));
    segment_max_decode_kernel<<<p2w_cdiv((long)B * F, 256), 256, 0, s>>>(reinterpret_cast<unsigned*>(out), B * F);
    return P2W_LAUNCH_STATUS();
}

// one wave per row: 16-byte loads, lane-strided, shuffle reduction
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, int ldx, int F, const float* __restrict__ w,
                                                     float b, int m, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= m) return;
    float acc = 0.f;
    for (int c = lane * 4; c < F; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(&x[(size_t)row * ldx + c]);
        const float4 ww = *reinterpret_cast<const float4*>(&w[c]);
        acc = fmaf(v.x, ww.x, acc); acc = fmaf(v.y, ww.y, acc); acc = fmaf(v.z, ww.z, acc); acc = fmaf(v.w, ww.w, acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) out[row] = acc + b;
}

extern "C" int32_t p2w_rowdot(const float* x, int32_t ldx, int32_t F, const float* w, float b, int32_t m, float* out,
                              p2w_stream_t stream) {
    if (m == 0) return P2W_OK;
    P2W_CHECK_PTR(x); P2W_CHECK_PTR(w); P2W_CHECK_PTR(out); P2W_CHECK_ALIGN16(x); P2W_CHECK_ALIGN16(w);
    if (m < 0 || F <= 0 || (F & 3) || (ldx & 3) || ldx < F) return P2W_EINVAL;
    rowdot_kernel<<<p2w_cdiv(m, 4), 256, 0, p2w_s(stream)>>>(x, ldx, F, w, b, m, out);
    return P2W_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
extern "C" int32_t p2w_version(void) { return 100; }

extern "C" const char* p2w_strerror(int32_t code) {
    switch (code) {
        case P2W_OK: return "ok";
        case P2W_EINVAL: return "p2w: invalid argument (size, stride or k out of range)";
        case P2W_ENULL: return "p2w: required pointer is NULL";
        case P2W_EALIGN: return "p2w: pointer or stride is not 16-byte aligned";
        case P2W_EWORKSPACE: return "p2w: workspace too small";
        case P2W_EUNSUPPORTED: return "p2w: unsupported configuration";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "p2w: unknown error";
}
