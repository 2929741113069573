// Feature kernels: stem, fp32 MFMA GEMM + fused epilogue, fused PointNetConv (fp32 parity form), kNN-interpolation +
// concat, segment max, row dot, and the extern "C" entry points of the H (fp16 / bf16 plane) family whose kernels
// live in p2w_hgemm.h.  gfx950 only.
//
// MFMA core: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate - the parity mode).
//   A operand: lane l supplies A[row = l&31][k = l>>5];  B operand: B[k = l>>5][col = l&31];
//   C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5), reg in [0,16).
// Both operands are staged k-contiguous in LDS ([row][k] and [col][k], row stride 36 floats = one
// ds_read_b128 of padding -> conflict-free) so one 16-byte LDS read feeds 4 MFMA steps: within an
// 8-wide k group, lane half h takes k = 4h..4h+3 and step s uses element s (the k order inside the
// group is permuted identically for A and B, which leaves the sum unchanged).
#include "p2w_hgemm.h"

constexpr int G_BM = 128, G_BN = 128, G_BK = 32, G_LD = 36;

extern "C" void p2w_packed_dims(int32_t N, int32_t K, int32_t* N_pad, int32_t* K_pad) {
    if (N_pad) *N_pad = (N + 255) / 256 * 256;  // widest column tile of any kernel
    if (K_pad) *K_pad = (K + G_BK - 1) / G_BK * G_BK;
}
extern "C" int32_t p2w_packed_dims_h(int32_t prec, int32_t N, int32_t K, int32_t* N_pad, int32_t* K_pad) {
    if (prec < P2W_PREC_F16X3 || prec > P2W_PREC_BF16) return P2W_EINVAL;
    const int ka = prec == P2W_PREC_F16X3 ? 32 : 64;   // single-plane slabs are 64 k wide
    if (N_pad) *N_pad = (N + 255) / 256 * 256;
    if (K_pad) *K_pad = (K + ka - 1) / ka * ka;
    return P2W_OK;
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
        if (epi->interp) return P2W_EUNSUPPORTED;   // (the interpolated residual is p2w_gemm_h2's)
        ep = {epi->bias, epi->sc0, epi->sh0, epi->sc1, epi->sh1, epi->residual,
              epi->ldr, epi->relu0, epi->relu1, epi->relu2, epi->relu_final, nullptr};
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
// layer-2 bias + ReLU + BN affine, then max over the target's valid neighbour slots (rows of the 32-row MFMA tile)
__device__ __forceinline__ void sa_epilogue(const f32x16 (&acc)[2][2], float wscale, int t0, int n0, int wr, int wc, int lane,
                                            int M, int kw, const int* __restrict__ deg, int C2, const float* __restrict__ b2,
                                            const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                            float* __restrict__ out, int ldo) {
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
            if (cv && h == 0) out[(size_t)tgt * ldo + col] = vmax;
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
// H family entry points: argument checks here, kernels in p2w_hgemm.h (f16x3 instantiated in this file, the
// single-plane precisions in p2w_feat_h1.hip)
// ------------------------------------------------------------------------------------------------
static int32_t gemm_h2_checked(int32_t prec, const void* A_h, int32_t ldh_a, const void* Wh, float wscale, int32_t M, int32_t N,
                               int32_t K, const p2w_epilogue* epi, float* out_f32, int32_t ldo, void* out_h, int32_t ldh_o,
                               void* ws, size_t ws_bytes, int32_t flags, p2w_stream_t stream) {
    if (prec < P2W_PREC_F16X3 || prec > P2W_PREC_BF16) return P2W_EINVAL;
    if (ws) P2W_CHECK_ALIGN16(ws);
    if ((flags & P2W_GEMM_STREAMK) && (flags & P2W_GEMM_NO_STREAMK)) return P2W_EINVAL;
    if (M == 0) return P2W_OK;
    P2W_CHECK_PTR(A_h); P2W_CHECK_PTR(Wh);
    if (!out_f32 && !out_h) return P2W_ENULL;
    P2W_CHECK_ALIGN16(A_h); P2W_CHECK_ALIGN16(Wh);
    if (out_h) P2W_CHECK_ALIGN16(out_h);
    if (M < 0 || N <= 0 || K <= 0 || ldh_a < K || !(wscale > 0.f)) return P2W_EINVAL;
    if (out_f32 && ldo < N) return P2W_EINVAL;
    if (out_h && ldh_o < N) return P2W_EINVAL;
    if ((flags & P2W_GEMM_TILE_128) && (flags & P2W_GEMM_TILE_256)) return P2W_EINVAL;
    EpiArgs ep = {};
    if (epi) {
        if ((epi->sc0 && !epi->sh0) || (epi->sc1 && !epi->sh1)) return P2W_ENULL;
        if (epi->residual && epi->ldr < N) return P2W_EINVAL;
        ep = {epi->bias, epi->sc0, epi->sh0, epi->sc1, epi->sh1, epi->residual,
              epi->ldr, epi->relu0, epi->relu1, epi->relu2, epi->relu_final, nullptr, epi->range, nullptr};
        if (epi->interp) {   // the residual's rows are interpolated from the [interp_rows, ldr] matrix `residual`
            if (!epi->residual) return P2W_ENULL;
            if (flags & P2W_GEMM_RESIDUAL_H) return P2W_EUNSUPPORTED;
            if (reinterpret_cast<uintptr_t>(epi->interp) & 15u) return P2W_EALIGN;
            if (epi->interp_rows <= 0 || (size_t)epi->interp_rows * (size_t)epi->ldr >= ((size_t)1 << 31)) return P2W_EINVAL;   // (32-bit element offsets)
            ep.imeta = static_cast<const int4*>(epi->interp);
        }
        if ((flags & P2W_GEMM_RESIDUAL_H) && epi->residual) {   // the residual is an H tensor of this precision, ldr its row pitch
            if ((epi->ldr & 7) || (reinterpret_cast<uintptr_t>(epi->residual) & 15u)) return P2W_EALIGN;
            ep.res_h = reinterpret_cast<const _Float16*>(epi->residual);
            ep.residual = nullptr;
        }
    }
    const _Float16* Ah = static_cast<const _Float16*>(A_h);
    const _Float16* Wp = static_cast<const _Float16*>(Wh);
    if (prec == P2W_PREC_F16X3)
        return launch_gemm_h<0>(Ah, ldh_a, Wp, wscale, M, N, K, ep, out_f32, ldo, static_cast<_Float16*>(out_h), ldh_o, flags,
                                p2w_s(stream), nullptr, nullptr, 0, static_cast<float*>(ws), ws_bytes);
    return p2w_gemm_h1_impl(prec, Ah, ldh_a, Wp, wscale, M, N, K, ep, out_f32, ldo, static_cast<_Float16*>(out_h), ldh_o, flags,
                            p2w_s(stream), nullptr, nullptr, 0, static_cast<float*>(ws), ws_bytes);
}
extern "C" int32_t p2w_gemm_h2(int32_t prec, const void* A_h, int32_t ldh_a, const void* Wh, float wscale, int32_t M, int32_t N,
                               int32_t K, const p2w_epilogue* epi, float* out_f32, int32_t ldo, void* out_h, int32_t ldh_o,
                               int32_t flags, p2w_stream_t stream) {
    return gemm_h2_checked(prec, A_h, ldh_a, Wh, wscale, M, N, K, epi, out_f32, ldo, out_h, ldh_o, nullptr, 0, flags, stream);
}
// workspace of the split-K tail: one 128 x 128 fp32 piece per workgroup of the tail launch (at most two per CU)
extern "C" size_t p2w_gemm_h2_sk_ws_bytes(void) { return (size_t)2 * p2w_cu_count() * 128 * 128 * sizeof(float); }
extern "C" int32_t p2w_gemm_h2_sk(int32_t prec, const void* A_h, int32_t ldh_a, const void* Wh, float wscale, int32_t M, int32_t N,
                                  int32_t K, const p2w_epilogue* epi, float* out_f32, int32_t ldo, void* out_h, int32_t ldh_o,
                                  void* ws, size_t ws_bytes, int32_t flags, p2w_stream_t stream) {
    if (!ws && ws_bytes) return P2W_ENULL;
    return gemm_h2_checked(prec, A_h, ldh_a, Wh, wscale, M, N, K, epi, out_f32, ldo, out_h, ldh_o, ws, ws_bytes, flags, stream);
}

// conv1 + BN + ReLU + conv2 with one output channel (model.py:241-243) without the [M, N] intermediate: the GEMM's epilogue
// leaves, per row and 64-column slice, the slice's share of dot(row, dot_w) in ws; this pass adds the slices in fixed order.
__global__ __launch_bounds__(256) void rowdot_finish_kernel(const float* __restrict__ part, int ldpart, int nslots, float b, int M,
                                                            float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float acc = 0.f;
    for (int s = 0; s < nslots; ++s) acc = acc + part[(size_t)s * ldpart + i];
    out[i] = acc + b;
}
static inline int rowdot_ldpart(int M) { return (M + 255) / 256 * 256; }
extern "C" size_t p2w_gemm_h2_rowdot_ws_bytes(int32_t M, int32_t N) {
    if (M < 0 || N <= 0) return 0;
    return (size_t)((N + 63) / 64) * rowdot_ldpart(M) * sizeof(float);
}
extern "C" int32_t p2w_gemm_h2_rowdot(int32_t prec, const void* A_h, int32_t ldh_a, const void* Wh, float wscale, int32_t M, int32_t N,
                                      int32_t K, const p2w_epilogue* epi, const float* dot_w, float dot_b, float* out, void* ws,
                                      size_t ws_bytes, int32_t flags, p2w_stream_t stream) {
    if (prec < P2W_PREC_F16X3 || prec > P2W_PREC_BF16) return P2W_EINVAL;
    if (M == 0) return P2W_OK;
    P2W_CHECK_PTR(A_h); P2W_CHECK_PTR(Wh); P2W_CHECK_PTR(dot_w); P2W_CHECK_PTR(out); P2W_CHECK_PTR(ws);
    P2W_CHECK_ALIGN16(A_h); P2W_CHECK_ALIGN16(Wh); P2W_CHECK_ALIGN16(ws);
    if (M < 0 || N <= 0 || K <= 0 || ldh_a < K || !(wscale > 0.f)) return P2W_EINVAL;
    if ((flags & P2W_GEMM_TILE_128) && (flags & P2W_GEMM_TILE_256)) return P2W_EINVAL;
    if (ws_bytes < p2w_gemm_h2_rowdot_ws_bytes(M, N)) return P2W_EWORKSPACE;
    EpiArgs ep = {};
    if (epi) {
        if ((epi->sc0 && !epi->sh0) || (epi->sc1 && !epi->sh1)) return P2W_ENULL;
        if (epi->residual || epi->interp) return P2W_EUNSUPPORTED;
        ep = {epi->bias, epi->sc0, epi->sh0, epi->sc1, epi->sh1, nullptr, 0, epi->relu0, epi->relu1, epi->relu2, epi->relu_final, nullptr, epi->range, nullptr};
    }
    const _Float16* Ah = static_cast<const _Float16*>(A_h);
    const _Float16* Wp = static_cast<const _Float16*>(Wh);
    float* part = static_cast<float*>(ws);
    const int ldpart = rowdot_ldpart(M);
    const int32_t rc = prec == P2W_PREC_F16X3
        ? launch_gemm_h<0>(Ah, ldh_a, Wp, wscale, M, N, K, ep, nullptr, 0, nullptr, 0, flags, p2w_s(stream), dot_w, part, ldpart)
        : p2w_gemm_h1_impl(prec, Ah, ldh_a, Wp, wscale, M, N, K, ep, nullptr, 0, nullptr, 0, flags, p2w_s(stream), dot_w, part, ldpart);
    if (rc != P2W_OK) return rc;
    rowdot_finish_kernel<<<p2w_cdiv(M, 256), 256, 0, p2w_s(stream)>>>(part, ldpart, (N + 63) / 64, dot_b, M, out);
    return P2W_LAUNCH_STATUS();
}

// Rows of the fused PointNetConv's GEMM, 32 per MFMA tile, G per target (G = 32: a tile per target; G = 8: four targets per
// tile).  Target of slot group gi: list[gi] (gi < *n_list_dev) or gi itself.  Writes per row the source's P row offset and
// the normalised offset g = (rel / (dmax + 1e-8), refl_j) (pointnet.py:119-129), per group the descriptor
// (target << 6) | neighbour count, -1 for a group without a target.
__global__ __launch_bounds__(256) void sa_edge_meta_kernel(const float4* __restrict__ xyzr, const int* __restrict__ idx,
                                                           const int* __restrict__ batch_dst, const float* __restrict__ sf,
                                                           const int* __restrict__ nbr, const int* __restrict__ deg, int kw,
                                                           int M, int n_src, int ldp4, int G, const int* __restrict__ list,
                                                           const int* __restrict__ n_list_dev, const int* __restrict__ src_row,
                                                           int* __restrict__ meta_j, float4* __restrict__ meta_g, int* __restrict__ desc,
                                                           float4* __restrict__ zero_row) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (zero_row && g < ldp4) zero_row[g] = make_float4(0.f, 0.f, 0.f, 0.f);   // P's row n_src: what empty neighbour slots gather (ldp4 <= 128 < 256)
    const int n_groups = list ? *n_list_dev : M;                 // slot groups that have a target
    const int gpt = 32 / G;
    const long n_tiles = ((long)n_groups + gpt - 1) / gpt;
    if (g >= n_tiles * 32) return;                               // whole 64-lane waves leave together (32 | 64)
    const int gi = (int)(g / G), slot = (int)(g % G);
    const int tgt = gi < n_groups ? (list ? list[gi] : gi) : -1;
    int j = 0, d = 0;
    float rx = 0.f, ry = 0.f, rz = 0.f, rf = 0.f, nrm = 0.f;
    bool valid = false;
    if (tgt >= 0) {
        d = min(deg[tgt], kw);
        const int self = idx[tgt];
        const float s = sf[batch_dst[tgt]];
        const float4 pi = xyzr[self];
        valid = slot < d;
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
    for (int off = G >> 1; off >= 1; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));  // G lanes = one target
    const float den = dmax + 1e-8f;
    // offset of the source's P row in float4 units; empty slot: P's all-zero row + a zero offset give a zero A row (the epilogue masks it)
    // (src_row: the P row of source point j when P's rows are not in the points' own order - p2w_sa_conv_h_rows)
    meta_j[g] = (valid ? (src_row ? src_row[j] : j) : n_src) * ldp4;
    meta_g[g] = make_float4(rx / den, ry / den, rz / den, rf);
    if (slot == 0) desc[gi] = tgt >= 0 ? ((tgt << 6) | d) : -1;
}

// P2W_SA_PACK8: stable partition of the targets by neighbour count (<= 8: four to a tile) - block counts, one-block scan, scatter
__global__ __launch_bounds__(SA_PART_BLOCK) void sa_part_count_kernel(const int* __restrict__ deg, int kw, int M, int* __restrict__ blk_small) {
    const int t = blockIdx.x * SA_PART_BLOCK + threadIdx.x;
    const bool small = t < M && min(deg[t], kw) <= 8;
    const int c = __syncthreads_count(small);
    if (threadIdx.x == 0) blk_small[blockIdx.x] = c;
}
__global__ __launch_bounds__(1024) void sa_part_scan_kernel(int* __restrict__ blk_small, int nblk, int M, int* __restrict__ counts) {
    __shared__ int part[1024];
    const int per = (nblk + 1023) / 1024, b0 = threadIdx.x * per, b1 = min(b0 + per, nblk);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += blk_small[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan of the per-thread sums
        const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;            // exclusive prefix of this thread's blocks
    for (int b = b0; b < b1; ++b) { const int c = blk_small[b]; blk_small[b] = run; run += c; }
    if (threadIdx.x == 1023) {
        const int ns = part[1023], nl = M - ns;
        counts[0] = ns; counts[1] = nl; counts[2] = (ns + 3) / 4; counts[3] = nl;
    }
}
__global__ __launch_bounds__(SA_PART_BLOCK) void sa_part_scatter_kernel(const int* __restrict__ deg, int kw, int M, const int* __restrict__ blk_small,
                                                                     int* __restrict__ list_small, int* __restrict__ list_large) {
    __shared__ int wsum[SA_PART_BLOCK / 64];
    const int t = blockIdx.x * SA_PART_BLOCK + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in = t < M, small = in && min(deg[t], kw) <= 8;
    const unsigned long long bal = __ballot(small);
    const int r_in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int before = 0;                                   // small targets of this block in earlier waves
    for (int w = 0; w < wave; ++w) before += wsum[w];
    const int off_s = blk_small[blockIdx.x];          // small targets in earlier blocks
    const int rank_s = before + r_in_wave;            // this target's rank among the block's small ones (if small)
    if (small) list_small[off_s + rank_s] = t;
    else if (in) list_large[(blockIdx.x * SA_PART_BLOCK - off_s) + (threadIdx.x - rank_s)] = t;   // stable: earlier large targets
}

extern "C" size_t p2w_sa_conv_h_ws_bytes(int32_t M, int32_t flags) { return sa_conv_ws_bytes(M < 0 ? 0 : M, flags); }

extern "C" int32_t p2w_sa_conv_h_rows(int32_t prec, const float* P, int32_t ldp, int32_t n_src, const float* xyzr_src, const int32_t* idx,
                                 const int32_t* batch_dst, const float* sf, const int32_t* nbr, const int32_t* deg,
                                 int32_t kw, int32_t M, const float* w1r4, const void* W2h, float wscale, int32_t C1,
                                 int32_t C2, const float* b2, const float* bn_s, const float* bn_t, float* out,
                                 int32_t ldo, void* out_h, int32_t ldh, void* ws, size_t ws_bytes, int32_t flags,
                                 const int32_t* src_row, uint32_t* range, p2w_stream_t stream) {
    if (prec < P2W_PREC_F16X3 || prec > P2W_PREC_BF16) return P2W_EINVAL;
    if (M == 0) return P2W_OK;
    P2W_CHECK_PTR(P); P2W_CHECK_PTR(xyzr_src); P2W_CHECK_PTR(idx); P2W_CHECK_PTR(batch_dst); P2W_CHECK_PTR(sf);
    P2W_CHECK_PTR(nbr); P2W_CHECK_PTR(deg); P2W_CHECK_PTR(w1r4); P2W_CHECK_PTR(W2h); P2W_CHECK_PTR(b2);
    P2W_CHECK_PTR(bn_s); P2W_CHECK_PTR(bn_t);
    if (!out && !out_h) return P2W_ENULL;
    P2W_CHECK_ALIGN16(P); P2W_CHECK_ALIGN16(xyzr_src); P2W_CHECK_ALIGN16(w1r4); P2W_CHECK_ALIGN16(W2h);
    if (M < 0 || n_src < 0 || kw <= 0 || kw > 32 || C1 <= 0 || C2 <= 0 || (C1 & 3) || (ldp & 3) || ldp < C1 || (out && ldo < C2) ||
        (out_h && (ldh < C2 || (ldh & 7))) || !(wscale > 0.f))
        return P2W_EINVAL;
    const _Float16* W2 = static_cast<const _Float16*>(W2h);
    if (prec == P2W_PREC_F16X3)
        return launch_sa_conv_h<0>(P, ldp, n_src, xyzr_src, idx, batch_dst, sf, nbr, deg, kw, M, w1r4, W2, wscale, C1, C2, b2, bn_s, bn_t,
                                   out, ldo, static_cast<_Float16*>(out_h), ldh, ws, ws_bytes, flags, p2w_s(stream), src_row, range);
    return p2w_sa_conv_h1_impl(prec, P, ldp, n_src, xyzr_src, idx, batch_dst, sf, nbr, deg, kw, M, w1r4, W2, wscale, C1, C2, b2, bn_s,
                               bn_t, out, ldo, static_cast<_Float16*>(out_h), ldh, ws, ws_bytes, flags, p2w_s(stream), src_row, range);
}

extern "C" int32_t p2w_sa_conv_h(int32_t prec, const float* P, int32_t ldp, int32_t n_src, const float* xyzr_src, const int32_t* idx,
                                 const int32_t* batch_dst, const float* sf, const int32_t* nbr, const int32_t* deg,
                                 int32_t kw, int32_t M, const float* w1r4, const void* W2h, float wscale, int32_t C1,
                                 int32_t C2, const float* b2, const float* bn_s, const float* bn_t, float* out,
                                 int32_t ldo, void* out_h, int32_t ldh, void* ws, size_t ws_bytes, int32_t flags,
                                 p2w_stream_t stream) {
    return p2w_sa_conv_h_rows(prec, P, ldp, n_src, xyzr_src, idx, batch_dst, sf, nbr, deg, kw, M, w1r4, W2h, wscale, C1, C2, b2, bn_s, bn_t,
                              out, ldo, out_h, ldh, ws, ws_bytes, flags, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------
// small HBM-bound kernels
// ------------------------------------------------------------------------------------------------
// 4 consecutive output columns of one row -> fp32 row (pitch ldo) and/or H2 row (pitch ldh)
template <int PREC>
__device__ __forceinline__ void store4(const OutArgs& o, size_t row, int c, const float (&v)[4]) {
    if (o.f32 && c < o.ldo) *reinterpret_cast<float4*>(&o.f32[row * o.ldo + c]) = make_float4(v[0], v[1], v[2], v[3]);
    if (o.h2 && c < o.hcols) h_store4<PREC>(o.h2, o.ldh, row, c, v);
}
// launch a kernel template instantiated for the three H precisions
#define P2W_LAUNCH_PREC(prec, KERNEL, grid, block, stream, ...)                                          \
    do {                                                                                                 \
        if ((prec) == P2W_PREC_F16) KERNEL<1><<<grid, block, 0, stream>>>(__VA_ARGS__);                  \
        else if ((prec) == P2W_PREC_BF16) KERNEL<2><<<grid, block, 0, stream>>>(__VA_ARGS__);            \
        else KERNEL<0><<<grid, block, 0, stream>>>(__VA_ARGS__);                                         \
    } while (0)

// INDEXED: the records come in another (e.g. cell-sorted) order and carry their own row in .w: the fp32 output goes to that row,
// the H output to the record's position (p2w_stem_h2_indexed)
template <int PREC, bool INDEXED = false>
__global__ __launch_bounds__(256) void stem_kernel(const float4* __restrict__ xyzr, int n, const float* __restrict__ w,
                                                   const float* __restrict__ b, int C, int q4, OutArgs o, unsigned* __restrict__ range) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per (row, 4 channels incl. zero padding)
    float amax = 0.f;
    if (g < (long)n * q4) {
        const int row = (int)(g / q4), c0 = (int)(g % q4) * 4;
        const float4 p = xyzr[row];
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            v[e] = (c < C) ? fmaxf(fmaf(p.z, w[c * 3 + 2], fmaf(p.y, w[c * 3 + 1], fmaf(p.x, w[c * 3 + 0], b[c]))), 0.f) : 0.f;
        }
        amax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));   // (ReLU outputs: >= 0)
        if constexpr (INDEXED) {
            if (o.f32 && c0 < o.ldo) *reinterpret_cast<float4*>(&o.f32[(size_t)__float_as_int(p.w) * o.ldo + c0]) = make_float4(v[0], v[1], v[2], v[3]);
            if (o.h2 && c0 < o.hcols) h_store4<PREC>(o.h2, o.ldh, (size_t)row, c0, v);
        } else {
            store4<PREC>(o, (size_t)row, c0, v);
        }
    }
    if (range) range_commit_max(range, amax, threadIdx.x & 63);
}


static int32_t stem_launch(int32_t prec, const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out,
                           void* out_h2, int32_t ldh, p2w_stream_t stream, bool indexed = false, uint32_t* range = nullptr) {
    if (prec < P2W_PREC_F16X3 || prec > P2W_PREC_BF16) return P2W_EINVAL;
    if (n == 0) return P2W_OK;
    P2W_CHECK_PTR(xyzr); P2W_CHECK_PTR(w); P2W_CHECK_PTR(b); P2W_CHECK_ALIGN16(xyzr);
    if (!out && !out_h2) return P2W_ENULL;
    if (n < 0 || C <= 0 || (C & 3) || (out_h2 && (ldh < C || (ldh & 7)))) return P2W_EINVAL;
    const int ka = prec == P2W_PREC_F16X3 ? 32 : 64;
    const int hcols = out_h2 ? (ldh < (C + ka - 1) / ka * ka ? ldh : (C + ka - 1) / ka * ka) : 0;   // C channels + zero pad to the K-slab boundary
    OutArgs o = {out, C, static_cast<_Float16*>(out_h2), out_h2 ? ldh : 0, hcols, nullptr, nullptr, 0};
    const int q4 = (out_h2 ? hcols : C) >> 2;
    if (indexed) {
        const int grid = p2w_cdiv((long)n * q4, 256);
        const float4* x4 = reinterpret_cast<const float4*>(xyzr);
        if (prec == P2W_PREC_F16) stem_kernel<1, true><<<grid, 256, 0, p2w_s(stream)>>>(x4, n, w, b, C, q4, o, range);
        else if (prec == P2W_PREC_BF16) stem_kernel<2, true><<<grid, 256, 0, p2w_s(stream)>>>(x4, n, w, b, C, q4, o, range);
        else stem_kernel<0, true><<<grid, 256, 0, p2w_s(stream)>>>(x4, n, w, b, C, q4, o, range);
        return P2W_LAUNCH_STATUS();
    }
    P2W_LAUNCH_PREC(prec, stem_kernel, p2w_cdiv((long)n * q4, 256), 256, p2w_s(stream), reinterpret_cast<const float4*>(xyzr), n, w, b,
                    C, q4, o, range);
    return P2W_LAUNCH_STATUS();
}
extern "C" int32_t p2w_stem(const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out,
                            p2w_stream_t stream) {
    P2W_CHECK_PTR(out);
    return stem_launch(P2W_PREC_F16X3, xyzr, n, w, b, C, out, nullptr, 0, stream);
}
extern "C" int32_t p2w_stem_h2(int32_t prec, const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out,
                               void* out_h2, int32_t ldh, uint32_t* range, p2w_stream_t stream) {
    return stem_launch(prec, xyzr, n, w, b, C, out, out_h2, ldh, stream, false, range);
}
extern "C" int32_t p2w_stem_h2_indexed(int32_t prec, const float* xyzr, int32_t n, const float* w, const float* b, int32_t C, float* out,
                                       void* out_h2, int32_t ldh, uint32_t* range, p2w_stream_t stream) {
    return stem_launch(prec, xyzr, n, w, b, C, out, out_h2, ldh, stream, true, range);
}

// One wave per output row: the row's neighbours, their inverse-square-distance weights and the denominator are
// wave-uniform (scalar loads, computed once per row instead of once per 4-column chunk); the lanes then sweep the row
// 256 columns at a time with coalesced 16-byte loads of the coarse features.  The arithmetic per element is the same
// as torch-scatter's: products and sums in neighbour order from 0, then a literal division.
#ifndef P2W_IC_ROWS
#define P2W_IC_ROWS 1   // swept 1..16 on the bench forward: 447 / 473 / 500 / 582 / 682 us for 1 / 2 / 4 / 8 / 16
#endif
constexpr int IC_ROWS = P2W_IC_ROWS;   // rows per wave (consecutive)
#ifndef P2W_IC_PAIRS
#define P2W_IC_PAIRS 1   // 0: one chunk per step (A/B: interpolation 0.355 -> 0.316 ms per bench step)
#endif
template <int PREC>
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
#if P2W_IC_PAIRS
        // the common shape (k <= 2 neighbours, interpolated columns only): two 256-column chunks per step with all their loads
        // issued before any arithmetic - the kernel waits for memory (VALU active 5 % of its wave cycles), so what counts is
        // bytes in flight per lane
        int c_first = 4 * lane;
        if (d >= 1 && d <= 2 && Fs == 0) {
            for (; c_first + 256 < Fc && c_first + 256 < 4 * q4; c_first += 512) {
                float4 xa[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    xa[u][0] = *reinterpret_cast<const float4*>(&xc[(size_t)js[0] * Fc + c_first + 256 * u]);
                    xa[u][1] = d > 1 ? *reinterpret_cast<const float4*>(&xc[(size_t)js[1] * Fc + c_first + 256 * u]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
                    num.x = num.x + xa[u][0].x * ws[0]; num.y = num.y + xa[u][0].y * ws[0]; num.z = num.z + xa[u][0].z * ws[0]; num.w = num.w + xa[u][0].w * ws[0];
                    if (d > 1) { num.x = num.x + xa[u][1].x * ws[1]; num.y = num.y + xa[u][1].y * ws[1]; num.z = num.z + xa[u][1].z * ws[1]; num.w = num.w + xa[u][1].w * ws[1]; }
                    const float v2[4] = {num.x / den, num.y / den, num.z / den, num.w / den};
                    store4<PREC>(o, (size_t)q, c_first + 256 * u, v2);
                }
            }
        }
        for (int c = c_first; c < 4 * q4; c += 256) {
#else
        for (int c = 4 * lane; c < 4 * q4; c += 256) {
#endif
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
            store4<PREC>(o, (size_t)q, c, v);
        }
    }
}

static int32_t interp_launch(int32_t prec, const float* xc, int32_t Fc, const float* xyzr_c, const float* xyzr_f, const int32_t* nbr,
                             const int32_t* deg, int32_t kw, const float* skip, int32_t Fs, int32_t m, float* out, int32_t ldo,
                             void* out_h2, int32_t ldh, p2w_stream_t stream) {
    if (prec < P2W_PREC_F16X3 || prec > P2W_PREC_BF16) return P2W_EINVAL;
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
    const int ka = prec == P2W_PREC_F16X3 ? 32 : 64;
    const int hfull = (Fc + Fs + ka - 1) / ka * ka;
    const int hcols = out_h2 ? (ldh < hfull ? ldh : hfull) : 0;   // ldh is the pitch: the row may continue with columns another producer owns
    const int width = (out ? ldo : 0) > hcols ? ldo : hcols;
    OutArgs o = {out, out ? ldo : 0, static_cast<_Float16*>(out_h2), out_h2 ? ldh : 0, hcols, nullptr, nullptr, 0};
    P2W_LAUNCH_PREC(prec, interp_concat_kernel, p2w_cdiv(m, 4 * IC_ROWS), 256, p2w_s(stream),
        xc, Fc, reinterpret_cast<const float4*>(xyzr_c), reinterpret_cast<const float4*>(xyzr_f), nbr, deg, kw, skip, Fs, m,
        width >> 2, o);
    return P2W_LAUNCH_STATUS();
}
extern "C" int32_t p2w_interp_concat(const float* xc, int32_t Fc, const float* xyzr_c, const float* xyzr_f, const int32_t* nbr,
                                     const int32_t* deg, int32_t kw, const float* skip, int32_t Fs, int32_t m, float* out,
                                     int32_t ldo, p2w_stream_t stream) {
    P2W_CHECK_PTR(out);
    return interp_launch(P2W_PREC_F16X3, xc, Fc, xyzr_c, xyzr_f, nbr, deg, kw, skip, Fs, m, out, ldo, nullptr, 0, stream);
}
extern "C" int32_t p2w_interp_concat_h2(int32_t prec, const float* xc, int32_t Fc, const float* xyzr_c, const float* xyzr_f,
                                        const int32_t* nbr, const int32_t* deg, int32_t kw, const float* skip, int32_t Fs,
                                        int32_t m, void* out_h2, int32_t ldh, p2w_stream_t stream) {
    P2W_CHECK_PTR(out_h2);
    return interp_launch(prec, xc, Fc, xyzr_c, xyzr_f, nbr, deg, kw, skip, Fs, m, nullptr, 0, out_h2, ldh, stream);
}

// knn_interpolate's weights as one record per fine row (k <= 2): {n0, n1, a0, a1} with a_s = w_s / (w_0 + w_1), w_s as in
// interp_concat_kernel (1 / max(d2, 1e-16), d2 = (dx dx + dy dy) + dz dz in fp32).  A row with one neighbour: {n0, n0, 1, 0}.
__global__ __launch_bounds__(256) void interp_weights_kernel(const float4* __restrict__ xyzr_c, const float4* __restrict__ xyzr_f,
                                                             const int* __restrict__ nbr, const int* __restrict__ deg, int kw, int m,
                                                             int4* __restrict__ rec) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    const int d = min(deg[q], kw);
    const float4 pf = xyzr_f[q];
    int js[2] = {0, 0};
    float ws[2] = {0.f, 0.f};
    for (int s = 0; s < d && s < 2; ++s) {
        const int j = nbr[(size_t)q * kw + s];
        const float4 pc = xyzr_c[j];
        const float dx = pc.x - pf.x, dy = pc.y - pf.y, dz = pc.z - pf.z;
        const float d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
        js[s] = j; ws[s] = 1.0f / fmaxf(d2, 1e-16f);
    }
    if (d < 2) js[1] = js[0];
    const float den = ws[0] + ws[1];
    const float a0 = d > 0 ? ws[0] / den : 0.f, a1 = d > 1 ? ws[1] / den : 0.f;
    rec[q] = make_int4(js[0], js[1], __float_as_int(a0), __float_as_int(a1));
}
extern "C" int32_t p2w_interp_weights(const float* xyzr_c, const float* xyzr_f, const int32_t* nbr, const int32_t* deg, int32_t kw,
                                      int32_t m, void* records, p2w_stream_t stream) {
    if (m == 0) return P2W_OK;
    P2W_CHECK_PTR(xyzr_c); P2W_CHECK_PTR(xyzr_f); P2W_CHECK_PTR(nbr); P2W_CHECK_PTR(deg); P2W_CHECK_PTR(records);
    P2W_CHECK_ALIGN16(xyzr_c); P2W_CHECK_ALIGN16(xyzr_f); P2W_CHECK_ALIGN16(records);
    if (m < 0 || kw <= 0) return P2W_EINVAL;
    if (kw > 2) return P2W_EUNSUPPORTED;
    interp_weights_kernel<<<p2w_cdiv(m, 256), 256, 0, p2w_s(stream)>>>(reinterpret_cast<const float4*>(xyzr_c), reinterpret_cast<const float4*>(xyzr_f),
                                                                         nbr, deg, kw, m, static_cast<int4*>(records));
    return P2W_LAUNCH_STATUS();
}

template <int PREC>
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
    store4<PREC>(o, (size_t)q, c, v);
}

static int32_t concat_launch(int32_t prec, const float* x, int32_t F, const float* xyzr, int32_t m, float* out, int32_t ldo, void* out_h2,
                             int32_t ldh, p2w_stream_t stream) {
    if (prec < P2W_PREC_F16X3 || prec > P2W_PREC_BF16) return P2W_EINVAL;
    if (m == 0) return P2W_OK;
    P2W_CHECK_PTR(x); P2W_CHECK_PTR(xyzr);
    if (!out && !out_h2) return P2W_ENULL;
    P2W_CHECK_ALIGN16(x); P2W_CHECK_ALIGN16(xyzr);
    if (m < 0 || F <= 0 || (F & 3)) return P2W_EINVAL;
    if (out && ((ldo & 3) || ldo < F + 4)) return P2W_EINVAL;
    if (out_h2 && ((ldh & 7) || ldh < F + 4)) return P2W_EINVAL;
    const int width = (out ? ldo : 0) > (out_h2 ? ldh : 0) ? ldo : ldh;
    OutArgs o = {out, out ? ldo : 0, static_cast<_Float16*>(out_h2), out_h2 ? ldh : 0, out_h2 ? ldh : 0, nullptr, nullptr, 0};
    P2W_LAUNCH_PREC(prec, concat_xyz_kernel, p2w_cdiv((long)m * (width >> 2), 256), 256, p2w_s(stream),
        x, F, reinterpret_cast<const float4*>(xyzr), m, width >> 2, o);
    return P2W_LAUNCH_STATUS();
}
extern "C" int32_t p2w_concat_xyz(const float* x, int32_t F, const float* xyzr, int32_t m, float* out, int32_t ldo,
                                  p2w_stream_t stream) {
    P2W_CHECK_PTR(out);
    return concat_launch(P2W_PREC_F16X3, x, F, xyzr, m, out, ldo, nullptr, 0, stream);
}
extern "C" int32_t p2w_concat_xyz_h2(int32_t prec, const float* x, int32_t F, const float* xyzr, int32_t m, void* out_h2, int32_t ldh,
                                     p2w_stream_t stream) {
    P2W_CHECK_PTR(out_h2);
    return concat_launch(prec, x, F, xyzr, m, nullptr, 0, out_h2, ldh, stream);
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
extern "C" int32_t p2w_version(void) { return 600; }

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
