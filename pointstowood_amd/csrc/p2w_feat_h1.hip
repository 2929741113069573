// Single-plane precisions of the H family (P2W_PREC_F16, P2W_PREC_BF16): one v_mfma_f32_32x32x16_{f16,bf16} per product,
// fp32 accumulate - the arithmetic the reference's own GPU path uses (torch.cuda.amp.autocast,
// pointstowood/src/predicter.py:197).  The kernels are the templates of p2w_hgemm.h; this translation unit only
// instantiates them, so that their register allocation cannot perturb the f16x3 (parity) kernels in p2w_feat.hip.
#include "p2w_hgemm.h"

int32_t p2w_gemm_h1_impl(int32_t prec, const _Float16* Ah, int32_t ldh_a, const _Float16* Wp, float wscale, int32_t M, int32_t N,
                         int32_t K, const EpiArgs& ep, float* out_f32, int32_t ldo, _Float16* out_h2, int32_t ldh_o,
                         int32_t flags, hipStream_t stream, const float* dotw, float* part, int32_t ldpart, float* skws,
                         size_t skws_bytes) {
    if (prec == P2W_PREC_F16)
        return launch_gemm_h<1>(Ah, ldh_a, Wp, wscale, M, N, K, ep, out_f32, ldo, out_h2, ldh_o, flags, stream, dotw, part, ldpart, skws, skws_bytes);
    return launch_gemm_h<2>(Ah, ldh_a, Wp, wscale, M, N, K, ep, out_f32, ldo, out_h2, ldh_o, flags, stream, dotw, part, ldpart, skws, skws_bytes);
}

int32_t p2w_sa_conv_h1_impl(int32_t prec, const float* P, int32_t ldp, int32_t n_src, const float* xyzr_src, const int32_t* idx,
                            const int32_t* batch_dst, const float* sf, const int32_t* nbr, const int32_t* deg, int32_t kw,
                            int32_t M, const float* w1r4, const _Float16* W2h, float wscale, int32_t C1, int32_t C2,
                            const float* b2, const float* bn_s, const float* bn_t, float* out, int32_t ldo, _Float16* out_h2,
                            int32_t ldh, void* ws, size_t ws_bytes, int32_t flags, hipStream_t stream, const int32_t* src_row,
                            unsigned* range) {
    if (prec == P2W_PREC_F16)
        return launch_sa_conv_h<1>(P, ldp, n_src, xyzr_src, idx, batch_dst, sf, nbr, deg, kw, M, w1r4, W2h, wscale, C1, C2, b2, bn_s, bn_t,
                                   out, ldo, out_h2, ldh, ws, ws_bytes, flags, stream, src_row, range);
    return launch_sa_conv_h<2>(P, ldp, n_src, xyzr_src, idx, batch_dst, sf, nbr, deg, kw, M, w1r4, W2h, wscale, C1, C2, b2, bn_s, bn_t,
                               out, ldo, out_h2, ldh, ws, ws_bytes, flags, stream, src_row, range);
}
