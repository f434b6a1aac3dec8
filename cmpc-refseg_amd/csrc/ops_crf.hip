// Fully connected CRF post-processing of the evaluation script (/root/reference/test.py:309-322: pydensecrf's DenseCRF2D with a
// Gaussian (sxy 3, compat 3) and a bilateral (sxy 20, srgb 3, compat 10) Potts term, 5 mean-field iterations, argmax).
//
// pydensecrf (a third-party dependency of the reference, not part of its tree) filters through a permutohedral lattice, an
// approximation of the Gaussian kernels.  On the GPU the kernels are evaluated directly: one thread per pixel walks the window of
// radius 4 sigma (e^-8 of the kernel's peak beyond it; the whole image when it is smaller), the separable spatial factor comes from
// two LDS tables, the colour factor is one v_exp_f32 per pair -- 2.7e9 pairs per bilateral pass at 320 x 320, about a millisecond.
// Same mean-field recursion and symmetric normalisation as the library (oracle/dense_crf_numpy.py restates them); the masks are
// therefore parity-unpinned against pydensecrf itself.
#include "cmpc_common.h"
#include "../../include/cmpc.h"

namespace {

__global__ __launch_bounds__(256) void crf_init_kernel(const float* __restrict__ sigm, const unsigned char* __restrict__ rgb, float2* __restrict__ U,
                                                       float2* __restrict__ Q, uint32_t* __restrict__ pix, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float p = sigm[i];
    const float u0 = -logf(fmaxf(1.0f - p, 1e-30f)), u1 = -logf(fmaxf(p, 1e-30f));      // test.py:312-315
    U[i] = make_float2(u0, u1);
    const float m = fminf(u0, u1), e0 = expf(m - u0), e1 = expf(m - u1), r = 1.0f / (e0 + e1);
    Q[i] = make_float2(e0 * r, e1 * r);
    pix[i] = (uint32_t)rgb[3 * i] | ((uint32_t)rgb[3 * i + 1] << 8) | ((uint32_t)rgb[3 * i + 2] << 16);
}

// out_i = sum over the window of k_ij * v_j,  k_ij = exp(-a_xy (dx^2 + dy^2)) * (BILATERAL ? exp(-a_c |I_i - I_j|^2) : 1), j = i included.
// v == nullptr: v_j = 1 and the result is stored as the symmetric normaliser 1 / sqrt(sum + 1e-20) in nrm_out.
template <bool BILATERAL>
__global__ __launch_bounds__(256) void crf_filter_kernel(const float2* __restrict__ v, const uint32_t* __restrict__ pix, float2* __restrict__ out,
                                                         float* __restrict__ nrm_out, int H, int W, int radius, float a_xy, float a_c) {
    __shared__ float wtab[512];                            // exp(-a_xy d^2), d = 0..radius  (radius <= 511: checked by the launcher)
    for (int d = threadIdx.x; d <= radius; d += 256) wtab[d] = __expf(-a_xy * (float)(d * d));
    __syncthreads();
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const uint32_t c0 = pix[y * W + x];
    const float r0 = (float)(c0 & 255u), g0 = (float)((c0 >> 8) & 255u), b0 = (float)(c0 >> 16);
    const float cl2 = a_c * 1.4426950408889634f;          // exp(-a t) = exp2(-a log2(e) t)
    float s0 = 0.f, s1 = 0.f;
    const int y_lo = max(0, y - radius), y_hi = min(H - 1, y + radius), x_lo = max(0, x - radius), x_hi = min(W - 1, x + radius);
    for (int yy = y_lo; yy <= y_hi; ++yy) {
        const float wy = wtab[abs(yy - y)];
        const long row = (long)yy * W;
        float t0 = 0.f, t1 = 0.f;
        for (int xx = x_lo; xx <= x_hi; ++xx) {
            float k = wtab[abs(xx - x)];
            if (BILATERAL) {
                const uint32_t c = pix[row + xx];
                const float dr = (float)(c & 255u) - r0, dg = (float)((c >> 8) & 255u) - g0, db = (float)(c >> 16) - b0;
                k *= __builtin_amdgcn_exp2f(-cl2 * (dr * dr + dg * dg + db * db));
            }
            if (v) { const float2 q = v[row + xx]; t0 += k * q.x; t1 += k * q.y; }
            else t0 += k;
        }
        s0 += wy * t0; s1 += wy * t1;
    }
    const long i = (long)y * W + x;
    if (v) out[i] = make_float2(s0, s1);
    else nrm_out[i] = rsqrtf(s0 + 1e-20f);
}

// v_m = Q * nrm_m: the pre-normalised inputs of the two filters
__global__ __launch_bounds__(256) void crf_scale_kernel(const float2* __restrict__ Q, const float* __restrict__ ng, const float* __restrict__ nb,
                                                        float2* __restrict__ vg, float2* __restrict__ vb, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 q = Q[i];
    vg[i] = make_float2(q.x * ng[i], q.y * ng[i]);
    vb[i] = make_float2(q.x * nb[i], q.y * nb[i]);
}

// Q <- softmax(-U + w_g * nrm_g * F_g + w_b * nrm_b * F_b)   (Potts: -w on equal labels, subtracted from -U)
__global__ __launch_bounds__(256) void crf_update_kernel(const float2* __restrict__ U, const float2* __restrict__ fg, const float* __restrict__ ng,
                                                         const float2* __restrict__ fb, const float* __restrict__ nb, float wg, float wb,
                                                         float2* __restrict__ Q, float* __restrict__ q_out, unsigned char* __restrict__ mask, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 u = U[i], a = fg[i], b = fb[i];
    const float cg = wg * ng[i], cb = wb * nb[i];
    const float e0 = -u.x + cg * a.x + cb * b.x, e1 = -u.y + cg * a.y + cb * b.y;
    const float m = fmaxf(e0, e1), x0 = expf(e0 - m), x1 = expf(e1 - m), r = 1.0f / (x0 + x1);
    Q[i] = make_float2(x0 * r, x1 * r);
    if (q_out) { q_out[i] = x0 * r; q_out[n + i] = x1 * r; }
    if (mask) mask[i] = x1 > x0 ? 1 : 0;                   // np.argmax(Q, axis=0): the first maximum wins a tie
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int cmpc_dense_crf(const float* sigm, const unsigned char* rgb, int H, int W, float sxy_g, float compat_g, float sxy_b, float srgb,
                              float compat_b, int iters, float* q_out, unsigned char* mask_out, void* stream) {
    if (!sigm || !rgb || H <= 0 || W <= 0 || iters < 0 || sxy_g <= 0.f || sxy_b <= 0.f || srgb <= 0.f || (!q_out && !mask_out)) {
        cmpc_set_error("dense_crf: bad args"); return CMPC_EINVAL;
    }
    const int n = H * W;
    const int rg = min(max(H, W) - 1, (int)ceilf(4.0f * sxy_g)), rb = min(max(H, W) - 1, (int)ceilf(4.0f * sxy_b));
    if (rg > 511 || rb > 511) { cmpc_set_error("dense_crf: kernel radius %d > 511", max(rg, rb)); return CMPC_EINVAL; }
    // scratch: U, Q, vg, vb, fg, fb (float2 [n]) + ng, nb (float [n]) + pix (uint32 [n])
    char* ws = (char*)cmpc_ws((size_t)n * (6 * sizeof(float2) + 2 * sizeof(float) + sizeof(uint32_t)), ST);
    if (!ws) return CMPC_EHIP;
    float2* U = (float2*)ws; float2* Q = U + n; float2* vg = Q + n; float2* vb = vg + n; float2* fg = vb + n; float2* fb = fg + n;
    float* ng = (float*)(fb + n); float* nb = ng + n; uint32_t* pix = (uint32_t*)(nb + n);
    const dim3 g1((n + 255) / 256), g2((W + 31) / 32, (H + 7) / 8);
    const float ag = 0.5f / (sxy_g * sxy_g), ab = 0.5f / (sxy_b * sxy_b), ac = 0.5f / (srgb * srgb);
    hipLaunchKernelGGL(crf_init_kernel, g1, dim3(256), 0, ST, sigm, rgb, U, Q, pix, n);
    hipLaunchKernelGGL((crf_filter_kernel<false>), g2, dim3(256), 0, ST, (const float2*)nullptr, pix, (float2*)nullptr, ng, H, W, rg, ag, 0.f);
    hipLaunchKernelGGL((crf_filter_kernel<true>), g2, dim3(256), 0, ST, (const float2*)nullptr, pix, (float2*)nullptr, nb, H, W, rb, ab, ac);
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(crf_scale_kernel, g1, dim3(256), 0, ST, Q, ng, nb, vg, vb, n);
        hipLaunchKernelGGL((crf_filter_kernel<false>), g2, dim3(256), 0, ST, vg, pix, fg, (float*)nullptr, H, W, rg, ag, 0.f);
        hipLaunchKernelGGL((crf_filter_kernel<true>), g2, dim3(256), 0, ST, vb, pix, fb, (float*)nullptr, H, W, rb, ab, ac);
        const bool last = it + 1 == iters;
        hipLaunchKernelGGL(crf_update_kernel, g1, dim3(256), 0, ST, U, fg, ng, fb, nb, compat_g, compat_b, Q, last ? q_out : nullptr,
                           last ? mask_out : nullptr, n);
    }
    if (iters == 0) {                                       // Q = softmax(-U): written through the update kernel with zero pairwise weight
        hipLaunchKernelGGL(crf_update_kernel, g1, dim3(256), 0, ST, U, U, ng, U, nb, 0.f, 0.f, Q, q_out, mask_out, n);
    }
    return cmpc_check_launch("dense_crf");
}
