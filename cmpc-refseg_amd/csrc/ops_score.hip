// Score heads: 3x3 SAME convolution M->1, legacy tf.image.resize_bilinear, sigmoid, weighed
// logistic loss and the in-graph mIoU counters (reference CMPC_model.py:128-142,440-447,486-490;
// util/loss.py:6-16).  Everything here is fp32 (parity-critical) except the feature map read.
#include "cmpc_common.h"
#include "../../include/cmpc.h"

namespace {

constexpr int WPB = 4;
constexpr int MB = 2;

// Each lane owns the same 8 channels (per 512-block) for every pixel its wave visits, so the 9 x 8
// filter taps it needs live in registers for the whole grid-stride loop.
template <typename T, int NB>
__global__ __launch_bounds__(256) void score_conv_fwd_kernel(const T* __restrict__ feat, const float* __restrict__ Wk, const float* __restrict__ bias,
                                                            float* __restrict__ score, int B, int h, int w, int ld, int M) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int total = B * h * w;
    float wk[NB][9][8];
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int c = k * 512 + lane * 8 + e; wk[k][t][e] = (c < M) ? Wk[t * M + c] : 0.f; }
    for (int p = blockIdx.x * WPB + wv; p < total; p += gridDim.x * WPB) {
        const int b = p / (h * w), y = (p / w) % h, x = p % w;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float v[9][8];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    if (yy >= 0 && yy < h && xx >= 0 && xx < w) ld8<T>(feat + ((long)(b * h + yy) * w + xx) * ld + c0, v[t]);
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[t][e] = 0.f;
                    }
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc += v[t][e] * wk[k][t][e];
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) score[p] = acc + bias[0];
    }
}

// dfeat[b,y,x,:] (+)= sum_taps dscore[b, y-dy, x-dx] * W[tap,:]
template <typename T, int NB>
__global__ __launch_bounds__(256) void score_conv_bwd_data_kernel(const float* __restrict__ dscore, const float* __restrict__ Wk, T* __restrict__ dfeat,
                                                                 int accumulate, int B, int h, int w, int ld, int M) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int total = B * h * w;
    float wk[NB][9][8];
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int c = k * 512 + lane * 8 + e; wk[k][t][e] = (c < M) ? Wk[t * M + c] : 0.f; }
    for (int p = blockIdx.x * WPB + wv; p < total; p += gridDim.x * WPB) {
        const int b = p / (h * w), y = (p / w) % h, x = p % w;
        float ds[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1, yy = y - dy, xx = x - dx;
            ds[t] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? dscore[(b * h + yy) * w + xx] : 0.f;
        }
        T* d = dfeat + (long)p * ld;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float o[8];
                if (accumulate) ld8<T>(d + c0, o);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = 0.f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) v += ds[t] * wk[k][t][e];
                    o[e] = accumulate ? o[e] + v : v;
                }
                st8<T>(d + c0, o);
            }
        }
    }
}

// ---- row-based forms (ld <= 512): one workgroup per image row, each of its 4 waves walks a run of consecutive
// pixels.  The 72 filter taps of a lane are loaded once per run (the per-pixel kernels above reload 18 KB of taps
// for every 1-2 pixels: 83 us for a 13 MB map), and forward keeps a sliding window of three neighbour columns in
// registers (3 new 16-B loads per pixel instead of 9, the column after next already in flight).
template <typename T>
__global__ __launch_bounds__(256) void score_conv_fwd_row_kernel(const T* __restrict__ feat, const float* __restrict__ Wk, const float* __restrict__ bias,
                                                                float* __restrict__ score, int h, int w, int ld, int M) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x / h, y = blockIdx.x % h;
    const int per = (w + WPB - 1) / WPB, x0 = wv * per, x1 = min(w, x0 + per);
    if (x0 >= x1) return;
    const int c0 = lane * 8;
    const bool act = c0 < ld;
    float wk[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int c = c0 + e; wk[t][e] = (c < M) ? Wk[t * M + c] : 0.f; }
    auto load_col = [&](int xx, float (&c)[3][8]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            if (act && yy >= 0 && yy < h && xx >= 0 && xx < w) ld8<T>(feat + ((long)(b * h + yy) * w + xx) * ld + c0, c[r]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) c[r][e] = 0.f;
            }
        }
    };
    float cL[3][8], cM[3][8], cR[3][8], cN[3][8];
    load_col(x0 - 1, cL); load_col(x0, cM); load_col(x0 + 1, cR);
    const float bs = bias[0];
    for (int x = x0; x < x1; ++x) {
        load_col(x + 2, cN);
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc += cL[r][e] * wk[r * 3][e] + cM[r][e] * wk[r * 3 + 1][e] + cR[r][e] * wk[r * 3 + 2][e];
        acc = wave_sum(acc);
        if (lane == 0) score[(b * h + y) * w + x] = acc + bs;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) { cL[r][e] = cM[r][e]; cM[r][e] = cR[r][e]; cR[r][e] = cN[r][e]; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void score_conv_bwd_data_row_kernel(const float* __restrict__ dscore, const float* __restrict__ Wk, T* __restrict__ dfeat,
                                                                     int accumulate, int h, int w, int ld, int M) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x / h, y = blockIdx.x % h;
    const int per = (w + WPB - 1) / WPB, x0 = wv * per, x1 = min(w, x0 + per);
    const int c0 = lane * 8;
    if (x0 >= x1 || c0 >= ld) return;
    float wk[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int c = c0 + e; wk[t][e] = (c < M) ? Wk[t * M + c] : 0.f; }
    for (int x = x0; x < x1; ++x) {
        float ds[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);
            ds[t] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? dscore[(b * h + yy) * w + xx] : 0.f;
        }
        T* d = dfeat + ((long)(b * h + y) * w + x) * ld + c0;
        float o[8];
        if (accumulate) ld8<T>(d, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) v += ds[t] * wk[t][e];
            o[e] = accumulate ? o[e] + v : v;
        }
        st8<T>(d, o);
    }
}

// dW[tap,m] += sum_pixels dscore[b, y-dy, x-dx] * feat[b,y,x,m];  dbias += sum dscore
template <typename T>
__global__ __launch_bounds__(256) void score_conv_bwd_w_kernel(const float* __restrict__ dscore, const T* __restrict__ feat, float* part, float* bias_part,
                                                              int B, int h, int w, int ld, int M) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int total = B * h * w;
    float acc[9][MB][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < MB; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[t][k][e] = 0.f;
    float sb = 0.f;
    for (int p = blockIdx.x * WPB + wv; p < total; p += gridDim.x * WPB) {
        const int b = p / (h * w), y = (p / w) % h, x = p % w;
        float ds[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1, yy = y - dy, xx = x - dx;
            ds[t] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? dscore[(b * h + yy) * w + xx] : 0.f;
        }
        sb += dscore[p];
        const T* f = feat + (long)p * ld;
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float v[8];
                ld8<T>(f + c0, v);
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[t][k][e] += ds[t] * v[e];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
#pragma unroll
                for (int e = 0; e < 8; ++e) lds[wv * ld + c0 + e] = acc[t][k][e];
            }
        }
        __syncthreads();
        float* pr = part + ((long)blockIdx.x * 9 + t) * ld;                          // this workgroup's partial row of tap t
        for (int c = threadIdx.x; c < ld; c += 256)
            pr[c] = (c < M) ? lds[c] + lds[ld + c] + lds[2 * ld + c] + lds[3 * ld + c] : 0.f;
    }
    // bias gradient: this workgroup's partial (its 4 waves in a fixed order), folded over the workgroups by reduce_parts
    __syncthreads();
    if (lane == 0) lds[wv] = sb;
    __syncthreads();
    if (threadIdx.x == 0) bias_part[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// tf.image.resize_bilinear (align_corners=False, legacy): in = out_idx * (in_size / out_size) in
// float32; lo = floor(in); hi = min(lo + 1, in_size - 1); lerp = in - lo.
__device__ __forceinline__ void interp_coef(int o, float scale, int n_in, int& lo, int& hi, float& lerp) {
    const float in = (float)o * scale;
    lo = (int)floorf(in);
    hi = min(lo + 1, n_in - 1);
    lerp = in - (float)lo;
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ score, float* __restrict__ up, float* __restrict__ sigm,
                                                          const float* __restrict__ target, float* loss_part, int* inter, int* uni,
                                                          int h, int w, int H, int W) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const float* s = score + (long)b * h * w;
    float ls = 0.f; int ci = 0, cu = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const int Y = i / W, X = i % W;
        int y0, y1, x0, x1; float fy, fx;
        interp_coef(Y, sy, h, y0, y1, fy);
        interp_coef(X, sx, w, x0, x1, fx);
        const float tl = s[y0 * w + x0], tr = s[y0 * w + x1], bl = s[y1 * w + x0], br = s[y1 * w + x1];
        const float top = tl + (tr - tl) * fx;
        const float bot = bl + (br - bl) * fx;
        const float u = top + (bot - top) * fy;
        const long o = (long)b * H * W + i;
        up[o] = u;
        if (sigm) sigm[o] = 1.0f / (1.0f + expf(-u));
        if (target) {
            const float z = target[o];
            ls += fmaxf(u, 0.f) - u * z + log1pf(expf(-fabsf(u)));     // sigmoid_cross_entropy_with_logits
            const bool pr = u > 0.f, gt = z != 0.f;
            ci += (pr && gt) ? 1 : 0; cu += (pr || gt) ? 1 : 0;
        }
    }
    if (target) {
        ls = block_sum_256(ls, red);
        const float fi = block_sum_256((float)ci, red);
        const float fu = block_sum_256((float)cu, red);
        // loss: this workgroup's partial (folded in a fixed order by reduce_parts); the pixel counters are integers: atomics are exact
        if (threadIdx.x == 0) { loss_part[(long)b * gridDim.x + blockIdx.x] = ls; atomicAdd(inter + b, (int)(fi + 0.5f)); atomicAdd(uni + b, (int)(fu + 0.5f)); }
    }
}

// One WAVE per low-resolution pixel: its lanes share the (at most ~20 x 20) window of high-resolution pixels whose
// legacy-bilinear footprint can touch it (consecutive lanes = consecutive X: coalesced reads), then one wave
// reduction.  (One THREAD per pixel walked that window alone with scattered reads: 50 workgroups, 73 us.)
__global__ __launch_bounds__(256) void upsample_loss_bwd_kernel(const float* __restrict__ up, const float* __restrict__ target, float* __restrict__ dscore,
                                                               float wscale, int B, int h, int w, int H, int W) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= B * h * w) return;
    const int b = p / (h * w), y = (p / w) % h, x = p % w;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int Ylo = max(0, (int)floorf((float)(y - 1) / sy) - 1), Yhi = min(H - 1, (int)ceilf((float)(y + 1) / sy) + 1);
    const int Xlo = max(0, (int)floorf((float)(x - 1) / sx) - 1), Xhi = min(W - 1, (int)ceilf((float)(x + 1) / sx) + 1);
    const int nX = Xhi - Xlo + 1, cnt = (Yhi - Ylo + 1) * nX;
    float acc = 0.f;
    for (int i = lane; i < cnt; i += 64) {
        const int Y = Ylo + i / nX, X = Xlo + i % nX;
        int y0, y1, x0, x1; float fy, fx;
        interp_coef(Y, sy, h, y0, y1, fy);
        interp_coef(X, sx, w, x0, x1, fx);
        const float wy = (y0 == y ? 1.f - fy : 0.f) + (y1 == y ? fy : 0.f);
        const float wx = (x0 == x ? 1.f - fx : 0.f) + (x1 == x ? fx : 0.f);
        if (wy != 0.f && wx != 0.f) {
            const long o = ((long)b * H + Y) * W + X;
            acc += wy * wx * (1.0f / (1.0f + expf(-up[o])) - target[o]);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) dscore[p] = acc * wscale;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int cmpc_score_conv_fwd(int dt, const void* feat, const float* Wk, const float* bias, float* score,
                                   int B, int h, int w, int ld, int M, void* stream) {
    if (ld % 8 || M > ld || ld > MB * 512) { cmpc_set_error("score_conv_fwd: bad ld/M"); return CMPC_EINVAL; }
    const int total = B * h * w, g = (total + 3) / 4 > 2048 ? 2048 : (total + 3) / 4;
    CMPC_DISPATCH_DT(dt, {
        if (ld <= 512) hipLaunchKernelGGL((score_conv_fwd_row_kernel<T>), dim3(B * h), dim3(256), 0, ST, (const T*)feat, Wk, bias, score, h, w, ld, M);
        else hipLaunchKernelGGL((score_conv_fwd_kernel<T, 2>), dim3(g), dim3(256), 0, ST, (const T*)feat, Wk, bias, score, B, h, w, ld, M);
    });
    return cmpc_check_launch("score_conv_fwd");
}

extern "C" int cmpc_score_conv_bwd(int dt, const float* dscore, const void* feat, const float* Wk, void* dfeat, int accumulate,
                                   float* dWk, float* dbias, int B, int h, int w, int ld, int M, void* stream) {
    cmpc_op_scope op_("score_conv_bwd");
    if (ld % 8 || M > ld || ld > MB * 512) { cmpc_set_error("score_conv_bwd: bad ld/M"); return CMPC_EINVAL; }
    const int total = B * h * w, g = (total + 3) / 4 > 2048 ? 2048 : (total + 3) / 4;
    CMPC_DISPATCH_DT(dt, {
        if (dfeat && ld <= 512) hipLaunchKernelGGL((score_conv_bwd_data_row_kernel<T>), dim3(B * h), dim3(256), 0, ST, dscore, Wk, (T*)dfeat, accumulate, h, w, ld, M);
        else if (dfeat) hipLaunchKernelGGL((score_conv_bwd_data_kernel<T, 2>), dim3(g), dim3(256), 0, ST, dscore, Wk, (T*)dfeat, accumulate, B, h, w, ld, M);
    });
    if (dWk) {
        const int gw = g > 256 ? 256 : g;
        float* part = (float*)cmpc_ws(((size_t)gw * 9 * ld + gw) * sizeof(float), ST);
        if (!part) return CMPC_EHIP;
        float* bias_part = part + (size_t)gw * 9 * ld;
        CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((score_conv_bwd_w_kernel<T>), dim3(gw), dim3(256), WPB * ld * sizeof(float), ST,
                                                 dscore, (const T*)feat, part, bias_part, B, h, w, ld, M));
        if (cmpc_reduce_parts_f32(part, 9L * ld, 1, gw, 9, ld, M, dWk, 0, M, 1, ST)) return CMPC_EHIP;
        if (dbias && cmpc_reduce_parts_f32(bias_part, 1, 1, gw, 1, 1, 1, dbias, 0, 0, 1, ST)) return CMPC_EHIP;
    }
    return cmpc_check_launch("score_conv_bwd");
}

extern "C" int cmpc_upsample_fwd(const float* score, float* up, float* sigm, const float* target, float* loss,
                                 int* inter, int* uni, int B, int h, int w, int H, int W, void* stream) {
    cmpc_op_scope op_("upsample_fwd");
    if (target && (!loss || !inter || !uni)) { cmpc_set_error("upsample_fwd: loss/inter/uni required with target"); return CMPC_EINVAL; }
    const int gx = (H * W + 255) / 256 > 64 ? 64 : (H * W + 255) / 256;
    float* loss_part = nullptr;
    if (target) { loss_part = (float*)cmpc_ws((size_t)B * gx * sizeof(float), ST); if (!loss_part) return CMPC_EHIP; }
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(gx, B), dim3(256), 0, ST, score, up, sigm, target, loss_part, inter, uni, h, w, H, W);
    if (target && cmpc_reduce_parts_f32(loss_part, 1, B, gx, 1, 1, 1, loss, 1, 0, 1, ST)) return CMPC_EHIP;       // loss[b] += sum over the workgroups
    return cmpc_check_launch("upsample_fwd");
}

extern "C" int cmpc_upsample_loss_bwd(const float* up, const float* target, float* dscore, float wscale,
                                      int B, int h, int w, int H, int W, void* stream) {
    hipLaunchKernelGGL(upsample_loss_bwd_kernel, dim3((B * h * w + 3) / 4), dim3(256), 0, ST, up, target, dscore, wscale, B, h, w, H, W);
    return cmpc_check_launch("upsample_loss_bwd");
}
