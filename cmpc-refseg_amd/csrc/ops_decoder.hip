// Stage kernels of the CMPCv5_BiLSTM graph that CMPC_model does not have (reference: /root/reference/CMPCv5_BiLSTM_model.py, "v5:",
// and CMPCv5_BiLSTM_HSV_model.py, "hsv:"): slim batch_norm in training and inference mode (v5:190-251 under resnet_arg_scope), legacy
// bilinear resizing of whole feature maps (v5:201,246), tf.image.rgb_to_hsv of the input image (hsv:120-126), array_ops.reverse_sequence
// for the backward LSTM direction (v5:170-174), the sequence mask of the concatenated BiLSTM outputs (v5:181) and the decoder's last
// 1x1 convolution to one channel (v5:205).  All of them are HBM-bound streams over [rows, channels] maps: a lane owns 8 consecutive
// channels (16-byte accesses), column sums go through per-workgroup partial rows in float64 that ONE later workgroup folds in a fixed
// order (no atomics: results do not depend on which workgroup finishes first).
#include "cmpc_common.h"
#include "../../include/cmpc.h"
#include <math.h>

namespace {

// lanes of a wave as (row_sub, column group): lpr = lanes per row = min(64, ld / 8) when that divides 64, else 64
__host__ __device__ inline int lanes_per_row(int ld) {
    const int cg = ld / 8;
    return (cg <= 64 && (64 % cg) == 0) ? cg : 64;
}
constexpr int BN_ROWS = 128;      // rows per workgroup of the statistics kernels (4 waves x 32)
constexpr int BN_MB = 4;          // column blocks of 512 per lane: ld <= 2048

// sums over the rows of s0 = f(x) and s1 = g(x) per channel; MODE 0: (x, x^2) for the forward statistics;
// MODE 1: (dy*[y>0], dy*[y>0]*xhat) for the backward pass (relu gate through the OUTPUT y)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                                        const float* __restrict__ mean_rstd, int Cpad, int relu, double* __restrict__ part, int R, int C) {
    extern __shared__ double lds[];            // [4][2][Cpad]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ldw = (C + 7) / 8 * 8, lpr = lanes_per_row(ldw), rsub = 64 / lpr, rs = lane / lpr, cgp = lane % lpr;
    float a0[BN_MB][8], a1[BN_MB][8], mu[BN_MB][8], rsd[BN_MB][8];
#pragma unroll
    for (int k = 0; k < BN_MB; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a0[k][e] = 0.f; a1[k][e] = 0.f; mu[k][e] = 0.f; rsd[k][e] = 0.f; }
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < BN_MB; ++k) {
            const int c0 = k * 512 + cgp * 8;
            if (c0 < ldw) { ld8<float>(mean_rstd + c0, mu[k]); ld8<float>(mean_rstd + Cpad + c0, rsd[k]); }
        }
    }
    const int r_begin = blockIdx.x * BN_ROWS + w * (BN_ROWS / 4), r_end = min(R, r_begin + BN_ROWS / 4);
    for (int r = r_begin + rs; r < r_end; r += rsub) {
#pragma unroll
        for (int k = 0; k < BN_MB; ++k) {
            const int c0 = k * 512 + cgp * 8;
            if (c0 < ldw) {
                float xv[8];
                ld8<T>(x + (long)r * ldx + c0, xv);
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { a0[k][e] += xv[e]; a1[k][e] += xv[e] * xv[e]; }
                } else {
                    float dv[8], yv[8];
                    ld8<T>(dy + (long)r * lddy + c0, dv);
                    if (relu) ld8<T>(y + (long)r * ldy + c0, yv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float g = (!relu || yv[e] > 0.f) ? dv[e] : 0.f;
                        a0[k][e] += g; a1[k][e] += g * (xv[e] - mu[k][e]) * rsd[k][e];
                    }
                }
            }
        }
    }
    // across the row_sub lanes of the wave (fp32: <= 32 rows each), then across the 4 waves in float64
#pragma unroll
    for (int k = 0; k < BN_MB; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            for (int o = lpr; o < 64; o <<= 1) { a0[k][e] += __shfl_xor(a0[k][e], o, 64); a1[k][e] += __shfl_xor(a1[k][e], o, 64); }
    if (rs == 0) {
#pragma unroll
        for (int k = 0; k < BN_MB; ++k) {
            const int c0 = k * 512 + cgp * 8;
            if (c0 < ldw)
#pragma unroll
                for (int e = 0; e < 8; ++e) { lds[(w * 2) * Cpad + c0 + e] = (double)a0[k][e]; lds[(w * 2 + 1) * Cpad + c0 + e] = (double)a1[k][e]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ldw; c += 256) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { s0 += lds[(ww * 2) * Cpad + c]; s1 += lds[(ww * 2 + 1) * Cpad + c]; }
        part[((long)blockIdx.x * 2) * Cpad + c] = s0;
        part[((long)blockIdx.x * 2 + 1) * Cpad + c] = s1;
    }
}

// Fold of the per-workgroup partial rows.  A workgroup owns FOLD_C channels; FOLD_S threads per channel each add a contiguous run of
// the partials in ascending order, thread 0 of the channel adds the FOLD_S results in a fixed order: deterministic, and 1024 partials
// (131 072 rows at the decoder's resolution) are 64 dependent additions deep instead of 1024 (250 us -> 20 us per batch-norm).
constexpr int FOLD_C = 16, FOLD_S = 16;
__device__ __forceinline__ bool fold_parts(const double* __restrict__ part, int nparts, int C, int Cpad, int& c, double& s0, double& s1) {
    __shared__ double red[2][FOLD_S][FOLD_C];
    const int cl = threadIdx.x % FOLD_C, sub = threadIdx.x / FOLD_C;
    c = blockIdx.x * FOLD_C + cl;
    const int per = (nparts + FOLD_S - 1) / FOLD_S, i0 = sub * per, i1 = min(nparts, i0 + per);
    double a0 = 0.0, a1 = 0.0;
    if (c < C) for (int i = i0; i < i1; ++i) { a0 += part[((long)i * 2) * Cpad + c]; a1 += part[((long)i * 2 + 1) * Cpad + c]; }
    red[0][sub][cl] = a0; red[1][sub][cl] = a1;
    __syncthreads();
    if (sub != 0 || c >= Cpad) return false;
    s0 = 0.0; s1 = 0.0;
#pragma unroll
    for (int k = 0; k < FOLD_S; ++k) { s0 += red[0][k][cl]; s1 += red[1][k][cl]; }
    return true;
}
// fold of the forward partials: sums[2][Cpad] (float64: kept for the moving-statistics update) and mean_rstd[2][Cpad]
__global__ __launch_bounds__(FOLD_C * FOLD_S) void bn_stats_finish_kernel(const double* __restrict__ part, int nparts, double* __restrict__ sums, float* __restrict__ mean_rstd,
                                                                        int R, int C, int Cpad, float eps) {
    int c; double s0, s1;
    if (!fold_parts(part, nparts, C, Cpad, c, s0, s1)) return;
    sums[c] = s0; sums[Cpad + c] = s1;
    const double m = s0 / R;
    double var = s1 / R - m * m;
    if (var < 0.0) var = 0.0;
    mean_rstd[c] = c < C ? (float)m : 0.f;
    mean_rstd[Cpad + c] = c < C ? (float)(1.0 / sqrt(var + (double)eps)) : 0.f;
}
// inference mode: mean / rstd from the moving statistics
__global__ void bn_from_moving_kernel(const float* __restrict__ mm, const float* __restrict__ mv, float* __restrict__ mean_rstd, int C, int Cpad, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Cpad) return;
    mean_rstd[c] = c < C ? mm[c] : 0.f;
    mean_rstd[Cpad + c] = c < C ? (float)(1.0 / sqrt((double)mv[c] + (double)eps)) : 0.f;
}
// moving = decay * moving + (1 - decay) * batch; the variance with Bessel's correction (fused_batch_norm)
__global__ void bn_update_moving_kernel(const double* __restrict__ sums, int R, float decay, float* __restrict__ mm, float* __restrict__ mv, int C, int Cpad) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double m = sums[c] / R;
    double var = sums[Cpad + c] / R - m * m;
    if (var < 0.0) var = 0.0;
    var *= (double)R / (double)(R > 1 ? R - 1 : 1);
    mm[c] = (float)((double)mm[c] * decay + m * (1.0 - (double)decay));
    mv[c] = (float)((double)mv[c] * decay + var * (1.0 - (double)decay));
}

// y = act(gamma * (x - mean) * rstd + beta) on channels < C, 0 on the pad channels C .. Cy of the destination block.
// Same lane -> (row_sub, 8 channels) mapping as the statistics kernel: the per-channel scale / shift live in registers, no index division.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ mean_rstd, int Cpad, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y, int ldy, int Cy, int R, int C, int relu) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lpr = lanes_per_row(Cy), rsub = 64 / lpr, rs = lane / lpr, cgp = lane % lpr, cin = (C + 7) / 8 * 8;
    float mu[BN_MB][8], rsd[BN_MB][8], ga[BN_MB][8], be[BN_MB][8];
#pragma unroll
    for (int k = 0; k < BN_MB; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = k * 512 + cgp * 8 + e;
            mu[k][e] = rsd[k][e] = ga[k][e] = be[k][e] = 0.f;
            if (c < C) { mu[k][e] = mean_rstd[c]; rsd[k][e] = mean_rstd[Cpad + c]; ga[k][e] = gamma[c]; be[k][e] = beta[c]; }
        }
    const int r_begin = blockIdx.x * BN_ROWS + w * (BN_ROWS / 4), r_end = min(R, r_begin + BN_ROWS / 4);
    for (int r = r_begin + rs; r < r_end; r += rsub) {
#pragma unroll
        for (int k = 0; k < BN_MB; ++k) {
            const int c0 = k * 512 + cgp * 8;
            if (c0 >= Cy) continue;
            float v[8];
            if (c0 < cin) {
                float xv[8];
                ld8<T>(x + (long)r * ldx + c0, xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float o = (c0 + e < C) ? (xv[e] - mu[k][e]) * rsd[k][e] * ga[k][e] + be[k][e] : 0.f;
                    v[e] = relu ? fmaxf(o, 0.f) : o;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            st8<T>(y + (long)r * ldy + c0, v);
        }
    }
}

// fold of the backward partials -> dbeta, dgamma (one writer per element) and the two means the apply pass needs
__global__ __launch_bounds__(FOLD_C * FOLD_S) void bn_bwd_finish_kernel(const double* __restrict__ part, int nparts, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                      float* __restrict__ means, int R, int C, int Cpad) {
    int c; double s0, s1;
    if (!fold_parts(part, nparts, C, Cpad, c, s0, s1)) return;
    if (c < C) { dbeta[c] += (float)s0; dgamma[c] += (float)s1; }
    means[c] = (float)(s0 / R); means[Cpad + c] = (float)(s1 / R);
}
// dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * [y > 0]
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy, const T* __restrict__ x, int ldx,
                                                          const float* __restrict__ mean_rstd, const float* __restrict__ means, int Cpad, const float* __restrict__ gamma,
                                                          T* __restrict__ dx, int lddx, int R, int C, int relu) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cin = (C + 7) / 8 * 8, lpr = lanes_per_row(cin), rsub = 64 / lpr, rs = lane / lpr, cgp = lane % lpr;
    float mu[BN_MB][8], rsd[BN_MB][8], gr[BN_MB][8], m0[BN_MB][8], m1[BN_MB][8];
#pragma unroll
    for (int k = 0; k < BN_MB; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = k * 512 + cgp * 8 + e;
            mu[k][e] = rsd[k][e] = gr[k][e] = m0[k][e] = m1[k][e] = 0.f;
            if (c < C) { mu[k][e] = mean_rstd[c]; rsd[k][e] = mean_rstd[Cpad + c]; gr[k][e] = gamma[c] * rsd[k][e]; m0[k][e] = means[c]; m1[k][e] = means[Cpad + c]; }
        }
    const int r_begin = blockIdx.x * BN_ROWS + w * (BN_ROWS / 4), r_end = min(R, r_begin + BN_ROWS / 4);
    for (int r = r_begin + rs; r < r_end; r += rsub) {
#pragma unroll
        for (int k = 0; k < BN_MB; ++k) {
            const int c0 = k * 512 + cgp * 8;
            if (c0 >= cin) continue;
            float dv[8], yv[8], xv[8], o[8];
            ld8<T>(dy + (long)r * lddy + c0, dv);
            if (relu) ld8<T>(y + (long)r * ldy + c0, yv);
            ld8<T>(x + (long)r * ldx + c0, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = (!relu || yv[e] > 0.f) ? dv[e] : 0.f;
                const float xh = (xv[e] - mu[k][e]) * rsd[k][e];
                o[e] = (c0 + e < C) ? gr[k][e] * (g - m0[k][e] - xh * m1[k][e]) : 0.f;
            }
            st8<T>(dx + (long)r * lddx + c0, o);
        }
    }
}

// tf.image.resize_bilinear (align_corners=False, legacy): in = out_idx * (in_size / out_size) in float32; lo = floor(in);
// hi = min(lo + 1, in_size - 1); lerp = in - lo; value = top + (bot - top) * ly with top = tl + (tr - tl) * lx
__device__ __forceinline__ void interp_coef(int o, float scale, int n_in, int& lo, int& hi, float& lerp) {
    const float in = (float)o * scale;
    lo = (int)floorf(in);
    hi = min(lo + 1, n_in - 1);
    lerp = in - (float)lo;
}
template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int B, int h, int w, int H, int W, int C) {
    const int cg = C / 8;
    const long total = (long)B * H * W * cg;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cg; const int c0 = (int)(i - p * cg) * 8;
        const int X = (int)(p % W), Y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        int y0, y1, x0, x1; float fy, fx;
        interp_coef(Y, sy, h, y0, y1, fy);
        interp_coef(X, sx, w, x0, x1, fx);
        const T* base = x + (long)b * h * w * ldx + c0;
        float tl[8], tr[8], bl[8], br[8], o[8];
        ld8<T>(base + ((long)y0 * w + x0) * ldx, tl); ld8<T>(base + ((long)y0 * w + x1) * ldx, tr);
        ld8<T>(base + ((long)y1 * w + x0) * ldx, bl); ld8<T>(base + ((long)y1 * w + x1) * ldx, br);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float top = tl[e] + (tr[e] - tl[e]) * fx, bot = bl[e] + (br[e] - bl[e]) * fx;
            o[e] = top + (bot - top) * fy;
        }
        st8<T>(y + p * ldy + c0, o);
    }
}
// gradient of the resize: every INPUT pixel gathers the output pixels whose footprint contains it (one writer per element)
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int B, int h, int w, int H, int W, int C) {
    const int cg = C / 8;
    const long total = (long)B * h * w * cg;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cg; const int c0 = (int)(i - p * cg) * 8;
        const int x = (int)(p % w), y = (int)((p / w) % h), b = (int)(p / ((long)w * h));
        const int Ylo = max(0, (int)floorf((float)(y - 1) / sy) - 1), Yhi = min(H - 1, (int)ceilf((float)(y + 1) / sy) + 1);
        const int Xlo = max(0, (int)floorf((float)(x - 1) / sx) - 1), Xhi = min(W - 1, (int)ceilf((float)(x + 1) / sx) + 1);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int Y = Ylo; Y <= Yhi; ++Y) {
            int y0, y1; float fy;
            interp_coef(Y, sy, h, y0, y1, fy);
            const float wy = (y0 == y ? 1.f - fy : 0.f) + (y1 == y ? fy : 0.f);
            if (wy == 0.f) continue;
            for (int X = Xlo; X <= Xhi; ++X) {
                int x0, x1; float fx;
                interp_coef(X, sx, w, x0, x1, fx);
                const float wx = (x0 == x ? 1.f - fx : 0.f) + (x1 == x ? fx : 0.f);
                if (wx == 0.f) continue;
                float d[8];
                ld8<T>(dy + (((long)b * H + Y) * W + X) * lddy + c0, d);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += wy * wx * d[e];
            }
        }
        st8<T>(dx + p * lddx + c0, acc);
    }
}

// tf.image.rgb_to_hsv of one pixel (core/kernels/colorspace_op.h), any value range
__device__ __forceinline__ void rgb2hsv(float r, float g, float b, float& hh, float& ss, float& vv) {
    const float v = fmaxf(r, fmaxf(g, b)), rng = v - fminf(r, fminf(g, b));
    ss = v > 0.f ? rng / v : 0.f;
    const float norm = 1.0f / (6.0f * rng);
    float h = (r == v) ? norm * (g - b) : ((g == v) ? norm * (b - r) + 2.0f / 6.0f : norm * (r - g) + 4.0f / 6.0f);
    h = rng > 0.f ? h : 0.f;
    hh = h < 0.f ? h + 1.0f : h;
    vv = v;
}
// hsv:120-126: im (BGR minus mean) + mean, reversed to RGB, rgb_to_hsv, legacy bilinear to [h, w]; out [B*h*w, ld] (3 channels, rest 0)
template <typename T>
__global__ __launch_bounds__(256) void hsv_map_kernel(const float* __restrict__ im, T* __restrict__ out, int ld, int B, int H, int W, int h, int w,
                                                     float mu_b, float mu_g, float mu_r) {
    const long total = (long)B * h * w;
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const int x = (int)(p % w), y = (int)((p / w) % h), b = (int)(p / ((long)w * h));
        int y0, y1, x0, x1; float fy, fx;
        interp_coef(y, sy, H, y0, y1, fy);
        interp_coef(x, sx, W, x0, x1, fx);
        float c[4][3];
        const int ys[4] = {y0, y0, y1, y1}, xs[4] = {x0, x1, x0, x1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* px = im + (((long)b * H + ys[k]) * W + xs[k]) * 3;
            rgb2hsv(px[2] + mu_r, px[1] + mu_g, px[0] + mu_b, c[k][0], c[k][1], c[k][2]);
        }
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const float top = c[0][e] + (c[1][e] - c[0][e]) * fx, bot = c[2][e] + (c[3][e] - c[2][e]) * fx;
            v[e] = top + (bot - top) * fy;
        }
        st8<T>(out + p * ld, v);
        const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c0 = 8; c0 < ld; c0 += 8) st8<T>(out + p * ld + c0, z);
    }
}

// array_ops.reverse_sequence on [B, T, ld] float rows: out[b, t] = in[b, t < len ? len - 1 - t : t]
__global__ void reverse_sequence_kernel(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ seq_len, int T, int ld) {
    const int b = blockIdx.y, t = blockIdx.x;
    const int n = min(max(seq_len[b], 0), T), s = t < n ? n - 1 - t : t;
    for (int c = threadIdx.x; c < ld; c += blockDim.x) out[((long)b * T + t) * ld + c] = in[((long)b * T + s) * ld + c];
}
// words_tb[t * B + b] = words[b, t < len ? len - 1 - t : t]   (time-major ids of the reversed sequences)
__global__ void reverse_words_tb_kernel(const int* __restrict__ words, const int* __restrict__ seq_len, int* __restrict__ out, int B, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i - b * T;
    const int n = min(max(seq_len[b], 0), T), s = t < n ? n - 1 - t : t;
    out[t * B + b] = words[b * T + s];
}
// seq_mask of v5:181: 1 where sum |[fw | bw]| != 0
__global__ void rows_nonzero2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ mask, int ld, int C) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += fabsf(a[(long)r * ld + c]) + fabsf(b[(long)r * ld + c]);
    s = wave_sum(s);
    if (lane == 0) mask[r] = s != 0.f ? 1.f : 0.f;
}

// decoder's last 1x1 convolution to ONE channel (v5:205): out[r] = x[r, :] . w + bias; one wave per row
template <typename T>
__global__ __launch_bounds__(256) void conv_to1_fwd_kernel(const T* __restrict__ x, int ld, const float* __restrict__ wv, const float* __restrict__ bias, float* __restrict__ out,
                                                          long R, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float b0 = bias[0];
    for (long r = (long)blockIdx.x * 4 + w; r < R; r += (long)gridDim.x * 4) {
        float s = 0.f;
        for (int c0 = lane * 8; c0 < C; c0 += 512) {
            float xv[8];
            ld8<T>(x + r * ld + c0, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (c0 + e < C) s += xv[e] * wv[c0 + e];
        }
        s = wave_sum(s);
        if (lane == 0) out[r] = s + b0;
    }
}
// dx[r, :] = d[r] * w; per-workgroup partial rows of dw[c] = sum_r d[r] x[r, c] and db = sum_r d[r] (folded by reduce_parts)
template <typename T>
__global__ __launch_bounds__(256) void conv_to1_bwd_kernel(const float* __restrict__ d, const T* __restrict__ x, int ld, const float* __restrict__ wv, T* __restrict__ dx,
                                                          float* __restrict__ part, long R, int C) {
    extern __shared__ float lds_f[];           // [4][ld + 8]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float aw[BN_MB][8], ab = 0.f, wr[BN_MB][8];
#pragma unroll
    for (int k = 0; k < BN_MB; ++k) {
        const int c0 = k * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { aw[k][e] = 0.f; wr[k][e] = (c0 + e < C) ? wv[c0 + e] : 0.f; }
    }
    for (long r = (long)blockIdx.x * 4 + w; r < R; r += (long)gridDim.x * 4) {
        const float dr = d[r];
        ab += dr;
#pragma unroll
        for (int k = 0; k < BN_MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float xv[8], o[8];
                ld8<T>(x + r * ld + c0, xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) { aw[k][e] += dr * xv[e]; o[e] = dr * wr[k][e]; }
                st8<T>(dx + r * ld + c0, o);
            }
        }
    }
    const int stride = ld + 8;
#pragma unroll
    for (int k = 0; k < BN_MB; ++k) {
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld)
#pragma unroll
            for (int e = 0; e < 8; ++e) lds_f[w * stride + c0 + e] = aw[k][e];
    }
    if (lane == 0) lds_f[w * stride + ld] = ab;       // every lane of the wave saw the same rows
    __syncthreads();
    for (int c = threadIdx.x; c <= ld; c += 256)
        part[(long)blockIdx.x * stride + c] = (lds_f[c] + lds_f[stride + c]) + (lds_f[2 * stride + c] + lds_f[3 * stride + c]);
}

inline int grid_for(long items, int cap = 4096) { long g = (items + 255) / 256; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }
bool bn_ok(const char* what, int ld, int C, int Cpad) {
    if (C < 1 || Cpad % 8 || Cpad < C || Cpad > 512 * BN_MB || ld % 8 || ld < (C + 7) / 8 * 8) {
        cmpc_set_error("%s: need 1 <= C <= Cpad <= %d, Cpad %% 8 == 0, row stride %% 8 == 0 and >= roundup(C, 8)", what, 512 * BN_MB); return false;
    }
    return true;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int cmpc_bn_stats(int dt, const void* x, int ldx, int R, int C, int Cpad, float eps, double* sums, float* mean_rstd, void* stream) {
    if (!x || !sums || !mean_rstd || R < 1 || !bn_ok("bn_stats", ldx, C, Cpad)) { if (R < 1 || !x || !sums || !mean_rstd) cmpc_set_error("bn_stats: bad args"); return CMPC_EINVAL; }
    const int nparts = (R + BN_ROWS - 1) / BN_ROWS;
    double* part = (double*)cmpc_ws((size_t)nparts * 2 * Cpad * sizeof(double), ST);
    if (!part) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((bn_partial_kernel<T, 0>), dim3(nparts), dim3(256), 8 * Cpad * sizeof(double), ST, (const T*)x, ldx, (const T*)nullptr, 0,
                                             (const T*)nullptr, 0, (const float*)nullptr, Cpad, 0, part, R, C));
    if (cmpc_check_launch("bn_stats") != CMPC_OK) return CMPC_EHIP;
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((Cpad + FOLD_C - 1) / FOLD_C), dim3(FOLD_C * FOLD_S), 0, ST, part, nparts, sums, mean_rstd, R, C, Cpad, eps);
    return cmpc_check_launch("bn_stats_finish");
}
extern "C" int cmpc_bn_from_moving(const float* moving_mean, const float* moving_var, int C, int Cpad, float eps, float* mean_rstd, void* stream) {
    if (!moving_mean || !moving_var || !mean_rstd || C < 1 || Cpad < C) { cmpc_set_error("bn_from_moving: bad args"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(bn_from_moving_kernel, dim3((Cpad + 255) / 256), dim3(256), 0, ST, moving_mean, moving_var, mean_rstd, C, Cpad, eps);
    return cmpc_check_launch("bn_from_moving");
}
extern "C" int cmpc_bn_update_moving(const double* sums, int R, float decay, float* moving_mean, float* moving_var, int C, int Cpad, void* stream) {
    if (!sums || !moving_mean || !moving_var || R < 1 || C < 1 || Cpad < C) { cmpc_set_error("bn_update_moving: bad args"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(bn_update_moving_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, sums, R, decay, moving_mean, moving_var, C, Cpad);
    return cmpc_check_launch("bn_update_moving");
}
extern "C" int cmpc_bn_apply_fwd(int dt, const void* x, int ldx, const float* mean_rstd, int Cpad, const float* gamma, const float* beta, void* y, int ldy, int Cy,
                                 int R, int C, int relu, void* stream) {
    if (!x || !y || !mean_rstd || !gamma || !beta || R < 1 || Cy % 8 || Cy < C || ldy % 8 || !bn_ok("bn_apply_fwd", ldx, C, Cpad)) {
        cmpc_set_error("bn_apply_fwd: bad args (Cy %% 8 == 0, Cy >= C, 16-B aligned rows)"); return CMPC_EINVAL;
    }
    if (Cy > 512 * BN_MB) { cmpc_set_error("bn_apply_fwd: Cy <= %d", 512 * BN_MB); return CMPC_EINVAL; }
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((bn_apply_fwd_kernel<T>), dim3((R + BN_ROWS - 1) / BN_ROWS), dim3(256), 0, ST, (const T*)x, ldx, mean_rstd, Cpad, gamma, beta,
                                             (T*)y, ldy, Cy, R, C, relu));
    return cmpc_check_launch("bn_apply_fwd");
}
extern "C" int cmpc_bn_bwd(int dt, const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx, const float* mean_rstd, int Cpad, const float* gamma,
                           void* dx, int lddx, float* dgamma, float* dbeta, float* means_scratch, int R, int C, int relu, void* stream) {
    if (!dy || !x || !dx || !mean_rstd || !gamma || !dgamma || !dbeta || !means_scratch || (relu && !y) || R < 1 || lddy % 8 || lddx % 8 || (relu && ldy % 8) ||
        !bn_ok("bn_bwd", ldx, C, Cpad)) { cmpc_set_error("bn_bwd: bad args"); return CMPC_EINVAL; }
    const int nparts = (R + BN_ROWS - 1) / BN_ROWS;
    double* part = (double*)cmpc_ws((size_t)nparts * 2 * Cpad * sizeof(double), ST);
    if (!part) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((bn_partial_kernel<T, 1>), dim3(nparts), dim3(256), 8 * Cpad * sizeof(double), ST, (const T*)x, ldx, (const T*)dy, lddy,
                                             (const T*)y, ldy, mean_rstd, Cpad, relu, part, R, C));
    if (cmpc_check_launch("bn_bwd_partial") != CMPC_OK) return CMPC_EHIP;
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((Cpad + FOLD_C - 1) / FOLD_C), dim3(FOLD_C * FOLD_S), 0, ST, part, nparts, dgamma, dbeta, means_scratch, R, C, Cpad);
    if (cmpc_check_launch("bn_bwd_finish") != CMPC_OK) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((bn_apply_bwd_kernel<T>), dim3((R + BN_ROWS - 1) / BN_ROWS), dim3(256), 0, ST, (const T*)dy, lddy, (const T*)y, ldy,
                                             (const T*)x, ldx, mean_rstd, means_scratch, Cpad, gamma, (T*)dx, lddx, R, C, relu));
    return cmpc_check_launch("bn_bwd_apply");
}

extern "C" int cmpc_resize_bilinear_fwd(int dt, const void* x, int ldx, void* y, int ldy, int B, int h, int w, int H, int W, int C, void* stream) {
    if (!x || !y || B < 1 || h < 1 || w < 1 || H < 1 || W < 1 || C < 8 || C % 8 || ldx % 8 || ldy % 8) { cmpc_set_error("resize_bilinear_fwd: bad args (C %% 8 == 0, 16-B rows)"); return CMPC_EINVAL; }
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((resize_fwd_kernel<T>), dim3(grid_for((long)B * H * W * (C / 8), 16384)), dim3(256), 0, ST, (const T*)x, ldx, (T*)y, ldy, B, h, w, H, W, C));
    return cmpc_check_launch("resize_bilinear_fwd");
}
extern "C" int cmpc_resize_bilinear_bwd(int dt, const void* dy, int lddy, void* dx, int lddx, int B, int h, int w, int H, int W, int C, void* stream) {
    if (!dy || !dx || B < 1 || h < 1 || w < 1 || H < 1 || W < 1 || C < 8 || C % 8 || lddy % 8 || lddx % 8) { cmpc_set_error("resize_bilinear_bwd: bad args"); return CMPC_EINVAL; }
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((resize_bwd_kernel<T>), dim3(grid_for((long)B * h * w * (C / 8), 16384)), dim3(256), 0, ST, (const T*)dy, lddy, (T*)dx, lddx, B, h, w, H, W, C));
    return cmpc_check_launch("resize_bilinear_bwd");
}
extern "C" int cmpc_hsv_map(int dt, const float* im, void* out, int ld, int B, int H, int W, int h, int w, void* stream) {
    if (!im || !out || B < 1 || ld < 8 || ld % 8) { cmpc_set_error("hsv_map: bad args"); return CMPC_EINVAL; }
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((hsv_map_kernel<T>), dim3(grid_for((long)B * h * w)), dim3(256), 0, ST, im, (T*)out, ld, B, H, W, h, w,
                                             104.00698793f, 116.66876762f, 122.67891434f));
    return cmpc_check_launch("hsv_map");
}
extern "C" int cmpc_reverse_sequence(const float* in, float* out, const int* seq_len, int B, int T, int ld, void* stream) {
    if (!in || !out || !seq_len || in == out || B < 1 || T < 1) { cmpc_set_error("reverse_sequence: bad args (not in place)"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(reverse_sequence_kernel, dim3(T, B), dim3(256), 0, ST, in, out, seq_len, T, ld);
    return cmpc_check_launch("reverse_sequence");
}
extern "C" int cmpc_reverse_words_tb(const int* words, const int* seq_len, int* out_tb, int B, int T, void* stream) {
    if (!words || !seq_len || !out_tb) { cmpc_set_error("reverse_words_tb: bad args"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(reverse_words_tb_kernel, dim3((B * T + 255) / 256), dim3(256), 0, ST, words, seq_len, out_tb, B, T);
    return cmpc_check_launch("reverse_words_tb");
}
extern "C" int cmpc_rows_nonzero2(const float* a, const float* b, float* mask, int rows, int ld, int C, void* stream) {
    if (!a || !b || !mask || rows < 1) { cmpc_set_error("rows_nonzero2: bad args"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(rows_nonzero2_kernel, dim3(rows), dim3(64), 0, ST, a, b, mask, ld, C);
    return cmpc_check_launch("rows_nonzero2");
}
extern "C" int cmpc_conv_to1_fwd(int dt, const void* x, int ld, const float* w, const float* bias, float* out, int R, int C, void* stream) {
    if (!x || !w || !bias || !out || R < 1 || ld % 8 || C > ld) { cmpc_set_error("conv_to1_fwd: bad args"); return CMPC_EINVAL; }
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((conv_to1_fwd_kernel<T>), dim3(grid_for((long)R * 64, 8192)), dim3(256), 0, ST, (const T*)x, ld, w, bias, out, (long)R, C));
    return cmpc_check_launch("conv_to1_fwd");
}
extern "C" int cmpc_conv_to1_bwd(int dt, const float* d, const void* x, int ld, const float* w, void* dx, float* dw, float* dbias, int R, int C, void* stream) {
    cmpc_op_scope op_("conv_to1_bwd");
    if (!d || !x || !w || !dx || !dw || !dbias || R < 1 || ld % 8 || C > ld || ld > 512 * BN_MB) { cmpc_set_error("conv_to1_bwd: bad args"); return CMPC_EINVAL; }
    const int gx = (int)std::min<long>(512, ((long)R + 3) / 4), stride = ld + 8;
    float* part = (float*)cmpc_ws((size_t)gx * stride * sizeof(float), ST);
    if (!part) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((conv_to1_bwd_kernel<T>), dim3(gx), dim3(256), 4 * stride * sizeof(float), ST, d, (const T*)x, ld, w, (T*)dx, part, (long)R, C));
    if (cmpc_check_launch("conv_to1_bwd") != CMPC_OK) return CMPC_EHIP;
    if (cmpc_reduce_parts_f32(part, stride, 1, gx, 1, ld, C, dw, 0, 0, 1, ST)) return CMPC_EHIP;
    return cmpc_reduce_parts_f32(part + ld, stride, 1, gx, 1, 1, 1, dbias, 0, 0, 1, ST);
}
