// Mutan fusion, cross-modal graph softmaxes and gated-exchange elementwise kernels.
#include "cmpc_common.h"
#include "../../include/cmpc.h"

namespace {

constexpr int WPB = 4;
constexpr int MB = 2;          // column blocks of 512: ld <= 1024 for the per-head register tiles
__host__ inline int rows_grid(int N, int cap) { int g = (N + WPB - 1) / WPB; return g < 1 ? 1 : (g > cap ? cap : g); }

// ------------------------------------------------------------------------------------------
// mutan_fusion (CMPC_model.py:295-328)
// ------------------------------------------------------------------------------------------
// FULL = (ld == MB * 512): every lane owns live columns in every block, so the `c0 < ld` guards vanish at compile time.
// With the guards each (head, block) piece is its own exec-masked region and hipcc waits for its load before the next
// region's load is issued: ten dependent round trips per row (mutan_bwd ran at 2.1 TB/s).
// PRE: P already holds tanh(vis_trans_h) (the producing GEMM applied it as its epilogue) and is only read; otherwise P holds the
// pre-activations and is overwritten by their tanh (what mutan_bwd expects to find).
template <typename T, bool FULL, bool PRE>
__global__ __launch_bounds__(256) void mutan_fwd_kernel(T* __restrict__ P, const float* __restrict__ g, T* __restrict__ X1,
                                                       float* __restrict__ rstd, int N, int ld, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    const float* gb = g + (long)b * 5 * ld;
    // the language gates of this sample are the same for every node row: keep this lane's 5 x 16 in registers
    float gv[5][MB][8];
#pragma unroll
    for (int h = 0; h < 5; ++h)
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (FULL || c0 < ld) ld8<float>(gb + h * ld + c0, gv[h][k]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[h][k][e] = 0.f;
            }
        }
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long r = (long)b * N + n;
        T* Pr = P + r * 5 * ld;
        float q[MB][8];
#pragma unroll
        for (int k = 0; k < MB; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) q[k][e] = 0.f;
        // all ten pieces of the row are loaded before the first in-place store: load -> tanh -> store per piece was a
        // chain of ten dependent round trips (the compiler cannot move a load above a store to the same array)
        float pv[5][MB][8];
#pragma unroll
        for (int h = 0; h < 5; ++h)
#pragma unroll
            for (int k = 0; k < MB; ++k) {
                const int c0 = k * 512 + lane * 8;
                if (FULL || c0 < ld) ld8<T>(Pr + h * ld + c0, pv[h][k]);
            }
#pragma unroll
        for (int h = 0; h < 5; ++h) {
#pragma unroll
            for (int k = 0; k < MB; ++k) {
                const int c0 = k * 512 + lane * 8;
                if (FULL || c0 < ld) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float th = (c0 + e < C) ? (PRE ? pv[h][k][e] : cmpc_tanh(pv[h][k][e])) : 0.f;
                        pv[h][k][e] = th;
                        q[k][e] += th * gv[h][k][e];
                    }
                    if (!PRE) st8<T>(Pr + h * ld + c0, pv[h][k]);
                }
            }
        }
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < MB; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = k * 512 + lane * 8 + e;
                q[k][e] = (c < C) ? cmpc_tanh(q[k][e]) : 0.f;
                ss += q[k][e] * q[k][e];
            }
        ss = wave_sum(ss);
        const bool clamped = ss < 1e-12f;
        const float rs = rsqrtf(fmaxf(ss, 1e-12f));
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (FULL || c0 < ld) {
#pragma unroll
                for (int e = 0; e < 8; ++e) q[k][e] *= rs;
                st8<T>(X1 + r * ld + c0, q[k]);
            }
        }
        if (lane == 0) rstd[r] = clamped ? -rs : rs;
    }
}

template <typename T, bool FULL>
__global__ __launch_bounds__(256) void mutan_bwd_kernel(T* __restrict__ Th, const float* __restrict__ g, const T* __restrict__ X1,
                                                       const float* __restrict__ rstd, const T* __restrict__ dX1, float* part,
                                                       int N, int ld, int C) {
    extern __shared__ float lds[];     // [WPB][ld]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    const float* gb = g + (long)b * 5 * ld;
    float gv[5][MB][8];
#pragma unroll
    for (int h = 0; h < 5; ++h)
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (FULL || c0 < ld) ld8<float>(gb + h * ld + c0, gv[h][k]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[h][k][e] = 0.f;
            }
        }
    float acc[5][MB][8];
#pragma unroll
    for (int h = 0; h < 5; ++h)
#pragma unroll
        for (int k = 0; k < MB; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[h][k][e] = 0.f;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long r = (long)b * N + n;
        float d[MB][8], xv[MB][8];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (FULL || c0 < ld) {
                ld8<T>(dX1 + r * ld + c0, d[k]); ld8<T>(X1 + r * ld + c0, xv[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (c0 + e >= C) d[k][e] = 0.f; dot += d[k][e] * xv[k][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { d[k][e] = 0.f; xv[k][e] = 0.f; }
            }
        }
        dot = wave_sum(dot);
        const float rs = rstd[r];
        const float a = fabsf(rs);
        if (rs < 0.f) dot = 0.f;
        // dq = l2norm-bwd * (1 - tq^2), tq = X1 / |rs|
#pragma unroll
        for (int k = 0; k < MB; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float tq = xv[k][e] / a;
                d[k][e] = a * (d[k][e] - xv[k][e] * dot) * (1.f - tq * tq);
            }
        T* Tr = Th + r * 5 * ld;
        // per column block: the five heads' pieces are loaded before the first in-place store (see mutan_fwd)
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (FULL || c0 < ld) {
                float tv[5][8];
#pragma unroll
                for (int h = 0; h < 5; ++h) ld8<T>(Tr + h * ld + c0, tv[h]);
#pragma unroll
                for (int h = 0; h < 5; ++h) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float th = tv[h][e];
                        acc[h][k][e] += d[k][e] * th;
                        tv[h][e] = (c0 + e < C) ? d[k][e] * gv[h][k][e] * (1.f - th * th) : 0.f;
                    }
                    st8<T>(Tr + h * ld + c0, tv[h]);
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 5; ++h) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (FULL || c0 < ld) {
#pragma unroll
                for (int e = 0; e < 8; ++e) lds[w * ld + c0 + e] = acc[h][k][e];
            }
        }
        __syncthreads();
        float* pr = part + (((long)b * gridDim.x + blockIdx.x) * 5 + h) * ld;      // this workgroup's partial row
        for (int c = threadIdx.x; c < ld; c += 256)
            pr[c] = (c < C) ? lds[c] + lds[ld + c] + lds[2 * ld + c] + lds[3 * ld + c] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// build_spa_graph softmaxes: one 1024-thread workgroup per sample; lane = word index t.
// ------------------------------------------------------------------------------------------
constexpr float F32_MIN = -3.4028234663852886e38f;     // tf.float32.min

// Chunked over the N nodes: grid (chunks, B), 4 waves per workgroup, one node row per wave step,
// lane = word index.  Pass 1 writes gw_w and per-chunk online column statistics (max, sum exp);
// pass 2 folds the chunk statistics and writes gw_v.  (One workgroup per sample was 8 workgroups
// on a 256-CU chip.)
constexpr int GS_ROWS = 64;     // node rows per workgroup

template <typename T>
__global__ __launch_bounds__(256) void graph_softmax_fwd1_kernel(const float* __restrict__ A0, const float* __restrict__ pr, const float* __restrict__ mask,
                                                                float* __restrict__ gw_w, T* __restrict__ gw_w_t, float* __restrict__ cstat,
                                                                int N, int Tn, int Tp, int mask_after) {
    __shared__ float cmax[4][64], csum[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y, ch = blockIdx.x;
    const bool tv = lane < Tn;
    const float prt = tv ? pr[b * Tn + lane] : 0.f;
    const float mk = tv ? mask[b * Tn + lane] : 0.f;
    const long sb = (long)b * N * Tp;
    float m_run = -INFINITY, s_run = 0.f;
    const int n1 = min(N, (ch + 1) * GS_ROWS);
    for (int n = ch * GS_ROWS + w; n < n1; n += 4) {
        const float a = tv ? prt * A0[sb + (long)n * Tp + lane] : 0.f;
        // CMPC_model.py:389-394: padded words leave the softmax (logit float32.min); CMPCv5_BiLSTM_model.py:486-487: they stay in it
        // (with logit parse_R * affinity = 0) and the result is masked afterwards
        const float lg = tv ? (mask_after ? a : (mk * a + (1.f - mk) * F32_MIN)) : -INFINITY;
        const float mx = wave_max(lg);
        const float ex = tv ? expf(lg - mx) : 0.f;
        const float sm = wave_sum(ex);
        const float pz = mask_after ? mk * (ex / sm) : ex / sm;
        if (lane < Tp) {
            gw_w[sb + (long)n * Tp + lane] = pz;
            Elem<T>::st(gw_w_t + sb + (long)n * Tp + lane, pz);
        }
        if (tv) {
            const float mn = fmaxf(m_run, a);
            s_run = s_run * expf(m_run - mn) + expf(a - mn);
            m_run = mn;
        }
    }
    cmax[w][lane] = m_run; csum[w][lane] = s_run;
    __syncthreads();
    if (w == 0) {
        float M = fmaxf(fmaxf(cmax[0][lane], cmax[1][lane]), fmaxf(cmax[2][lane], cmax[3][lane]));
        float S = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) S += (cmax[i][lane] == -INFINITY) ? 0.f : csum[i][lane] * expf(cmax[i][lane] - M);
        float* st = cstat + (((long)b * gridDim.x + ch) * 64 + lane) * 2;
        st[0] = M; st[1] = S;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void graph_softmax_fwd2_kernel(const float* __restrict__ A0, const float* __restrict__ pr, const float* __restrict__ mask,
                                                                const float* __restrict__ cstat, float* __restrict__ gw_v, T* __restrict__ gw_v_t,
                                                                int N, int Tn, int Tp) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y, ch = blockIdx.x;
    const bool tv = lane < Tn;
    const float prt = tv ? pr[b * Tn + lane] : 0.f;
    const float mk = tv ? mask[b * Tn + lane] : 0.f;
    const long sb = (long)b * N * Tp;
    float M = -INFINITY, S = 0.f;
    for (int c = 0; c < (int)gridDim.x; ++c) {
        const float* st = cstat + (((long)b * gridDim.x + c) * 64 + lane) * 2;
        const float m2 = st[0], s2 = st[1];
        if (m2 == -INFINITY) continue;
        const float mn = fmaxf(M, m2);
        S = S * expf(M - mn) + s2 * expf(m2 - mn);
        M = mn;
    }
    const int n1 = min(N, (ch + 1) * GS_ROWS);
    for (int n = ch * GS_ROWS + w; n < n1; n += 4) {
        float v = 0.f;
        if (tv) v = expf(prt * A0[sb + (long)n * Tp + lane] - M) / S * mk;
        if (lane < Tp) {
            gw_v[sb + (long)n * Tp + lane] = v;
            Elem<T>::st(gw_v_t + sb + (long)n * Tp + lane, v);
        }
    }
}

// backward pass 1: per-chunk column dots  sum_n gw_v * dgw_v * mask
__global__ __launch_bounds__(256) void graph_softmax_bwd1_kernel(const float* __restrict__ dgw_v, const float* __restrict__ gw_v,
                                                                const float* __restrict__ mask, float* __restrict__ cdot, int N, int Tn, int Tp) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y, ch = blockIdx.x;
    const bool tv = lane < Tn;
    const float mk = tv ? mask[b * Tn + lane] : 0.f;
    const long sb = (long)b * N * Tp;
    float cd = 0.f;
    const int n1 = min(N, (ch + 1) * GS_ROWS);
    for (int n = ch * GS_ROWS + w; n < n1; n += 4)
        if (tv) cd += gw_v[sb + (long)n * Tp + lane] * dgw_v[sb + (long)n * Tp + lane] * mk;
    red[w][lane] = cd;
    __syncthreads();
    if (w == 0) cdot[((long)b * gridDim.x + ch) * 64 + lane] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

template <typename T>
__global__ __launch_bounds__(256) void graph_softmax_bwd2_kernel(const float* __restrict__ dgw_w, const float* __restrict__ dgw_v,
                                                                const float* __restrict__ gw_w, const float* __restrict__ gw_v,
                                                                const float* __restrict__ A0, const float* __restrict__ pr, const float* __restrict__ mask,
                                                                const float* __restrict__ cdot, float* __restrict__ dA0, T* __restrict__ dA0_t,
                                                                float* __restrict__ dpr_part, int N, int Tn, int Tp) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y, ch = blockIdx.x;
    const bool tv = lane < Tn;
    const float prt = tv ? pr[b * Tn + lane] : 0.f;
    const float mk = tv ? mask[b * Tn + lane] : 0.f;
    const long sb = (long)b * N * Tp;
    float CD = 0.f;
    for (int c = 0; c < (int)gridDim.x; ++c) CD += cdot[((long)b * gridDim.x + c) * 64 + lane];
    float dp = 0.f;
    const int n1 = min(N, (ch + 1) * GS_ROWS);
    for (int n = ch * GS_ROWS + w; n < n1; n += 4) {
        const long o = sb + (long)n * Tp + lane;
        const float pw = tv ? gw_w[o] : 0.f, dw = tv ? dgw_w[o] : 0.f;
        const float rd = wave_sum(pw * dw);
        float dA = 0.f;
        if (tv) {
            dA = mk * pw * (dw - rd);
            dA += gw_v[o] * (dgw_v[o] * mk - CD);
            dp += dA * A0[o];
        }
        if (lane < Tp) {
            const float v = tv ? dA * prt : 0.f;
            dA0[o] = v;
            Elem<T>::st(dA0_t + o, v);
        }
    }
    red[w][lane] = dp;
    __syncthreads();
    if (w == 0) {                 // this chunk's partial row of d(parse_R) (folded over the chunks by reduce_parts: one writer per word)
        const float sdp = tv ? red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane] : 0.f;
        dpr_part[((long)b * gridDim.x + ch) * 64 + lane] = sdp;
    }
}

// ------------------------------------------------------------------------------------------
// gated exchange helpers
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_n_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float m = -INFINITY;
    for (int n = threadIdx.x; n < N; n += 256) m = fmaxf(m, x[(long)b * N + n]);
    m = block_max_256(m, red);
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) s += expf(x[(long)b * N + n] - m);
    s = block_sum_256(s, red);
    for (int n = threadIdx.x; n < N; n += 256) y[(long)b * N + n] = expf(x[(long)b * N + n] - m) / s;
}

__global__ __launch_bounds__(256) void softmax_n_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, int N) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float d = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) d += dy[(long)b * N + n] * y[(long)b * N + n];
    d = block_sum_256(d, red);
    for (int n = threadIdx.x; n < N; n += 256) dx[(long)b * N + n] = y[(long)b * N + n] * (dy[(long)b * N + n] - d);
}

__global__ __launch_bounds__(256) void l2norm_all_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ rstd1, int n) {
    __shared__ float red[4];
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) ss += x[i] * x[i];
    ss = block_sum_256(ss, red);
    const float rs = rsqrtf(fmaxf(ss, 1e-12f));
    for (int i = threadIdx.x; i < n; i += 256) y[i] = x[i] * rs;
    if (threadIdx.x == 0) rstd1[0] = (ss < 1e-12f) ? -rs : rs;
}

__global__ __launch_bounds__(256) void l2norm_all_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ rstd1,
                                                            float* __restrict__ dx, int n) {
    __shared__ float red[4];
    float d = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) d += dy[i] * y[i];
    d = block_sum_256(d, red);
    const float rs = rstd1[0], a = fabsf(rs);
    if (rs < 0.f) d = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) dx[i] = a * (dy[i] - y[i] * d);
}

template <typename T>
__global__ __launch_bounds__(256) void exch_combine_fwd_kernel(const T* __restrict__ feat, const T* __restrict__ r1, const T* __restrict__ r2,
                                                              const float* __restrict__ g1, const float* __restrict__ g2, int ld_g,
                                                              T* __restrict__ out, float* __restrict__ rstd, int N, int ld, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    const bool two = r2 != nullptr;        // CMPCv5_BiLSTM_model.py:343-346 has ONE gated branch
    float gv1[MB][8], gv2[MB][8];
#pragma unroll
    for (int k = 0; k < MB; ++k) {
        const int c0 = k * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) gv2[k][e] = 0.f;
        if (c0 < ld) { ld8<float>(g1 + (long)b * ld_g + c0, gv1[k]); if (two) ld8<float>(g2 + (long)b * ld_g + c0, gv2[k]); }
    }
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
        float v[MB][8];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float f[8], a[8], c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                ld8<T>(feat + base + c0, f); ld8<T>(r1 + base + c0, a); if (two) ld8<T>(r2 + base + c0, c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int cc = c0 + e;
                    v[k][e] = (cc < C) ? f[e] + a[e] * gv1[k][e] + c[e] * gv2[k][e] : 0.f;
                    ss += v[k][e] * v[k][e];
                }
            }
        }
        ss = wave_sum(ss);
        const bool clamped = ss < 1e-12f;
        const float rs = rsqrtf(fmaxf(ss, 1e-12f));
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[k][e] *= rs;
                st8<T>(out + base + c0, v[k]);
            }
        }
        if (lane == 0) rstd[(long)b * N + n] = clamped ? -rs : rs;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void exch_combine_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ out, const float* __restrict__ rstd,
                                                              const T* __restrict__ r1, const T* __restrict__ r2,
                                                              const float* __restrict__ g1, const float* __restrict__ g2, int ld_g,
                                                              T* __restrict__ dfeat, int accumulate, T* __restrict__ dp1, T* __restrict__ dp2,
                                                              float* part, int N, int ld, int C) {
    extern __shared__ float lds[];     // [WPB][4*ld]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    const bool two = r2 != nullptr;
    float a1[MB][8], a2[MB][8], gv1[MB][8], gv2[MB][8];
    float s1c[MB][8], s2c[MB][8];       // column sums of dp1 / dp2 as stored: the trans_feat convolutions' bias gradients (no second pass over the maps)
#pragma unroll
    for (int k = 0; k < MB; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { a1[k][e] = 0.f; a2[k][e] = 0.f; gv2[k][e] = 0.f; s1c[k][e] = 0.f; s2c[k][e] = 0.f; }
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld) { ld8<float>(g1 + (long)b * ld_g + c0, gv1[k]); if (two) ld8<float>(g2 + (long)b * ld_g + c0, gv2[k]); }
    }
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
        float d[MB][8], ov[MB][8];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                ld8<T>(dout + base + c0, d[k]); ld8<T>(out + base + c0, ov[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (c0 + e >= C) d[k][e] = 0.f; dot += d[k][e] * ov[k][e]; }
            }
        }
        dot = wave_sum(dot);
        const float rs = rstd[(long)b * N + n], a = fabsf(rs);
        if (rs < 0.f) dot = 0.f;
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float x1[8], x2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, o1[8], o2[8], df[8];
                ld8<T>(r1 + base + c0, x1); if (two) ld8<T>(r2 + base + c0, x2);
                if (accumulate) ld8<T>(dfeat + base + c0, df);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int cc = c0 + e;
                    const float dE = (cc < C) ? a * (d[k][e] - ov[k][e] * dot) : 0.f;
                    a1[k][e] += dE * x1[e]; a2[k][e] += dE * x2[e];
                    o1[e] = (cc < C && x1[e] > 0.f) ? dE * gv1[k][e] : 0.f;
                    o2[e] = (cc < C && x2[e] > 0.f) ? dE * gv2[k][e] : 0.f;
                    df[e] = accumulate ? df[e] + dE : dE;
                    s1c[k][e] += stored_value<T>(o1[e]); s2c[k][e] += stored_value<T>(o2[e]);
                }
                st8<T>(dp1 + base + c0, o1); if (two) st8<T>(dp2 + base + c0, o2); st8<T>(dfeat + base + c0, df);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MB; ++k) {
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                lds[(w * 4) * ld + c0 + e] = a1[k][e]; lds[(w * 4 + 1) * ld + c0 + e] = a2[k][e];
                lds[(w * 4 + 2) * ld + c0 + e] = s1c[k][e]; lds[(w * 4 + 3) * ld + c0 + e] = s2c[k][e];
            }
        }
    }
    __syncthreads();
    float* pr = part + ((long)b * gridDim.x + blockIdx.x) * 4 * ld;                 // partial rows [dg1 | dg2 | db1 | db2]
    for (int c = threadIdx.x; c < ld; c += 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sq = 0.f;
#pragma unroll
            for (int ww = 0; ww < WPB; ++ww) sq += lds[(ww * 4 + q) * ld + c];
            pr[q * ld + c] = (c < C) ? sq : 0.f;
        }
    }
}

bool map_ok(const char* what, int ld, int C, int dt) {
    if (ld <= 0 || C <= 0 || C > ld || ld % 8 || ld > MB * 512) {
        cmpc_set_error("%s: need 0 < C <= ld <= %d, ld %% 8 == 0 (got C=%d ld=%d)", what, MB * 512, C, ld);
        return false;
    }
    if (dt != DT_F32 && dt != DT_BF16 && dt != DT_F16) { cmpc_set_error("%s: bad dtype %d", what, dt); return false; }
    return true;
}

// ------------------------------------------------------------------------------------------
// lowrank_nn: C[b][m, n] (+)= alpha * sum_{k < Kv} A[b][m, k] * Bk[b][k, n] for a SHORT reduction (Kv <= 24: the word axis of the
// cross-modal graph, T = 20 -- Y = gw_w . Z, dX1 += gw_v . dZ, dX1 += scale * dA0 . PT of build_spa_graph / graph_conv,
// CMPC_model.py:359-410).  20 MACs per output element: the product is a stream of C, not a GEMM -- an MFMA tile pipeline spends
// its time in prologue / epilogue and holds a whole CU per workgroup (24 us per launch for 26 MB of output).  Here a thread owns 8
// consecutive columns whose weights stay in registers as k-pairs (Bk is k-major, so the loads are whole lines; two rows are zipped into pairs once), every wave reads a row of A
// as wave-uniform 16-B loads and v_dot2_f32_{f16,bf16} accumulate in fp32; 64 VGPR-light waves per CU stream C at the store rate
// and leave room for a neighbour stream's GEMM.
// ------------------------------------------------------------------------------------------
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ float dot2acc(uint32_t a, uint32_t b, float c);
template <> __device__ __forceinline__ float dot2acc<f16_t>(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2v, a), __builtin_bit_cast(h2v, b), c, false);
}
template <> __device__ __forceinline__ float dot2acc<bf16_t>(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, a), __builtin_bit_cast(bf2v, b), c, false);
}

template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<f16_t>(const f16_t* p, float (&v)[4]) {
    typedef _Float16 h4v __attribute__((ext_vector_type(4)));
    const h4v a = *reinterpret_cast<const h4v*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)a[e];
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u); v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<f16_t>(f16_t* p, const float (&v)[4]) {
    typedef _Float16 h4v __attribute__((ext_vector_type(4)));
    h4v a;
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = (f16_t)v[e];
    *reinterpret_cast<h4v*>(p) = a;
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    uint2 a;
    a.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); a.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = a;
}

template <typename T, int KC>      // KC = 16-B chunks of k per row (8 elements each)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8)))
void lowrank_nn_kernel(const T* __restrict__ A, int lda, long sA, const T* __restrict__ Bk, int ldb, long sB,
                       T* __restrict__ C, int ldc, long sC, int M, int N, int n_valid, int Kv,
                       float alpha, int accumulate, int rows_per_block) {
    const int tpr = N >> 2;                                   // threads per output row (a power of two <= 256: checked by the launcher)
    const int tid = threadIdx.x, cg = tid & (tpr - 1), rsub = tid / tpr, rstep = 256 / tpr;
    const int b = blockIdx.y, c0 = cg * 4;
    A += b * sA; Bk += b * sB; C += b * sC;
    uint32_t w[4][KC * 4];                                    // this thread's 4 columns, k-pairs (k, k+1) in the low / high half
#pragma unroll
    for (int p = 0; p < KC * 4; ++p) {
        uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);          // rows 2p and 2p+1 of Bk, columns c0..c0+3 (8 B, coalesced)
        if (2 * p < Kv) lo = *reinterpret_cast<const uint2*>(Bk + (long)(2 * p) * ldb + c0);
        if (2 * p + 1 < Kv) hi = *reinterpret_cast<const uint2*>(Bk + (long)(2 * p + 1) * ldb + c0);
        w[0][p] = (lo.x & 0xffffu) | (hi.x << 16); w[1][p] = (lo.x >> 16) | (hi.x & 0xffff0000u);
        w[2][p] = (lo.y & 0xffffu) | (hi.y << 16); w[3][p] = (lo.y >> 16) | (hi.y & 0xffff0000u);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c0 + e >= n_valid) {
#pragma unroll
            for (int p = 0; p < KC * 4; ++p) w[e][p] = 0u;
        }
    const int r_end = min(M, (int)(blockIdx.x + 1) * rows_per_block);
    // The row loop is a chain of dependent loads (A row, the old C when accumulating): RB rows per trip, every load of the trip
    // issued before the first dot product, so that a wave has RB rows of latency in flight instead of one.
    constexpr int RB = 4;               // (scalar A loads with 6-8 rows in flight, possible when a workgroup shares every row, measured slower)
    for (int m0 = blockIdx.x * rows_per_block + rsub; m0 < r_end; m0 += RB * rstep) {
        uint32_t a[RB][KC * 4];
        float o[RB][4];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int m = min(m0 + r * rstep, M - 1);           // rows past the range: a valid address, result dropped
            const T* ar = A + (long)m * lda;
#pragma unroll
            for (int q = 0; q < KC; ++q) {
                const uint4 v = *reinterpret_cast<const uint4*>(ar + q * 8);
                a[r][q * 4] = v.x; a[r][q * 4 + 1] = v.y; a[r][q * 4 + 2] = v.z; a[r][q * 4 + 3] = v.w;
            }
#pragma unroll
            for (int p = 0; p < KC * 4; ++p) { if (2 * p >= Kv) a[r][p] = 0u; else if (2 * p + 1 >= Kv) a[r][p] &= 0xffffu; }   // pad columns of A may hold anything
#pragma unroll
            for (int e = 0; e < 4; ++e) o[r][e] = 0.f;
            if (accumulate) ld4<T>(C + (long)m * ldc + c0, o[r]);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int m = m0 + r * rstep;
            if (m < r_end) {
                float acc[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s0 = 0.f, s1 = 0.f;                  // two chains per column
#pragma unroll
                    for (int p = 0; p < KC * 4; p += 2) { s0 = dot2acc<T>(a[r][p], w[e][p], s0); s1 = dot2acc<T>(a[r][p + 1], w[e][p + 1], s1); }
                    acc[e] = (s0 + s1) * alpha + o[r][e];
                }
                st4<T>(C + (long)m * ldc + c0, acc);
            }
        }
    }
}
}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int cmpc_lowrank_nn(int dt, const void* A, int lda, int64_t sA, const void* Bk, int ldb, int64_t sB, void* C, int ldc, int64_t sC,
                               int M, int N, int n_valid, int Kv, int batch, float alpha, int accumulate, void* stream) {
    if (!A || !Bk || !C || M <= 0 || N <= 0 || batch <= 0 || Kv <= 0) { cmpc_set_error("lowrank_nn: bad args"); return CMPC_EINVAL; }
    if (dt != DT_BF16 && dt != DT_F16) { cmpc_set_error("lowrank_nn: 16-bit storage only"); return CMPC_EINVAL; }
    const int tpr = N / 4, kc = (Kv + 7) / 8;
    if (N % 4 || tpr > 256 || (tpr & (tpr - 1)) || Kv > 24 || lda < kc * 8 || ldb < N || lda % 8 || ldb % 4 || ldc % 4 || sA % 8 || sB % 4 || sC % 4 ||
        (uintptr_t)A % 16 || ((uintptr_t)Bk | (uintptr_t)C) % 8) {
        cmpc_set_error("lowrank_nn: need N = 4 * 2^j <= 1024, Kv <= 24, lda >= 8*ceil(Kv/8) and aligned rows (N=%d Kv=%d lda=%d ldb=%d ldc=%d)", N, Kv, lda, ldb, ldc);
        return CMPC_EINVAL;
    }
    // ~4 workgroups per CU (the row loop is a chain of dependent loads); a workgroup re-reads its weights (Kv x N, L2) per ~12 rows
    const int rstep = 256 / tpr;
    const int rb = 4;                                        // rows per trip of the kernel
    int per = (M * batch + 1023) / 1024; per = per < rb * rstep ? rb * rstep : per;
    per = (per + rb * rstep - 1) / (rb * rstep) * (rb * rstep);
    const dim3 grid((M + per - 1) / per, batch);
#define LR_LAUNCH(TT, KC) hipLaunchKernelGGL((lowrank_nn_kernel<TT, KC>), grid, dim3(256), 0, ST, (const TT*)A, lda, (long)sA, (const TT*)Bk, ldb, (long)sB, \
                                             (TT*)C, ldc, (long)sC, M, N, n_valid, Kv, alpha, accumulate, per)
    if (dt == DT_F16) { if (kc <= 2) LR_LAUNCH(f16_t, 2); else LR_LAUNCH(f16_t, 3); }
    else { if (kc <= 2) LR_LAUNCH(bf16_t, 2); else LR_LAUNCH(bf16_t, 3); }
#undef LR_LAUNCH
    return cmpc_check_launch("lowrank_nn");
}

extern "C" int cmpc_mutan_fwd(int dt, void* P, const float* g, void* X1, float* rstd, int B, int N, int ld, int C, int pre_tanh, void* stream) {
    if (!map_ok("mutan_fwd", ld, C, dt)) return CMPC_EINVAL;
    const dim3 grid(rows_grid(N, 400), B);
    CMPC_DISPATCH_DT(dt, {
        if (ld == MB * 512) {
            if (pre_tanh) hipLaunchKernelGGL((mutan_fwd_kernel<T, true, true>), grid, dim3(256), 0, ST, (T*)P, g, (T*)X1, rstd, N, ld, C);
            else hipLaunchKernelGGL((mutan_fwd_kernel<T, true, false>), grid, dim3(256), 0, ST, (T*)P, g, (T*)X1, rstd, N, ld, C);
        } else {
            if (pre_tanh) hipLaunchKernelGGL((mutan_fwd_kernel<T, false, true>), grid, dim3(256), 0, ST, (T*)P, g, (T*)X1, rstd, N, ld, C);
            else hipLaunchKernelGGL((mutan_fwd_kernel<T, false, false>), grid, dim3(256), 0, ST, (T*)P, g, (T*)X1, rstd, N, ld, C);
        }
    });
    return cmpc_check_launch("mutan_fwd");
}

extern "C" int cmpc_mutan_bwd(int dt, void* Th, const float* g, const void* X1, const float* rstd, const void* dX1,
                              float* dg, int B, int N, int ld, int C, void* stream) {
    cmpc_op_scope op_("mutan_bwd");
    if (!map_ok("mutan_bwd", ld, C, dt)) return CMPC_EINVAL;
    const int gx = rows_grid(N, 64);
    float* part = (float*)cmpc_ws((size_t)B * gx * 5 * ld * sizeof(float), ST);
    if (!part) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, {
        if (ld == MB * 512) hipLaunchKernelGGL((mutan_bwd_kernel<T, true>), dim3(gx, B), dim3(256), WPB * ld * sizeof(float), ST,
                                               (T*)Th, g, (const T*)X1, rstd, (const T*)dX1, part, N, ld, C);
        else hipLaunchKernelGGL((mutan_bwd_kernel<T, false>), dim3(gx, B), dim3(256), WPB * ld * sizeof(float), ST,
                                (T*)Th, g, (const T*)X1, rstd, (const T*)dX1, part, N, ld, C);
    });
    if (cmpc_reduce_parts_f32(part, 5L * ld, B, gx, 5, ld, C, dg, 5L * ld, ld, 1, ST)) return CMPC_EHIP;
    return cmpc_check_launch("mutan_bwd");
}

extern "C" int cmpc_graph_softmax_fwd(int dt, int mask_after, const float* A0, const float* pr, const float* mask, float* gw_w, float* gw_v,
                                      void* gw_w_t, void* gw_v_t, float* scratch, int B, int N, int T_, int Tp, void* stream) {
    if (T_ <= 0 || T_ > 64 || Tp < T_ || Tp > 64) { cmpc_set_error("graph_softmax: need 0 < T <= Tp <= 64"); return CMPC_EINVAL; }
    if (!scratch) { cmpc_set_error("graph_softmax: scratch (B*ceil(N/64)*128 floats) required"); return CMPC_EINVAL; }
    const int ch = (N + GS_ROWS - 1) / GS_ROWS;
    CMPC_DISPATCH_DT(dt, {
        hipLaunchKernelGGL((graph_softmax_fwd1_kernel<T>), dim3(ch, B), dim3(256), 0, ST, A0, pr, mask, gw_w, (T*)gw_w_t, scratch, N, T_, Tp, mask_after);
        hipLaunchKernelGGL((graph_softmax_fwd2_kernel<T>), dim3(ch, B), dim3(256), 0, ST, A0, pr, mask, scratch, gw_v, (T*)gw_v_t, N, T_, Tp);
    });
    return cmpc_check_launch("graph_softmax_fwd");
}

extern "C" int cmpc_graph_softmax_bwd(int dt, const float* dgw_w, const float* dgw_v, const float* gw_w, const float* gw_v,
                                      const float* A0, const float* pr, const float* mask, float* dA0, void* dA0_t, float* dpr,
                                      float* scratch, int B, int N, int T_, int Tp, void* stream) {
    cmpc_op_scope op_("graph_softmax_bwd");
    if (T_ <= 0 || T_ > 64 || Tp < T_ || Tp > 64) { cmpc_set_error("graph_softmax: need 0 < T <= Tp <= 64"); return CMPC_EINVAL; }
    if (!scratch) { cmpc_set_error("graph_softmax: scratch (B*ceil(N/64)*128 floats) required"); return CMPC_EINVAL; }
    const int ch = (N + GS_ROWS - 1) / GS_ROWS;
    float* dpr_part = scratch + (long)B * ch * 64;        // second half of the scratch: per-chunk partial rows of d(parse_R)
    hipLaunchKernelGGL(graph_softmax_bwd1_kernel, dim3(ch, B), dim3(256), 0, ST, dgw_v, gw_v, mask, scratch, N, T_, Tp);
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((graph_softmax_bwd2_kernel<T>), dim3(ch, B), dim3(256), 0, ST, dgw_w, dgw_v, gw_w, gw_v, A0, pr, mask,
                                             scratch, dA0, (T*)dA0_t, dpr_part, N, T_, Tp));
    if (cmpc_reduce_parts_f32(dpr_part, 64, B, ch, 1, 64, T_, dpr, T_, 0, 0, ST)) return CMPC_EHIP;      // dpr[b, t] = sum over the chunks
    return cmpc_check_launch("graph_softmax_bwd");
}

extern "C" int cmpc_softmax_n_fwd(const float* logits, float* attn, int B, int N, void* stream) {
    hipLaunchKernelGGL(softmax_n_fwd_kernel, dim3(B), dim3(256), 0, ST, logits, attn, N);
    return cmpc_check_launch("softmax_n_fwd");
}
extern "C" int cmpc_softmax_n_bwd(const float* dattn, const float* attn, float* dlogits, int B, int N, void* stream) {
    hipLaunchKernelGGL(softmax_n_bwd_kernel, dim3(B), dim3(256), 0, ST, dattn, attn, dlogits, N);
    return cmpc_check_launch("softmax_n_bwd");
}
extern "C" int cmpc_l2norm_all_fwd(const float* x, float* y, float* rstd1, int n, void* stream) {
    hipLaunchKernelGGL(l2norm_all_fwd_kernel, dim3(1), dim3(256), 0, ST, x, y, rstd1, n);
    return cmpc_check_launch("l2norm_all_fwd");
}
extern "C" int cmpc_l2norm_all_bwd(const float* dy, const float* y, const float* rstd1, float* dx, int n, void* stream) {
    hipLaunchKernelGGL(l2norm_all_bwd_kernel, dim3(1), dim3(256), 0, ST, dy, y, rstd1, dx, n);
    return cmpc_check_launch("l2norm_all_bwd");
}

extern "C" int cmpc_exchange_combine_fwd(int dt, const void* feat, const void* r1, const void* r2, const float* g1, const float* g2,
                                         int ld_g, void* out, float* rstd, int B, int N, int ld, int C, void* stream) {
    if (!map_ok("exchange_combine_fwd", ld, C, dt)) return CMPC_EINVAL;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((exch_combine_fwd_kernel<T>), dim3(rows_grid(N, 400), B), dim3(256), 0, ST,
                                             (const T*)feat, (const T*)r1, (const T*)r2, g1, g2, ld_g, (T*)out, rstd, N, ld, C));
    return cmpc_check_launch("exchange_combine_fwd");
}

extern "C" int cmpc_exchange_combine_bwd(int dt, const void* dout, const void* out, const float* rstd, const void* r1, const void* r2,
                                         const float* g1, const float* g2, int ld_g, void* dfeat, int accumulate_dfeat,
                                         void* dp1, void* dp2, float* dg1, float* dg2, float* db1, float* db2, int B, int N, int ld, int C, void* stream) {
    cmpc_op_scope op_("exchange_combine_bwd");
    if (!map_ok("exchange_combine_bwd", ld, C, dt)) return CMPC_EINVAL;
    const int gx = rows_grid(N, 64);
    float* part = (float*)cmpc_ws((size_t)B * gx * 4 * ld * sizeof(float), ST);
    if (!part) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((exch_combine_bwd_kernel<T>), dim3(gx, B), dim3(256), WPB * 4 * ld * sizeof(float), ST,
                                             (const T*)dout, (const T*)out, rstd, (const T*)r1, (const T*)r2, g1, g2, ld_g,
                                             (T*)dfeat, accumulate_dfeat, (T*)dp1, (T*)dp2, part, N, ld, C));
    if (cmpc_reduce_parts_f32(part, 4L * ld, B, gx, 1, ld, C, dg1, ld_g, 0, 1, ST)) return CMPC_EHIP;
    if (r2 && cmpc_reduce_parts_f32(part + ld, 4L * ld, B, gx, 1, ld, C, dg2, ld_g, 0, 1, ST)) return CMPC_EHIP;
    // db1 / db2 [C] += column sums of dp1 / dp2 over every row of the batch (gradient targets: folded with the bucket's other folds)
    if (db1 && cmpc_reduce_parts_f32(part + 2 * ld, 4L * ld, 1, B * gx, 1, ld, C, db1, 0, 0, 1, ST)) return CMPC_EHIP;
    if (r2 && db2 && cmpc_reduce_parts_f32(part + 3 * ld, 4L * ld, 1, B * gx, 1, ld, C, db2, 0, 0, 1, ST)) return CMPC_EHIP;
    return cmpc_check_launch("exchange_combine_bwd");
}
