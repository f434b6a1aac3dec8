// ConvLSTMCell.call (reference util/cell.py:36-79) around its 1x1-convolution GEMM, one step.
// Gate pre-activations Yg [R, 4*ld] (blocks j,i,f,o), peepholes [N, M] fp32, five whole-sample
// LayerNorms (j,i,f,o,c).  Wave per spatial row; fp32 math; float64 per-sample sums.
#include "cmpc_common.h"
#include "../../include/cmpc.h"

namespace {

constexpr int WPB = 4;
constexpr int MB = 2;     // ld <= 1024
__host__ inline int rows_grid(int N, int cap) { int g = (N + WPB - 1) / WPB; return g < 1 ? 1 : (g > cap ? cap : g); }

struct Sum2 { double a, b; };

// this workgroup's pair of statistic `sidx` into the stat block (cmpc_common.h: the consumers add the pairs)
__device__ __forceinline__ void flush_sums(double s1, double s2, double* stats, int sidx, double (*red)[WPB], int slot) {
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[2 * slot][w] = s1; red[2 * slot + 1][w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0)
        stat_store(stats, sidx, blockIdx.x, gridDim.x, red[2 * slot][0] + red[2 * slot][1] + red[2 * slot][2] + red[2 * slot][3],
                   red[2 * slot + 1][0] + red[2 * slot + 1][1] + red[2 * slot + 1][2] + red[2 * slot + 1][3]);
    __syncthreads();
}

template <int NB>
__device__ __forceinline__ void colflush(const float (&acc)[NB][8], float* out, int ld, int C, float* lds) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld) {
#pragma unroll
            for (int e = 0; e < 8; ++e) lds[w * ld + c0 + e] = acc[k][e];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ld; c += 256)      // this workgroup's partial row (plain stores)
        out[c] = (c < C) ? lds[c] + lds[ld + c] + lds[2 * ld + c] + lds[3 * ld + c] : 0.f;
}

// ---- forward A: peepholes on i,f (in place) + stats j,i,f -----------------------------------
template <typename T, int NB>
__global__ __launch_bounds__(256) void clstm_a_kernel(T* __restrict__ Yg, const T* __restrict__ c_prev, const float* __restrict__ W_ci,
                                                     const float* __restrict__ W_cf, double* sums, int B, int N, int ld, int M) {
    __shared__ double red[6][WPB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    double sj1 = 0, sj2 = 0, si1 = 0, si2 = 0, sf1 = 0, sf2 = 0;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long r = (long)b * N + n;
        T* y = Yg + r * 4 * ld;
        float aj1 = 0, aj2 = 0, ai1 = 0, ai2 = 0, af1 = 0, af2 = 0;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int c0 = kb * 512 + lane * 8;
            if (c0 >= ld) continue;
            float j[8], i[8], f[8];
            ld8<T>(y + c0, j); ld8<T>(y + ld + c0, i); ld8<T>(y + 2 * ld + c0, f);
            if (c_prev) {
                float c[8], wi[8], wf[8];
                ld8<T>(c_prev + r * ld + c0, c);
                if (c0 < M) { ld8<float>(W_ci + (long)n * M + c0, wi); ld8<float>(W_cf + (long)n * M + c0, wf); }   // M % 8 == 0 or tail masked below
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int m = c0 + e;
                    if (m < M) { i[e] += wi[e] * c[e]; f[e] += wf[e] * c[e]; }
                }
                st8<T>(y + ld + c0, i); st8<T>(y + 2 * ld + c0, f);
                // statistics must see the values as stored (bf16 rounding included)
                ld8<T>(y + ld + c0, i); ld8<T>(y + 2 * ld + c0, f);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (c0 + e < M) {
                    aj1 += j[e]; aj2 += j[e] * j[e]; ai1 += i[e]; ai2 += i[e] * i[e]; af1 += f[e]; af2 += f[e] * f[e];
                }
            }
        }
        sj1 += aj1; sj2 += aj2; si1 += ai1; si2 += ai2; sf1 += af1; sf2 += af2;
    }
    flush_sums(sj1, sj2, sums, 0 * B + b, red, 0);
    flush_sums(si1, si2, sums, 1 * B + b, red, 1);
    flush_sums(sf1, sf2, sums, 2 * B + b, red, 2);
}


// this lane's slice of a per-channel fp32 vector (LayerNorm gamma / beta): loaded once per wave
template <int NB>
__device__ __forceinline__ void load_vec(const float* v, float (&out)[NB][8], int ld) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld) ld8<float>(v + c0, out[k]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) out[k][e] = 0.f;
        }
    }
}

struct LnP { const float* beta[5]; const float* gamma[5]; };
struct LnG { float* dbeta[5]; float* dgamma[5]; };

// ---- forward B ---------------------------------------------------------------------------------
template <typename T, int NB>
__global__ __launch_bounds__(256) void clstm_b_kernel(T* __restrict__ Yg, const T* __restrict__ c_prev, const float* __restrict__ W_co,
                                                     LnP ln, const double* sums, double* stats_out, T* __restrict__ c_pre, int B, int N, int ld, int M) {
    __shared__ double red[4][WPB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    const double cnt = (double)N * M;
    float mj, rj, mi, ri, mf, rf;
    ln_stats(sums, 0 * B + b, cnt, mj, rj);
    ln_stats(sums, 1 * B + b, cnt, mi, ri);
    ln_stats(sums, 2 * B + b, cnt, mf, rf);
    float gj[NB][8], bj[NB][8], gi[NB][8], bi[NB][8], gf[NB][8], bf_[NB][8];
    load_vec<NB>(ln.gamma[0], gj, ld); load_vec<NB>(ln.beta[0], bj, ld); load_vec<NB>(ln.gamma[1], gi, ld);
    load_vec<NB>(ln.beta[1], bi, ld); load_vec<NB>(ln.gamma[2], gf, ld); load_vec<NB>(ln.beta[2], bf_, ld);
    double so1 = 0, so2 = 0, sc1 = 0, sc2 = 0;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long r = (long)b * N + n;
        T* y = Yg + r * 4 * ld;
        float ao1 = 0, ao2 = 0, ac1 = 0, ac2 = 0;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int c0 = kb * 512 + lane * 8;
            if (c0 >= ld) continue;
            float j[8], i[8], f[8], o[8], c[8], cp[8];
            ld8<T>(y + c0, j); ld8<T>(y + ld + c0, i); ld8<T>(y + 2 * ld + c0, f); ld8<T>(y + 3 * ld + c0, o);
            if (c_prev) ld8<T>(c_prev + r * ld + c0, c);
            float wo[8];
            if (c0 < M) ld8<float>(W_co + (long)n * M + c0, wo);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = c0 + e;
                if (m < M) {
                    const float jn = (j[e] - mj) * rj * gj[kb][e] + bj[kb][e];
                    const float in = (i[e] - mi) * ri * gi[kb][e] + bi[kb][e];
                    const float fn = (f[e] - mf) * rf * gf[kb][e] + bf_[kb][e];
                    const float fg = sigmoid_fast(fn + 1.0f), ig = sigmoid_fast(in), jt = cmpc_tanh(jn);
                    cp[e] = (c_prev ? c[e] * fg : 0.f) + ig * jt;
                    o[e] += wo[e] * cp[e];
                } else { cp[e] = 0.f; o[e] = 0.f; }
            }
            st8<T>(c_pre + r * ld + c0, cp); st8<T>(y + 3 * ld + c0, o);
            // the statistics are those of the STORED (bf16-rounded) values, which is what clstm_c normalises; rounding
            // in registers instead of reading the two vectors back saves a dependent memory round trip per row
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c0 + e < M) {
                    const float ov = stored_value<T>(o[e]), cv = stored_value<T>(cp[e]);
                    ao1 += ov; ao2 += ov * ov; ac1 += cv; ac2 += cv * cv;
                }
        }
        so1 += ao1; so2 += ao2; sc1 += ac1; sc2 += ac2;
    }
    flush_sums(so1, so2, stats_out, 3 * B + b, red, 0);
    flush_sums(sc1, sc2, stats_out, 4 * B + b, red, 1);
}

// ---- forward C ---------------------------------------------------------------------------------
template <typename T, int NB>
__global__ __launch_bounds__(256) void clstm_c_kernel(const T* __restrict__ Yg, const T* __restrict__ c_pre, LnP ln, const double* sums,
                                                     T* __restrict__ c_new, T* __restrict__ h, int B, int N, int ld, int M) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    const double cnt = (double)N * M;
    float mo, ro, mc, rc;
    ln_stats(sums, 3 * B + b, cnt, mo, ro);
    ln_stats(sums, 4 * B + b, cnt, mc, rc);
    float go[NB][8], bo[NB][8], gc[NB][8], bc[NB][8];
    load_vec<NB>(ln.gamma[3], go, ld); load_vec<NB>(ln.beta[3], bo, ld); load_vec<NB>(ln.gamma[4], gc, ld); load_vec<NB>(ln.beta[4], bc, ld);
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long r = (long)b * N + n;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int c0 = kb * 512 + lane * 8;
            if (c0 >= ld) continue;
            float o[8], cp[8], cn[8], hh[8];
            ld8<T>(Yg + r * 4 * ld + 3 * ld + c0, o); ld8<T>(c_pre + r * ld + c0, cp);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = c0 + e;
                if (m < M) {
                    const float on = (o[e] - mo) * ro * go[kb][e] + bo[kb][e];
                    cn[e] = (cp[e] - mc) * rc * gc[kb][e] + bc[kb][e];
                    hh[e] = sigmoid_fast(on) * cmpc_tanh(cn[e]);
                } else { cn[e] = 0.f; hh[e] = 0.f; }
            }
            st8<T>(c_new + r * ld + c0, cn); st8<T>(h + r * ld + c0, hh);
        }
    }
}

// ---- backward pass 1: through h = sig(LN o) * tanh(LN c) up to the LN inputs' dxhat ------------
template <typename T, int NB>
__global__ __launch_bounds__(256) void clstm_bwd1_kernel(const T* __restrict__ dh, const T* __restrict__ dc_new, const T* __restrict__ Yg,
                                                        const T* __restrict__ c_pre, LnP ln, const double* sums, T* __restrict__ dYg,
                                                        T* __restrict__ scr, float* part, double* stats_out, int B, int N, int ld, int M) {
    extern __shared__ float lds[];
    __shared__ double red[4][WPB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    const double cnt = (double)N * M;
    float mo, ro, mc, rc;
    ln_stats(sums, 3 * B + b, cnt, mo, ro);
    ln_stats(sums, 4 * B + b, cnt, mc, rc);
    float go[NB][8], bo[NB][8], gc[NB][8], bc[NB][8];
    load_vec<NB>(ln.gamma[3], go, ld); load_vec<NB>(ln.beta[3], bo, ld); load_vec<NB>(ln.gamma[4], gc, ld); load_vec<NB>(ln.beta[4], bc, ld);
    float ago[NB][8], abo[NB][8], agc[NB][8], abc[NB][8];
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ago[k][e] = abo[k][e] = agc[k][e] = abc[k][e] = 0.f; }
    double so1 = 0, so2 = 0, sc1 = 0, sc2 = 0;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long r = (long)b * N + n;
        float ao1 = 0, ao2 = 0, ac1 = 0, ac2 = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float g[8], dcn[8], o[8], cp[8], xo[8], xc[8];
                ld8<T>(dh + r * ld + c0, g); ld8<T>(Yg + r * 4 * ld + 3 * ld + c0, o); ld8<T>(c_pre + r * ld + c0, cp);
                if (dc_new) ld8<T>(dc_new + r * ld + c0, dcn);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int m = c0 + e;
                    if (m < M) {
                        const float xho = (o[e] - mo) * ro, xhc = (cp[e] - mc) * rc;
                        const float on = xho * go[k][e] + bo[k][e];
                        const float cn = xhc * gc[k][e] + bc[k][e];
                        const float so = sigmoid_fast(on), tc = cmpc_tanh(cn);
                        const float don = g[e] * tc * so * (1.f - so);
                        const float dcc = g[e] * so * (1.f - tc * tc) + (dc_new ? dcn[e] : 0.f);
                        ago[k][e] += don * xho; abo[k][e] += don; agc[k][e] += dcc * xhc; abc[k][e] += dcc;
                        xo[e] = don * go[k][e]; xc[e] = dcc * gc[k][e];
                        ao1 += xo[e]; ao2 += xo[e] * xho; ac1 += xc[e]; ac2 += xc[e] * xhc;
                    } else { xo[e] = 0.f; xc[e] = 0.f; }
                }
                st8<T>(dYg + r * 4 * ld + 3 * ld + c0, xo); st8<T>(scr + r * ld + c0, xc);
            }
        }
        so1 += ao1; so2 += ao2; sc1 += ac1; sc2 += ac2;
    }
    flush_sums(so1, so2, stats_out, 3 * B + b, red, 0);
    flush_sums(sc1, sc2, stats_out, 4 * B + b, red, 1);
    const long wg = (long)b * gridDim.x + blockIdx.x;
    float* pr = part + wg * 4 * ld;                 // [dgamma_o, dbeta_o, dgamma_c, dbeta_c]
    colflush(ago, pr, ld, M, lds); colflush(abo, pr + ld, ld, M, lds);
    colflush(agc, pr + 2 * ld, ld, M, lds); colflush(abc, pr + 3 * ld, ld, M, lds);
}

// ---- backward pass 2: finish LN(o), LN(c); through the cell update to dxhat of j,i,f -----------
template <typename T, int NB>
__global__ __launch_bounds__(256) void clstm_bwd2_kernel(const T* __restrict__ Yg, const T* __restrict__ c_prev, const T* __restrict__ c_pre,
                                                        const float* __restrict__ W_co, LnP ln, const double* sums, const double* bsums_in,
                                                        T* __restrict__ dYg, const T* __restrict__ scr, T* __restrict__ dc_prev,
                                                        float* part, double* stats_out, int B, int N, int ld, int M) {
    extern __shared__ float lds[];
    __shared__ double red[6][WPB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    const double cnt = (double)N * M;
    float mj, rj, mi, ri, mf, rf, mo, ro, mc, rc;
    ln_stats(sums, 0 * B + b, cnt, mj, rj);
    ln_stats(sums, 1 * B + b, cnt, mi, ri);
    ln_stats(sums, 2 * B + b, cnt, mf, rf);
    ln_stats(sums, 3 * B + b, cnt, mo, ro);
    ln_stats(sums, 4 * B + b, cnt, mc, rc);
    float o_m1, o_m2, c_m1, c_m2;
    stat_means(bsums_in, 3 * B + b, cnt, o_m1, o_m2);
    stat_means(bsums_in, 4 * B + b, cnt, c_m1, c_m2);
    float gq[3][NB][8], bq[3][NB][8];
#pragma unroll
    for (int q = 0; q < 3; ++q) { load_vec<NB>(ln.gamma[q], gq[q], ld); load_vec<NB>(ln.beta[q], bq[q], ld); }
    float ag[3][NB][8], ab[3][NB][8];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { ag[q][k][e] = 0.f; ab[q][k][e] = 0.f; }
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long r = (long)b * N + n;
        float a[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float j[8], i[8], f[8], o[8], cp[8], c[8], dxo[8], dxc[8], oj[8], oi[8], of[8], dcp[8];
                const T* y = Yg + r * 4 * ld;
                ld8<T>(y + c0, j); ld8<T>(y + ld + c0, i); ld8<T>(y + 2 * ld + c0, f); ld8<T>(y + 3 * ld + c0, o);
                ld8<T>(c_pre + r * ld + c0, cp);
                if (c_prev) ld8<T>(c_prev + r * ld + c0, c);
                ld8<T>(dYg + r * 4 * ld + 3 * ld + c0, dxo); ld8<T>(scr + r * ld + c0, dxc);
                float wo[8];
                if (c0 < M) ld8<float>(W_co + (long)n * M + c0, wo);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int m = c0 + e;
                    if (m < M) {
                        const float xho = (o[e] - mo) * ro, xhc = (cp[e] - mc) * rc;
                        const float dop = ro * (dxo[e] - o_m1 - xho * o_m2);
                        float dc = rc * (dxc[e] - c_m1 - xhc * c_m2);
                        const float wco = wo[e];
                        dc += dop * wco;
                        dxo[e] = dop;
                        const float xhj = (j[e] - mj) * rj, xhi = (i[e] - mi) * ri, xhf = (f[e] - mf) * rf;
                        const float jn = xhj * gq[0][k][e] + bq[0][k][e];
                        const float in = xhi * gq[1][k][e] + bq[1][k][e];
                        const float fn = xhf * gq[2][k][e] + bq[2][k][e];
                        const float fg = sigmoid_fast(fn + 1.0f), ig = sigmoid_fast(in), jt = cmpc_tanh(jn);
                        const float cpv = c_prev ? c[e] : 0.f;
                        const float dfn = dc * cpv * fg * (1.f - fg);
                        const float din = dc * jt * ig * (1.f - ig);
                        const float djn = dc * ig * (1.f - jt * jt);
                        dcp[e] = dc * fg;
                        ag[0][k][e] += djn * xhj; ab[0][k][e] += djn;
                        ag[1][k][e] += din * xhi; ab[1][k][e] += din;
                        ag[2][k][e] += dfn * xhf; ab[2][k][e] += dfn;
                        oj[e] = djn * gq[0][k][e]; oi[e] = din * gq[1][k][e]; of[e] = dfn * gq[2][k][e];
                        a[0] += oj[e]; a[1] += oj[e] * xhj; a[2] += oi[e]; a[3] += oi[e] * xhi; a[4] += of[e]; a[5] += of[e] * xhf;
                    } else { dxo[e] = 0.f; oj[e] = oi[e] = of[e] = 0.f; dcp[e] = 0.f; }
                }
                T* dy = dYg + r * 4 * ld;
                st8<T>(dy + c0, oj); st8<T>(dy + ld + c0, oi); st8<T>(dy + 2 * ld + c0, of); st8<T>(dy + 3 * ld + c0, dxo);
                if (c_prev) st8<T>(dc_prev + r * ld + c0, dcp);
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) s[q] += a[q];
    }
    flush_sums(s[0], s[1], stats_out, 0 * B + b, red, 0);
    flush_sums(s[2], s[3], stats_out, 1 * B + b, red, 1);
    flush_sums(s[4], s[5], stats_out, 2 * B + b, red, 2);
    const long wg = (long)b * gridDim.x + blockIdx.x;
    float* pr = part + wg * 6 * ld;                 // [dgamma_q, dbeta_q] for q = j, i, f
#pragma unroll
    for (int q = 0; q < 3; ++q) { colflush(ag[q], pr + (2 * q) * ld, ld, M, lds); colflush(ab[q], pr + (2 * q + 1) * ld, ld, M, lds); }
}

// ---- backward pass 3: finish LN(j), LN(i), LN(f); peepholes on i,f ----------------------------
template <typename T, int NB>
__global__ __launch_bounds__(256) void clstm_bwd3_kernel(const T* __restrict__ Yg, const T* __restrict__ c_prev, const float* __restrict__ W_ci,
                                                        const float* __restrict__ W_cf, const double* sums, const double* bsums,
                                                        T* __restrict__ dYg, T* __restrict__ dc_prev, float* dW_ci, float* dW_cf,
                                                        int B, int N, int ld, int M) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    const double cnt = (double)N * M;
    float mean[3], rstd[3], m1[3], m2[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        ln_stats(sums, q * B + b, cnt, mean[q], rstd[q]);
        stat_means(bsums, q * B + b, cnt, m1[q], m2[q]);
    }
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long r = (long)b * N + n;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int c0 = kb * 512 + lane * 8;
            if (c0 >= ld) continue;
            float x[3][8], d[3][8], c[8], dcp[8];
#pragma unroll
            for (int q = 0; q < 3; ++q) { ld8<T>(Yg + r * 4 * ld + q * ld + c0, x[q]); ld8<T>(dYg + r * 4 * ld + q * ld + c0, d[q]); }
            float wi[8], wf[8];
            if (c_prev) { ld8<T>(c_prev + r * ld + c0, c); ld8<T>(dc_prev + r * ld + c0, dcp); if (c0 < M) { ld8<float>(W_ci + (long)n * M + c0, wi); ld8<float>(W_cf + (long)n * M + c0, wf); } }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = c0 + e;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float xh = (x[q][e] - mean[q]) * rstd[q];
                    d[q][e] = (m < M) ? rstd[q] * (d[q][e] - m1[q] - xh * m2[q]) : 0.f;
                }
                if (c_prev && m < M) dcp[e] += d[1][e] * wi[e] + d[2][e] * wf[e];
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) st8<T>(dYg + r * 4 * ld + q * ld + c0, d[q]);
            if (c_prev) st8<T>(dc_prev + r * ld + c0, dcp);
        }
    }
}


// ---- peephole gradients: dW_c*[n,m] += sum_b dgate[b,n,m] * c[b,n,m]; one wave per spatial row n,
// the batch loop inside, so no atomics (the three time steps are sequential launches) --------------
template <typename T, int NB>
__global__ __launch_bounds__(256) void clstm_peephole_grad_kernel(const T* __restrict__ dYg, const T* __restrict__ c_prev, const T* __restrict__ c_pre,
                                                                 float* __restrict__ dW_ci, float* __restrict__ dW_cf, float* __restrict__ dW_co,
                                                                 int B, int N, int ld, int M) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int c0 = kb * 512 + lane * 8;
            if (c0 >= ld) continue;
            float ai[8], af[8], ao[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { ai[e] = af[e] = ao[e] = 0.f; }
            for (int b = 0; b < B; ++b) {
                const long r = (long)b * N + n;
                float di[8], df[8], dop[8], cp[8], cq[8];
                ld8<T>(dYg + r * 4 * ld + 3 * ld + c0, dop); ld8<T>(c_pre + r * ld + c0, cq);
#pragma unroll
                for (int e = 0; e < 8; ++e) ao[e] += dop[e] * cq[e];
                if (c_prev) {
                    ld8<T>(dYg + r * 4 * ld + ld + c0, di); ld8<T>(dYg + r * 4 * ld + 2 * ld + c0, df); ld8<T>(c_prev + r * ld + c0, cp);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ai[e] += di[e] * cp[e]; af[e] += df[e] * cp[e]; }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = c0 + e;
                if (m < M) {
                    dW_co[(long)n * M + m] += ao[e];
                    if (c_prev) { dW_ci[(long)n * M + m] += ai[e]; dW_cf[(long)n * M + m] += af[e]; }
                }
            }
        }
    }
}

bool ok(const char* what, int dt, int ld, int M) {
    if (ld <= 0 || M <= 0 || M > ld || ld % 8 || ld > MB * 512) { cmpc_set_error("%s: need 0 < M <= ld <= %d, ld %% 8 == 0", what, MB * 512); return false; }
    if (dt != DT_F32 && dt != DT_BF16 && dt != DT_F16) { cmpc_set_error("%s: bad dtype", what); return false; }
    return true;
}
LnP to_lnp(const cmpc_convlstm_ln* l) { LnP p; for (int i = 0; i < 5; ++i) { p.beta[i] = l->beta[i]; p.gamma[i] = l->gamma[i]; } return p; }
LnG to_lng(const cmpc_convlstm_dln* l) { LnG p; for (int i = 0; i < 5; ++i) { p.dbeta[i] = l->dbeta[i]; p.dgamma[i] = l->dgamma[i]; } return p; }

}  // namespace

#define ST ((hipStream_t)stream)
// kernels are instantiated for 1 or 2 column blocks of 512
#define CLSTM_NB(ld, ...) do { if ((ld) <= 512) { constexpr int NBX = 1; __VA_ARGS__; } else { constexpr int NBX = 2; __VA_ARGS__; } } while (0)

// workspace: fp32 column partials [nwg][ncol][ld] followed by fp64 pair partials [npair][nwg][2]
static int clstm_ws(long nwg, int ncol, int npair, int ld, float** part, double** dpart, hipStream_t st) {
    const size_t fbytes = ((size_t)nwg * ncol * ld * sizeof(float) + 15) / 16 * 16;
    char* ws = (char*)cmpc_ws(fbytes + (size_t)npair * nwg * 2 * sizeof(double), st);
    if (!ws) return CMPC_EHIP;
    *part = (float*)ws; *dpart = (double*)(ws + fbytes);
    return CMPC_OK;
}

extern "C" int cmpc_convlstm_a(int dt, void* Yg, const void* c_prev, const float* W_ci, const float* W_cf, double* sums,
                               int B, int N, int ld, int M, void* stream) {
    if (!ok("convlstm_a", dt, ld, M)) return CMPC_EINVAL;
    const int gx = rows_grid(N, 100);                       // <= STAT_PARTS workgroups per sample
    CLSTM_NB(ld, CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((clstm_a_kernel<T, NBX>), dim3(gx, B), dim3(256), 0, ST,
                                             (T*)Yg, (const T*)c_prev, W_ci, W_cf, sums, B, N, ld, M)));
    return cmpc_check_launch("convlstm_a");
}

extern "C" int cmpc_convlstm_b(int dt, void* Yg, const void* c_prev, const float* W_co, const cmpc_convlstm_ln* ln,
                               double* sums, void* c_pre, int B, int N, int ld, int M, void* stream) {
    if (!ok("convlstm_b", dt, ld, M)) return CMPC_EINVAL;
    const int gx = rows_grid(N, 100);
    CLSTM_NB(ld, CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((clstm_b_kernel<T, NBX>), dim3(gx, B), dim3(256), 0, ST,
                                             (T*)Yg, (const T*)c_prev, W_co, to_lnp(ln), sums, sums, (T*)c_pre, B, N, ld, M)));
    return cmpc_check_launch("convlstm_b");
}

extern "C" int cmpc_convlstm_c(int dt, const void* Yg, const void* c_pre, const cmpc_convlstm_ln* ln, const double* sums,
                               void* c_new, void* h, int B, int N, int ld, int M, void* stream) {
    if (!ok("convlstm_c", dt, ld, M)) return CMPC_EINVAL;
    CLSTM_NB(ld, CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((clstm_c_kernel<T, NBX>), dim3(rows_grid(N, 200), B), dim3(256), 0, ST,
                                             (const T*)Yg, (const T*)c_pre, to_lnp(ln), sums, (T*)c_new, (T*)h, B, N, ld, M)));
    return cmpc_check_launch("convlstm_c");
}

extern "C" int cmpc_convlstm_bwd(int dt, const void* dh, const void* dc_new, const void* Yg, const void* c_prev, const void* c_pre,
                                 const float* W_ci, const float* W_cf, const float* W_co,
                                 const cmpc_convlstm_ln* ln, const double* sums, void* dYg, void* dc_prev,
                                 float* dW_ci, float* dW_cf, float* dW_co, const cmpc_convlstm_dln* dln, void* scr, double* bsums,
                                 int B, int N, int ld, int M, void* stream) {
    cmpc_op_scope op_("convlstm_bwd");
    if (!ok("convlstm_bwd", dt, ld, M)) return CMPC_EINVAL;
    const size_t lds = WPB * ld * sizeof(float);
    const int gx = rows_grid(N, 64);
    const long nwg = (long)B * gx;
    float* part; double* dpart;
    if (clstm_ws(nwg, 6, 0, ld, &part, &dpart, ST)) return CMPC_EHIP;
    // pass 1: LN(o), LN(c) dxhat + their statistics / parameter gradients
    CLSTM_NB(ld, CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((clstm_bwd1_kernel<T, NBX>), dim3(gx, B), dim3(256), lds, ST,
                           (const T*)dh, (const T*)dc_new, (const T*)Yg, (const T*)c_pre, to_lnp(ln), sums, (T*)dYg, (T*)scr, part, bsums, B, N, ld, M)));
    for (int q = 0; q < 2; ++q) {
        if (cmpc_reduce_parts_f32(part + (2 * q) * ld, 4 * ld, 1, (int)nwg, 1, ld, M, dln->dgamma[3 + q], 0, 0, 1, ST)) return CMPC_EHIP;
        if (cmpc_reduce_parts_f32(part + (2 * q + 1) * ld, 4 * ld, 1, (int)nwg, 1, ld, M, dln->dbeta[3 + q], 0, 0, 1, ST)) return CMPC_EHIP;
    }
    // pass 2: finish o, c; cell update; LN(j,i,f) dxhat + statistics.  Fresh partial rows: pass 1's may still be waiting for
    // a deferred fold (cmpc_fold_begin); without a collector this returns the same per-stream block, as before.
    if (clstm_ws(nwg, 6, 0, ld, &part, &dpart, ST)) return CMPC_EHIP;
    CLSTM_NB(ld, CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((clstm_bwd2_kernel<T, NBX>), dim3(gx, B), dim3(256), lds, ST,
                           (const T*)Yg, (const T*)c_prev, (const T*)c_pre, W_co, to_lnp(ln), sums, bsums, (T*)dYg, (const T*)scr, (T*)dc_prev,
                           part, bsums, B, N, ld, M)));
    for (int q = 0; q < 3; ++q) {
        if (cmpc_reduce_parts_f32(part + (2 * q) * ld, 6 * ld, 1, (int)nwg, 1, ld, M, dln->dgamma[q], 0, 0, 1, ST)) return CMPC_EHIP;
        if (cmpc_reduce_parts_f32(part + (2 * q + 1) * ld, 6 * ld, 1, (int)nwg, 1, ld, M, dln->dbeta[q], 0, 0, 1, ST)) return CMPC_EHIP;
    }
    CLSTM_NB(ld, CMPC_DISPATCH_DT(dt, {
        hipLaunchKernelGGL((clstm_bwd3_kernel<T, NBX>), dim3(rows_grid(N, 200), B), dim3(256), 0, ST,
                           (const T*)Yg, (const T*)c_prev, W_ci, W_cf, sums, bsums, (T*)dYg, (T*)dc_prev, dW_ci, dW_cf, B, N, ld, M);
        hipLaunchKernelGGL((clstm_peephole_grad_kernel<T, NBX>), dim3(rows_grid(N, 512)), dim3(256), 0, ST,
                           (const T*)dYg, (const T*)c_prev, (const T*)c_pre, dW_ci, dW_cf, dW_co, B, N, ld, M);
    }));
    return cmpc_check_launch("convlstm_bwd");
}
