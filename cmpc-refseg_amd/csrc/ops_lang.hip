// Language-side kernels (reference CMPC_model.py:144-192,347-357) and parameter maintenance
// (packing fp32 masters into padded GEMM operands; TF-Adam, CMPC_model.py:450-478).
// The language tensors are tiny ([B*T, .]); they stay fp32 end to end.
#include "cmpc_common.h"
#include "../../include/cmpc.h"

namespace {

__global__ void embed_gather_kernel(const float* __restrict__ table, const int* __restrict__ words, float* __restrict__ out,
                                    int G, int ld_out, int vocab) {
    const int r = blockIdx.x;
    int wd = words[r];
    wd = wd < 0 ? 0 : (wd >= vocab ? vocab - 1 : wd);
    for (int c = threadIdx.x; c < ld_out; c += blockDim.x) out[(long)r * ld_out + c] = (c < G) ? table[(long)wd * G + c] : 0.f;
}

// d(embedding table): row `wd` += sum of dout over every occurrence of word wd.  The block of the FIRST occurrence of a word adds all
// its occurrences in index order (n_words <= a few hundred: the scan is free); the others return -- one writer per table row, no
// atomics, a sum that does not depend on scheduling.
__global__ void embed_scatter_kernel(const float* __restrict__ dout, int ld, const int* __restrict__ words, float* dtable, int n_words, int G, int vocab) {
    const int r = blockIdx.x;
    auto clampw = [&](int w) { return w < 0 ? 0 : (w >= vocab ? vocab - 1 : w); };
    const int wd = clampw(words[r]);
    for (int q = 0; q < r; ++q) if (clampw(words[q]) == wd) return;          // not the first occurrence
    for (int c = threadIdx.x; c < G; c += blockDim.x) {
        float g = 0.f;
        for (int q = r; q < n_words; ++q) if (clampw(words[q]) == wd) g += dout[(long)q * ld + c];
        dtable[(long)wd * G + c] += g;
    }
}

// tf LSTMCell: i, j, f, o = split(gates); c' = sig(f + 1) c + sig(i) tanh(j); h' = sig(o) tanh(c')
__global__ void lstm_cell_fwd_kernel(float* __restrict__ gates, const float* __restrict__ c_prev, const float* __restrict__ h_prev,
                                     const int* __restrict__ seq_len, int t, float* __restrict__ c_out, float* __restrict__ h_out,
                                     float* __restrict__ out_t, int ld_out, int ld, int R) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ld) return;
    float* g = gates + (long)b * 4 * ld;
    const bool live = t < seq_len[b];
    float cn = 0.f, hn = 0.f;
    if (c < R) {
        const float is = sigmoidf_(g[c]), jt = tanhf(g[ld + c]), fs = sigmoidf_(g[2 * ld + c] + 1.0f), os = sigmoidf_(g[3 * ld + c]);
        g[c] = is; g[ld + c] = jt; g[2 * ld + c] = fs; g[3 * ld + c] = os;
        cn = fs * c_prev[(long)b * ld + c] + is * jt;
        hn = os * tanhf(cn);
    } else { g[c] = 0.f; g[ld + c] = 0.f; g[2 * ld + c] = 0.f; g[3 * ld + c] = 0.f; }
    const long o = (long)b * ld + c;
    c_out[o] = live ? cn : c_prev[o];
    h_out[o] = live ? hn : h_prev[o];
    out_t[(long)b * ld_out + c] = live ? hn : 0.f;
}

__global__ void lstm_cell_bwd_kernel(const float* __restrict__ ga, const float* __restrict__ c_prev, const float* __restrict__ c_out,
                                     const int* __restrict__ seq_len, int t, const float* __restrict__ dout_t, int ld_dout,
                                     float* __restrict__ dh, float* __restrict__ dc, float* __restrict__ dgates, int ld, int R) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ld) return;
    const bool live = t < seq_len[b];
    const long o = (long)b * ld + c;
    float* dg = dgates + (long)b * 4 * ld;
    if (!live || c >= R) {
        dg[c] = 0.f; dg[ld + c] = 0.f; dg[2 * ld + c] = 0.f; dg[3 * ld + c] = 0.f;
        if (c >= R) { dh[o] = 0.f; dc[o] = 0.f; }
        return;   // state gradients pass through unchanged for finished sequences
    }
    const float* g = ga + (long)b * 4 * ld;
    const float is = g[c], jt = g[ld + c], fs = g[2 * ld + c], os = g[3 * ld + c];
    const float cn = c_out[o], tc = tanhf(cn);
    const float dhn = dout_t[(long)b * ld_dout + c] + dh[o];
    const float dcn = dc[o] + dhn * os * (1.f - tc * tc);
    dg[c] = dcn * jt * is * (1.f - is);
    dg[ld + c] = dcn * is * (1.f - jt * jt);
    dg[2 * ld + c] = dcn * c_prev[o] * fs * (1.f - fs);
    dg[3 * ld + c] = dhn * tc * os * (1.f - os);
    dc[o] = dcn * fs;
    dh[o] = 0.f;          // the caller adds dgates . W_h^T
}

// One step of the LSTM backward recurrence in ONE launch: dh = (dh kept for finished sequences) + dgates[t] . W_h^T, then
// the cell backward of step t-1 on that dh (what cmpc_gemm_nt(skinny) + cmpc_lstm_cell_bwd did in two launches of a
// host-paced serial chain).  A workgroup owns 4 hidden units: its 4 waves split K = 4*ld (float4 loads, all issued before
// the first use), the 8 x 4 partial sums are reduced across lanes, and threads 0..31 then run the cell backward of their
// (sample, unit).  B <= 8.
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const float* __restrict__ dg_t, const float* __restrict__ Wn, int ldw,
                                                           const float* __restrict__ ga, const float* __restrict__ c_prev, const float* __restrict__ c_out,
                                                           const int* __restrict__ seq_len, int tm1, const float* __restrict__ dout, int ld_dout,
                                                           float* __restrict__ dh, float* __restrict__ dc, float* __restrict__ dg_out, int B, int ld, int R) {
    constexpr int NC = 4, MMAX = 8;
    __shared__ float red[4][MMAX * NC];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k0 = blockIdx.x * NC, K = 4 * ld;
    float acc[MMAX][NC];
#pragma unroll
    for (int m = 0; m < MMAX; ++m)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[m][c] = 0.f;
    for (int k = (wid * 64 + lane) * 4; k < K; k += 1024) {
        float4 w[NC], a[MMAX];
#pragma unroll
        for (int c = 0; c < NC; ++c) w[c] = *reinterpret_cast<const float4*>(Wn + (long)min(k0 + c, ld - 1) * ldw + k);
#pragma unroll
        for (int m = 0; m < MMAX; ++m) a[m] = *reinterpret_cast<const float4*>(dg_t + (long)min(m, B - 1) * K + k);
#pragma unroll
        for (int m = 0; m < MMAX; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[m][c] += a[m].x * w[c].x + a[m].y * w[c].y + a[m].z * w[c].z + a[m].w * w[c].w;
    }
#pragma unroll
    for (int m = 0; m < MMAX; ++m)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float v = wave_sum(acc[m][c]);
            if (lane == 0) red[wid][m * NC + c] = v;
        }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= MMAX * NC) return;
    const int b = t / NC, c = k0 + t % NC;
    if (b >= B || c >= ld) return;
    const long o = (long)b * ld + c;
    float* dg = dg_out + (long)b * 4 * ld;
    // dh after the product of step t (columns >= R of W_h^T are zero rows of the packed operand, so they add 0)
    const float dh_in = dh[o] + (c < R ? (red[0][t] + red[1][t] + red[2][t] + red[3][t]) : 0.f);
    const bool live = tm1 < seq_len[b];
    if (!live || c >= R) {
        dg[c] = 0.f; dg[ld + c] = 0.f; dg[2 * ld + c] = 0.f; dg[3 * ld + c] = 0.f;
        if (c >= R) { dh[o] = 0.f; dc[o] = 0.f; } else dh[o] = dh_in;       // state gradients pass through for finished sequences
        return;
    }
    const float* g = ga + (long)b * 4 * ld;
    const float is = g[c], jt = g[ld + c], fs = g[2 * ld + c], os = g[3 * ld + c];
    const float cn = c_out[o], tc = tanhf(cn);
    const float dhn = dout[(long)b * ld_dout + c] + dh_in;
    const float dcn = dc[o] + dhn * os * (1.f - tc * tc);
    dg[c] = dcn * jt * is * (1.f - is);
    dg[ld + c] = dcn * is * (1.f - jt * jt);
    dg[2 * ld + c] = dcn * c_prev[o] * fs * (1.f - fs);
    dg[3 * ld + c] = dhn * tc * os * (1.f - os);
    dc[o] = dcn * fs;
    dh[o] = 0.f;
}

// ------------------------------------------------------------------------------------------
// The whole LSTM recurrence in ONE launch (dynamic_rnn over T steps, CMPC_model.py:144-164 / bidirectional_dynamic_rnn,
// CMPCv5_BiLSTM_model.py:170-174): a wave owns ONE hidden unit -- its four gate rows of W_h (forward) or its row of W_h^T (backward) stay
// in registers for all T steps, so the 16 MB of recurrent weights are read once per launch instead of once per step -- a workgroup owns 4
// units, 256 workgroups cover the (padded) 1024 units.  The previous step's h (forward) / gate gradients (backward) are staged through
// LDS; lane b < B of the wave runs the cell update of (sample b, unit) and keeps c (forward) or dh / dc (backward) in registers.  Steps
// are separated by a grid barrier: an agent-scope release / acquire counter in global memory (each workgroup adds 1 per step and waits for
// G * step).  Every spin loop has a watchdog (SEQ_SPIN_LIMIT polls, ~1 s): on expiry the workgroup raises *abort_flag, which releases
// every other waiter, and the grid drains -- a launch that cannot make progress ends instead of hanging the GPU.  Co-residency: 256
// workgroups of 256 threads with <= 64 KB of LDS fit twice per CU, so two such launches (the two directions, two processes on one GPU) can
// be resident together.  Needs B <= 8 and ld <= 1024; the engine keeps the per-step path for anything else.
// ------------------------------------------------------------------------------------------
constexpr long SEQ_SPIN_LIMIT = 1L << 24;
__device__ __forceinline__ void grid_arrive(unsigned* bar) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// returns false when the launch was aborted (watchdog of this or another workgroup)
__device__ __forceinline__ bool grid_wait(unsigned* bar, unsigned target, int* abort_flag) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        long spins = 0;
        int ok = 1;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255) == 0) {
                if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                if (spins > SEQ_SPIN_LIMIT) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    const bool ok = ok_s != 0;
    __syncthreads();
    return ok;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

constexpr int SEQ_B = 8, SEQ_KI = 4;        // B <= 8 samples, ld <= 256 * SEQ_KI
__global__ __launch_bounds__(256) void lstm_seq_fwd_kernel(const float* __restrict__ xg, const float* __restrict__ Wh, int ldw, const int* __restrict__ seq_len,
                                                          float* __restrict__ gates, float* __restrict__ h_all, float* __restrict__ c_all,
                                                          float* __restrict__ outs, unsigned* bar, int* abort_flag, int B, int T, int ld, int R) {
    extern __shared__ float hs[];                 // [SEQ_B][ld]: h of the previous step
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, u = blockIdx.x * 4 + wv;
    const bool unit_ok = u < R;
    float4 w[4][SEQ_KI];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < SEQ_KI; ++i) {
            const int k = lane * 4 + 256 * i;
            w[g][i] = (unit_ok && k < ld) ? *reinterpret_cast<const float4*>(Wh + (long)(g * ld + u) * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    float c_reg = 0.f;                            // lane b: cell state of (sample b, unit u)
    const int len = lane < B ? seq_len[lane] : 0;
    for (int t = 0; t < T; ++t) {
        if (t > 0 && !grid_wait(bar, gridDim.x * (unsigned)t, abort_flag)) return;
        const float* hp = h_all + (long)t * B * ld;
        for (int i = threadIdx.x * 4; i < B * ld; i += 1024) *reinterpret_cast<float4*>(hs + i) = *reinterpret_cast<const float4*>(hp + i);
        __syncthreads();
        float pre[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < SEQ_B; ++b) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            if (b < B) {
#pragma unroll
                for (int i = 0; i < SEQ_KI; ++i) {
                    const int k = lane * 4 + 256 * i;
                    if (k < ld) {
                        const float4 hv = *reinterpret_cast<const float4*>(hs + b * ld + k);
#pragma unroll
                        for (int g = 0; g < 4; ++g) a[g] += dot4(w[g][i], hv);
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) { const float v = wave_sum(a[g]); if (lane == b) pre[g] = v; }
        }
        if (lane < B && u < ld) {
            const int b = lane;
            const long go = ((long)t * B + b) * 4 * ld + u, so = ((long)(t + 1) * B + b) * ld + u;
            const bool live = t < len;
            float is = 0.f, jt = 0.f, fs = 0.f, os = 0.f, cn = 0.f, hn = 0.f;
            const float h_prev = hs[b * ld + u];
            if (unit_ok) {
                is = sigmoidf_(pre[0] + xg[go]); jt = tanhf(pre[1] + xg[go + ld]);
                fs = sigmoidf_(pre[2] + xg[go + 2 * ld] + 1.0f); os = sigmoidf_(pre[3] + xg[go + 3 * ld]);
                cn = fs * c_reg + is * jt;
                hn = os * tanhf(cn);
            }
            gates[go] = is; gates[go + ld] = jt; gates[go + 2 * ld] = fs; gates[go + 3 * ld] = os;
            c_reg = live ? cn : c_reg;
            c_all[so] = c_reg;
            h_all[so] = live ? hn : h_prev;
            outs[((long)b * T + t) * ld + u] = live ? hn : 0.f;
        }
        if (t + 1 < T) grid_arrive(bar); else __syncthreads();
    }
}

// dgates [T, B, 4 ld] <- the backward recurrence over `douts` [B, T, ld] (gradient of the outputs); Wn [ld rows][4 ld]: row k = W_h^T row of
// hidden unit k (the packed input-major operand)
__global__ __launch_bounds__(256) void lstm_seq_bwd_kernel(const float* __restrict__ Wn, int ldw, const float* __restrict__ gates, const float* __restrict__ c_all,
                                                          const int* __restrict__ seq_len, const float* __restrict__ douts, float* __restrict__ dgates,
                                                          unsigned* bar, int* abort_flag, int B, int T, int ld, int R) {
    extern __shared__ float dgs[];                // [4][4 ld]: the gate gradients of 4 samples of the previous step (two passes for B = 8)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, u = blockIdx.x * 4 + wv;
    const bool unit_ok = u < R;
    const int K4 = 4 * ld;
    float4 wn[4 * SEQ_KI];
#pragma unroll
    for (int i = 0; i < 4 * SEQ_KI; ++i) {
        const int k = lane * 4 + 256 * i;
        wn[i] = (unit_ok && k < K4) ? *reinterpret_cast<const float4*>(Wn + (long)u * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float dh_carry = 0.f, dc = 0.f;               // lane b: state gradients of (sample b, unit u)
    const int len = lane < B ? seq_len[lane] : 0;
    for (int t = T - 1; t >= 0; --t) {
        float mv = 0.f;
        if (t < T - 1) {
            if (!grid_wait(bar, gridDim.x * (unsigned)(T - 1 - t), abort_flag)) return;
            const float* dg = dgates + (long)(t + 1) * B * K4;
            for (int half = 0; half < (B + 3) / 4; ++half) {
                const int nb = min(4, B - half * 4);
                __syncthreads();
                for (int i = threadIdx.x * 4; i < nb * K4; i += 1024) *reinterpret_cast<float4*>(dgs + i) = *reinterpret_cast<const float4*>(dg + (long)half * 4 * K4 + i);
                __syncthreads();
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    float a = 0.f;
                    if (bb < nb) {
#pragma unroll
                        for (int i = 0; i < 4 * SEQ_KI; ++i) {
                            const int k = lane * 4 + 256 * i;
                            if (k < K4) a += dot4(wn[i], *reinterpret_cast<const float4*>(dgs + bb * K4 + k));
                        }
                    }
                    const float v = wave_sum(a);
                    if (lane == half * 4 + bb) mv = v;
                }
            }
        }
        if (lane < B && u < ld) {
            const int b = lane;
            const long go = ((long)t * B + b) * K4 + u, so = ((long)t * B + b) * ld + u;      // c_all[t] = c_prev, c_all[t + 1] = c_out
            const float dh_in = dh_carry + (unit_ok ? mv : 0.f);
            const bool live = t < len;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
            if (!unit_ok) { dh_carry = 0.f; dc = 0.f; }
            else if (!live) dh_carry = dh_in;      // state gradients pass through for finished sequences
            else {
                const float is = gates[go], jt = gates[go + ld], fs = gates[go + 2 * ld], os = gates[go + 3 * ld];
                const float cn = c_all[so + (long)B * ld], tc = tanhf(cn);
                const float dhn = douts[((long)b * T + t) * ld + u] + dh_in;
                const float dcn = dc + dhn * os * (1.f - tc * tc);
                d0 = dcn * jt * is * (1.f - is);
                d1 = dcn * is * (1.f - jt * jt);
                d2 = dcn * c_all[so] * fs * (1.f - fs);
                d3 = dhn * tc * os * (1.f - os);
                dc = dcn * fs;
                dh_carry = 0.f;
            }
            dgates[go] = d0; dgates[go + ld] = d1; dgates[go + 2 * ld] = d2; dgates[go + 3 * ld] = d3;
        }
        if (t > 0) grid_arrive(bar);
    }
}

__global__ void parse_softmax_fwd_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ mask, float* __restrict__ parse, int n, int ncls) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* l = logits + (long)r * ld;
    float m = l[0];
    for (int k = 1; k < ncls; ++k) m = fmaxf(m, l[k]);
    float e[8], s = 0.f;
    for (int k = 0; k < ncls; ++k) { e[k] = expf(l[k] - m); s += e[k]; }
    const float mk = mask[r];
    for (int k = 0; k < ncls; ++k) parse[r * ncls + k] = e[k] / s * mk;
}

__global__ void parse_softmax_bwd_kernel(const float* __restrict__ dparse, const float* __restrict__ parse, const float* __restrict__ mask,
                                         float* __restrict__ dlogits, int ld, int n, int ncls) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float mk = mask[r];
    float p[8], d[8], dot = 0.f;
    for (int k = 0; k < ncls; ++k) { p[k] = parse[r * ncls + k]; d[k] = dparse[r * ncls + k] * mk; dot += p[k] * d[k]; }
    for (int k = 0; k < ncls; ++k) dlogits[(long)r * ld + k] = (mk != 0.f) ? p[k] * (d[k] - dot) : 0.f;
    for (int k = ncls; k < ld; ++k) dlogits[(long)r * ld + k] = 0.f;
}


// valid_lang / nec_lang (CMPC_model.py:166-192): v[b] = l2norm(sum_t w[b,t] wf[b,t,:]), w = sum of the first ncls parser classes.
// One 1024-thread block per sample: thread = 1..4 columns, the T word weights staged in LDS, the T loads of a column independent.
__global__ __launch_bounds__(1024) void lang_pool_fwd_kernel(const float* __restrict__ parse, const float* __restrict__ wf, float* __restrict__ v,
                                                            float* __restrict__ rstd, int T, int ld, int R, int ncls, int lo, int ps) {
    __shared__ float wgt[64];
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < T) {
        float w = 0.f;
        for (int k = 0; k < ncls; ++k) w += parse[((long)b * T + tid) * ps + lo + k];
        wgt[tid] = w;
    }
    __syncthreads();
    float mine[2] = {0.f, 0.f};                 // ld <= 2048
    float ss = 0.f;
    for (int i = 0, c = tid; c < ld; c += 1024, ++i) {
        float acc = 0.f;
        if (c < R) {
            const float* col = wf + (long)b * T * ld + c;
#pragma unroll 4
            for (int t = 0; t < T; ++t) acc += wgt[t] * col[(long)t * ld];
        }
        mine[i] = acc;
        ss += acc * acc;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) ss += red[w];
    const float rs = rsqrtf(fmaxf(ss, 1e-12f));
    for (int i = 0, c = tid; c < ld; c += 1024, ++i) v[(long)b * ld + c] = mine[i] * rs;
    if (tid == 0) rstd[b] = (ss < 1e-12f) ? -rs : rs;
}

// backward: one block per (sample, word).  draw = |rstd| (dv - v (v.dv)) is recomputed per word (R values);
// dparse[b,t,k<ncls] += draw . wf[b,t,:];  dwf[b,t,:] += w[b,t] draw.   (both accumulate: the two pools share the buffers)
__global__ __launch_bounds__(256) void lang_pool_bwd_kernel(const float* __restrict__ dv, const float* __restrict__ v, const float* __restrict__ rstd,
                                                           const float* __restrict__ parse, const float* __restrict__ wf,
                                                           float* __restrict__ dparse, float* __restrict__ dwf, int T, int ld, int R, int ncls, int lo, int ps) {
    __shared__ float red[4];
    const int b = blockIdx.x, t = blockIdx.y;
    float dot = 0.f;
    for (int c = threadIdx.x; c < R; c += 256) dot += dv[(long)b * ld + c] * v[(long)b * ld + c];
    dot = block_sum_256(dot, red);
    const float rs = rstd[b], a = fabsf(rs);
    if (rs < 0.f) dot = 0.f;
    float wgt = 0.f;
    for (int k = 0; k < ncls; ++k) wgt += parse[((long)b * T + t) * ps + lo + k];
    float dw = 0.f;
    for (int c = threadIdx.x; c < R; c += 256) {
        const float draw = a * (dv[(long)b * ld + c] - v[(long)b * ld + c] * dot);
        const long o = ((long)b * T + t) * ld + c;
        dw += draw * wf[o];
        dwf[o] += wgt * draw;
    }
    dw = block_sum_256(dw, red);
    if (threadIdx.x == 0)
        for (int k = 0; k < ncls; ++k) dparse[((long)b * T + t) * ps + lo + k] += dw;
}

// ------------------------------------------------------------------------------------------
// pack: padded operand (k, n) <- master[src_row(k), src_col(n)] through a 32x32 LDS tile so that
// both the read (n contiguous) and the transposed write (k contiguous) are coalesced.
// ------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ int seg_map(int x, int ns, const int (&src)[N], const int (&len)[N], const int (&dst)[N]) {
    // fully unrolled with static indices: a runtime-indexed loop made hipcc keep the whole descriptor in scratch
    // (168 B per lane written and re-read by every block: more traffic than the tile itself)
    int r = -1;
#pragma unroll
    for (int s = 0; s < N; ++s) if (s < ns && x >= dst[s] && x < dst[s] + len[s]) r = src[s] + (x - dst[s]);
    return r;
}

__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ master, char* __restrict__ arena, const cmpc_pack_desc* __restrict__ descs,
                                                  const int* __restrict__ tile_prefix, int ndesc, int tile_begin,
                                                  const int* __restrict__ tile_desc) {
    // tile = 64 (k) x 128 (n): float4 reads along n (every segment boundary is a multiple of 4); writes along n
    // (natural, 8 / 16 B per lane) or along k through LDS (transposed: 8 k-values = one 16-B bf16 store per lane, a
    // full 128-B line per destination row).  LDS rows are 129 floats: both the scalar stores (element order rotated
    // per 8-lane group) and the column reads (lane -> 4 k-groups x 8 rows per half wave) are bank-conflict free.
    __shared__ float tile[64][129];
    int lo = 0, hi = ndesc;
    const int gt = (int)blockIdx.x + tile_begin;
    // tile -> descriptor: one table lookup instead of a binary search (~9 dependent global loads)
    if (tile_desc) lo = tile_desc[gt];
    else while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_prefix[mid] <= gt) lo = mid; else hi = mid; }
    const cmpc_pack_desc d = descs[lo];
    const int t = gt - tile_prefix[lo];
    const int Kp = d.transpose ? d.cols : d.rows, Np = d.transpose ? d.rows : d.cols;
    const int tn = (Np + 127) / 128;
    const int k0 = (t / tn) * 64, n0 = (t % tn) * 128;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    char* dst = arena + d.dst_off;
    const int n = n0 + 4 * tx;
    const int sc = (n < Np) ? seg_map(n, d.nns, d.ns_src, d.ns_len, d.ns_dst) : -1;
    const int rot = tx >> 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int kk = ty + 8 * i, k = k0 + kk;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < Kp && sc >= 0) {
            const int sr = seg_map(k, d.nks, d.ks_src, d.ks_len, d.ks_dst);
            if (sr >= 0) {
                const float* src = master + d.src_off + (long)sr * d.ld_src + sc;
                if ((d.ld_src & 3) == 0) v = *reinterpret_cast<const float4*>(src);
                else {          // master rows that are not 16-B aligned (the 5-class parser of the video model, [500, 5]): scalar reads clipped to the row
                    const int lim = d.ld_src - sc;
                    v.x = src[0]; v.y = lim > 1 ? src[1] : 0.f; v.z = lim > 2 ? src[2] : 0.f; v.w = lim > 3 ? src[3] : 0.f;
                }
            }
        }
        if (!d.transpose) {
            if (k < Kp && n < Np) {
                const long o = (long)k * d.ld_dst + n;
                if (d.dst_dt == DT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + o) = v;
                else if (d.dst_dt == DT_F16) {
                    f16_t* q = reinterpret_cast<f16_t*>(dst) + o;
                    typedef __attribute__((ext_vector_type(4))) _Float16 h4v;
                    *reinterpret_cast<h4v*>(q) = h4v{(f16_t)v.x, (f16_t)v.y, (f16_t)v.z, (f16_t)v.w};
                } else {
                    uint2 u;
                    u.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
                    u.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(dst) + o) = u;
                }
            }
        } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cc = (c + rot) & 3;          // 8-lane groups start at different elements: 32 distinct banks
                tile[kk][4 * tx + cc] = e[cc];
            }
        }
    }
    if (!d.transpose) return;
    __syncthreads();
    // transposed write: row n of the destination holds k contiguous; a lane owns 8 consecutive k of one row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kq = ((lane & 3) + 4 * (lane >> 5)) * 8, nsub = (lane >> 2) & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = nsub + 8 * wv + 32 * j, nn = n0 + nl, k = k0 + kq;
        if (nn < Np && k < Kp) {
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = tile[kq + e][nl];
            const long o = (long)nn * d.ld_dst + k;
            // rows of a destination operand are padded to multiples of 64 elements, so k..k+7 stay inside the row
            if (d.dst_dt == DT_F32) st8<float>(reinterpret_cast<float*>(dst) + o, a);
            else if (d.dst_dt == DT_F16) st8<f16_t>(reinterpret_cast<f16_t*>(dst) + o, a);
            else st8<bf16_t>(reinterpret_cast<bf16_t*>(dst) + o, a);
        }
    }
}

// tf.train.AdamOptimizer:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr_t m / (sqrt(v) + eps)
// Overflow guard (f16 storage saturates to inf): an element whose gradient is not finite is left untouched -- parameter and both
// moments -- and counted in *nonfinite, so that one overflowing batch cannot poison the Adam state; the count is surfaced by the handle.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  const cmpc_adam_seg* __restrict__ segs, float lr_t, float b1, float b2, float eps, float gscale,
                                                  int* __restrict__ nonfinite) {
    const cmpc_adam_seg s = segs[blockIdx.x];
    for (int i = threadIdx.x; i < s.count; i += 256) {
        const long o = s.off + i;
        const float pv = p[o];
        const float gin = g[o];
        if (!(fabsf(gin) <= 3.4028235e38f)) { if (nonfinite) atomicAdd(nonfinite, 1); continue; }      // inf or nan
        const float gr = gin * gscale * s.gmult + s.wd * pv;
        const float mn = b1 * m[o] + (1.f - b1) * gr;
        const float vn = b2 * v[o] + (1.f - b2) * gr * gr;
        m[o] = mn; v[o] = vn;
        p[o] = pv - lr_t * mn / (sqrtf(vn) + eps);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int cmpc_embed_gather(const float* table, const int* words, float* out, int n_words, int G, int ld_out, int vocab, void* stream) {
    if (n_words <= 0) return CMPC_OK;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(n_words), dim3(128), 0, ST, table, words, out, G, ld_out, vocab);
    return cmpc_check_launch("embed_gather");
}
extern "C" int cmpc_embed_scatter(const float* dout, int ld, const int* words, float* dtable, int n_words, int G, int vocab, void* stream) {
    if (n_words <= 0) return CMPC_OK;
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(n_words), dim3(128), 0, ST, dout, ld, words, dtable, n_words, G, vocab);
    return cmpc_check_launch("embed_scatter");
}
extern "C" int cmpc_lstm_cell_fwd(float* gates, const float* c_prev, const float* h_prev, const int* seq_len, int t,
                                  float* c_out, float* h_out, float* out_t, int ld_out, int B, int ld, int R, void* stream) {
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((ld + 255) / 256, B), dim3(256), 0, ST, gates, c_prev, h_prev, seq_len, t, c_out, h_out, out_t, ld_out, ld, R);
    return cmpc_check_launch("lstm_cell_fwd");
}
extern "C" int cmpc_lstm_cell_bwd(const float* gates_act, const float* c_prev, const float* c_out, const int* seq_len, int t,
                                  const float* dout_t, int ld_dout, float* dh, float* dc, float* dgates,
                                  int B, int ld, int R, void* stream) {
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((ld + 255) / 256, B), dim3(256), 0, ST, gates_act, c_prev, c_out, seq_len, t, dout_t, ld_dout, dh, dc, dgates, ld, R);
    return cmpc_check_launch("lstm_cell_bwd");
}
extern "C" int cmpc_lstm_bwd_step(const float* dgates_t, const float* Wn, int ldw, const float* gates_act_tm1, const float* c_prev, const float* c_out,
                                  const int* seq_len, int tm1, const float* dout_tm1, int ld_dout, float* dh, float* dc, float* dgates_tm1,
                                  int B, int ld, int R, void* stream) {
    if (B < 1 || B > 8 || ld % 4 || ldw % 4 || !dgates_t || !Wn) { cmpc_set_error("lstm_bwd_step: need 1 <= B <= 8 and 16-B aligned rows"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(lstm_bwd_step_kernel, dim3((ld + 3) / 4), dim3(256), 0, ST, dgates_t, Wn, ldw, gates_act_tm1, c_prev, c_out, seq_len, tm1,
                       dout_tm1, ld_dout, dh, dc, dgates_tm1, B, ld, R);
    return cmpc_check_launch("lstm_bwd_step");
}
// One launch for the T steps of a direction (see lstm_seq_fwd_kernel).  xg [T, B, 4 ld]: the x-side pre-activations incl. bias; Wh: the gate rows
// of W_h, [4 ld rows][ldw], k contiguous; h_all / c_all [(T + 1), B, ld] with slice 0 = the initial state; sync: 8 bytes of device memory the
// caller zeroed (barrier counter, abort flag).  Returns CMPC_EINVAL when B > 8 or ld > 1024 (use the per-step entry points then).
extern "C" int cmpc_lstm_seq_fwd(const float* xg, const float* Wh, int ldw, const int* seq_len, float* gates, float* h_all, float* c_all, float* outs,
                                 void* sync, int B, int T, int ld, int R, void* stream) {
    if (!xg || !Wh || !seq_len || !gates || !h_all || !c_all || !outs || !sync || B < 1 || B > SEQ_B || T < 1 || ld < 64 || ld % 64 || ld > 256 * SEQ_KI || R > ld || ldw % 4) {
        cmpc_set_error("lstm_seq_fwd: need 1 <= B <= %d, ld a multiple of 64 and <= %d, 16-B aligned rows", SEQ_B, 256 * SEQ_KI); return CMPC_EINVAL;
    }
    hipLaunchKernelGGL(lstm_seq_fwd_kernel, dim3(ld / 4), dim3(256), (size_t)SEQ_B * ld * sizeof(float), ST, xg, Wh, ldw, seq_len, gates, h_all, c_all, outs,
                       (unsigned*)sync, (int*)sync + 1, B, T, ld, R);
    return cmpc_check_launch("lstm_seq_fwd");
}
extern "C" int cmpc_lstm_seq_bwd(const float* Wn, int ldw, const float* gates, const float* c_all, const int* seq_len, const float* douts, float* dgates,
                                 void* sync, int B, int T, int ld, int R, void* stream) {
    if (!Wn || !gates || !c_all || !seq_len || !douts || !dgates || !sync || B < 1 || B > SEQ_B || T < 1 || ld < 64 || ld % 64 || ld > 256 * SEQ_KI || R > ld || ldw % 4) {
        cmpc_set_error("lstm_seq_bwd: need 1 <= B <= %d, ld a multiple of 64 and <= %d, 16-B aligned rows", SEQ_B, 256 * SEQ_KI); return CMPC_EINVAL;
    }
    const size_t lds = (size_t)4 * 4 * ld * sizeof(float);
    static const bool attr = ((void)hipFuncSetAttribute((const void*)lstm_seq_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4 * 256 * SEQ_KI * (int)sizeof(float)), true);
    (void)attr;
    hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3(ld / 4), dim3(256), lds, ST, Wn, ldw, gates, c_all, seq_len, douts, dgates, (unsigned*)sync, (int*)sync + 1, B, T, ld, R);
    return cmpc_check_launch("lstm_seq_bwd");
}
extern "C" int cmpc_parse_softmax_fwd(const float* logits, int ld, const float* mask, float* parse, int n, int ncls, void* stream) {
    if (ncls < 2 || ncls > 8 || ncls > ld) { cmpc_set_error("parse_softmax: 2 <= ncls <= 8"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(parse_softmax_fwd_kernel, dim3((n + 63) / 64), dim3(64), 0, ST, logits, ld, mask, parse, n, ncls);
    return cmpc_check_launch("parse_softmax_fwd");
}
extern "C" int cmpc_parse_softmax_bwd(const float* dparse, const float* parse, const float* mask, float* dlogits, int ld, int n, int ncls, void* stream) {
    if (ncls < 2 || ncls > 8 || ncls > ld) { cmpc_set_error("parse_softmax: 2 <= ncls <= 8"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(parse_softmax_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, ST, dparse, parse, mask, dlogits, ld, n, ncls);
    return cmpc_check_launch("parse_softmax_bwd");
}
extern "C" int cmpc_lang_pool_fwd(const float* parse, const float* wf, float* v, float* rstd, int B, int T, int ld, int R, int ncls, int cls_lo, int pstride, void* stream) {
    if (ncls < 1 || cls_lo < 0 || cls_lo + ncls > pstride || pstride > 8) { cmpc_set_error("lang_pool: classes [cls_lo, cls_lo + ncls) must lie inside a parse row of pstride <= 8"); return CMPC_EINVAL; }
    if (T > 64 || ld > 2048) { cmpc_set_error("lang_pool: T <= 64 and ld <= 2048"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(lang_pool_fwd_kernel, dim3(B), dim3(1024), 0, ST, parse, wf, v, rstd, T, ld, R, ncls, cls_lo, pstride);
    return cmpc_check_launch("lang_pool_fwd");
}
extern "C" int cmpc_lang_pool_bwd(const float* dv, const float* v, const float* rstd, const float* parse, const float* wf,
                                  float* dparse, float* dwf, int B, int T, int ld, int R, int ncls, int cls_lo, int pstride, void* stream) {
    if (ncls < 1 || cls_lo < 0 || cls_lo + ncls > pstride || pstride > 8) { cmpc_set_error("lang_pool: classes [cls_lo, cls_lo + ncls) must lie inside a parse row of pstride <= 8"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(lang_pool_bwd_kernel, dim3(B, T), dim3(256), 0, ST, dv, v, rstd, parse, wf, dparse, dwf, T, ld, R, ncls, cls_lo, pstride);
    return cmpc_check_launch("lang_pool_bwd");
}

extern "C" int cmpc_pack_weights_range(const float* master, void* arena, const cmpc_pack_desc* descs_dev, const int* tile_prefix_dev,
                                       const int* tile_desc_dev, int ndesc, int tile_begin, int tile_end, void* stream) {
    if (ndesc <= 0 || tile_end <= tile_begin) return CMPC_OK;
    if (tile_begin < 0) { cmpc_set_error("pack_weights: bad tile range"); return CMPC_EINVAL; }
    hipLaunchKernelGGL(pack_kernel, dim3(tile_end - tile_begin), dim3(256), 0, ST, master, (char*)arena, descs_dev, tile_prefix_dev, ndesc, tile_begin, tile_desc_dev);
    return cmpc_check_launch("pack_weights");
}

extern "C" int cmpc_pack_weights(const float* master, void* arena, const cmpc_pack_desc* descs_dev, const int* tile_prefix_dev,
                                 int ndesc, int total_tiles, void* stream) {
    return cmpc_pack_weights_range(master, arena, descs_dev, tile_prefix_dev, nullptr, ndesc, 0, total_tiles, stream);
}

extern "C" int cmpc_adam_step(float* params, const float* grads, float* m, float* v, const cmpc_adam_seg* segs_dev, int nseg,
                              float lr_t, float beta1, float beta2, float eps, float gscale, int* nonfinite, void* stream) {
    if (nseg <= 0) return CMPC_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(nseg), dim3(256), 0, ST, params, grads, m, v, segs_dev, lr_t, beta1, beta2, eps, gscale, nonfinite);
    return cmpc_check_launch("adam_step");
}
