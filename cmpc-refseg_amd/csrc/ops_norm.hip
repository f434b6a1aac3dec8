// Row / whole-sample normalisation kernels and small map utilities (HBM-bound, wave-per-row).
// Layout: map [B*N, ld] of T (bf16 or f32), C valid channels, 16-byte vector accesses, fp32 math,
// wave64 shuffle reductions; per-sample sums as float64 stat blocks (one pair per workgroup, summed by the consumers:
// cmpc_common.h), per-column sums as per-workgroup partial rows folded in a fixed order (reduce_parts_*): no atomics.
#include "cmpc_common.h"
#include "../../include/cmpc.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <vector>

// ------------------------------------------------------------------------------------------
// error plumbing shared by all translation units
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void cmpc_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
long g_cmpc_launches = 0;          // host-side count of checked launches (cmpc_launch_count)
// Launch trace (cmpc_launch_trace): while enabled, every checked launch is followed by a hipEvent on the ONE stream the caller named, so
// that on an in-order stream the interval between consecutive events is that launch's duration (what rocprofv3 --kernel-trace reports,
// without a profiler).  Durations are summed per launch name (string literals: the pointers are stable).
static bool g_trace_on = false;
static hipStream_t g_trace_stream = nullptr;
static std::vector<std::pair<const char*, hipEvent_t>> g_trace;
int cmpc_check_launch(const char* what) {
    ++g_cmpc_launches;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cmpc_set_error("%s: %s", what, hipGetErrorString(e)); return CMPC_EHIP; }
    if (g_trace_on) {
        hipEvent_t ev = nullptr;
        if (hipEventCreate(&ev) == hipSuccess) { (void)hipEventRecord(ev, g_trace_stream); g_trace.push_back({what, ev}); }
    }
    return CMPC_OK;
}
static thread_local const char* t_op = nullptr;
static thread_local bool t_op_marked = false;
cmpc_op_scope::cmpc_op_scope(const char* name) : prev(t_op), prev_marked(t_op_marked) { t_op = name; t_op_marked = false; }
cmpc_op_scope::~cmpc_op_scope() { t_op = prev; t_op_marked = prev_marked; }
void cmpc_trace_producer() {
    if (!g_trace_on || !t_op || t_op_marked) return;
    t_op_marked = true;
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) == hipSuccess) { (void)hipEventRecord(ev, g_trace_stream); g_trace.push_back({t_op, ev}); }
}
extern "C" int cmpc_launch_trace(int enable, void* stream) {
    for (auto& t : g_trace) (void)hipEventDestroy(t.second);
    g_trace.clear();
    g_trace_on = false;
    if (enable) {
        g_trace_stream = (hipStream_t)stream;
        g_trace_on = true;
        cmpc_check_launch("(start)");          // the reference point of the first interval
        --g_cmpc_launches;
    }
    return CMPC_OK;
}
// per-name totals of the trace recorded so far: index 0 .. n-1 (CMPC_EINVAL past the end); synchronises the traced stream's events
extern "C" int cmpc_launch_trace_read(int index, const char** name, double* ms, int64_t* launches) {
    static std::vector<std::pair<const char*, std::pair<double, int64_t>>> agg;
    if (index == 0) {
        agg.clear();
        g_trace_on = false;
        for (size_t i = 1; i < g_trace.size(); ++i) {
            if (hipEventSynchronize(g_trace[i].second) != hipSuccess) { cmpc_set_error("launch_trace_read: event"); return CMPC_EHIP; }
            float d = 0.f;
            if (hipEventElapsedTime(&d, g_trace[i - 1].second, g_trace[i].second) != hipSuccess) d = 0.f;
            bool found = false;
            for (auto& a : agg) if (a.first == g_trace[i].first || strcmp(a.first, g_trace[i].first) == 0) { a.second.first += d; a.second.second += 1; found = true; break; }
            if (!found) agg.push_back({g_trace[i].first, {(double)d, 1}});
        }
    }
    if (index < 0 || index >= (int)agg.size()) return CMPC_EINVAL;
    if (name) *name = agg[index].first;
    if (ms) *ms = agg[index].second.first;
    if (launches) *launches = agg[index].second.second;
    return CMPC_OK;
}
extern "C" const char* cmpc_last_error(void) { return g_err; }

struct WsSlot { hipStream_t st; void* p; size_t bytes; bool used; };
static WsSlot g_ws[256];
static thread_local cmpc_fold_ctx* t_fold = nullptr;
void cmpc_fold_begin(cmpc_fold_ctx* ctx) { t_fold = ctx; if (ctx) { ctx->off = 0; ctx->n = 0; } }
// a stream that is about to be destroyed gives its slot (and block) back: a process that creates handle after handle would otherwise run out
// of the 64 slots (3 lane streams per handle)
static std::mutex g_ws_mu;
void cmpc_ws_release(hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (auto& w : g_ws) if (w.used && w.st == st) { if (w.p) (void)hipFree(w.p); w.p = nullptr; w.bytes = 0; w.used = false; w.st = nullptr; }
}
void* cmpc_ws(size_t bytes, hipStream_t st) {
    if (t_fold) {                       // collecting deferred folds: partial rows must survive until cmpc_fold_flush
        const size_t need = (bytes + 255) / 256 * 256;
        if (t_fold->off + need <= t_fold->cap) { void* p = t_fold->arena + t_fold->off; t_fold->off += need; return p; }
    }
    std::lock_guard<std::mutex> lock(g_ws_mu);          // the slot table is shared by every handle / host thread of the process
    WsSlot* slot = nullptr;
    for (auto& w : g_ws) if (w.used && w.st == st) { slot = &w; break; }
    if (!slot) {
        for (auto& w : g_ws) if (!w.used) { slot = &w; w.used = true; w.st = st; w.p = nullptr; w.bytes = 0; break; }
        if (!slot) { cmpc_set_error("workspace: more than 256 streams in use"); return nullptr; }
    }
    if (bytes > slot->bytes) {
        // growth only happens while shapes are first seen (never inside a steady-state step).  It cannot happen
        // inside a stream capture (hipMalloc is illegal there), and a block that is outgrown is retired, not
        // freed: a HIP graph captured earlier may still hold its address.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            cmpc_set_error("workspace of %zu bytes needed during stream capture: run one eager pass on the same streams first", bytes);
            return nullptr;
        }
        const size_t want = bytes < ((size_t)32 << 20) ? ((size_t)32 << 20) : bytes * 2;
        if (hipMalloc(&slot->p, want) != hipSuccess) { slot->p = nullptr; slot->bytes = 0; cmpc_set_error("workspace allocation of %zu bytes failed", want); return nullptr; }
        slot->bytes = want;
    }
    return slot->p;
}

// Column-parallel fold of per-workgroup partial rows: a block owns 64 columns (lane = column, so every part row is read as one 256-B
// line) of one outer index and walks ALL its partial rows, its 4 waves taking rows w, w+4, ... and meeting in LDS in a fixed order:
// exactly one writer per output element, no atomics, a sum that does not depend on scheduling.
__global__ __launch_bounds__(256) void reduce_parts_f32_kernel(const float* __restrict__ part, long part_stride, int ninner, int nseg, int seg_ld, int seg_C,
                                                              float* __restrict__ out, long ld_out, long out_seg, int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int o = blockIdx.y;
    const bool ok = j < nseg * seg_ld;
    const int seg = ok ? j / seg_ld : 0, c = ok ? j - seg * seg_ld : 0;
    const bool valid = ok && c < seg_C;
    const float* p = part + (long)o * ninner * part_stride + j;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (valid) {
        int i = w;
        for (; i + 12 < ninner; i += 16) {
            s0 += p[(long)i * part_stride]; s1 += p[(long)(i + 4) * part_stride];
            s2 += p[(long)(i + 8) * part_stride]; s3 += p[(long)(i + 12) * part_stride];
        }
        for (; i < ninner; i += 4) s0 += p[(long)i * part_stride];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && valid) {
        const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        float* dst = out + (long)o * ld_out + (long)seg * out_seg + c;
        *dst = accumulate ? *dst + s : s;
    }
}
// every recorded fold in one launch: a block owns 64 columns of one (fold, outer index) and walks ALL its partial rows in a fixed
// order (no split of a column between blocks), so a fold's result does not depend on scheduling; folds that share a target are
// chained behind one head and summed by the same block
__global__ __launch_bounds__(256) void reduce_parts_grouped_kernel(const cmpc_fold_desc* __restrict__ table, int ndesc) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int lo = 0, hi = ndesc;
    const int blk = blockIdx.x;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (table[mid].blk_begin <= blk) lo = mid; else hi = mid; }
    const cmpc_fold_desc d = table[lo];
    const int cols = d.nseg * d.seg_ld, cb = (cols + 63) / 64;
    const int local = blk - d.blk_begin, o = local / cb, j = (local % cb) * 64 + lane;
    const bool ok = j < cols;
    const int seg = ok ? j / d.seg_ld : 0, c = ok ? j - seg * d.seg_ld : 0;
    const bool valid = ok && c < d.seg_C;
    float tot = 0.f;
    // descriptors chained behind d fold into the same target (e.g. the three ConvLSTM steps' LayerNorm gradients): summed here in
    // list order, one writer per element
    for (int q = 0; q <= d.chain; ++q) {
        const cmpc_fold_desc& e = table[lo + q];
        const float* p = e.part + (long)o * e.ninner * e.part_stride + j;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (valid) {
            int i = w;
            for (; i + 12 < e.ninner; i += 16) {
                s0 += p[(long)i * e.part_stride]; s1 += p[(long)(i + 4) * e.part_stride];
                s2 += p[(long)(i + 8) * e.part_stride]; s3 += p[(long)(i + 12) * e.part_stride];
            }
            for (; i < e.ninner; i += 4) s0 += p[(long)i * e.part_stride];
        }
        __syncthreads();
        red[w][lane] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        tot += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    }
    if (w == 0 && valid) d.out[(long)o * d.ld_out + (long)seg * d.out_seg + c] += tot;
}
#define FOLD_UPLOAD 48
struct FoldUploadArgs { int n, base; cmpc_fold_desc d[FOLD_UPLOAD]; };
__global__ void fold_desc_upload_kernel(const FoldUploadArgs ua, cmpc_fold_desc* table) {
    constexpr int W = (int)(sizeof(cmpc_fold_desc) / sizeof(int));
    const int* src = reinterpret_cast<const int*>(&ua.d[0]);
    int* dst = reinterpret_cast<int*>(table + ua.base);
    for (int i = threadIdx.x; i < ua.n * W; i += blockDim.x) dst[i] = src[i];
}
static int fold_launch(cmpc_fold_desc* descs, int n, cmpc_fold_desc* table_dev, cmpc_fold_desc* shadow, int* shadow_n, hipStream_t st) {
    if (n == 0) return CMPC_OK;
    // folds with the same target and shape become one chain (head first, `chain` = number of followers): a single block sums them
    std::vector<cmpc_fold_desc> sorted;
    std::vector<char> used(n, 0);
    sorted.reserve(n);
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (used[i]) continue;
        const size_t head = sorted.size();
        for (int k = i; k < n; ++k) {
            const cmpc_fold_desc &a = descs[i], &b = descs[k];
            if (used[k] || a.out != b.out || a.nouter != b.nouter || a.nseg != b.nseg || a.seg_ld != b.seg_ld || a.seg_C != b.seg_C ||
                a.ld_out != b.ld_out || a.out_seg != b.out_seg) continue;
            used[k] = 1;
            sorted.push_back(b);
        }
        const int nb = ((descs[i].nseg * descs[i].seg_ld + 63) / 64) * descs[i].nouter;
        for (size_t k = head; k < sorted.size(); ++k) { sorted[k].chain = 0; sorted[k].blk_begin = blocks + nb; }   // followers are never searched for
        sorted[head].chain = (int)(sorted.size() - head) - 1;
        sorted[head].blk_begin = blocks;
        blocks += nb;
    }
    memcpy(descs, sorted.data(), sizeof(cmpc_fold_desc) * n);
    if (*shadow_n != n || memcmp(shadow, descs, sizeof(cmpc_fold_desc) * n) != 0) {
        for (int c0 = 0; c0 < n; c0 += FOLD_UPLOAD) {     // through the kernel-argument segment: asynchronous, no host buffer lifetime issue
            FoldUploadArgs ua;
            ua.n = n - c0 < FOLD_UPLOAD ? n - c0 : FOLD_UPLOAD;
            ua.base = c0;
            for (int i = 0; i < ua.n; ++i) ua.d[i] = descs[c0 + i];
            hipLaunchKernelGGL(fold_desc_upload_kernel, dim3(1), dim3(256), 0, st, ua, table_dev);
        }
        memcpy(shadow, descs, sizeof(cmpc_fold_desc) * n);
        *shadow_n = n;
    }
    hipLaunchKernelGGL(reduce_parts_grouped_kernel, dim3(blocks), dim3(256), 0, st, table_dev, n);
    return cmpc_check_launch("reduce_parts_grouped");
}
int cmpc_fold_flush(cmpc_fold_ctx* ctx, hipStream_t st) {
    t_fold = nullptr;
    if (!ctx || ctx->n == 0) return CMPC_OK;
    const int n = ctx->n;
    ctx->n = 0;
    return fold_launch(ctx->descs, n, ctx->table_dev, ctx->shadow, &ctx->shadow_n, st);
}
int cmpc_fold_flush_ranges(cmpc_fold_ctx* ctx, const float* const* lo, const float* const* hi, int nr, cmpc_fold_desc* table_dev,
                           cmpc_fold_desc* shadow, int* shadow_n, hipStream_t st) {
    if (!ctx || ctx->n == 0) return CMPC_OK;
    // stable partition: selected descriptors to the back (they are launched from there), the rest keep their order in front
    std::vector<cmpc_fold_desc> sel, rest;
    for (int i = 0; i < ctx->n; ++i) {
        bool in = false;
        for (int r = 0; r < nr && !in; ++r) in = ctx->descs[i].out >= lo[r] && ctx->descs[i].out < hi[r];
        (in ? sel : rest).push_back(ctx->descs[i]);
    }
    for (size_t i = 0; i < rest.size(); ++i) ctx->descs[i] = rest[i];
    ctx->n = (int)rest.size();
    return fold_launch(sel.data(), (int)sel.size(), table_dev, shadow, shadow_n, st);
}
int cmpc_reduce_parts_f32(const float* part, long part_stride, int nouter, int ninner, int nseg, int seg_ld, int seg_C,
                          float* out, long ld_out, long out_seg, int accumulate, hipStream_t st) {
    cmpc_trace_producer();        // closes the producing stage kernel's trace interval under ITS name (launch trace only)
    const int cols = nseg * seg_ld;
    if (accumulate && t_fold && out >= t_fold->lo && out < t_fold->hi && t_fold->n < t_fold->table_cap &&
        (const char*)part >= t_fold->arena && (const char*)part < t_fold->arena + t_fold->cap) {
        // the target is read by nothing before the optimizer and the partial rows are not recycled: fold later, with all the others
        t_fold->descs[t_fold->n++] = cmpc_fold_desc{part, part_stride, nouter, ninner, nseg, seg_ld, seg_C, 0, out, ld_out, out_seg, 0, 0};
        return CMPC_OK;
    }
    hipLaunchKernelGGL(reduce_parts_f32_kernel, dim3((cols + 63) / 64, nouter), dim3(256), 0, st, part, part_stride, ninner, nseg, seg_ld, seg_C,
                       out, ld_out, out_seg, accumulate);
    return cmpc_check_launch("reduce_parts_f32");
}
extern "C" int cmpc_abi_version(void) { return CMPC_ABI_VERSION; }

namespace {

constexpr int MAXBLK = 4;      // ld <= 2048 for kernels that keep per-column registers
constexpr int WPB = 4;         // waves per 256-thread block

__host__ inline int rows_grid(int N) { int g = (N + WPB - 1) / WPB; return g < 1 ? 1 : (g > 400 ? 400 : g); }

// per-lane column accumulators (lane owns columns blk*512 + lane*8 + e) -> this workgroup's partial row
// out[0..ld) (plain stores; a reduce_parts launch folds the rows: no contended atomics)
__device__ __forceinline__ void colsum_flush(const float (&acc)[MAXBLK][8], float* out, int ld, int C, float* lds) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k) {
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld) {
#pragma unroll
            for (int e = 0; e < 8; ++e) lds[w * ld + c0 + e] = acc[k][e];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ld; c += 256)
        out[c] = (c < C) ? lds[c] + lds[ld + c] + lds[2 * ld + c] + lds[3 * ld + c] : 0.f;
}

// ------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        Elem<TD>::st(d + i, Elem<TS>::ld(s + i));
}

template <typename T>
__global__ void axpy_kernel(const T* __restrict__ x, T* __restrict__ y, float a, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float xv[8], yv[8];
        ld8<T>(x + i * 8, xv); ld8<T>(y + i * 8, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) yv[e] += a * xv[e];
        st8<T>(y + i * 8, yv);
    }
}

// dpre = dy * act'(y); db[c] += sum; dsb[b][c] += per-sample sums
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dpre,
                                                     int act, int N, int stride, int ld, int C, float* part) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    float acc[MAXBLK][8];
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * stride;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float g[8], yv[8];
                ld8<T>(dy + base + c0, g);
                if (act != ACT_NONE) {
                    ld8<T>(y + base + c0, yv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] *= act_grad_from_out(yv[e], act);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (c0 + e >= C) g[e] = 0.f; acc[k][e] += g[e]; }
                if (dpre) st8<T>(dpre + base + c0, g);
            }
        }
    }
    if (part) colsum_flush(acc, part + ((long)b * gridDim.x + blockIdx.x) * ld, ld, C, lds);
}

// out[b,c] += sum_n w[b,n] * x[b,n,c]
template <typename T>
__global__ __launch_bounds__(256) void wcolsum_kernel(const T* __restrict__ x, const float* __restrict__ wgt, float* part,
                                                     int N, int ld, int C, float scale) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    float acc[MAXBLK][8];
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const float wv = wgt[(long)b * N + n] * scale;
        const long base = ((long)b * N + n) * ld;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float xv[8];
                ld8<T>(x + base + c0, xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[k][e] += wv * xv[e];
            }
        }
    }
    colsum_flush(acc, part + ((long)b * gridDim.x + blockIdx.x) * ld, ld, C, lds);
}

// s[b,n] = scale * x[b,n,:] . v[b,:]
template <typename T>
__global__ __launch_bounds__(256) void rowdot1_kernel(const T* __restrict__ x, const float* __restrict__ v, int ld_v, float* __restrict__ s,
                                                     int N, int ld, int C, float scale) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
        float acc = 0.f;
        for (int c0 = lane * 8; c0 < ld; c0 += 512) {
            float xv[8], vv[8];
            ld8<T>(x + base + c0, xv); ld8<float>(v + (long)b * ld_v + c0, vv);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (c0 + e < C) acc += xv[e] * vv[e];
        }
        acc = wave_sum(acc);
        if (lane == 0) s[(long)b * N + n] = acc * scale;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rank1_kernel(T* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ v1,
                                                   const float* __restrict__ w2, const float* __restrict__ v2, int ld_v,
                                                   float s1, float s2, int N, int ld, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
        const float a1 = s1 * w1[(long)b * N + n];
        const float a2 = w2 ? s2 * w2[(long)b * N + n] : 0.f;
        for (int c0 = lane * 8; c0 < ld; c0 += 512) {
            float xv[8], p1[8], p2[8];
            ld8<T>(x + base + c0, xv); ld8<float>(v1 + (long)b * ld_v + c0, p1);
            if (w2) ld8<float>(v2 + (long)b * ld_v + c0, p2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (c0 + e < C) {
                    xv[e] += a1 * p1[e];
                    if (w2) xv[e] += a2 * p2[e];
                }
            }
            st8<T>(x + base + c0, xv);
        }
    }
}


// y = relu?(y + bias[c] (+ res))  on an NHWC map [R, C] (C % 8 == 0): the frozen-BN shift, the
// residual add and the ReLU of a DeepLab-ResNet bottleneck in one pass over the conv output.
template <typename T>
__global__ void bias_act_res_kernel(T* __restrict__ y, const float* __restrict__ bias, const T* __restrict__ res, int relu, long n8, int C) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)((i * 8) % C);
        float v[8], r[8];
        ld8<T>(y + i * 8, v);
        if (res) ld8<T>(res + i * 8, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = v[e] + bias[c0 + e];
            if (res) x += r[e];
            v[e] = relu ? fmaxf(x, 0.f) : x;
        }
        st8<T>(y + i * 8, v);
    }
}

// ------------------------------------------------------------------------------------------
// l2_normalize over channels.  rstd is stored NEGATIVE when sum(x^2) < eps (clamped branch:
// y = x / sqrt(eps) and d y / d x is the constant 1/sqrt(eps)).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ rstd,
                                                        float* __restrict__ nz, int R, int ld, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = blockIdx.x * WPB + w; r < R; r += gridDim.x * WPB) {
        const long base = (long)r * ld;
        float xv[MAXBLK][8];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                ld8<T>(x + base + c0, xv[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (c0 + e >= C) xv[k][e] = 0.f; ss += xv[k][e] * xv[k][e]; }
            }
        }
        ss = wave_sum(ss);
        const bool clamped = ss < 1e-12f;
        const float rs = rsqrtf(fmaxf(ss, 1e-12f));
        float sa = 0.f;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xv[k][e] *= rs; sa += fabsf(xv[k][e]); }
                st8<T>(y + base + c0, xv[k]);
            }
        }
        if (nz) sa = wave_sum(sa);
        if (lane == 0) {
            rstd[r] = clamped ? -rs : rs;
            if (nz) nz[r] = (sa != 0.f) ? 1.f : 0.f;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, const float* __restrict__ rstd,
                                                        T* __restrict__ dx, int R, int ld, int C, int accumulate) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = blockIdx.x * WPB + w; r < R; r += gridDim.x * WPB) {
        const long base = (long)r * ld;
        float g[MAXBLK][8], yv[MAXBLK][8];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                ld8<T>(dy + base + c0, g[k]); ld8<T>(y + base + c0, yv[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (c0 + e >= C) g[k][e] = 0.f; dot += g[k][e] * yv[k][e]; }
            }
        }
        dot = wave_sum(dot);
        const float rs = rstd[r];
        const float a = fabsf(rs);
        if (rs < 0.f) dot = 0.f;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float o[8];
                if (accumulate) ld8<T>(dx + base + c0, o);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = a * (g[k][e] - yv[k][e] * dot);
                    o[e] = accumulate ? o[e] + v : v;
                }
                st8<T>(dx + base + c0, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// per-sample {sum x, sum x^2} over (n, c < C)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sample_stats_kernel(const T* __restrict__ x, double* __restrict__ dpart, int N, int ld, int C) {
    __shared__ double red[2][WPB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    double s1 = 0.0, s2 = 0.0;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
        float a1 = 0.f, a2 = 0.f;
        for (int c0 = lane * 8; c0 < ld; c0 += 512) {
            float xv[8];
            ld8<T>(x + base + c0, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (c0 + e < C) { a1 += xv[e]; a2 += xv[e] * xv[e]; }
        }
        s1 += (double)a1; s2 += (double)a2;
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane == 0) { red[0][w] = s1; red[1][w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0)
        stat_store(dpart, b, blockIdx.x, gridDim.x, red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
}

// ------------------------------------------------------------------------------------------
// graph_conv elementwise parts
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gconv_pre_fwd_kernel(const T* __restrict__ Y, const T* __restrict__ X, const double* __restrict__ sums,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           T* __restrict__ G, int N, int ld, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    float mean, rstd;
    ln_stats(sums, b, (double)N * C, mean, rstd);
    // per-channel LayerNorm parameters of this lane's columns (the masters are padded past C inside the
    // flat buffer, values beyond C are never used)
    float gmv[MAXBLK][8], btv[MAXBLK][8];
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k) {
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld) { ld8<float>(gamma + c0, gmv[k]); if (beta) ld8<float>(beta + c0, btv[k]); }
    }
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float yv[8], xv[8], o[8];
                ld8<T>(Y + base + c0, yv); ld8<T>(X + base + c0, xv);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = (c0 + e < C) ? fmaxf(xv[e] + (yv[e] - mean) * rstd * gmv[k][e] + btv[k][e], 0.f) : 0.f;
                st8<T>(G + base + c0, o);
            }
        }
    }
}

// pass 1 of the LN backward: dxh = dG*[G>0]*gamma (written to dY), sample sums of dxh and dxh*xhat,
// dgamma += dZ*xhat, dbeta += dZ; dX (+)= dZ
template <typename T>
__global__ __launch_bounds__(256) void gconv_pre_bwd1_kernel(const T* __restrict__ dG, const T* __restrict__ G, const T* __restrict__ Y,
                                                            const double* __restrict__ sums, const float* __restrict__ gamma,
                                                            T* __restrict__ dX, int accumulate_dX, T* __restrict__ dY,
                                                            float* part, double* dpart, int N, int ld, int C) {
    extern __shared__ float lds[];
    __shared__ double red[2][WPB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    float mean, rstd;
    ln_stats(sums, b, (double)N * C, mean, rstd);
    float ag[MAXBLK][8], ab[MAXBLK][8], gmv[MAXBLK][8];
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; }
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld) ld8<float>(gamma + c0, gmv[k]);
    }
    double s1 = 0.0, s2 = 0.0;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float g[8], gv[8], yv[8], o[8], dxo[8];
                ld8<T>(dG + base + c0, g); ld8<T>(G + base + c0, gv); ld8<T>(Y + base + c0, yv);
                if (accumulate_dX) ld8<T>(dX + base + c0, dxo);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c0 + e;
                    const float dz = (c < C && gv[e] > 0.f) ? g[e] : 0.f;
                    const float xh = (yv[e] - mean) * rstd;
                    const float dxh = (c < C) ? dz * gmv[k][e] : 0.f;
                    ag[k][e] += dz * xh; ab[k][e] += dz;
                    a1 += dxh; a2 += dxh * xh;
                    o[e] = dxh;
                    dxo[e] = accumulate_dX ? dxo[e] + dz : dz;
                }
                st8<T>(dY + base + c0, o);
                st8<T>(dX + base + c0, dxo);
            }
        }
        s1 += (double)a1; s2 += (double)a2;
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane == 0) { red[0][w] = s1; red[1][w] = s2; }
    __syncthreads();
    const long wg = (long)b * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0)
        stat_store(dpart, b, blockIdx.x, gridDim.x, red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    colsum_flush(ag, part + wg * 2 * ld, ld, C, lds);
    colsum_flush(ab, part + wg * 2 * ld + ld, ld, C, lds);
}

// pass 2: dY = rstd * (dxh - mean(dxh) - xhat * mean(dxh*xhat))   (in place on dY; xsrc = pre-LN input)
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd2_kernel(T* __restrict__ dY, const T* __restrict__ xsrc, const double* __restrict__ sums,
                                                     const double* __restrict__ bsums, int N, int ld, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    float mean, rstd;
    const double cnt = (double)N * C;
    ln_stats(sums, b, cnt, mean, rstd);
    float m1, m2;
    stat_means(bsums, b, cnt, m1, m2);
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
        for (int c0 = lane * 8; c0 < ld; c0 += 512) {
            float d[8], xv[8];
            ld8<T>(dY + base + c0, d); ld8<T>(xsrc + base + c0, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[e] - mean) * rstd;
                d[e] = (c0 + e < C) ? rstd * (d[e] - m1 - xh * m2) : 0.f;
            }
            st8<T>(dY + base + c0, d);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gconv_post_fwd_kernel(const T* __restrict__ U, const double* __restrict__ sums,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ out, float* __restrict__ rstd_row, int N, int ld, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    float mean, rstd;
    ln_stats(sums, b, (double)N * C, mean, rstd);
    // per-channel LayerNorm parameters of this lane's columns (the masters are padded past C inside the
    // flat buffer, values beyond C are never used)
    float gmv[MAXBLK][8], btv[MAXBLK][8];
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k) {
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld) { ld8<float>(gamma + c0, gmv[k]); if (beta) ld8<float>(beta + c0, btv[k]); }
    }
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
        float hv[MAXBLK][8];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float uv[8];
                ld8<T>(U + base + c0, uv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c0 + e;
                    const float h = (c < C) ? fmaxf((uv[e] - mean) * rstd * gmv[k][e] + btv[k][e], 0.f) : 0.f;
                    hv[k][e] = h; ss += h * h;
                }
            }
        }
        ss = wave_sum(ss);
        const bool clamped = ss < 1e-12f;
        const float rs = rsqrtf(fmaxf(ss, 1e-12f));
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[k][e] *= rs;
                st8<T>(out + base + c0, hv[k]);
            }
        }
        if (lane == 0) rstd_row[(long)b * N + n] = clamped ? -rs : rs;
    }
}

// pass 1: dH = l2norm-bwd(dout), dUn = dH*[out>0], dxh = dUn*gamma -> dU; sums; dgamma/dbeta
template <typename T>
__global__ __launch_bounds__(256) void gconv_post_bwd1_kernel(const T* __restrict__ dout, const T* __restrict__ out, const float* __restrict__ rstd_row,
                                                             const T* __restrict__ U, const double* __restrict__ sums, const float* __restrict__ gamma,
                                                             T* __restrict__ dU, float* part, double* dpart, int N, int ld, int C) {
    extern __shared__ float lds[];
    __shared__ double red[2][WPB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
    float mean, rstd;
    ln_stats(sums, b, (double)N * C, mean, rstd);
    float ag[MAXBLK][8], ab[MAXBLK][8], gmv[MAXBLK][8];
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; }
        const int c0 = k * 512 + lane * 8;
        if (c0 < ld) ld8<float>(gamma + c0, gmv[k]);
    }
    double s1 = 0.0, s2 = 0.0;
    for (int n = blockIdx.x * WPB + w; n < N; n += gridDim.x * WPB) {
        const long base = ((long)b * N + n) * ld;
        float g[MAXBLK][8], ov[MAXBLK][8];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                ld8<T>(dout + base + c0, g[k]); ld8<T>(out + base + c0, ov[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (c0 + e >= C) g[k][e] = 0.f; dot += g[k][e] * ov[k][e]; }
            }
        }
        dot = wave_sum(dot);
        const float rs = rstd_row[(long)b * N + n];
        const float a = fabsf(rs);
        if (rs < 0.f) dot = 0.f;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < MAXBLK; ++k) {
            const int c0 = k * 512 + lane * 8;
            if (c0 < ld) {
                float uv[8], o[8];
                ld8<T>(U + base + c0, uv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c0 + e;
                    const float dh = a * (g[k][e] - ov[k][e] * dot);
                    const float dz = (c < C && ov[k][e] > 0.f) ? dh : 0.f;
                    const float xh = (uv[e] - mean) * rstd;
                    const float dxh = (c < C) ? dz * gmv[k][e] : 0.f;
                    ag[k][e] += dz * xh; ab[k][e] += dz;
                    a1 += dxh; a2 += dxh * xh;
                    o[e] = dxh;
                }
                st8<T>(dU + base + c0, o);
            }
        }
        s1 += (double)a1; s2 += (double)a2;
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane == 0) { red[0][w] = s1; red[1][w] = s2; }
    __syncthreads();
    const long wg = (long)b * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0)
        stat_store(dpart, b, blockIdx.x, gridDim.x, red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    colsum_flush(ag, part + wg * 2 * ld, ld, C, lds);
    colsum_flush(ab, part + wg * 2 * ld + ld, ld, C, lds);
}

bool map_ok(const char* what, int ld, int C, int dt) {
    if (ld <= 0 || C <= 0 || C > ld || ld % 8 || ld > MAXBLK * 512) {
        cmpc_set_error("%s: need 0 < C <= ld <= %d, ld %% 8 == 0 (got C=%d ld=%d)", what, MAXBLK * 512, C, ld);
        return false;
    }
    if (dt != DT_F32 && dt != DT_BF16 && dt != DT_F16) { cmpc_set_error("%s: bad dtype %d", what, dt); return false; }
    return true;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int cmpc_cast(int src_dt, const void* src, int dst_dt, void* dst, int64_t n, void* stream) {
    if (n <= 0) return CMPC_OK;
    const int g = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
#define CAST_CASE(SD, ST_, DD, DT_) if (src_dt == SD && dst_dt == DD) { hipLaunchKernelGGL((cast_kernel<ST_, DT_>), dim3(g), dim3(256), 0, ST, (const ST_*)src, (DT_*)dst, (long)n); return cmpc_check_launch("cast"); }
    CAST_CASE(DT_F32, float, DT_BF16, bf16_t) CAST_CASE(DT_BF16, bf16_t, DT_F32, float) CAST_CASE(DT_F32, float, DT_F32, float)
    CAST_CASE(DT_BF16, bf16_t, DT_BF16, bf16_t) CAST_CASE(DT_F32, float, DT_F16, f16_t) CAST_CASE(DT_F16, f16_t, DT_F32, float)
    CAST_CASE(DT_F16, f16_t, DT_F16, f16_t) CAST_CASE(DT_F16, f16_t, DT_BF16, bf16_t) CAST_CASE(DT_BF16, bf16_t, DT_F16, f16_t)
#undef CAST_CASE
    cmpc_set_error("cast: bad dtypes");
    return CMPC_EINVAL;
}

extern "C" int cmpc_axpy(int dt, const void* x, void* y, float a, int64_t n, void* stream) {
    if (n % 8) { cmpc_set_error("axpy: n must be a multiple of 8"); return CMPC_EINVAL; }
    if (n == 0) return CMPC_OK;
    const long n8 = n / 8;
    const int g = (int)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((axpy_kernel<T>), dim3(g), dim3(256), 0, ST, (const T*)x, (T*)y, a, n8));
    return cmpc_check_launch("axpy");
}

extern "C" int cmpc_bias_act_res(int dt, void* y, const float* bias, const void* res, int relu, int64_t R, int C, void* stream) {
    if (C <= 0 || C % 8 || !y || !bias) { cmpc_set_error("bias_act_res: C must be a positive multiple of 8"); return CMPC_EINVAL; }
    const long n8 = R * C / 8;
    if (n8 == 0) return CMPC_OK;
    const int g = (int)((n8 + 255) / 256 > 8192 ? 8192 : (n8 + 255) / 256);
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((bias_act_res_kernel<T>), dim3(g), dim3(256), 0, ST, (T*)y, bias, (const T*)res, relu, n8, C));
    return cmpc_check_launch("bias_act_res");
}

extern "C" int cmpc_act_bwd(int dt, const void* dy, const void* y, void* dpre, int act, int R, int stride, int ld, int C,
                            float* db, float* dsb, int ld_dsb, int rows_per_sample, void* stream) {
    cmpc_op_scope op_("act_bwd");
    if (!map_ok("act_bwd", ld, C, dt)) return CMPC_EINVAL;
    if (stride < ld || stride % 8) { cmpc_set_error("act_bwd: stride must be >= ld and a multiple of 8"); return CMPC_EINVAL; }
    int N = R, B = 1;
    if (dsb) { if (rows_per_sample <= 0 || R % rows_per_sample) { cmpc_set_error("act_bwd: bad rows_per_sample"); return CMPC_EINVAL; } N = rows_per_sample; B = R / N; }
    if (R == 0) return CMPC_OK;
    const int gx = dsb ? (rows_grid(N) > 64 ? 64 : rows_grid(N)) : ((R + 3) / 4 > 512 ? 512 : (R + 3) / 4);
    float* part = nullptr;
    if (db || dsb) { part = (float*)cmpc_ws((size_t)B * gx * ld * sizeof(float), ST); if (!part) return CMPC_EHIP; }
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(gx, B), dim3(256), WPB * ld * sizeof(float), ST,
                                             (const T*)dy, (const T*)y, (T*)dpre, act, N, stride, ld, C, part));
    if (db && cmpc_reduce_parts_f32(part, ld, 1, B * gx, 1, ld, C, db, 0, 0, 1, ST)) return CMPC_EHIP;
    if (dsb && cmpc_reduce_parts_f32(part, ld, B, gx, 1, ld, C, dsb, ld_dsb, 0, 1, ST)) return CMPC_EHIP;
    return cmpc_check_launch("act_bwd");
}

extern "C" int cmpc_wcolsum(int dt, const void* x, const float* w, float* out, int ld_out, int B, int N, int ld, int C, float scale, void* stream) {
    cmpc_op_scope op_("wcolsum");
    if (!map_ok("wcolsum", ld, C, dt)) return CMPC_EINVAL;
    const int gx = rows_grid(N) > 64 ? 64 : rows_grid(N);
    float* part = (float*)cmpc_ws((size_t)B * gx * ld * sizeof(float), ST);
    if (!part) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((wcolsum_kernel<T>), dim3(gx, B), dim3(256), WPB * ld * sizeof(float), ST,
                                             (const T*)x, w, part, N, ld, C, scale));
    if (cmpc_reduce_parts_f32(part, ld, B, gx, 1, ld, C, out, ld_out, 0, 1, ST)) return CMPC_EHIP;
    return cmpc_check_launch("wcolsum");
}

extern "C" int cmpc_rowdot1(int dt, const void* x, const float* v, int ld_v, float* s, int B, int N, int ld, int C, float scale, void* stream) {
    if (!map_ok("rowdot1", ld, C, dt)) return CMPC_EINVAL;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((rowdot1_kernel<T>), dim3(rows_grid(N), B), dim3(256), 0, ST, (const T*)x, v, ld_v, s, N, ld, C, scale));
    return cmpc_check_launch("rowdot1");
}

extern "C" int cmpc_rank1_update(int dt, void* x, const float* w1, const float* v1, const float* w2, const float* v2,
                                 int ld_v, float s1, float s2, int B, int N, int ld, int C, void* stream) {
    if (!map_ok("rank1_update", ld, C, dt)) return CMPC_EINVAL;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((rank1_kernel<T>), dim3(rows_grid(N), B), dim3(256), 0, ST, (T*)x, w1, v1, w2, v2, ld_v, s1, s2, N, ld, C));
    return cmpc_check_launch("rank1_update");
}

extern "C" int cmpc_l2norm_rows_fwd(int dt, const void* x, void* y, float* rstd, float* nz_mask, int R, int ld, int C, void* stream) {
    if (!map_ok("l2norm_rows_fwd", ld, C, dt)) return CMPC_EINVAL;
    if (R == 0) return CMPC_OK;
    const int g = (R + 3) / 4 > 2048 ? 2048 : (R + 3) / 4;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((l2norm_fwd_kernel<T>), dim3(g), dim3(256), 0, ST, (const T*)x, (T*)y, rstd, nz_mask, R, ld, C));
    return cmpc_check_launch("l2norm_rows_fwd");
}

extern "C" int cmpc_l2norm_rows_bwd(int dt, const void* dy, const void* y, const float* rstd, void* dx, int R, int ld, int C, int accumulate, void* stream) {
    if (!map_ok("l2norm_rows_bwd", ld, C, dt)) return CMPC_EINVAL;
    if (R == 0) return CMPC_OK;
    const int g = (R + 3) / 4 > 2048 ? 2048 : (R + 3) / 4;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((l2norm_bwd_kernel<T>), dim3(g), dim3(256), 0, ST, (const T*)dy, (const T*)y, rstd, (T*)dx, R, ld, C, accumulate));
    return cmpc_check_launch("l2norm_rows_bwd");
}

extern "C" int cmpc_sample_stats(int dt, const void* x, double* sums, int B, int N, int ld, int C, void* stream) {
    if (ld <= 0 || C > ld || ld % 8) { cmpc_set_error("sample_stats: bad ld/C"); return CMPC_EINVAL; }
    const int gx = rows_grid(N) > 128 ? 128 : rows_grid(N);
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((sample_stats_kernel<T>), dim3(gx, B), dim3(256), 0, ST, (const T*)x, sums, N, ld, C));
    return cmpc_check_launch("sample_stats");
}

extern "C" int cmpc_gconv_pre_fwd(int dt, const void* Y, const void* X, const double* sums, const float* gamma, const float* beta,
                                  void* G, int B, int N, int ld, int C, void* stream) {
    if (!map_ok("gconv_pre_fwd", ld, C, dt)) return CMPC_EINVAL;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((gconv_pre_fwd_kernel<T>), dim3(rows_grid(N), B), dim3(256), 0, ST,
                                             (const T*)Y, (const T*)X, sums, gamma, beta, (T*)G, N, ld, C));
    return cmpc_check_launch("gconv_pre_fwd");
}

// workspace of the LN-backward first passes: [B*gx][2][ld] fp32 column partials (the fp64 pairs go to the caller's stat block)
static int ln_bwd_ws(int B, int gx, int ld, float** part, hipStream_t st) {
    *part = (float*)cmpc_ws((size_t)B * gx * 2 * ld * sizeof(float), st);
    return *part ? CMPC_OK : CMPC_EHIP;
}

extern "C" int cmpc_gconv_pre_bwd(int dt, const void* dG, const void* G, const void* Y, const double* sums, const float* gamma,
                                  void* dX, int accumulate_dX, void* dY, float* dgamma, float* dbeta, double* bsums,
                                  int B, int N, int ld, int C, void* stream) {
    cmpc_op_scope op_("gconv_pre_bwd");
    if (!map_ok("gconv_pre_bwd", ld, C, dt)) return CMPC_EINVAL;
    const int gx = rows_grid(N) > 64 ? 64 : rows_grid(N);
    float* part;
    if (ln_bwd_ws(B, gx, ld, &part, ST)) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((gconv_pre_bwd1_kernel<T>), dim3(gx, B), dim3(256), WPB * ld * sizeof(float), ST,
                           (const T*)dG, (const T*)G, (const T*)Y, sums, gamma, (T*)dX, accumulate_dX, (T*)dY, part, bsums, N, ld, C));
    if (cmpc_reduce_parts_f32(part, 2 * ld, 1, B * gx, 1, ld, C, dgamma, 0, 0, 1, ST)) return CMPC_EHIP;
    if (cmpc_reduce_parts_f32(part + ld, 2 * ld, 1, B * gx, 1, ld, C, dbeta, 0, 0, 1, ST)) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((ln_bwd2_kernel<T>), dim3(rows_grid(N), B), dim3(256), 0, ST, (T*)dY, (const T*)Y, sums, bsums, N, ld, C));
    return cmpc_check_launch("gconv_pre_bwd");
}

extern "C" int cmpc_gconv_post_fwd(int dt, const void* U, const double* sums, const float* gamma, const float* beta,
                                   void* out, float* rstd_row, int B, int N, int ld, int C, void* stream) {
    if (!map_ok("gconv_post_fwd", ld, C, dt)) return CMPC_EINVAL;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((gconv_post_fwd_kernel<T>), dim3(rows_grid(N), B), dim3(256), 0, ST,
                                             (const T*)U, sums, gamma, beta, (T*)out, rstd_row, N, ld, C));
    return cmpc_check_launch("gconv_post_fwd");
}

extern "C" int cmpc_gconv_post_bwd(int dt, const void* dout, const void* out, const float* rstd_row, const void* U,
                                   const double* sums, const float* gamma, void* dU, float* dgamma, float* dbeta, double* bsums,
                                   int B, int N, int ld, int C, void* stream) {
    cmpc_op_scope op_("gconv_post_bwd");
    if (!map_ok("gconv_post_bwd", ld, C, dt)) return CMPC_EINVAL;
    const int gx = rows_grid(N) > 64 ? 64 : rows_grid(N);
    float* part;
    if (ln_bwd_ws(B, gx, ld, &part, ST)) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((gconv_post_bwd1_kernel<T>), dim3(gx, B), dim3(256), WPB * ld * sizeof(float), ST,
                           (const T*)dout, (const T*)out, rstd_row, (const T*)U, sums, gamma, (T*)dU, part, bsums, N, ld, C));
    if (cmpc_reduce_parts_f32(part, 2 * ld, 1, B * gx, 1, ld, C, dgamma, 0, 0, 1, ST)) return CMPC_EHIP;
    if (cmpc_reduce_parts_f32(part + ld, 2 * ld, 1, B * gx, 1, ld, C, dbeta, 0, 0, 1, ST)) return CMPC_EHIP;
    CMPC_DISPATCH_DT(dt, hipLaunchKernelGGL((ln_bwd2_kernel<T>), dim3(rows_grid(N), B), dim3(256), 0, ST, (T*)dU, (const T*)U, sums, bsums, N, ld, C));
    return cmpc_check_launch("gconv_post_bwd");
}
