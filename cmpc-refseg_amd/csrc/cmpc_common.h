// Shared device helpers for the CMPC gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#define CMPC_OK 0
#define CMPC_EINVAL (-1)
#define CMPC_EHIP (-2)

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8v;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(8))) short s8;

typedef uint16_t bf16_t;   // storage type of a bf16 element
typedef _Float16 f16_t;    // IEEE half: same MFMA rate as bf16, 3 more mantissa bits (2^-12 against 2^-9 rounding), range +-65504
typedef __attribute__((ext_vector_type(8))) _Float16 h8v;

enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3 };

void cmpc_set_error(const char* fmt, ...);
int cmpc_check_launch(const char* what);
// Launch-trace attribution for entry points that launch a stage kernel and THEN fold its partial rows (cmpc_reduce_parts_f32 checks its
// own launch): `cmpc_op_scope op("mutan_bwd");` at the top of the entry point makes the first fold inside it close the stage kernel's
// interval under that name (cmpc_trace_producer), so the kernel's time is not booked on the fold.
struct cmpc_op_scope { const char* prev; bool prev_marked; cmpc_op_scope(const char* name); ~cmpc_op_scope(); };
void cmpc_trace_producer();
// Library-owned scratch for per-workgroup partial sums: one growing buffer PER STREAM (up to 64
// streams), so stage operators running on different streams never share partial rows.
void* cmpc_ws(size_t bytes, hipStream_t st);
void cmpc_ws_release(hipStream_t st);      // before destroying a stream the library created: frees its workspace slot
// out[o*ld_out + seg*out_seg + c] (+)= sum_{i<ninner} part[(o*ninner+i)*part_stride + seg*seg_ld + c], c < seg_C   (one writer per element)
int cmpc_reduce_parts_f32(const float* part, long part_stride, int nouter, int ninner, int nseg, int seg_ld, int seg_C,
                          float* out, long ld_out, long out_seg, int accumulate, hipStream_t st);
// cmpc_gemm_tn_grouped with `nslots` caller-owned persistent descriptor tables (device) and host copies of their contents: a step
// whose descriptors equal a cached table reuses it; otherwise slot *victim (round robin) is rewritten (asynchronously)
int cmpc_gemm_tn_grouped_cached(const void* args /* cmpc_gemm_tn_args[n] */, int n, void* const* tables_dev, int nslots, int* victim, size_t table_bytes,
                                std::vector<char>* shadows, hipStream_t st);
// Deferred folds.  Column sums whose target lies in [lo, hi) (the flat gradient buffer: bias, LayerNorm and peephole gradients, read by
// nothing before the optimizer) need not be folded right behind their producer: between cmpc_fold_begin and cmpc_fold_flush (same host
// thread) cmpc_ws hands out NON-recycled pieces of `arena`, cmpc_reduce_parts_f32 records such folds instead of launching them, and
// cmpc_fold_flush folds them all in ONE launch.  table_dev: device buffer for the descriptors (>= 64 B each); it is re-uploaded only
// when the recorded list differs from the previous flush (shapes are static, so after the first step it never does).
struct cmpc_fold_desc { const float* part; long part_stride; int nouter, ninner, nseg, seg_ld, seg_C, blk_begin; float* out; long ld_out, out_seg; int chain, pad_; };
struct cmpc_fold_ctx {
    char* arena = nullptr; size_t cap = 0, off = 0; const float* lo = nullptr; const float* hi = nullptr;
    cmpc_fold_desc* table_dev = nullptr; int table_cap = 0;
    cmpc_fold_desc* descs = nullptr; int n = 0;            // host list of this step (table_cap entries)
    cmpc_fold_desc* shadow = nullptr; int shadow_n = -1;   // what table_dev holds
};
void cmpc_fold_begin(cmpc_fold_ctx* ctx);
int cmpc_fold_flush(cmpc_fold_ctx* ctx, hipStream_t st);
// Partial flush: fold (and forget) only the recorded folds whose target lies in one of the nr ranges [lo[i], hi[i]); the collector stays
// on.  table_dev / shadow / shadow_n: the persistent descriptor table of THIS flush point (its own cache, see cmpc_fold_flush).
int cmpc_fold_flush_ranges(cmpc_fold_ctx* ctx, const float* const* lo, const float* const* hi, int nr, cmpc_fold_desc* table_dev,
                           cmpc_fold_desc* shadow, int* shadow_n, hipStream_t st);
// out[o*nval + v] = sum_{i<ninner} part[(o*ninner+i)*nval + v]

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN-preserving
    return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

template <> struct Elem<f16_t> {
    static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }      // v_cvt_f16_f32: RNE, overflow -> inf
};

// 8 consecutive elements <-> float[8] (16-byte vector accesses; p must be 16-B aligned for
// bf16 and 32-B aligned for f32).
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void ld8<f16_t>(const f16_t* p, float (&v)[8]) {
    const h8v a = *reinterpret_cast<const h8v*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    uint4 a;
    a.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    a.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    a.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
    a.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = a;
}

template <> __device__ __forceinline__ void st8<f16_t>(f16_t* p, const float (&v)[8]) {
    h8v a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (f16_t)v[e];
    *reinterpret_cast<h8v*>(p) = a;
}

// the value a stored element will read back as (bf16 storage rounds; fp32 storage is exact)
template <typename T> __device__ __forceinline__ float stored_value(float v);
template <> __device__ __forceinline__ float stored_value<float>(float v) { return v; }
template <> __device__ __forceinline__ float stored_value<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float stored_value<f16_t>(float v) { return (float)(f16_t)v; }

// wave64 reductions via cross-lane shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block reductions for 256-thread blocks (4 waves); red must hold >= 4 floats per value
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// the bulk elementwise paths (ConvLSTM gates: 13 M evaluations per step): v_exp_f32 and v_rcp_f32 (~1 ulp each) instead of the IEEE division's
// ten instructions; |error| < 2e-7.  The text LSTM keeps sigmoidf_: its gradients pass through 20-25 steps of cancellation (CMPCv5 embedding
// gradient: 5e-3 relative against the oracle with this form, 4e-4 with the exact one).
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// tanh for the bulk elementwise paths (65 M evaluations per level in the Mutan heads): 1 - 2 / (exp(2|x|) + 1) on the hardware
// exp2 / rcp (~1 ulp each), an odd polynomial below 1/8 where that form cancels.  |error| < 1e-7 absolute, < 5e-7 relative:
// below an ulp of the 16-bit storage types and two orders under the fp32 parity tolerance; about a third of tanhf's instructions.
__device__ __forceinline__ float cmpc_tanh(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * ax);       // exp(2|x|) on v_exp_f32; inf for large |x| -> t = 1
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);           // v_rcp_f32
    const float p = ax * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
    return copysignf(ax < 0.125f ? p : t, x);
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_TANH: return tanhf(v);
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    switch (act) {
        case ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case ACT_TANH: return 1.0f - y * y;
        case ACT_SIGMOID: return y * (1.0f - y);
        default: return 1.0f;
    }
}

// Whole-sample statistics (LayerNorm sums, their backward counterparts) live in "stat blocks": double [n_stats][STAT_PARTS][2], one
// (sum, sum of squares) pair per producing workgroup, unused slots zero.  The producer's workgroups store their pairs (stat_store);
// every consumer wave adds the 128 pairs itself in a fixed order (stat_load) -- no fold launch between the two kernels, and the
// result does not depend on which workgroup finished first.
constexpr int STAT_PARTS = 128;
// called by ONE thread of workgroup `wgx` of `ngx` (<= STAT_PARTS) that feed statistic `sidx`; workgroup 0 also clears the unused slots
__device__ __forceinline__ void stat_store(double* stats, int sidx, int wgx, int ngx, double s1, double s2) {
    double* d = stats + ((long)sidx * STAT_PARTS + wgx) * 2;
    d[0] = s1; d[1] = s2;
    if (wgx == 0) for (int i = ngx; i < STAT_PARTS; ++i) { d[2 * i] = 0.0; d[2 * i + 1] = 0.0; }
}
// every lane of a full wave must take part
__device__ __forceinline__ void stat_load(const double* stats, int sidx, double& s1, double& s2) {
    const double* p = stats + (long)sidx * STAT_PARTS * 2;
    const int lane = threadIdx.x & 63;
    const double a1 = p[2 * lane], a2 = p[2 * lane + 1], b1 = p[2 * (lane + 64)], b2 = p[2 * (lane + 64) + 1];
    s1 = wave_sum_d(a1 + b1); s2 = wave_sum_d(a2 + b2);
}
// whole-sample LayerNorm statistics (tf.contrib.layers.layer_norm, eps 1e-12) from statistic `sidx` of a stat block
__device__ __forceinline__ void ln_stats(const double* stats, int sidx, double count, float& mean, float& rstd) {
    double s1, s2;
    stat_load(stats, sidx, s1, s2);
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + 1e-12));
}
// the two means of a backward statistic
__device__ __forceinline__ void stat_means(const double* stats, int sidx, double count, float& m1, float& m2) {
    double s1, s2;
    stat_load(stats, sidx, s1, s2);
    m1 = (float)(s1 / count); m2 = (float)(s2 / count);
}

#define CMPC_DISPATCH_DT(dt, ...)                                  \
    do {                                                           \
        if ((dt) == DT_F32) { typedef float T; __VA_ARGS__; }      \
        else if ((dt) == DT_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else if ((dt) == DT_F16) { typedef f16_t T; __VA_ARGS__; } \
        else { cmpc_set_error("bad dtype %d", (int)(dt)); return CMPC_EINVAL; } \
    } while (0)
