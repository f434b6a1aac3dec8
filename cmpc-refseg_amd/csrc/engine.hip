// engine.hip -- whole-path entry points of the C ABI (include/cmpc.h: cmpc_create / cmpc_forward / cmpc_backward /
// cmpc_optimizer_step / cmpc_destroy): LSTM_model.build_graph() + train_op() of the reference
// (/root/reference/CMPC_model.py:89-142,426-492) as ONE C++ object that owns the parameters, the packed GEMM
// operands, a static workspace for every intermediate of a step, three lane streams and their events, and
// enqueues the stage kernels of this library directly -- no Python, no autograd, no allocation and no environment
// lookups in a step.  The backward pass is hand-sequenced (reverse stage order, fan-out gradients summed explicitly).
//
// Stage order and stream plan (forward; backward mirrors it):
//   main : text encoder (lstm, :144-164) -> parser (:347-357) -> valid_lang (:166-178)      [overlaps the backbone]
//   lanes: c5 | c4 | c3 : lateral+l2norm (:108-113) -> mutan (:295-328) -> spa_graph (:359-410) -> fusion (:338-344)
//                         -> score_cX + upsample + BCE (:128-133,440-443)
//   main : nec_lang (:180-192)
//   lanes: gated exchange round 1 (:271-276), join, round 2 (:278-284)
//   main : ConvLSTM over (c3, c4, c5) (:287-290, util/cell.py:36-79) -> score + upsample + sigmoid + BCE + mIoU
#include "cmpc_common.h"
#include "../../include/cmpc.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <array>
#include <map>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

extern long g_cmpc_launches;     // ops_norm.hip: bumped by cmpc_check_launch

namespace {

#define CK(x) do { const int _rc = (x); if (_rc != CMPC_OK) return _rc; } while (0)
#define HCK(x) do { const hipError_t _e = (x); if (_e != hipSuccess) { cmpc_set_error("%s: %s", #x, hipGetErrorString(_e)); return CMPC_EHIP; } } while (0)

inline int pad64(int x) { return (x + 63) / 64 * 64; }
inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

// ------------------------------------------------------------------------------------------
// small kernels of the orchestration itself (everything else lives in the stage files)
// ------------------------------------------------------------------------------------------
struct AddArgs { const void* src[8]; int n; };

// dst[i] = (acc ? dst[i] : 0) + sum_k src_k[i]   (fan-out gradients of a stage output; 8 elements per lane)
template <typename T>
__global__ __launch_bounds__(256) void add_n_kernel(T* __restrict__ dst, AddArgs a, int acc, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float s[8];
        if (acc) ld8<T>(dst + i * 8, s);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < a.n) {
                float v[8];
                ld8<T>(reinterpret_cast<const T*>(a.src[k]) + i * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[e];
            }
        }
        st8<T>(dst + i * 8, s);
    }
}

// dst[c][r] = (T) src[r][c]  (src f32 [rows][cols], dst [cols][rows]); 32 x 32 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) Elem<T>::st(dst + (long)c * rows + r, tile[tx][ty + 8 * i]);
    }
}

__global__ void transpose_i32_kernel(const int* __restrict__ src, int* __restrict__ dst, int rows, int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // dst[c][r] = src[r][c]
    if (i < rows * cols) { const int r = i / cols, c = i - r * cols; dst[c * rows + r] = src[i]; }
}
// dst[i] = src[i*stride + col]
__global__ void col_get_kernel(const float* __restrict__ src, int stride, int col, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[(long)i * stride + col];
}
// dst[i*stride + col] += a[i] + b[i] + c[i]
__global__ void col_add3_kernel(float* __restrict__ dst, int stride, int col, const float* __restrict__ a, const float* __restrict__ b,
                                const float* __restrict__ c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[(long)i * stride + col] += a[i] + b[i] + c[i];
}
// dst[r, :] = src[0, :] for r < rows (tf.tile of a language vector over the sampled frames, CMPC_video_mm_tgraph_allvec.py:341)
__global__ void tile_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * ld) dst[i] = src[i % ld];
}
// out[c] = sum_{r < rows} src[r, c]
__global__ void sum_rows_kernel(const float* __restrict__ src, float* __restrict__ out, int rows, int ld, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ld) return;
    float s = accumulate ? out[c] : 0.f;
    for (int r = 0; r < rows; ++r) s += src[(long)r * ld + c];
    out[c] = s;
}
// The temporal graph's message passing on its F <= 8 nodes (CMPC_video_mm_tgraph_allvec.py:484-497): adj = softmax_g(scale * q[f] . k[g]),
// Y[f] = sum_g adj[f, g] X[g].  One workgroup; everything is [F, ld] float32.
__global__ __launch_bounds__(256) void tgraph_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ X, float* __restrict__ adj,
                                                        float* __restrict__ Y, int Fr, int ld, int C, float scale) {
    __shared__ float sa[8][8];
    __shared__ float red[4];
    for (int f = 0; f < Fr; ++f)
        for (int g = 0; g < Fr; ++g) {
            float d = 0.f;
            for (int c = threadIdx.x; c < C; c += 256) d += q[(long)f * ld + c] * k[(long)g * ld + c];
            d = block_sum_256(d, red);
            if (threadIdx.x == 0) sa[f][g] = d * scale;
        }
    __syncthreads();
    if (threadIdx.x < Fr) {
        const int f = threadIdx.x;
        float m = sa[f][0];
        for (int g = 1; g < Fr; ++g) m = fmaxf(m, sa[f][g]);
        float sum = 0.f;
        for (int g = 0; g < Fr; ++g) { sa[f][g] = expf(sa[f][g] - m); sum += sa[f][g]; }
        for (int g = 0; g < Fr; ++g) { sa[f][g] /= sum; adj[f * 8 + g] = sa[f][g]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ld; c += 256)
        for (int f = 0; f < Fr; ++f) {
            float y = 0.f;
            if (c < C) for (int g = 0; g < Fr; ++g) y += sa[f][g] * X[(long)g * ld + c];
            Y[(long)f * ld + c] = y;
        }
}
// dX[g] (+)= sum_f adj[f, g] dY[f];  dadj = dY . X^T -> softmax backward -> dS;  dq[f] = scale * sum_g dS[f, g] k[g];  dk[g] = scale * sum_f dS[f, g] q[f]
__global__ __launch_bounds__(256) void tgraph_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ adj, const float* __restrict__ X, const float* __restrict__ q,
                                                        const float* __restrict__ k, float* __restrict__ dX, int acc_dX, float* __restrict__ dq, float* __restrict__ dk,
                                                        int Fr, int ld, int C, float scale) {
    __shared__ float da[8][8];
    __shared__ float red[4];
    for (int f = 0; f < Fr; ++f)
        for (int g = 0; g < Fr; ++g) {
            float d = 0.f;
            for (int c = threadIdx.x; c < C; c += 256) d += dY[(long)f * ld + c] * X[(long)g * ld + c];
            d = block_sum_256(d, red);
            if (threadIdx.x == 0) da[f][g] = d;
        }
    __syncthreads();
    if (threadIdx.x < Fr) {
        const int f = threadIdx.x;
        float dot = 0.f;
        for (int g = 0; g < Fr; ++g) dot += da[f][g] * adj[f * 8 + g];
        for (int g = 0; g < Fr; ++g) da[f][g] = adj[f * 8 + g] * (da[f][g] - dot) * scale;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ld; c += 256)
        for (int a = 0; a < Fr; ++a) {
            float x = 0.f, vq = 0.f, vk = 0.f;
            if (c < C)
                for (int b = 0; b < Fr; ++b) {
                    x += adj[b * 8 + a] * dY[(long)b * ld + c];
                    vq += da[a][b] * k[(long)b * ld + c];
                    vk += da[b][a] * q[(long)b * ld + c];
                }
            dX[(long)a * ld + c] = acc_dX ? dX[(long)a * ld + c] + x : x;
            dq[(long)a * ld + c] = vq; dk[(long)a * ld + c] = vk;
        }
}
// scalars[0..5] = loss_all, loss_c3, loss_c4, loss_c5, loss_last, mIoU  (CMPC_model.py:440-447,486-490)
__global__ void scalars_kernel(const float* __restrict__ l_last, const float* __restrict__ l5, const float* __restrict__ l4, const float* __restrict__ l3,
                               const int* __restrict__ inter, const int* __restrict__ uni, int B, float w0, float w5, float w4, float w3,
                               float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float a = 0.f, b5 = 0.f, b4 = 0.f, b3 = 0.f;
    double iou = 0.0;
    for (int b = 0; b < B; ++b) { a += l_last[b]; b5 += l5[b]; b4 += l4[b]; b3 += l3[b]; iou += (double)inter[b] / (double)uni[b]; }
    a /= B; b5 /= B; b4 /= B; b3 /= B;
    out[0] = w0 * a + w5 * b5 + w4 * b4 + w3 * b3;
    out[1] = b3; out[2] = b4; out[3] = b5; out[4] = a;
    out[5] = (float)(iou / B);
}

// ------------------------------------------------------------------------------------------
struct ParamSpec { std::string name; int rank; int64_t shape[4]; int64_t count; int64_t off; float wd; float gmult; };
struct Operand { size_t off; int dt; int rows; int ld; };
struct Tap { std::string name; void* ptr; int dt; int rank; int64_t shape[4]; };

struct GemmOpt {
    int n_valid = -1, batch = 1; int64_t sC = 0; int c_f32 = 0;
    const float* bias = nullptr; const float* sbias = nullptr; int ld_sbias = 0;
    const float* pbias = nullptr; int ld_pbias = 0; int rows_per_sample = 0;
    int act = ACT_NONE; float alpha = 1.f; int accumulate = 0;
};
struct Seg { const void* A; int lda; const void* Bt; int ldb; int K; int64_t sA = 0; int64_t sB = 0; };

// bump allocator: first pass (base == nullptr) measures, second pass assigns
// CMPC_WS_GUARD=<bytes> (debugging): every allocation is followed by a guard of that many bytes that nothing may write; cmpc_debug_check_guards
// reports the allocations whose guard is no longer zero (an out-of-bounds write of the kernel that owns the buffer in front of it)
static size_t g_ws_guard = getenv("CMPC_WS_GUARD") ? (size_t)atol(getenv("CMPC_WS_GUARD")) : 0;
struct Bump {
    char* base = nullptr; size_t off = 0;
    std::vector<std::pair<char*, size_t>>* log = nullptr;      // (guard address, size of the allocation in front of it)
    void* take(size_t bytes) {
        void* p = base ? base + off : nullptr; off += up256(bytes);
        if (g_ws_guard) { if (base && log) log->push_back({base + off, bytes}); off += up256(g_ws_guard); }
        return p;
    }
};

struct LevelBuf {          // one pyramid level (c5 / c4 / c3), forward then backward
    int cin;
    const void* feat;
    void* X0t;             // model V5_BILSTM: tanh(lateral) before the l2-normalisation (kept for the backward pass)
    // model VIDEO (CMPC_video_mm_tgraph_allvec.py): temporal pooling / temporal graph / temporal context of one level; small tensors float32
    void *X1m, *X0m, *CTX, *GLO, *ctA_t, *ctGv_t, *ctPT, *ctPTt, *TGN16, *TGNt, *dCTX, *dGLO, *dctA0_t, *dX0f;
    float *g1, *lt, *kqv, *tlog, *tatt, *TG, *q, *k, *adj, *TY, *TG1, *TU, *TGN, *rrow_t, *ctv, *ctPTf, *ctk0s, *ctA0, *ctA, *ctGv, *ctsc, *ctx_rstd, *pb;
    float *dTGN, *dTU, *dTG1, *dTY, *dTG, *dq, *dk, *dtatt, *dtlog, *dkqv, *dlt, *dac, *dctv, *dctPT, *dctk0s, *dctA, *dctA0, *dprc, *ctsc2, *dea;
    double *tsums1, *tsums2, *tbs;
    void *X0, *P, *X1, *PT, *PTt, *gw_w_t, *gw_v_t, *Zt, *Y, *G, *U, *X2, *F;
    float *lat_rstd, *g, *mut_rstd, *Wd, *PTf, *k0s, *A0, *pr, *gw_w, *gw_v, *gsc, *Ztf, *rrow, *sb;
    double *sums1, *sums2;
    float *score, *up, *loss; int* iu;
    // backward
    void *dfus, *dpre, *dX1, *dX2, *dU, *dG, *dY, *Z, *dZ, *dZt, *dA0_t, *dX0, *dV;
    void* dfeat = nullptr;  // cfg.conv5: d cost / d (backbone tap) [R, cin]
    float *dscore, *dsb, *dvl, *Zf, *dgw_w, *dZf, *dZtf, *dgw_v, *dA0, *dpr, *gsc2, *dPT, *dk0s, *dWd, *dwf, *dg;
    double* bs;
};
struct ExgBuf {            // one gated_exchange_module
    float *q, *kq, *logits, *attn, *pooled, *gvpre, *gv, *rs1, *g[2], *rstd;
    void *r[2], *out;
    void *dfeat, *dp[2], *dfs[2];
    float *dg[2], *dgv, *dgvpre, *dpooled, *dnec, *dattn, *dlog, *dkq, *dq;
};
struct ClstmStep { void *Yg, *c_pre, *c_new, *h_new, *dYg, *dc_prev, *dx, *dh; double* sums; };
// one direction of the text encoder (CMPC_model: the only one; CMPCv5_BiLSTM: fw and bw)
struct LstmDir {
    std::string key, pk, pb;            // operand key ("lstm" / "lstm_fw" / "lstm_bw"), kernel / bias parameter names
    int* words_tb; float *emb, *xg, *gates, *h_all, *c_all, *outs, *douts, *dh, *dc, *dgt, *demb, *demb_parts;
    int *sync_f, *sync_b;               // {barrier counter, abort flag} of the one-launch recurrence kernels (zeroed with the forward / backward regions)
};
// one slim conv2d + batch_norm + relu layer (CMPCv5_BiLSTM_model.py:190-251): the convolution's output, its statistics, the backward scratch
struct BnLayer {
    std::string scope; int C, Cpad, R, dt, ldpre;
    void *pre, *dpre; double* sums; float *mr, *means, *mm, *mv;      // mm / mv: moving statistics (device, inside e->bn_state)
};

}  // namespace

struct cmpc_engine_s {
    cmpc_cfg cfg;
    int B, T, N, R, C, Cp, M, Mp, G, Gp, P, Pp, Tp, RNN, dt, esz, h, w, H, W, V;
    // parameters
    std::vector<ParamSpec> specs;
    std::unordered_map<std::string, int> pindex;
    int64_t total = 0;
    float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    // packed operands
    std::vector<cmpc_pack_desc> descs;
    std::unordered_map<std::string, Operand> ops;
    size_t arena_bytes = 0;
    char* arena = nullptr;
    cmpc_pack_desc* descs_dev = nullptr; int* tile_prefix_dev = nullptr; int* tile_desc_dev = nullptr;
    int ndesc = 0, total_tiles = 0, stage0_ndesc = 0, stage0_tiles = 0;
    cmpc_adam_seg* segs_dev = nullptr; int nseg = 0;
    int64_t step = 0;
    // workspace: [zero_fwd | zero_bwd | rest]
    char* ws = nullptr; size_t ws_bytes = 0, zf_bytes = 0, zb_bytes = 0;
    void* zero_page = nullptr;
    // streams / events
    hipStream_t lane[3] = {nullptr, nullptr, nullptr};       // the streams the three level / module chains run on (2 lanes: lane[2] = lane[0])
    hipStream_t own_lane[3] = {nullptr, nullptr, nullptr};   // the streams this handle created
    std::vector<hipEvent_t> evpool; size_t evnext = 0;
    hipEvent_t ev_opt0 = nullptr, ev_opt1 = nullptr; bool opt_pending = false;
    std::vector<std::pair<char*, size_t>> guards;
    hipEvent_t user_bwd_levels = nullptr;     // caller's event, recorded when the levels' backward is complete (cmpc_set_bwd_levels_event)
    // buffers
    void* spatial = nullptr;
    void* spatial1 = nullptr;        // the grid with a constant 1 in channel 8: A operand of the products whose row 8 is the bias gradient
    bool mutan_bias_row = false;     // every vis_trans head's bias sits right behind its DW: its gradient is row C + 8 of the spatial-rows product
    float *wf, *wf_rstd, *mask;
    float *h1, *lg, *parse, *dlg, *dh1;
    float *vl, *vl_rstd, *nec, *nec_rstd, *dparse, *dwf, *dvl, *dnec;
    // model variant (cfg.model): CMPC_model = 3 levels (c5, c4, c3), 3 exchange modules per round, 3 ConvLSTM steps; CMPCv5_BiLSTM = 2 / 2 / 2
    bool v5 = false; int nlev = 3, nex = 3, ncl = 3;
    bool vid = false; int Fr = 1, RF = 0, NC = 4;      // model VIDEO: sampled frames, rows of the per-frame maps (Fr * N, batch 1); parser classes (4 or 5)
    float *ac = nullptr, *ac_rstd = nullptr, *dac = nullptr, *ea_t = nullptr, *ones_t = nullptr, *zeros_nt = nullptr;
    LstmDir ldir[2]; int ndir = 1;
    float *outs_bw = nullptr, *douts_bw = nullptr, *wft = nullptr, *dwft = nullptr;     // v5: un-reversed backward outputs / their gradient; tanh(words_feat conv)
    void* hsv = nullptr;                                               // v5 hsv: [R, 64] map (3 channels)
    // v5: ASPP + decoder (CMPCv5_BiLSTM_model.py:190-251)
    int D = 0, Dp = 0, LOW = 0, CATp = 0, C2 = 0, h2 = 0, w2 = 0, R2 = 0;
    BnLayer bn[9];                  // aspp 1x1, 3x3 x3, image level, concat; decoder low level, 3x3 x2  (bn_scopes order)
    float* bn_state = nullptr; std::vector<std::pair<std::string, int64_t>> state_specs; std::unordered_map<std::string, int64_t> state_off; int64_t state_total = 0;
    const void* c2_feed = nullptr; const float* im_feed = nullptr;
    void *cat4, *enc, *deccat, *net1, *net2, *dcat4, *denc, *ddeccat, *dnet1, *dnet2;
    float *pooled, *img, *catsb, *dcatsb, *dimg, *dpooled, *ones_n, *zbias, *dpred, *zeros_bt = nullptr;
    LevelBuf lv[3];                 // c5, c4, c3
    ExgBuf ex[6];                   // c3, c4, c5, c3_2, c4_2, c5_2  (v5: c4, c5, c4_2, c5_2)
    void* de1[3];                   // gradients of the round-1 outputs (c3, c4, c5)
    ClstmStep cl[3]; void* cl_scr; double* cl_bs;
    float *score, *up, *sigm, *loss; int* iu; float *dscore; void* dfused;
    float* scalars; int* nonfinite;      // nonfinite[b]: gradient elements of bucket b the last optimizer step skipped (inf / nan)
    bool have_target = false;
    bool bwd_zeroed = false;      // the backward pass's accumulate-into buffers and the gradient buffer were cleared by the last cmpc_forward
    const int32_t* seq_len_feed = nullptr; const float* target_feed = nullptr;      // caller-owned feeds the backward pass re-reads
    hipStream_t last_main = nullptr; long l0 = 0;
    // Gradient buckets: contiguous ranges of the flat gradient buffer in the order they become final during cmpc_backward
    //   0: exchange modules x6 + ConvLSTM + final score   (after the round-1 exchange backward)
    //   1, 2, 3: pyramid level c5 / c4 / c3 (3 also: the three score_cX heads and the three laterals)   (after the levels' backward)
    //   4: text encoder + parser   (end of the backward pass)
    // The deferred dW products and bias / LayerNorm folds are issued per bucket, and an event per bucket tells a data-parallel
    // caller when it may start that bucket's all-reduce (cmpc_grad_bucket_wait).
    static constexpr int NBK = 5;
    struct Range { int64_t off, count; };
    std::vector<Range> bucket[NBK];
    hipEvent_t bucket_ev[NBK] = {};
    std::vector<std::pair<int, int>> bucket_segs[NBK];       // Adam segment index ranges [lo, hi) of each bucket
    std::vector<std::pair<int, int>> bucket_tiles[NBK];      // pack tile ranges [lo, hi) of the operands packed from each bucket's parameters
    std::vector<cmpc_adam_seg> segs_host; std::vector<int> tile_prefix_host;
    std::vector<cmpc_gemm_tn_args> deferred[NBK];
    // descriptor tables of the two grouped dW launches of a step; 4 cached variants each (the backbone taps alternate between two
    // buffer sets, so the lateral products' operand pointers alternate): a table that matches a cached one is not uploaded again
    void* tn_table[NBK][4] = {}; size_t tn_table_bytes = 0; std::vector<char> tn_shadow[NBK][4]; int tn_victim[NBK] = {};
    bool wgrad_overlap = true;          // issue the levels' / exchanges' dW beside the text encoder's backward chain
    bool lowrank = false;               // the graph's T-deep products through cmpc_lowrank_nn (16-bit storage, T <= 24, Cp = 4 * 2^j <= 1024)
    bool mutan_epilogue = true;         // the Mutan heads' tanh as the epilogue of their GEMM (P is written once, as tanh; mutan_fwd only reads it)
    bool lstm_seq = false;              // the text LSTM's T steps in ONE persistent launch per direction (B <= 8, Cp <= 1024; opt-in, see cmpc_create)
    cmpc_fold_ctx fold;                 // deferred bias / LayerNorm / peephole gradient folds (one launch per backward pass)
    std::vector<cmpc_fold_desc> fold_descs, fold_shadow[NBK]; cmpc_fold_desc* fold_table[NBK] = {}; int fold_shadow_n[NBK];
    std::vector<Tap> taps;
    std::unordered_map<std::string, int> tapindex;
    long launches_step = 0;
    // optional per-launch timing of the dominant kernel family (bench.py's roofline): event pairs around every bf16 MFMA gemm_nt
    bool marks_on = false; std::vector<std::pair<std::string, hipEvent_t>> marks;     // phase-boundary timestamps (cmpc_phase_marks)
    bool timing = false; std::vector<hipEvent_t> tev; std::vector<double> tflops, tbytes; std::vector<std::array<int, 4>> tshape;
};

namespace {
typedef cmpc_engine_s E;
// level / exchange-module names in build order: CMPC_model.py:120-125,271-283; CMPCv5_BiLSTM_model.py:134-137,364-375
const char* LEVELS_V1[3] = {"c5", "c4", "c3"};
const char* EXG_V1[6] = {"c3", "c4", "c5", "c3_2", "c4_2", "c5_2"};
const char* EXG_V5[6] = {"c4", "c5", "c4_2", "c5_2", "", ""};
inline const char* lvn(const cmpc_engine_s* e, int i) { return LEVELS_V1[i]; }                       // v5 uses the first two
inline const char* exn(const cmpc_engine_s* e, int i) { return e->v5 ? EXG_V5[i] : EXG_V1[i]; }      // i < 2 * e->nex
// slim conv2d + BatchNorm scopes in creation order (v5:234-249 then :196-204)
const char* BN_SCOPES[9] = {"aspp/conv_1x1", "aspp/conv_3x3_1", "aspp/conv_3x3_2", "aspp/conv_3x3_3", "aspp/image_level_features/conv_1x1",
                            "aspp/conv_1x1_concat", "decoder/low_level_features/conv_1x1", "decoder/upsampling_logits/conv_3x3_1",
                            "decoder/upsampling_logits/conv_3x3_2"};
enum { BN_A0 = 0, BN_A1 = 1, BN_A2 = 2, BN_A3 = 3, BN_IMG = 4, BN_CAT = 5, BN_LOW = 6, BN_D1 = 7, BN_D2 = 8 };

std::string fmt(const char* f, ...) {
    char buf[256]; va_list ap; va_start(ap, f); vsnprintf(buf, sizeof(buf), f, ap); va_end(ap); return buf;
}

// ------------------------------------------------------------------------------------------
// parameter manifest: names, shapes, creation order of the variables under scope "text_objseg"
// (CMPC_model.py:84; _conv :412-417; lstm :144-156; layer_norm :364,370; util/cell.py:42-66)
// ------------------------------------------------------------------------------------------
void add_param(E* e, const std::string& name, std::initializer_list<int64_t> shape, float wd, float gmult) {
    ParamSpec s; s.name = "text_objseg/" + name; s.rank = (int)shape.size(); s.count = 1; int i = 0;
    for (int64_t d : shape) { s.shape[i++] = d; s.count *= d; }
    for (; i < 4; ++i) s.shape[i] = 1;
    s.off = e->total; s.wd = wd; s.gmult = gmult;
    e->total += (s.count + 3) / 4 * 4;                    // every parameter 16-B aligned
    e->pindex[s.name] = (int)e->specs.size();
    e->specs.push_back(s);
}
void add_conv(E* e, const std::string& name, int k, int cin, int cout) {
    add_param(e, name + "/DW", {k, k, cin, cout}, e->cfg.weight_decay, 1.f);     // 'DW' -> L2 (:433)
    add_param(e, name + "/biases", {cout}, 0.f, 2.f);                            // 'biases' -> gradient x2 (:464-465)
}
void add_ln(E* e, const std::string& scope, int dim) {
    add_param(e, scope + "/beta", {dim}, 0.f, 1.f);
    add_param(e, scope + "/gamma", {dim}, 0.f, 1.f);
}
void build_manifest(E* e) {
    const int C = e->C, M = e->M, R = e->RNN;
    const int hx = (e->v5 && e->cfg.hsv) ? 3 : 0;
    add_param(e, "Variable", {e->V, e->G}, 0.f, 1.f);
    if (e->vid) {                                                                  // vid:105-133: MultiRNNCell([BasicLSTMCell]) under scope "RNN"
        add_param(e, "RNN/multi_rnn_cell/cell_0/basic_lstm_cell/kernel", {e->G + R, 4 * R}, 0.f, 1.f);
        add_param(e, "RNN/multi_rnn_cell/cell_0/basic_lstm_cell/bias", {4 * R}, 0.f, 1.f);
    } else if (!e->v5) {
        add_param(e, "rnn/lstm_cell/kernel", {e->G + R, 4 * R}, 0.f, 1.f);
        add_param(e, "rnn/lstm_cell/bias", {4 * R}, 0.f, 1.f);
    } else {                                                                       // BiLSTM(), v5:159-187
        for (const char* d : {"fw", "bw"}) {
            add_param(e, fmt("bidirectional_rnn/%s/lstm_cell/kernel", d), {e->G + R, 4 * R}, 0.f, 1.f);
            add_param(e, fmt("bidirectional_rnn/%s/lstm_cell/bias", d), {4 * R}, 0.f, 1.f);
        }
        add_conv(e, "words_feat", 1, 2 * R, R);
    }
    add_conv(e, "c5_lateral", 1, e->cfg.vf_dim + hx, C);
    add_conv(e, "c4_lateral", 1, e->cfg.c4_dim + hx, C);
    if (!e->v5) add_conv(e, "c3_lateral", 1, e->cfg.c3_dim, C);
    add_conv(e, "words_parse_1", 1, R, e->P);
    add_conv(e, "words_parse_2", 1, e->P, e->NC);
    for (int li = 0; li < e->nlev; ++li) {
        const char* lv = lvn(e, li);
        for (int hd = 1; hd <= 5; ++hd) {
            add_conv(e, fmt("vis_trans_%s_head%d", lv, hd), 1, C + 8, C);
            add_conv(e, fmt("lang_trans_%s_head%d", lv, hd), 1, R, C);
        }
        if (e->vid) {                                                              // build_temp_graph / build_temp_ctx, vid:458-530
            add_conv(e, fmt("tg_vtrans_%s", lv), 1, C, C);
            add_conv(e, fmt("tg_ltrans_%s", lv), 1, R, R);
            add_conv(e, fmt("tg_query_%s", lv), 1, C, C);
            add_conv(e, fmt("tg_key_%s", lv), 1, C, C);
            add_ln(e, fmt("gconv_feat_ln_temp_graph_%s", lv), C);
            add_conv(e, fmt("gconv_update_temp_graph_%s", lv), 1, C, C);
            add_ln(e, fmt("gconv_update_ln_temp_graph_%s", lv), C);
            add_conv(e, fmt("mm_trans_%s", lv), 1, C, C);
            add_conv(e, fmt("ctx_trans_%s", lv), 1, C, C);
        }
        add_conv(e, fmt("words_trans_%s", lv), 1, R, R);
        add_conv(e, fmt("spa_graph_trans2_%s", lv), 1, C, C);
        add_ln(e, fmt("gconv_feat_ln_spa_graph_%s", lv), C);
        add_conv(e, fmt("gconv_update_spa_graph_%s", lv), 1, C, C);
        add_ln(e, fmt("gconv_update_ln_spa_graph_%s", lv), C);
        add_conv(e, fmt("fusion_%s", lv), 1, (e->vid ? 3 : 2) * C + R + 8, M);        // vid:398-400: [lateral | spatial graph | temporal context | lang | grid]
    }
    for (int li = 0; li < e->nlev; ++li) add_conv(e, fmt("score_%s", lvn(e, li)), 3, M, 1);
    for (int xi = 0; xi < 2 * e->nex; ++xi) {
        const char* x = exn(e, xi);
        add_conv(e, fmt("spa_graph_key_%sgv_f1", x), 1, M, M);
        add_conv(e, fmt("lang_query_%sgv_f1", x), 1, R, M);
        add_conv(e, fmt("gv_lang_%sgv_f1", x), 1, M + R, M);
        add_conv(e, fmt("lang_feat_%s_f1", x), 1, M, M);
        add_conv(e, fmt("trans_feat_%s_f1", x), 1, M, M);
        if (!e->v5) {
            add_conv(e, fmt("lang_feat_%s_f2", x), 1, M, M);
            add_conv(e, fmt("trans_feat_%s_f2", x), 1, M, M);
        }
    }
    const std::string pre = "rnn/conv_lstm_cell";
    add_param(e, pre + "/kernel", {1, 1, 2 * M, 4 * M}, 0.f, 1.f);
    add_param(e, pre + "/W_ci", {e->h, e->w, M}, 0.f, 1.f);
    add_param(e, pre + "/W_cf", {e->h, e->w, M}, 0.f, 1.f);
    add_ln(e, pre + "/LayerNorm", M);
    add_ln(e, pre + "/LayerNorm_1", M);
    add_ln(e, pre + "/LayerNorm_2", M);
    add_param(e, pre + "/W_co", {e->h, e->w, M}, 0.f, 1.f);
    add_ln(e, pre + "/LayerNorm_3", M);
    add_ln(e, pre + "/LayerNorm_4", M);
    if (!e->v5) { add_conv(e, "score", 3, M, 1); return; }
    // atrous_spatial_pyramid_pooling + decoder (v5:190-251): slim conv2d `weights` (in reg_var_list: name ends in 'weights', v5:530) +
    // BatchNorm beta / gamma; the moving statistics are non-trainable state (cmpc_state_*)
    const int D = e->D, ksz[9] = {1, 3, 3, 3, 1, 1, 1, 3, 3}, cin[9] = {M, M, M, M, M, 5 * D, e->C2, D + e->LOW, D}, cout[9] = {D, D, D, D, D, D, e->LOW, D, D};
    for (int i = 0; i < 9; ++i) {
        add_param(e, std::string(BN_SCOPES[i]) + "/weights", {ksz[i], ksz[i], cin[i], cout[i]}, e->cfg.weight_decay, 1.f);
        add_ln(e, std::string(BN_SCOPES[i]) + "/BatchNorm", cout[i]);
        for (const char* sfx : {"moving_mean", "moving_variance"}) {
            const std::string n = std::string("text_objseg/") + BN_SCOPES[i] + "/BatchNorm/" + sfx;
            e->state_off[n] = e->state_total; e->state_specs.push_back({n, cout[i]}); e->state_total += (cout[i] + 3) / 4 * 4;
        }
    }
    add_param(e, "decoder/upsampling_logits/conv_1x1/weights", {1, 1, D, 1}, e->cfg.weight_decay, 1.f);      // v5:205
    add_param(e, "decoder/upsampling_logits/conv_1x1/biases", {1}, 0.f, 2.f);
}

inline const ParamSpec& spec(const E* e, const std::string& name) { return e->specs[e->pindex.at("text_objseg/" + name)]; }
inline int64_t poff(const E* e, const std::string& name) { return spec(e, name).off; }
inline float* pptr(const E* e, const std::string& name, int64_t elem = 0) { return e->params + poff(e, name) + elem; }
inline float* gptr(const E* e, const std::string& name, int64_t elem = 0) { return e->grads + poff(e, name) + elem; }

// ------------------------------------------------------------------------------------------
// operand plan: zero-padded packed copies of the masters, '<key>.t' = [Np][Kp] (output-major, forward) and
// '<key>.n' = [Kp][Np] (input-major, dX); concatenations of the reference are K-segments of one operand
// ------------------------------------------------------------------------------------------
typedef std::vector<std::array<int, 3>> Segs;      // (src, len, dst)

Operand& new_operand(E* e, const std::string& key, int dt, int rows, int ld) {
    const int esz = dt == DT_F32 ? 4 : 2;
    Operand op{e->arena_bytes, dt, rows, ld};
    e->arena_bytes += up256((size_t)rows * ld * esz);
    return e->ops[key] = op;
}
void add_desc(E* e, const Operand& op, const std::string& pname, int ld_src, int transpose, int row0, int col0, int rows, int cols,
              const Segs& ks, const Segs& ns, int64_t src_elem_off = 0) {
    const int esz = op.dt == DT_F32 ? 4 : 2;
    cmpc_pack_desc d; memset(&d, 0, sizeof(d));
    d.src_off = poff(e, pname) + src_elem_off; d.ld_src = ld_src;
    d.dst_off = (int64_t)op.off + ((int64_t)row0 * op.ld + col0) * esz;
    d.dst_dt = op.dt; d.transpose = transpose;
    d.rows = rows; d.cols = cols; d.ld_dst = op.ld;
    d.nks = (int)ks.size();
    for (size_t i = 0; i < ks.size(); ++i) { d.ks_src[i] = ks[i][0]; d.ks_len[i] = ks[i][1]; d.ks_dst[i] = ks[i][2]; }
    d.nns = (int)ns.size();
    for (size_t i = 0; i < ns.size(); ++i) { d.ns_src[i] = ns[i][0]; d.ns_len[i] = ns[i][1]; d.ns_dst[i] = ns[i][2]; }
    e->descs.push_back(d);
}
void linear(E* e, const std::string& key, const std::string& pname, int dt, int K, int N, int Kp, int Np, bool fwd = true, bool bwd = true,
            Segs ks = {}, Segs ns = {}) {
    if (ks.empty()) ks = {{0, K, 0}};
    if (ns.empty()) ns = {{0, N, 0}};
    const ParamSpec& s = spec(e, pname);
    const int ld_src = (int)s.shape[s.rank - 1];
    if (fwd) { Operand& op = new_operand(e, key + ".t", dt, Np, Kp); add_desc(e, op, pname, ld_src, 1, 0, 0, Np, Kp, ks, ns); }
    if (bwd) { Operand& op = new_operand(e, key + ".n", dt, Kp, Np); add_desc(e, op, pname, ld_src, 0, 0, 0, Kp, Np, ks, ns); }
}
// a 3x3 convolution's HWIO weights [3, 3, cin, cout] as implicit-GEMM operands (cmpc_conv_nhwc): '<key>.t' = [Np][9 * Kp], tap-major
// K (forward); '<key>.n' = [Kp][9 * Np] with the taps REVERSED (the input gradient is the same convolution of dY with the flipped,
// transposed kernel: dX[p] = sum_t dY[p - off_t] W_t^T, and -off_t = off_{8 - t})
void conv3(E* e, const std::string& key, const std::string& pname, int dt, int cin, int cout, int Kp, int Np, Segs ks = {}) {
    if (ks.empty()) ks = {{0, cin, 0}};
    Operand opt = new_operand(e, key + ".t", dt, Np, 9 * Kp);
    Operand opn = new_operand(e, key + ".n", dt, Kp, 9 * Np);
    for (int t = 0; t < 9; ++t) {
        add_desc(e, opt, pname, cout, 1, 0, t * Kp, Np, Kp, ks, {{0, cout, 0}}, (int64_t)t * cin * cout);
        add_desc(e, opn, pname, cout, 0, 0, (8 - t) * Np, Kp, Np, ks, {{0, cout, 0}}, (int64_t)t * cin * cout);
    }
}
void plan_operands(E* e) {
    const int C = e->C, M = e->M, R = e->RNN, G = e->G, P = e->P, Cp = e->Cp, Mp = e->Mp, Gp = e->Gp, Pp = e->Pp;
    const int V = e->dt, L = DT_F32;
    // text LSTM(s): kernel [G+R, 4R], gates i,j,f,o -> padded gate blocks of Cp
    Segs gate_ns; for (int g = 0; g < 4; ++g) gate_ns.push_back({g * R, R, g * Cp});
    for (int d = 0; d < e->ndir; ++d) {
        const LstmDir& D = e->ldir[d];
        linear(e, D.key, D.pk, L, G + R, 4 * R, Gp + Cp, 4 * Cp, true, true, {{0, G, 0}, {G, R, Gp}}, gate_ns);
        Operand& ob = new_operand(e, D.key + ".b", L, 1, 4 * Cp); add_desc(e, ob, D.pb, 4 * R, 0, 0, 0, 1, 4 * Cp, {{0, 1, 0}}, gate_ns);
    }
    if (e->v5) linear(e, "wfeat", "words_feat/DW", L, 2 * R, R, 2 * Cp, Cp, true, true, {{0, R, 0}, {R, R, Cp}});      // [fw | bw] -> R (v5:182)
    linear(e, "parse1", "words_parse_1/DW", L, R, P, Cp, Pp);
    linear(e, "parse2", "words_parse_2/DW", L, P, e->NC, Pp, 64);
    e->stage0_ndesc = (int)e->descs.size();       // what the text encoder + parser read: packed (and published) first
    const int cins[3] = {e->cfg.vf_dim, e->cfg.c4_dim, e->cfg.c3_dim};
    for (int i = 0; i < e->nlev; ++i) {
        // hsv: the three HSV channels are a K-segment of their own (rows cin .. cin+2 of the weight -> k = pad64(cin) ..), read from e->hsv
        Segs ks = {{0, cins[i], 0}};
        const bool hx = e->v5 && e->cfg.hsv;
        if (hx) ks.push_back({cins[i], 3, pad64(cins[i])});
        linear(e, fmt("lat_%s", lvn(e, i)), fmt("%s_lateral/DW", lvn(e, i)), V, cins[i] + (hx ? 3 : 0), C, pad64(cins[i]) + (hx ? 64 : 0), Cp, true, e->cfg.conv5 != 0, ks);
    }
    for (int li = 0; li < e->nlev; ++li) {
        const char* lv = lvn(e, li);
        // mutan: five heads side by side; forward operand [5Cp][Cp+64] (k: C visual rows, then the 8 spatial rows)
        Operand opt = new_operand(e, fmt("mutan_%s.t", lv), V, 5 * Cp, Cp + 64);
        Operand opn = new_operand(e, fmt("mutan_%s.n", lv), V, Cp, 5 * Cp);
        Operand lgt = new_operand(e, fmt("mlang_%s.t", lv), L, 5 * Cp, Cp);
        Operand lgn = new_operand(e, fmt("mlang_%s.n", lv), L, Cp, 5 * Cp);
        Operand opb = new_operand(e, fmt("mutan_%s.b", lv), L, 1, 5 * Cp);
        Operand lgb = new_operand(e, fmt("mlang_%s.b", lv), L, 1, 5 * Cp);
        for (int hd = 0; hd < 5; ++hd) {
            const std::string pn = fmt("vis_trans_%s_head%d/DW", lv, hd + 1), pl = fmt("lang_trans_%s_head%d/DW", lv, hd + 1);
            add_desc(e, opt, pn, C, 1, hd * Cp, 0, Cp, Cp + 64, {{0, C, 0}, {C, 8, Cp}}, {{0, C, 0}});
            add_desc(e, opn, pn, C, 0, 0, hd * Cp, Cp, Cp, {{0, C, 0}}, {{0, C, 0}});
            add_desc(e, lgt, pl, C, 1, hd * Cp, 0, Cp, Cp, {{0, R, 0}}, {{0, C, 0}});
            add_desc(e, lgn, pl, C, 0, 0, hd * Cp, Cp, Cp, {{0, R, 0}}, {{0, C, 0}});
            add_desc(e, opb, fmt("vis_trans_%s_head%d/biases", lv, hd + 1), C, 0, 0, hd * Cp, 1, Cp, {{0, 1, 0}}, {{0, C, 0}});
            add_desc(e, lgb, fmt("lang_trans_%s_head%d/biases", lv, hd + 1), C, 0, 0, hd * Cp, 1, Cp, {{0, 1, 0}}, {{0, C, 0}});
        }
        linear(e, fmt("wtrans_%s", lv), fmt("words_trans_%s/DW", lv), L, R, R, Cp, Cp);
        linear(e, fmt("t2_%s", lv), fmt("spa_graph_trans2_%s/DW", lv), L, C, C, Cp, Cp);
        linear(e, fmt("gupd_%s", lv), fmt("gconv_update_spa_graph_%s/DW", lv), V, C, C, Cp, Cp);
        const std::string fus = fmt("fusion_%s/DW", lv);
        if (!e->vid) {
            linear(e, fmt("fus_%s", lv), fus, V, 2 * C + R + 8, M, 2 * Cp + 64, Mp, true, true, {{0, C, 0}, {C, C, Cp}, {2 * C + R, 8, 2 * Cp}});
            linear(e, fmt("fusl_%s", lv), fus, L, R, M, Cp, Mp, true, true, {{2 * C, R, 0}});
        } else {
            // three visual K-segments; the tiled language vector is a per-sample bias and the grid a per-position bias (both float32 products)
            linear(e, fmt("fus_%s", lv), fus, V, 3 * C + R + 8, M, 3 * Cp, Mp, true, true, {{0, C, 0}, {C, C, Cp}, {2 * C, C, 2 * Cp}});
            linear(e, fmt("fusl_%s", lv), fus, L, R, M, Cp, Mp, true, true, {{3 * C, R, 0}});
            linear(e, fmt("fussp_%s", lv), fus, V, 8, M, 64, Mp, true, false, {{3 * C + R, 8, 0}});
            linear(e, fmt("tgv_%s", lv), fmt("tg_vtrans_%s/DW", lv), L, C, C, Cp, Cp);
            linear(e, fmt("tgl_%s", lv), fmt("tg_ltrans_%s/DW", lv), L, R, R, Cp, Cp);
            linear(e, fmt("tgq_%s", lv), fmt("tg_query_%s/DW", lv), L, C, C, Cp, Cp);
            linear(e, fmt("tgk_%s", lv), fmt("tg_key_%s/DW", lv), L, C, C, Cp, Cp);
            linear(e, fmt("tgu_%s", lv), fmt("gconv_update_temp_graph_%s/DW", lv), L, C, C, Cp, Cp);
            linear(e, fmt("mmt_%s", lv), fmt("mm_trans_%s/DW", lv), L, C, C, Cp, Cp);
            linear(e, fmt("ctxt_%s", lv), fmt("ctx_trans_%s/DW", lv), L, C, C, Cp, Cp);
        }
    }
    for (int xi = 0; xi < 2 * e->nex; ++xi) {
        const char* x = exn(e, xi);
        linear(e, fmt("key_%s", x), fmt("spa_graph_key_%sgv_f1/DW", x), L, M, M, Mp, Mp);
        linear(e, fmt("query_%s", x), fmt("lang_query_%sgv_f1/DW", x), L, R, M, Cp, Mp);
        linear(e, fmt("gv_%s", x), fmt("gv_lang_%sgv_f1/DW", x), L, M + R, M, Mp + Cp, Mp, true, true, {{0, M, 0}, {M, R, Mp}});
        for (const char* f : {"f1", "f2"}) {
            if (e->v5 && f[1] == '2') continue;
            linear(e, fmt("lfeat_%s_%s", x, f), fmt("lang_feat_%s_%s/DW", x, f), L, M, M, Mp, Mp);
            linear(e, fmt("tfeat_%s_%s", x, f), fmt("trans_feat_%s_%s/DW", x, f), V, M, M, Mp, Mp);
        }
    }
    Segs cl_ns; for (int g = 0; g < 4; ++g) cl_ns.push_back({g * M, M, g * Mp});
    linear(e, "clstm", "rnn/conv_lstm_cell/kernel", V, 2 * M, 4 * M, 2 * Mp, 4 * Mp, true, true, {{0, M, 0}, {M, M, Mp}}, cl_ns);
    if (!e->v5) return;
    // ASPP + decoder (v5:190-251)
    const int D = e->D, Dp = e->Dp, LOW = e->LOW, CATp = e->CATp, C2 = e->C2;
    auto w = [&](int i) { return std::string(BN_SCOPES[i]) + "/weights"; };
    linear(e, "aspp0", w(BN_A0), V, M, D, Mp, Dp);
    for (int k = 1; k <= 3; ++k) conv3(e, fmt("aspp%d", k), w(BN_A0 + k), V, M, D, Mp, Dp);
    linear(e, "aspp_img", w(BN_IMG), L, M, D, Mp, Dp);
    {   // conv_1x1_concat: K = [four branches | image level]; the image-level part is a per-sample bias (its input is constant over the map)
        Segs ks; for (int k = 0; k < 4; ++k) ks.push_back({k * D, D, k * Dp});
        linear(e, "aspp_cat", w(BN_CAT), V, 5 * D, D, 4 * Dp, Dp, true, true, ks);
        linear(e, "aspp_catl", w(BN_CAT), L, D, D, Dp, Dp, true, true, {{4 * D, D, 0}});
    }
    linear(e, "dec_low", w(BN_LOW), V, C2, LOW, pad64(C2), 64, true, false);
    conv3(e, "dec1", w(BN_D1), V, D + LOW, D, CATp, Dp, {{0, D, 0}, {D, LOW, Dp}});      // K = [upsampled encoder output | low-level features]
    conv3(e, "dec2", w(BN_D2), V, D, D, Dp, Dp);
}
inline const void* opp(const E* e, const std::string& key, int row = 0, int col = 0) {
    const Operand& o = e->ops.at(key);
    const int esz = o.dt == DT_F32 ? 4 : 2;
    return e->arena + o.off + ((size_t)row * o.ld + col) * esz;
}

int upload_tables(E* e) {
    const int n = (int)e->descs.size();
    e->ndesc = n;
    std::vector<int> pref(n + 1, 0), tdesc;
    for (int i = 0; i < n; ++i) {
        const cmpc_pack_desc& d = e->descs[i];
        const int kp = d.transpose ? d.cols : d.rows, np = d.transpose ? d.rows : d.cols;
        const int t = ((kp + 63) / 64) * ((np + 127) / 128);
        pref[i + 1] = pref[i] + t;
        for (int j = 0; j < t; ++j) tdesc.push_back(i);
    }
    e->total_tiles = pref[n]; e->stage0_tiles = pref[e->stage0_ndesc];
    HCK(hipMalloc(&e->descs_dev, sizeof(cmpc_pack_desc) * n));
    HCK(hipMemcpy(e->descs_dev, e->descs.data(), sizeof(cmpc_pack_desc) * n, hipMemcpyHostToDevice));
    HCK(hipMalloc(&e->tile_prefix_dev, sizeof(int) * (n + 1)));
    HCK(hipMemcpy(e->tile_prefix_dev, pref.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    HCK(hipMalloc(&e->tile_desc_dev, sizeof(int) * tdesc.size()));
    HCK(hipMemcpy(e->tile_desc_dev, tdesc.data(), sizeof(int) * tdesc.size(), hipMemcpyHostToDevice));
    std::vector<cmpc_adam_seg> segs;                       // <= 8192 elements each, inside one parameter
    for (const ParamSpec& s : e->specs) {
        // freeze_bn (v5:528-529): 'beta' / 'gamma' variables are not in the optimizer's list -> zero gradient, zero moments, no update
        const bool frozen = e->cfg.freeze_bn && (s.name.find("beta") != std::string::npos || s.name.find("gamma") != std::string::npos);
        for (int64_t o = 0; o < s.count; o += 8192)
            segs.push_back(cmpc_adam_seg{s.off + o, (int)std::min<int64_t>(8192, s.count - o), frozen ? 0.f : s.wd, frozen ? 0.f : s.gmult});
    }
    e->nseg = (int)segs.size();
    e->segs_host = segs; e->tile_prefix_host = pref;
    HCK(hipMalloc(&e->segs_dev, sizeof(cmpc_adam_seg) * segs.size()));
    HCK(hipMemcpy(e->segs_dev, segs.data(), sizeof(cmpc_adam_seg) * segs.size(), hipMemcpyHostToDevice));
    return CMPC_OK;
}

// ------------------------------------------------------------------------------------------
// workspace plan
// ------------------------------------------------------------------------------------------
void tap(E* e, const std::string& name, void* p, int dt, std::initializer_list<int64_t> shape) {
    Tap t; t.name = name; t.ptr = p; t.dt = dt; t.rank = (int)shape.size(); int i = 0;
    for (int64_t d : shape) t.shape[i++] = d;
    for (; i < 4; ++i) t.shape[i] = 1;
    auto it = e->tapindex.find(name);
    if (it != e->tapindex.end()) e->taps[it->second] = t;
    else { e->tapindex[name] = (int)e->taps.size(); e->taps.push_back(t); }
}

// zf: zeroed at the start of every forward; zb: zeroed at the start of every backward; g: written before read
void plan_workspace(E* e, Bump& zf, Bump& zb, Bump& g) {
    const int B = e->B, T = e->T, N = e->N, R = e->R, Cp = e->Cp, Mp = e->Mp, Gp = e->Gp, Pp = e->Pp, Tp = e->Tp, H = e->H, W = e->W;
    const size_t es = e->esz, F = 4, D = 8;
    const int vd = e->dt;
    const int RL = e->vid ? e->RF : R, BL = e->vid ? e->Fr : B;        // rows / "samples" of the per-frame maps (lateral, Mutan) of the video model
    e->zero_page = zf.take(256);
    e->spatial = g.take((size_t)RL * 64 * es);
    e->spatial1 = g.take((size_t)RL * 64 * es);
    // ---- text encoder / parser / language pools
    for (int d = 0; d < e->ndir; ++d) {
        LstmDir& D = e->ldir[d];
        D.words_tb = (int*)g.take((size_t)T * B * 4);
        D.emb = (float*)g.take((size_t)T * B * Gp * F);
        D.xg = (float*)g.take((size_t)T * B * 4 * Cp * F);
        D.gates = (float*)g.take((size_t)T * B * 4 * Cp * F);
        D.h_all = (float*)zf.take((size_t)(T + 1) * B * Cp * F);
        D.c_all = (float*)zf.take((size_t)(T + 1) * B * Cp * F);
        D.outs = (float*)g.take((size_t)B * T * Cp * F);
        D.douts = (float*)g.take((size_t)B * T * Cp * F);
        D.dh = (float*)zb.take((size_t)B * Cp * F);
        D.dc = (float*)zb.take((size_t)B * Cp * F);
        D.dgt = (float*)g.take((size_t)T * B * 4 * Cp * F);
        D.demb = (float*)g.take((size_t)T * B * Gp * F); D.demb_parts = (float*)g.take((size_t)8 * T * B * Gp * F);
        D.sync_f = (int*)zf.take(256); D.sync_b = (int*)zb.take(256);
        tap(e, fmt("lstm_sync_%d", d), D.sync_f, 3, {2}); tap(e, fmt("lstm_sync_bwd_%d", d), D.sync_b, 3, {2});
    }
    if (e->v5) {
        e->outs_bw = (float*)g.take((size_t)B * T * Cp * F); e->douts_bw = (float*)g.take((size_t)B * T * Cp * F);
        e->wft = (float*)g.take((size_t)B * T * Cp * F); e->dwft = (float*)g.take((size_t)B * T * Cp * F);
        if (e->cfg.hsv) e->hsv = g.take((size_t)R * 64 * es);
        tap(e, "bilstm_fw", e->ldir[0].outs, 0, {B * T, Cp}); tap(e, "bilstm_bw", e->outs_bw, 0, {B * T, Cp});
        if (e->cfg.hsv) tap(e, "hsv", e->hsv, vd, {R, 64});
    }
    e->wf = (float*)g.take((size_t)B * T * Cp * F);
    e->wf_rstd = (float*)g.take((size_t)B * T * F);
    e->mask = (float*)g.take((size_t)B * T * F);
    e->h1 = (float*)g.take((size_t)B * T * Pp * F);
    e->lg = (float*)g.take((size_t)B * T * 64 * F);
    e->parse = (float*)g.take((size_t)B * T * 8 * F);
    e->dlg = (float*)g.take((size_t)B * T * 64 * F);
    e->dh1 = (float*)g.take((size_t)B * T * Pp * F);
    e->vl = (float*)g.take((size_t)B * Cp * F); e->vl_rstd = (float*)g.take((size_t)B * F);
    e->nec = (float*)g.take((size_t)B * Cp * F); e->nec_rstd = (float*)g.take((size_t)B * F);
    e->dparse = (float*)zb.take((size_t)B * T * 8 * F);
    e->dwf = (float*)zb.take((size_t)B * T * Cp * F);
    e->dvl = (float*)g.take((size_t)B * Cp * F);
    e->dnec = (float*)g.take((size_t)B * Cp * F);
    tap(e, "words_feat", e->wf, 0, {B * T, Cp}); tap(e, "seq_mask", e->mask, 0, {B * T}); tap(e, "words_parse", e->parse, 0, {B * T, e->NC});
    tap(e, "valid_lang", e->vl, 0, {B, Cp}); tap(e, "nec_lang", e->nec, 0, {B, Cp}); tap(e, "spatial", e->spatial, vd, {RL, 64});
    if (e->vid) {
        e->ac = (float*)g.take((size_t)B * Cp * F); e->ac_rstd = (float*)g.take(256); e->dac = (float*)g.take((size_t)B * Cp * F);
        e->ea_t = (float*)g.take((size_t)e->Fr * Cp * F);
        e->ones_t = (float*)g.take((size_t)64 * F);                     // filled with 1 at create: pr / mask of the temporal-context softmax over the Fr graph nodes
        e->zeros_nt = (float*)g.take((size_t)N * Tp * F);               // never written: the absent column-softmax gradient of that softmax
        tap(e, "ac_lang", e->ac, 0, {B, Cp});
    }
    // ---- pyramid levels
    const int cins[3] = {e->cfg.vf_dim, e->cfg.c4_dim, e->cfg.c3_dim};
    const int nch = (N + 63) / 64;
    for (int i = 0; i < e->nlev; ++i) {
        LevelBuf& L = e->lv[i]; const char* n = lvn(e, i);
        L.cin = cins[i];
        L.X0 = g.take((size_t)RL * Cp * es); L.lat_rstd = (float*)g.take((size_t)RL * F);
        L.X0t = e->v5 ? g.take((size_t)R * Cp * es) : nullptr;
        L.g = (float*)g.take((size_t)BL * 5 * Cp * F); L.P = g.take((size_t)RL * 5 * Cp * es);
        L.X1 = g.take((size_t)RL * Cp * es); L.mut_rstd = (float*)g.take((size_t)RL * F);
        L.Wd = (float*)zf.take((size_t)B * Tp * Cp * F);
        L.PTf = (float*)g.take((size_t)B * Tp * Cp * F); L.PT = g.take((size_t)B * Tp * Cp * es);
        L.PTt = g.take((size_t)Cp * B * Tp * es);
        L.k0s = (float*)g.take((size_t)B * Tp * F);
        L.A0 = (float*)g.take((size_t)B * N * Tp * F); L.pr = (float*)g.take((size_t)B * T * F);
        L.gw_w = (float*)g.take((size_t)B * N * Tp * F); L.gw_v = (float*)g.take((size_t)B * N * Tp * F);
        L.gw_w_t = g.take((size_t)B * N * Tp * es); L.gw_v_t = g.take((size_t)B * N * Tp * es);
        L.gsc = (float*)g.take((size_t)B * nch * 128 * F);
        L.Ztf = (float*)zf.take((size_t)B * Cp * Tp * F); L.Zt = g.take((size_t)B * Cp * Tp * es);
        L.Y = g.take((size_t)R * Cp * es); L.sums1 = (double*)g.take((size_t)B * STAT_PARTS * 2 * D);
        L.G = g.take((size_t)R * Cp * es); L.U = g.take((size_t)R * Cp * es); L.sums2 = (double*)g.take((size_t)B * STAT_PARTS * 2 * D);
        L.X2 = g.take((size_t)R * Cp * es); L.rrow = (float*)g.take((size_t)R * F);
        L.sb = (float*)g.take((size_t)B * Mp * F); L.F = g.take((size_t)R * Mp * es);
        L.score = (float*)g.take((size_t)B * e->h * e->w * F); L.up = (float*)g.take((size_t)B * H * W * F);
        L.loss = (float*)zf.take((size_t)B * F); L.iu = (int*)zf.take((size_t)2 * B * 4);
        // backward
        L.dfus = g.take((size_t)R * Mp * es); L.dscore = (float*)g.take((size_t)B * e->h * e->w * F);
        L.dpre = g.take((size_t)R * Mp * es); L.dsb = (float*)zb.take((size_t)B * Mp * F);
        L.dX1 = (e->vid ? zb : g).take((size_t)RL * Cp * es); L.dX2 = g.take((size_t)R * Cp * es); L.dvl = (float*)g.take((size_t)B * Cp * F);
        L.bs = (double*)g.take((size_t)B * STAT_PARTS * 2 * D);
        L.dU = g.take((size_t)R * Cp * es); L.dG = g.take((size_t)R * Cp * es); L.dY = g.take((size_t)R * Cp * es);
        L.Zf = (float*)(e->lowrank ? zf : zb).take((size_t)B * Tp * Cp * F); L.Z = g.take((size_t)B * Tp * Cp * es);     // lowrank: Z = gw_v^T . X1 is a forward product
        L.dgw_w = (float*)g.take((size_t)B * N * Tp * F);
        L.dZf = (float*)zb.take((size_t)B * Tp * Cp * F); L.dZ = g.take((size_t)B * Tp * Cp * es);
        L.dZtf = (float*)zb.take((size_t)B * Cp * Tp * F); L.dZt = g.take((size_t)B * Cp * Tp * es);
        L.dgw_v = (float*)g.take((size_t)B * N * Tp * F);
        L.dA0 = (float*)g.take((size_t)B * N * Tp * F); L.dA0_t = g.take((size_t)B * N * Tp * es);
        L.dpr = (float*)g.take((size_t)B * T * F); L.gsc2 = (float*)g.take((size_t)B * nch * 128 * F);
        L.dPT = (float*)zb.take((size_t)B * Tp * Cp * F); L.dk0s = (float*)zb.take((size_t)B * Tp * F);
        L.dWd = (float*)g.take((size_t)B * Tp * Cp * F); L.dwf = (float*)g.take((size_t)B * T * Cp * F);
        L.dg = (float*)zb.take((size_t)BL * 5 * Cp * F);
        L.dX0 = g.take((size_t)RL * Cp * es); L.dV = g.take((size_t)RL * Cp * es);
        if (e->cfg.conv5) L.dfeat = g.take((size_t)RL * L.cin * es);
        if (e->vid) {
            const int Fr = e->Fr, mid = Fr / 2;
            L.X1m = (char*)L.X1 + (size_t)mid * N * Cp * es; L.X0m = (char*)L.X0 + (size_t)mid * N * Cp * es;
            L.g1 = (float*)g.take((size_t)5 * Cp * F);
            L.lt = (float*)g.take((size_t)Cp * F); L.kqv = (float*)g.take((size_t)Cp * F);
            L.tlog = (float*)g.take((size_t)Fr * N * F); L.tatt = (float*)g.take((size_t)Fr * N * F);
            L.TG = (float*)zf.take((size_t)Tp * Cp * F); L.q = (float*)g.take((size_t)Tp * Cp * F); L.k = (float*)g.take((size_t)Tp * Cp * F);
            L.adj = (float*)g.take(256); L.TY = (float*)g.take((size_t)Tp * Cp * F); L.TG1 = (float*)g.take((size_t)Tp * Cp * F);
            L.TU = (float*)g.take((size_t)Tp * Cp * F); L.TGN = (float*)g.take((size_t)Tp * Cp * F); L.rrow_t = (float*)g.take((size_t)Tp * F);
            L.tsums1 = (double*)g.take((size_t)STAT_PARTS * 2 * D); L.tsums2 = (double*)g.take((size_t)STAT_PARTS * 2 * D); L.tbs = (double*)g.take((size_t)STAT_PARTS * 2 * D);
            L.ctv = (float*)g.take((size_t)Tp * Cp * F); L.ctPTf = (float*)g.take((size_t)Tp * Cp * F); L.ctPT = g.take((size_t)Tp * Cp * es);
            L.ctPTt = g.take((size_t)Cp * Tp * es); L.ctk0s = (float*)g.take((size_t)Tp * F);
            L.ctA0 = (float*)g.take((size_t)N * Tp * F); L.ctA = (float*)g.take((size_t)N * Tp * F); L.ctGv = (float*)g.take((size_t)N * Tp * F);
            L.ctA_t = g.take((size_t)N * Tp * es); L.ctGv_t = g.take((size_t)N * Tp * es); L.ctsc = (float*)g.take((size_t)nch * 128 * F); L.ctsc2 = (float*)g.take((size_t)nch * 128 * F);
            L.TGN16 = g.take((size_t)Tp * Cp * es); L.TGNt = g.take((size_t)Cp * Tp * es);
            L.GLO = g.take((size_t)N * Cp * es); L.CTX = g.take((size_t)N * Cp * es); L.ctx_rstd = (float*)g.take((size_t)N * F);
            L.pb = (float*)g.take((size_t)N * Mp * F);
            // backward
            L.dCTX = g.take((size_t)N * Cp * es); L.dGLO = g.take((size_t)N * Cp * es); L.dX0f = g.take((size_t)N * Cp * es);
            L.dTGN = (float*)zb.take((size_t)Tp * Cp * F); L.dTU = (float*)g.take((size_t)Tp * Cp * F); L.dTG1 = (float*)g.take((size_t)Tp * Cp * F);
            L.dTY = (float*)g.take((size_t)Tp * Cp * F); L.dTG = (float*)g.take((size_t)Tp * Cp * F); L.dq = (float*)g.take((size_t)Tp * Cp * F); L.dk = (float*)g.take((size_t)Tp * Cp * F);
            L.dtatt = (float*)g.take((size_t)Fr * N * F); L.dtlog = (float*)g.take((size_t)Fr * N * F);
            L.dkqv = (float*)zb.take((size_t)Cp * F); L.dlt = (float*)g.take((size_t)Cp * F); L.dac = (float*)g.take((size_t)Cp * F);
            L.dctv = (float*)g.take((size_t)Tp * Cp * F); L.dctPT = (float*)zb.take((size_t)Tp * Cp * F); L.dctk0s = (float*)zb.take((size_t)Tp * F);
            L.dctA = (float*)g.take((size_t)N * Tp * F); L.dctA0 = (float*)g.take((size_t)N * Tp * F); L.dctA0_t = g.take((size_t)N * Tp * es);
            L.dprc = (float*)g.take((size_t)64 * F); L.dea = (float*)g.take((size_t)Cp * F);
            tap(e, fmt("mm_%s", n), L.X1, vd, {RL, Cp}); tap(e, fmt("tg_pool_%s", n), L.TG, 0, {Tp, Cp}); tap(e, fmt("tgraph_%s", n), L.TGN, 0, {Tp, Cp});
            tap(e, fmt("temp_ctx_%s", n), L.CTX, vd, {N, Cp});
        }
        tap(e, fmt("lat_%s", n), L.X0, vd, {RL, Cp}); tap(e, fmt("vis_la_sp_%s", n), L.X1, vd, {RL, Cp});
        if (e->cfg.conv5) tap(e, fmt("d%s", n), L.dfeat, vd, {RL, L.cin});
        tap(e, fmt("spa_graph_%s", n), L.X2, vd, {R, Cp}); tap(e, fmt("fusion_%s", n), L.F, vd, {R, Mp});
        tap(e, fmt("gw_w_%s", n), L.gw_w, 0, {B, N, Tp}); tap(e, fmt("gw_v_%s", n), L.gw_v, 0, {B, N, Tp});
        tap(e, fmt("score_%s", n), L.score, 0, {B, e->h, e->w, 1}); tap(e, fmt("up_%s", n), L.up, 0, {B, H, W, 1});
        tap(e, fmt("loss_vec_%s", n), L.loss, 0, {B});
    }
    // ---- gated exchange
    for (int i = 0; i < 2 * e->nex; ++i) {
        ExgBuf& X = e->ex[i];
        X.q = (float*)g.take((size_t)B * Mp * F); X.kq = (float*)g.take((size_t)B * Mp * F);
        X.logits = (float*)g.take((size_t)B * N * F); X.attn = (float*)g.take((size_t)B * N * F);
        X.pooled = (float*)zf.take((size_t)B * Mp * F);
        X.gvpre = (float*)g.take((size_t)B * Mp * F); X.gv = (float*)g.take((size_t)B * Mp * F); X.rs1 = (float*)g.take(256);
        for (int k = 0; k < 2; ++k) { X.g[k] = (float*)g.take((size_t)B * Mp * F); X.r[k] = g.take((size_t)R * Mp * es); }
        X.out = g.take((size_t)R * Mp * es); X.rstd = (float*)g.take((size_t)R * F);
        X.dfeat = g.take((size_t)R * Mp * es);
        for (int k = 0; k < 2; ++k) {
            X.dp[k] = g.take((size_t)R * Mp * es); X.dfs[k] = g.take((size_t)R * Mp * es);
            X.dg[k] = (float*)zb.take((size_t)B * Mp * F);
        }
        X.dgv = (float*)g.take((size_t)B * Mp * F); X.dgvpre = (float*)g.take((size_t)B * Mp * F);
        X.dpooled = (float*)g.take((size_t)B * Mp * F); X.dnec = (float*)g.take((size_t)B * Cp * F);
        X.dattn = (float*)g.take((size_t)B * N * F); X.dlog = (float*)g.take((size_t)B * N * F);
        X.dkq = (float*)zb.take((size_t)B * Mp * F); X.dq = (float*)g.take((size_t)B * Mp * F);
        tap(e, fmt("exg_%s", exn(e, i)), X.out, vd, {R, Mp});
    }
    for (int i = 0; i < e->nex; ++i) e->de1[i] = g.take((size_t)R * Mp * es);
    // ---- ConvLSTM + final score
    for (int s = 0; s < e->ncl; ++s) {
        ClstmStep& S = e->cl[s];
        S.Yg = g.take((size_t)R * 4 * Mp * es); S.sums = (double*)g.take((size_t)5 * B * STAT_PARTS * 2 * D);
        S.c_pre = g.take((size_t)R * Mp * es); S.c_new = g.take((size_t)R * Mp * es); S.h_new = g.take((size_t)R * Mp * es);
        S.dYg = g.take((size_t)R * 4 * Mp * es); S.dc_prev = g.take((size_t)R * Mp * es);
        S.dx = g.take((size_t)R * Mp * es); S.dh = g.take((size_t)R * Mp * es);
    }
    e->cl_scr = g.take((size_t)R * Mp * es); e->cl_bs = (double*)g.take((size_t)5 * B * STAT_PARTS * 2 * D);
    const int ph = e->v5 ? e->h2 : e->h, pw = e->v5 ? e->w2 : e->w;        // the map `pred` lives on (v5: the decoder's, H/4 x W/4)
    e->score = (float*)g.take((size_t)B * ph * pw * F); e->up = (float*)g.take((size_t)B * H * W * F);
    e->sigm = (float*)g.take((size_t)B * H * W * F);
    e->loss = (float*)zf.take((size_t)B * F); e->iu = (int*)zf.take((size_t)2 * B * 4);
    e->dscore = (float*)g.take((size_t)B * ph * pw * F); e->dfused = g.take((size_t)R * Mp * es);
    if (e->v5) {
        // ---- ASPP + decoder (v5:190-251)
        const int Dp = e->Dp, CATp = e->CATp, R2 = e->R2;
        const int rows[9] = {R, R, R, R, B, R, R2, R2, R2}, cpad[9] = {Dp, Dp, Dp, Dp, Dp, Dp, 64, Dp, Dp}, cval[9] = {e->D, e->D, e->D, e->D, e->D, e->D, e->LOW, e->D, e->D};
        for (int i = 0; i < 9; ++i) {
            BnLayer& L = e->bn[i];
            L.scope = BN_SCOPES[i]; L.C = cval[i]; L.Cpad = cpad[i]; L.R = rows[i]; L.dt = i == BN_IMG ? DT_F32 : vd; L.ldpre = cpad[i];
            const size_t esz_i = i == BN_IMG ? F : es;
            L.pre = g.take((size_t)rows[i] * cpad[i] * esz_i); L.dpre = g.take((size_t)rows[i] * cpad[i] * esz_i);
            L.sums = (double*)g.take((size_t)2 * cpad[i] * D); L.mr = (float*)g.take((size_t)2 * cpad[i] * F); L.means = (float*)g.take((size_t)2 * cpad[i] * F);
            L.mm = e->bn_state ? e->bn_state + e->state_off.at(std::string("text_objseg/") + BN_SCOPES[i] + "/BatchNorm/moving_mean") : nullptr;
            L.mv = e->bn_state ? e->bn_state + e->state_off.at(std::string("text_objseg/") + BN_SCOPES[i] + "/BatchNorm/moving_variance") : nullptr;
        }
        e->cat4 = g.take((size_t)R * 4 * Dp * es); e->enc = g.take((size_t)R * Dp * es);
        e->deccat = g.take((size_t)R2 * CATp * es); e->net1 = g.take((size_t)R2 * Dp * es); e->net2 = g.take((size_t)R2 * Dp * es);
        e->dcat4 = g.take((size_t)R * 4 * Dp * es); e->denc = g.take((size_t)R * Dp * es);
        e->ddeccat = g.take((size_t)R2 * CATp * es); e->dnet1 = g.take((size_t)R2 * Dp * es); e->dnet2 = g.take((size_t)R2 * Dp * es);
        e->pooled = (float*)zf.take((size_t)B * Mp * F); e->img = (float*)g.take((size_t)B * Dp * F); e->catsb = (float*)g.take((size_t)B * Dp * F);
        e->dcatsb = (float*)zb.take((size_t)B * Dp * F); e->dimg = (float*)g.take((size_t)B * Dp * F); e->dpooled = (float*)g.take((size_t)B * Mp * F);
        e->ones_n = (float*)g.take((size_t)B * N * F);              // 1/N per node: the global average pooling as an attention pool (filled once, at create)
        e->zbias = (float*)g.take((size_t)2048 * F);
        e->zeros_bt = (float*)g.take((size_t)std::max(B * T, 64) * F);      // never written: the absent third level's d(parse_R) and loss vector                // zero bias vector of the batch-normed convolutions (never written)
        tap(e, "aspp_branches", e->cat4, vd, {R, 4 * Dp}); tap(e, "aspp_image", e->img, 0, {B, Dp}); tap(e, "aspp", e->enc, vd, {R, Dp});
        tap(e, "dec_cat", e->deccat, vd, {R2, CATp}); tap(e, "dec_net2", e->net2, vd, {R2, Dp});
    }
    e->scalars = (float*)g.take(256);
    e->nonfinite = (int*)zb.take(256);
    e->tn_table_bytes = (size_t)128 << 10;
    for (int a = 0; a < E::NBK; ++a) for (int b = 0; b < 4; ++b) e->tn_table[a][b] = g.take(e->tn_table_bytes);
    // partial rows of the deferred folds (~2 MB each, ~110 per step at B*N = 12800, Cp = 1024) and the per-part slabs of the split dW
    // reductions: both scale with the rows x widest map of the plan; 2 GiB at the benchmark's sizes, 32 MiB floor for tiny handles.  A
    // request that does not fit is served from the per-stream scratch and folded at once (cmpc_ws / cmpc_reduce_parts_f32): same bits.
    e->fold.cap = up256(((size_t)32 << 20) + (size_t)(2147483648.0 * (((double)R * Cp + (e->v5 ? 4.0 * e->R2 * e->Dp : 0.0)) / (12800.0 * 1024.0))));
    e->fold.arena = (char*)g.take(e->fold.cap);
    e->fold.table_cap = 1024;
    for (int a = 0; a < E::NBK; ++a) e->fold_table[a] = (cmpc_fold_desc*)g.take(sizeof(cmpc_fold_desc) * e->fold.table_cap);
    e->fold.table_dev = e->fold_table[E::NBK - 1];
    tap(e, "fused", e->cl[e->ncl - 1].h_new, vd, {R, Mp});
    tap(e, "pred", e->score, 0, {B, ph, pw, 1}); tap(e, "up", e->up, 0, {B, H, W, 1}); tap(e, "sigm", e->sigm, 0, {B, H, W, 1});
    tap(e, "iu", e->iu, 3, {2, B}); tap(e, "loss_vec", e->loss, 0, {B}); tap(e, "scalars", e->scalars, 0, {6});
    tap(e, "grad_nonfinite", e->nonfinite, 3, {E::NBK});
}

// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
hipEvent_t next_event(E* e) { hipEvent_t ev = e->evpool[e->evnext]; e->evnext = (e->evnext + 1) % e->evpool.size(); return ev; }

// phase-boundary timestamp on a stream (only while cmpc_phase_marks is enabled; no profiler, so the overlap is the real one)
int mark(E* e, const char* name, hipStream_t st) {
    if (!e->marks_on) return CMPC_OK;
    hipEvent_t ev = nullptr;
    HCK(hipEventCreate(&ev));
    HCK(hipEventRecord(ev, st));
    e->marks.emplace_back(name, ev);
    return CMPC_OK;
}

// lane streams wait for everything queued so far on `from`
int fork_lanes(E* e, hipStream_t from, hipStream_t (&st)[3]) {
    if (e->cfg.n_lanes <= 1) { st[0] = st[1] = st[2] = from; return CMPC_OK; }
    hipEvent_t ev = next_event(e);
    HCK(hipEventRecord(ev, from));
    for (int i = 0; i < 3; ++i) { st[i] = e->lane[i]; HCK(hipStreamWaitEvent(e->lane[i], ev, 0)); }
    return CMPC_OK;
}
int join_lanes(E* e, hipStream_t into) {
    if (e->cfg.n_lanes <= 1) return CMPC_OK;
    for (int i = 0; i < 3; ++i) {
        hipEvent_t ev = next_event(e);
        HCK(hipEventRecord(ev, e->lane[i]));
        HCK(hipStreamWaitEvent(into, ev, 0));
    }
    return CMPC_OK;
}

thread_local E* t_cur = nullptr;       // the handle whose entry point is running on this thread (timing hook of gemm_nt)

// algorithmic (unpadded) extent of a padded dimension
inline int valid_extent(const E* e, int x) {
    if (x == e->Cp) return e->C;
    if (x == e->Mp) return e->M;
    if (x == 5 * e->Cp) return 5 * e->C;
    if (x == 4 * e->Mp) return 4 * e->M;
    return x;
}

cmpc_gemm_nt_args nt_args(int dt, std::initializer_list<Seg> segs, void* C, int ldc, int M, int N, const GemmOpt& o) {
    cmpc_gemm_nt_args a; memset(&a, 0, sizeof(a));
    a.dtype = dt; a.nseg = (int)segs.size();
    int i = 0;
    for (const Seg& s : segs) { a.A[i] = s.A; a.lda[i] = s.lda; a.Bt[i] = s.Bt; a.ldb[i] = s.ldb; a.K[i] = s.K; a.sA[i] = s.sA; a.sB[i] = s.sB; ++i; }
    a.C = C; a.ldc = ldc; a.sC = o.sC; a.c_f32 = o.c_f32;
    a.M = M; a.N = N; a.n_valid = o.n_valid < 0 ? N : o.n_valid; a.batch = o.batch;
    a.bias = o.bias; a.sbias = o.sbias; a.ld_sbias = o.ld_sbias; a.pbias = o.pbias; a.ld_pbias = o.ld_pbias;
    a.rows_per_sample = o.rows_per_sample; a.act = o.act; a.alpha = o.alpha; a.accumulate = o.accumulate;
    return a;
}

int gemm_nt(hipStream_t st, int dt, std::initializer_list<Seg> segs, void* C, int ldc, int M, int N, const GemmOpt& o = GemmOpt()) {
    E* te = t_cur;
    if (te && te->timing && dt != DT_F32 && N >= 128 && M >= 512) {        // the launches that dispatch to the 16-bit MFMA pipelines
        te->timing = false;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { cmpc_set_error("timing: hipEventCreate"); return CMPC_EHIP; }
        (void)hipEventRecord(e0, st);
        const int rc = gemm_nt(st, dt, segs, C, ldc, M, N, o);
        (void)hipEventRecord(e1, st);
        te->timing = true;
        double kalg = 0;
        for (const Seg& s : segs) kalg += (s.A == te->spatial && s.K == 64) ? 8 : valid_extent(te, s.K);
        const double nv = valid_extent(te, o.n_valid < 0 ? N : o.n_valid), rows = (double)M * o.batch;
        te->tev.push_back(e0); te->tev.push_back(e1);
        { int kp = 0; for (const Seg& s : segs) kp += s.K; te->tshape.push_back({M * o.batch, N, kp, (int)segs.size()}); }
        te->tflops.push_back(2.0 * rows * nv * kalg);
        te->tbytes.push_back(2.0 * (rows * (kalg + nv) + nv * kalg));        // A and C once per row, the weight once (bf16)
        return rc;
    }
    const cmpc_gemm_nt_args a = nt_args(dt, segs, C, ldc, M, N, o);
    return cmpc_gemm_nt(&a, st);
}

// two independent single-segment products of one shape in one launch (cmpc_gemm_nt_pair); the timing hook books them as one launch
struct NtJob { Seg seg; void* C; int ldc; GemmOpt o; };
int gemm_nt_pair(hipStream_t st, int dt, const NtJob& x, const NtJob& y, int M, int N) {
    E* te = t_cur;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = te && te->timing && dt != DT_F32 && N >= 128 && M >= 512;
    if (timed) {
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { cmpc_set_error("timing: hipEventCreate"); return CMPC_EHIP; }
        (void)hipEventRecord(e0, st);
    }
    const cmpc_gemm_nt_args a = nt_args(dt, {x.seg}, x.C, x.ldc, M, N, x.o), b = nt_args(dt, {y.seg}, y.C, y.ldc, M, N, y.o);
    const int rc = cmpc_gemm_nt_pair(&a, &b, st);
    if (timed) {
        (void)hipEventRecord(e1, st);
        double fl = 0, by = 0;
        for (const NtJob* j : {&x, &y}) {
            const double kalg = valid_extent(te, j->seg.K), nv = valid_extent(te, j->o.n_valid < 0 ? N : j->o.n_valid), rows = (double)M;
            fl += 2.0 * rows * nv * kalg; by += 2.0 * (rows * (kalg + nv) + nv * kalg);
        }
        te->tev.push_back(e0); te->tev.push_back(e1);
        te->tshape.push_back({2 * M, N, x.seg.K, 1});
        te->tflops.push_back(fl); te->tbytes.push_back(by);
    }
    return rc;
}

struct TnOpt { int nb2 = 1; int64_t a_bs = 0, d_bs = 0, o_bs = 0; float alpha = 1.f; bool defer = false; };
typedef std::vector<std::array<int64_t, 3>> Offs;

// the gradient bucket a gradient-buffer address belongs to
int bucket_of(const E* e, const float* p) {
    const int64_t o = p - e->grads;
    for (int b = 0; b < E::NBK; ++b)
        for (const E::Range& r : e->bucket[b]) if (o >= r.off && o < r.off + r.count) return b;
    return E::NBK - 1;
}

// out[k, n] += alpha * sum_r A[r, k] D[r, n]; defer: weight gradient, issued by flush_bucket() in one grouped launch per bucket
int gemm_tn(E* e, hipStream_t st, int dt, const void* A, int lda, int Ka, const void* D, int ldd, int Nd, float* out, int ldo,
            int R, int Kv, int Nv, const Offs& offs, const TnOpt& o = TnOpt()) {
    cmpc_gemm_tn_args a; memset(&a, 0, sizeof(a));
    a.dtype = dt; a.A = A; a.lda = lda; a.Ka = Ka; a.D = D; a.ldd = ldd; a.Nd = Nd; a.out = out; a.ldo = ldo;
    a.R = R; a.Kv = Kv; a.Nv = Nv;
    a.nb = (int)offs.size();
    for (size_t i = 0; i < offs.size(); ++i) { a.a_off[i] = offs[i][0]; a.d_off[i] = offs[i][1]; a.o_off[i] = offs[i][2]; }
    a.nb2 = o.nb2; a.a_bs = o.a_bs; a.d_bs = o.d_bs; a.o_bs = o.o_bs;
    const int tiles = ((Kv + 127) / 128) * ((Nv + 127) / 128) * a.nb * a.nb2;
    const int br = dt != DT_F32 ? 64 : 32;
    a.rsplit = std::max(1, std::min((R + 4 * br - 1) / (4 * br), (512 + tiles - 1) / tiles));      // split parts go through slabs + a fixed-order fold
    a.alpha = o.alpha; a.zeros = e->zero_page;
    if (o.defer) { e->deferred[bucket_of(e, out)].push_back(a); return CMPC_OK; }
    return cmpc_gemm_tn(&a, st);
}
const Offs OFF0 = {{0, 0, 0}};

// dpre = dy * act'(y) (optional), db[c] += column sums, dsb[b][c] += per-sample sums; column windows of <= 2048
int colsum(hipStream_t st, int dt, const void* dy, int R, int stride, int ld, int C, float* db, const void* y = nullptr, void* dpre = nullptr,
           int act = ACT_NONE, float* dsb = nullptr, int ld_dsb = 0, int rows_per_sample = 0) {
    const int es = dt == DT_F32 ? 4 : 2;
    for (int c0 = 0; c0 < ld; c0 += 2048) {
        const int wd = std::min(2048, ld - c0), cv = std::max(0, std::min(C - c0, wd));
        if (cv == 0) continue;
        CK(cmpc_act_bwd(dt, (const char*)dy + (size_t)c0 * es, y ? (const char*)y + (size_t)c0 * es : nullptr,
                        dpre ? (char*)dpre + (size_t)c0 * es : nullptr, act, R, stride, wd, cv, db ? db + c0 : nullptr,
                        dsb ? dsb + c0 : nullptr, ld_dsb, rows_per_sample, st));
    }
    return CMPC_OK;
}

int transpose_cast(hipStream_t st, const float* src, int dt, void* dst, int rows, int cols) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    if (dt == DT_F32) hipLaunchKernelGGL((transpose_cast_kernel<float>), grid, dim3(256), 0, st, src, (float*)dst, rows, cols);
    else if (dt == DT_BF16) hipLaunchKernelGGL((transpose_cast_kernel<bf16_t>), grid, dim3(256), 0, st, src, (bf16_t*)dst, rows, cols);
    else hipLaunchKernelGGL((transpose_cast_kernel<f16_t>), grid, dim3(256), 0, st, src, (f16_t*)dst, rows, cols);
    return cmpc_check_launch("transpose_cast");
}

int add_n(hipStream_t st, int dt, void* dst, std::initializer_list<const void*> srcs, bool acc, long n) {
    AddArgs a; a.n = 0;
    for (const void* s : srcs) a.src[a.n++] = s;
    for (int i = a.n; i < 8; ++i) a.src[i] = nullptr;
    const long n8 = n / 8;
    const int grid = (int)std::min<long>((n8 + 255) / 256, 2048);
    if (dt == DT_F32) hipLaunchKernelGGL((add_n_kernel<float>), dim3(grid), dim3(256), 0, st, (float*)dst, a, acc ? 1 : 0, n8);
    else if (dt == DT_BF16) hipLaunchKernelGGL((add_n_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (bf16_t*)dst, a, acc ? 1 : 0, n8);
    else hipLaunchKernelGGL((add_n_kernel<f16_t>), dim3(grid), dim3(256), 0, st, (f16_t*)dst, a, acc ? 1 : 0, n8);
    return cmpc_check_launch("add_n");
}

// ------------------------------------------------------------------------------------------
// stage: text encoder -- lstm(), CMPC_model.py:144-164
// ------------------------------------------------------------------------------------------
// one LSTM direction over its (already gathered) word ids: x-side product for all steps, then the recurrence
int lstm_dir_fwd(E* e, hipStream_t st, LstmDir& D, const int32_t* seq_len) {
    const int B = e->B, T = e->T, R = e->RNN, G = e->G, Cp = e->Cp, Gp = e->Gp, ldk = Gp + Cp;
    CK(cmpc_embed_gather(pptr(e, "Variable"), D.words_tb, D.emb, T * B, G, Gp, e->V, st));
    GemmOpt o; o.bias = (const float*)opp(e, D.key + ".b");
    CK(gemm_nt(st, DT_F32, {{D.emb, Gp, opp(e, D.key + ".t"), ldk, Gp}}, D.xg, 4 * Cp, T * B, 4 * Cp, o));
    const void* wh = opp(e, D.key + ".t", 0, Gp);
    if (e->lstm_seq)      // all T steps in one persistent launch (W_h in registers, a grid barrier per step)
        return cmpc_lstm_seq_fwd(D.xg, (const float*)wh, ldk, seq_len, D.gates, D.h_all, D.c_all, D.outs, D.sync_f, B, T, Cp, R, st);
    for (int t = 0; t < T; ++t) {
        float* gt = D.gates + (size_t)t * B * 4 * Cp;
        float *hp = D.h_all + (size_t)t * B * Cp, *cp = D.c_all + (size_t)t * B * Cp;
        GemmOpt s; s.sbias = D.xg + (size_t)t * B * 4 * Cp; s.ld_sbias = 4 * Cp; s.rows_per_sample = 1;
        CK(gemm_nt(st, DT_F32, {{hp, Cp, wh, ldk, Cp}}, gt, 4 * Cp, B, 4 * Cp, s));
        CK(cmpc_lstm_cell_fwd(gt, cp, hp, seq_len, t, cp + (size_t)B * Cp, hp + (size_t)B * Cp, D.outs + (size_t)t * Cp, T * Cp, B, Cp, R, st));
    }
    return CMPC_OK;
}
// in: D.douts (gradient of the direction's outputs [B, T, Cp]); out: kernel / bias gradients (deferred), D.demb (not yet scattered)
int lstm_dir_bwd(E* e, hipStream_t st, LstmDir& D, const int32_t* seq_len) {
    const int B = e->B, T = e->T, R = e->RNN, G = e->G, Cp = e->Cp, Gp = e->Gp;
    auto gat = [&](int t) { return D.gates + (size_t)t * B * 4 * Cp; };
    auto call = [&](int t) { return D.c_all + (size_t)t * B * Cp; };
    auto dgt = [&](int t) { return D.dgt + (size_t)t * B * 4 * Cp; };
    if (e->lstm_seq) {
        CK(cmpc_lstm_seq_bwd((const float*)opp(e, D.key + ".n", Gp, 0), 4 * Cp, D.gates, D.c_all, seq_len, D.douts, D.dgt, D.sync_b, B, T, Cp, R, st));
    } else if (B <= 8) {
        // one launch per step: dh += dg[t] . W_h^T fused with the cell backward of step t-1
        CK(cmpc_lstm_cell_bwd(gat(T - 1), call(T - 1), call(T), seq_len, T - 1, D.douts + (size_t)(T - 1) * Cp, T * Cp, D.dh, D.dc, dgt(T - 1), B, Cp, R, st));
        const float* wn = (const float*)opp(e, D.key + ".n", Gp, 0);
        for (int t = T - 1; t > 0; --t)
            CK(cmpc_lstm_bwd_step(dgt(t), wn, 4 * Cp, gat(t - 1), call(t - 1), call(t), seq_len, t - 1, D.douts + (size_t)(t - 1) * Cp, T * Cp,
                                  D.dh, D.dc, dgt(t - 1), B, Cp, R, st));
    } else {
        for (int t = T - 1; t >= 0; --t) {
            CK(cmpc_lstm_cell_bwd(gat(t), call(t), call(t + 1), seq_len, t, D.douts + (size_t)t * Cp, T * Cp, D.dh, D.dc, dgt(t), B, Cp, R, st));
            GemmOpt o; o.n_valid = R; o.accumulate = 1;
            CK(gemm_nt(st, DT_F32, {{dgt(t), 4 * Cp, opp(e, D.key + ".n", Gp, 0), 4 * Cp, 4 * Cp}}, D.dh, Cp, B, Cp, o));
        }
    }
    float* gk = gptr(e, D.pk);
    Offs o0, o1;
    for (int g = 0; g < 4; ++g) { o0.push_back({0, (int64_t)g * Cp, (int64_t)g * R}); o1.push_back({0, (int64_t)g * Cp, (int64_t)G * 4 * R + (int64_t)g * R}); }
    TnOpt d; d.defer = true;
    CK(gemm_tn(e, st, DT_F32, D.emb, Gp, Gp, D.dgt, 4 * Cp, Cp, gk, 4 * R, T * B, G, R, o0, d));
    CK(gemm_tn(e, st, DT_F32, D.h_all, Cp, Cp, D.dgt, 4 * Cp, Cp, gk, 4 * R, T * B, R, R, o1, d));
    float* gb = gptr(e, D.pb);
    for (int g = 0; g < 4; ++g) CK(colsum(st, DT_F32, D.dgt + (size_t)g * Cp, T * B, 4 * Cp, Cp, R, gb + (size_t)g * R));
    {   // demb = dg . W_x^T: a small output ([T*B, G]) over a long reduction (4 Cp) -> split-K as a batched product + one sum
        const int ks = (4 * Cp) % (8 * 32) == 0 ? 8 : 1, kc = 4 * Cp / ks;
        GemmOpt o; o.n_valid = G; o.batch = ks; o.sC = (int64_t)T * B * Gp;
        CK(gemm_nt(st, DT_F32, {{D.dgt, 4 * Cp, opp(e, D.key + ".n"), 4 * Cp, kc, (int64_t)kc, (int64_t)kc}}, D.demb_parts, Gp, T * B, Gp, o));
        const float* p = D.demb_parts; const size_t sl = (size_t)T * B * Gp;
        if (ks == 8) CK(add_n(st, DT_F32, D.demb, {p, p + sl, p + 2 * sl, p + 3 * sl, p + 4 * sl, p + 5 * sl, p + 6 * sl, p + 7 * sl}, false, (long)sl));
        else CK(add_n(st, DT_F32, D.demb, {p}, false, (long)sl));
    }
    return CMPC_OK;
}

// lstm(), CMPC_model.py:144-164 -- or BiLSTM(), CMPCv5_BiLSTM_model.py:159-187 (the backward direction on lane 1 beside the forward one)
int text_fwd(E* e, hipStream_t st, const int32_t* words, const int32_t* seq_len) {
    const int B = e->B, T = e->T, R = e->RNN, Cp = e->Cp;
    LstmDir& F = e->ldir[0];
    hipLaunchKernelGGL(transpose_i32_kernel, dim3((B * T + 255) / 256), dim3(256), 0, st, words, F.words_tb, B, T);
    CK(cmpc_check_launch("transpose_i32"));
    if (!e->v5) {
        CK(lstm_dir_fwd(e, st, F, seq_len));
        return cmpc_l2norm_rows_fwd(DT_F32, F.outs, e->wf, e->wf_rstd, e->mask, B * T, Cp, R, st);
    }
    LstmDir& W = e->ldir[1];
    hipStream_t sb = st;
    hipEvent_t back = nullptr;
    if (e->cfg.n_lanes > 1) {
        hipEvent_t ev = next_event(e);
        HCK(hipEventRecord(ev, st));
        sb = e->lane[1];
        HCK(hipStreamWaitEvent(sb, ev, 0));
    }
    // bidirectional_dynamic_rnn: the backward cell reads reverse_sequence(inputs) and its outputs are reversed back (v5:170-174)
    CK(cmpc_reverse_words_tb(words, seq_len, W.words_tb, B, T, sb));
    CK(lstm_dir_fwd(e, sb, W, seq_len));
    CK(cmpc_reverse_sequence(W.outs, e->outs_bw, seq_len, B, T, Cp, sb));
    if (sb != st) { back = next_event(e); HCK(hipEventRecord(back, sb)); }
    CK(lstm_dir_fwd(e, st, F, seq_len));
    if (back) HCK(hipStreamWaitEvent(st, back, 0));
    CK(cmpc_rows_nonzero2(F.outs, e->outs_bw, e->mask, B * T, Cp, R, st));                 // seq_mask (v5:181)
    GemmOpt o; o.n_valid = R; o.bias = pptr(e, "words_feat/biases"); o.act = ACT_TANH;     // words_feat conv + tanh (v5:182-183)
    CK(gemm_nt(st, DT_F32, {{F.outs, Cp, opp(e, "wfeat.t"), 2 * Cp, Cp}, {e->outs_bw, Cp, opp(e, "wfeat.t", 0, Cp), 2 * Cp, Cp}}, e->wft, Cp, B * T, Cp, o));
    return cmpc_l2norm_rows_fwd(DT_F32, e->wft, e->wf, e->wf_rstd, nullptr, B * T, Cp, R, st);          // v5:185
}
int text_bwd(E* e, hipStream_t st, const int32_t* seq_len) {
    const int B = e->B, T = e->T, R = e->RNN, G = e->G, Cp = e->Cp, Gp = e->Gp;
    LstmDir& F = e->ldir[0];
    if (!e->v5) {
        CK(cmpc_l2norm_rows_bwd(DT_F32, e->dwf, e->wf, e->wf_rstd, F.douts, B * T, Cp, R, 0, st));
        CK(lstm_dir_bwd(e, st, F, seq_len));
        return cmpc_embed_scatter(F.demb, Gp, F.words_tb, gptr(e, "Variable"), T * B, G, e->V, st);
    }
    LstmDir& W = e->ldir[1];
    TnOpt d; d.defer = true;
    // words_feat = l2norm(tanh([fw | bw] . W + b)) (v5:182-185); dpre is read again by the deferred weight-gradient products
    float* dpre = e->dwft;
    CK(cmpc_l2norm_rows_bwd(DT_F32, e->dwf, e->wf, e->wf_rstd, dpre, B * T, Cp, R, 0, st));
    CK(colsum(st, DT_F32, dpre, B * T, Cp, Cp, R, gptr(e, "words_feat/biases"), e->wft, dpre, ACT_TANH));
    float* gw = gptr(e, "words_feat/DW");
    CK(gemm_tn(e, st, DT_F32, F.outs, Cp, Cp, dpre, Cp, Cp, gw, R, B * T, R, R, OFF0, d));
    CK(gemm_tn(e, st, DT_F32, e->outs_bw, Cp, Cp, dpre, Cp, Cp, gw + (size_t)R * R, R, B * T, R, R, OFF0, d));
    GemmOpt o; o.n_valid = R;
    CK(gemm_nt(st, DT_F32, {{dpre, Cp, opp(e, "wfeat.n"), Cp, Cp}}, F.douts, Cp, B * T, Cp, o));
    CK(gemm_nt(st, DT_F32, {{dpre, Cp, opp(e, "wfeat.n", Cp, 0), Cp, Cp}}, e->douts_bw, Cp, B * T, Cp, o));
    hipStream_t sb = st;
    hipEvent_t back = nullptr;
    if (e->cfg.n_lanes > 1) {
        hipEvent_t ev = next_event(e);
        HCK(hipEventRecord(ev, st));
        sb = e->lane[1];
        HCK(hipStreamWaitEvent(sb, ev, 0));
    }
    CK(cmpc_reverse_sequence(e->douts_bw, W.douts, seq_len, B, T, Cp, sb));
    CK(lstm_dir_bwd(e, sb, W, seq_len));
    if (sb != st) { back = next_event(e); HCK(hipEventRecord(back, sb)); }
    CK(lstm_dir_bwd(e, st, F, seq_len));
    if (back) HCK(hipStreamWaitEvent(st, back, 0));
    // both directions add into the one embedding table: two scatters in a fixed order on one stream (one writer per row each)
    CK(cmpc_embed_scatter(F.demb, Gp, F.words_tb, gptr(e, "Variable"), T * B, G, e->V, st));
    return cmpc_embed_scatter(W.demb, Gp, W.words_tb, gptr(e, "Variable"), T * B, G, e->V, st);
}

// ------------------------------------------------------------------------------------------
// stage: build_lang_parser, CMPC_model.py:347-357
// ------------------------------------------------------------------------------------------
int parser_fwd(E* e, hipStream_t st) {
    const int BT = e->B * e->T, P = e->P, Cp = e->Cp, Pp = e->Pp;
    GemmOpt a; a.n_valid = P; a.bias = pptr(e, "words_parse_1/biases"); a.act = ACT_RELU;
    CK(gemm_nt(st, DT_F32, {{e->wf, Cp, opp(e, "parse1.t"), Cp, Cp}}, e->h1, Pp, BT, Pp, a));
    GemmOpt b; b.n_valid = e->NC; b.bias = pptr(e, "words_parse_2/biases");
    CK(gemm_nt(st, DT_F32, {{e->h1, Pp, opp(e, "parse2.t"), Pp, Pp}}, e->lg, 64, BT, 64, b));
    return cmpc_parse_softmax_fwd(e->lg, 64, e->mask, e->parse, BT, e->NC, st);
}
// dwf accumulates on top of e->dwf
int parser_bwd(E* e, hipStream_t st) {
    const int BT = e->B * e->T, R = e->RNN, P = e->P, Cp = e->Cp, Pp = e->Pp;
    TnOpt d; d.defer = true;
    CK(cmpc_parse_softmax_bwd(e->dparse, e->parse, e->mask, e->dlg, 64, BT, e->NC, st));
    CK(colsum(st, DT_F32, e->dlg, BT, 64, 64, e->NC, gptr(e, "words_parse_2/biases")));
    CK(gemm_tn(e, st, DT_F32, e->h1, Pp, Pp, e->dlg, 64, 64, gptr(e, "words_parse_2/DW"), e->NC, BT, P, e->NC, OFF0, d));
    GemmOpt a; a.n_valid = P;
    CK(gemm_nt(st, DT_F32, {{e->dlg, 64, opp(e, "parse2.n"), 64, 64}}, e->dh1, Pp, BT, Pp, a));
    CK(colsum(st, DT_F32, e->dh1, BT, Pp, Pp, P, gptr(e, "words_parse_1/biases"), e->h1, e->dh1, ACT_RELU));
    CK(gemm_tn(e, st, DT_F32, e->wf, Cp, Cp, e->dh1, Pp, Pp, gptr(e, "words_parse_1/DW"), P, BT, R, P, OFF0, d));
    GemmOpt b; b.n_valid = R; b.accumulate = 1;
    return gemm_nt(st, DT_F32, {{e->dh1, Pp, opp(e, "parse1.n"), Pp, Pp}}, e->dwf, Cp, BT, Cp, b);
}

// ------------------------------------------------------------------------------------------
// stages of one pyramid level
// ------------------------------------------------------------------------------------------
// The language side of a level -- everything that depends on the text encoder only (the five lang_trans heads, the word-side
// operands of build_spa_graph, the tiled-language share of the fusion conv): ~8 small fp32 launches that run while the backbone
// is still producing the visual features.
int level_lang_fwd(E* e, hipStream_t st, int li) {
    LevelBuf& L = e->lv[li]; const char* lv = lvn(e, li);
    const int B = e->B, T = e->T, C = e->C, Cp = e->Cp, Tp = e->Tp, M = e->M, Mp = e->Mp, dt = e->dt;
    { GemmOpt o; o.bias = (const float*)opp(e, fmt("mlang_%s.b", lv)); o.act = ACT_TANH;
      CK(gemm_nt(st, DT_F32, {{e->vl, Cp, opp(e, fmt("mlang_%s.t", lv)), Cp, Cp}}, L.g, 5 * Cp, B, 5 * Cp, o)); }
    { const float scale = 1.0f / sqrtf((float)C);
      const std::string t2n = fmt("t2_%s.n", lv);
      GemmOpt a; a.n_valid = C; a.batch = B; a.sC = (int64_t)Tp * Cp; a.bias = pptr(e, fmt("words_trans_%s/biases", lv));
      CK(gemm_nt(st, DT_F32, {{e->wf, Cp, opp(e, fmt("wtrans_%s.t", lv)), Cp, Cp, (int64_t)T * Cp, 0}}, L.Wd, Cp, T, Cp, a));
      GemmOpt b; b.n_valid = C;
      CK(gemm_nt(st, DT_F32, {{L.Wd, Cp, opp(e, t2n), Cp, Cp}}, L.PTf, Cp, B * Tp, Cp, b));
      CK(cmpc_cast(DT_F32, L.PTf, dt, L.PT, (int64_t)B * Tp * Cp, st));
      if (!e->lowrank) CK(transpose_cast(st, L.PTf, dt, L.PTt, B * Tp, Cp));   // PT^T [Cp][B*Tp] (GEMM operand of dX1 += dA0 . PT)
      CK(cmpc_rowdot1(DT_F32, L.Wd, pptr(e, fmt("spa_graph_trans2_%s/biases", lv)), 0, L.k0s, 1, B * Tp, Cp, C, scale, st));
      hipLaunchKernelGGL(col_get_kernel, dim3((B * T + 255) / 256), dim3(256), 0, st, e->parse, e->NC, 2, L.pr, B * T);
      CK(cmpc_check_launch("col_get")); }
    { GemmOpt s; s.n_valid = M;
      CK(gemm_nt(st, DT_F32, {{e->vl, Cp, opp(e, fmt("fusl_%s.t", lv)), Cp, Cp}}, L.sb, Mp, B, Mp, s)); }
    return CMPC_OK;
}

// the visual side (after level_lang_fwd, once the backbone taps are complete)
int level_fwd(E* e, hipStream_t st, int li, const float* target) {
    LevelBuf& L = e->lv[li]; const char* lv = lvn(e, li);
    const int B = e->B, N = e->N, T = e->T, R = e->R, C = e->C, Cp = e->Cp, Tp = e->Tp, M = e->M, Mp = e->Mp, dt = e->dt;
    // -- lateral 1x1 conv + l2_normalize (CMPC_model.py:108-113); CMPCv5_BiLSTM: tanh in between (v5:120-125), HSV channels as a K-segment (hsv:128-134)
    if (!e->v5) {
      GemmOpt o; o.n_valid = C; o.bias = pptr(e, fmt("%s_lateral/biases", lv));
      CK(gemm_nt(st, dt, {{L.feat, L.cin, opp(e, fmt("lat_%s.t", lv)), L.cin, L.cin}}, L.X0, Cp, R, Cp, o));
      CK(cmpc_l2norm_rows_fwd(dt, L.X0, L.X0, L.lat_rstd, nullptr, R, Cp, C, st));
    } else {
      GemmOpt o; o.n_valid = C; o.bias = pptr(e, fmt("%s_lateral/biases", lv)); o.act = ACT_TANH;
      const std::string k = fmt("lat_%s.t", lv);
      if (e->cfg.hsv) {
          const int ldk = L.cin + 64;
          CK(gemm_nt(st, dt, {{L.feat, L.cin, opp(e, k), ldk, L.cin}, {e->hsv, 64, opp(e, k, 0, L.cin), ldk, 64}}, L.X0t, Cp, R, Cp, o));
      } else CK(gemm_nt(st, dt, {{L.feat, L.cin, opp(e, k), L.cin, L.cin}}, L.X0t, Cp, R, Cp, o));
      CK(cmpc_l2norm_rows_fwd(dt, L.X0t, L.X0, L.lat_rstd, nullptr, R, Cp, C, st));
    }
    // -- mutan_fusion (:295-328)
    { const int ldk = Cp + 64; const std::string k = fmt("mutan_%s.t", lv);
      GemmOpt p; p.bias = (const float*)opp(e, fmt("mutan_%s.b", lv)); if (e->mutan_epilogue) p.act = ACT_TANH;
      CK(gemm_nt(st, dt, {{L.X0, Cp, opp(e, k), ldk, Cp}, {e->spatial, 64, opp(e, k, 0, Cp), ldk, 64}}, L.P, 5 * Cp, R, 5 * Cp, p));
      CK(cmpc_mutan_fwd(dt, L.P, L.g, L.X1, L.mut_rstd, B, N, Cp, C, e->mutan_epilogue ? 1 : 0, st)); }
    // -- build_spa_graph + graph_conv (:359-410); adjacency never formed, trans2 folded into the word side
    { const float scale = 1.0f / sqrtf((float)C);
      GemmOpt c; c.batch = B; c.sC = (int64_t)N * Tp; c.c_f32 = 1; c.alpha = scale; c.sbias = L.k0s; c.ld_sbias = Tp; c.rows_per_sample = N;
      CK(gemm_nt(st, dt, {{L.X1, Cp, L.PT, Cp, Cp, (int64_t)N * Cp, (int64_t)Tp * Cp}}, L.A0, Tp, N, Tp, c));
      CK(cmpc_graph_softmax_fwd(dt, e->v5 ? 1 : 0, L.A0, L.pr, e->mask, L.gw_w, L.gw_v, L.gw_w_t, L.gw_v_t, L.gsc, B, N, T, Tp, st));
      if (e->lowrank) {      // Z = gw_v^T . X1 [T, C] k-major (kept for the backward pass), Y = gw_w . Z as a stream of Y
          TnOpt zt; zt.nb2 = B; zt.a_bs = (int64_t)N * Tp; zt.d_bs = (int64_t)N * Cp; zt.o_bs = (int64_t)Tp * Cp;
          CK(gemm_tn(e, st, dt, L.gw_v_t, Tp, Tp, L.X1, Cp, Cp, L.Zf, Cp, N, T, C, OFF0, zt));
          CK(cmpc_cast(DT_F32, L.Zf, dt, L.Z, (int64_t)B * Tp * Cp, st));
          CK(cmpc_lowrank_nn(dt, L.gw_w_t, Tp, (int64_t)N * Tp, L.Z, Cp, (int64_t)Tp * Cp, L.Y, Cp, (int64_t)N * Cp, N, Cp, C, T, B, 1.0f, 0, st));
      } else {
          TnOpt z; z.nb2 = B; z.a_bs = (int64_t)N * Cp; z.d_bs = (int64_t)N * Tp; z.o_bs = (int64_t)Cp * Tp;     // Z^T = X1^T . gw_v
          CK(gemm_tn(e, st, dt, L.X1, Cp, Cp, L.gw_v_t, Tp, Tp, L.Ztf, Tp, N, C, T, OFF0, z));
          CK(cmpc_cast(DT_F32, L.Ztf, dt, L.Zt, (int64_t)B * Cp * Tp, st));
          GemmOpt y; y.n_valid = C; y.batch = B; y.sC = (int64_t)N * Cp;
          CK(gemm_nt(st, dt, {{L.gw_w_t, Tp, L.Zt, Tp, Tp, (int64_t)N * Tp, (int64_t)Cp * Tp}}, L.Y, Cp, N, Cp, y));
      }
      CK(cmpc_sample_stats(dt, L.Y, L.sums1, B, N, Cp, C, st));
      const std::string ln1 = fmt("gconv_feat_ln_spa_graph_%s", lv), ln2 = fmt("gconv_update_ln_spa_graph_%s", lv);
      CK(cmpc_gconv_pre_fwd(dt, L.Y, L.X1, L.sums1, pptr(e, ln1 + "/gamma"), pptr(e, ln1 + "/beta"), L.G, B, N, Cp, C, st));
      GemmOpt u; u.n_valid = C; u.bias = pptr(e, fmt("gconv_update_spa_graph_%s/biases", lv));
      CK(gemm_nt(st, dt, {{L.G, Cp, opp(e, fmt("gupd_%s.t", lv)), Cp, Cp}}, L.U, Cp, R, Cp, u));
      CK(cmpc_sample_stats(dt, L.U, L.sums2, B, N, Cp, C, st));
      CK(cmpc_gconv_post_fwd(dt, L.U, L.sums2, pptr(e, ln2 + "/gamma"), pptr(e, ln2 + "/beta"), L.X2, L.rrow, B, N, Cp, C, st)); }
    // -- fusion 1x1 over [vis_la_sp | spa_graph | tile(valid_lang) | spatial] (:338-344); the concat is never formed
    { const int ldk = 2 * Cp + 64; const std::string k = fmt("fus_%s.t", lv);
      GemmOpt f; f.n_valid = M; f.bias = pptr(e, fmt("fusion_%s/biases", lv)); f.sbias = L.sb; f.ld_sbias = Mp; f.rows_per_sample = N; f.act = ACT_RELU;
      CK(gemm_nt(st, dt, {{L.X1, Cp, opp(e, k), ldk, Cp}, {L.X2, Cp, opp(e, k, 0, Cp), ldk, Cp}, {e->spatial, 64, opp(e, k, 0, 2 * Cp), ldk, 64}},
                 L.F, Mp, R, Mp, f)); }
    // -- score_cX 3x3 + legacy bilinear + BCE (:128-133,440-443)
    CK(cmpc_score_conv_fwd(dt, L.F, pptr(e, fmt("score_%s/DW", lv)), pptr(e, fmt("score_%s/biases", lv)), L.score, B, e->h, e->w, Mp, M, st));
    return cmpc_upsample_fwd(L.score, L.up, nullptr, target, L.loss, L.iu, L.iu + B, B, e->h, e->w, e->H, e->W, st);
}

// dfus (the level's fusion-output gradient from the exchange modules) must already hold the sum of its consumers
int level_bwd(E* e, hipStream_t st, int li, const float* target) {
    LevelBuf& L = e->lv[li]; const char* lv = lvn(e, li);
    const int B = e->B, N = e->N, T = e->T, R = e->R, C = e->C, Cp = e->Cp, Tp = e->Tp, M = e->M, Mp = e->Mp, dt = e->dt, Rr = e->RNN;
    const int es = e->esz;
    TnOpt d; d.defer = true;
    // -- score head: dfus += score-conv backward of w_lv/B * (sigmoid(up) - target)
    CK(cmpc_upsample_loss_bwd(L.up, target, L.dscore, e->cfg.loss_w[1 + li] * e->cfg.loss_scale / B, B, e->h, e->w, e->H, e->W, st));
    CK(cmpc_score_conv_bwd(dt, L.dscore, L.F, pptr(e, fmt("score_%s/DW", lv)), L.dfus, 1, gptr(e, fmt("score_%s/DW", lv)),
                           gptr(e, fmt("score_%s/biases", lv)), B, e->h, e->w, Mp, M, st));
    // -- fusion
    { CK(colsum(st, dt, L.dfus, R, Mp, Mp, M, gptr(e, fmt("fusion_%s/biases", lv)), L.F, L.dpre, ACT_RELU, L.dsb, Mp, N));
      float* gw = gptr(e, fmt("fusion_%s/DW", lv));
      CK(gemm_tn(e, st, dt, L.X1, Cp, Cp, L.dpre, Mp, Mp, gw, M, R, C, M, OFF0, d));
      CK(gemm_tn(e, st, dt, L.X2, Cp, Cp, L.dpre, Mp, Mp, gw + (size_t)C * M, M, R, C, M, OFF0, d));
      CK(gemm_tn(e, st, dt, e->spatial, 64, 64, L.dpre, Mp, Mp, gw + (size_t)(2 * C + Rr) * M, M, R, 8, M, OFF0, d));
      CK(gemm_tn(e, st, DT_F32, e->vl, Cp, Cp, L.dsb, Mp, Mp, gw + (size_t)2 * C * M, M, B, Rr, M, OFF0, d));
      const std::string k = fmt("fus_%s.n", lv);
      GemmOpt o; o.n_valid = C;
      CK(gemm_nt(st, dt, {{L.dpre, Mp, opp(e, k), Mp, Mp}}, L.dX1, Cp, R, Cp, o));
      CK(gemm_nt(st, dt, {{L.dpre, Mp, opp(e, k, Cp, 0), Mp, Mp}}, L.dX2, Cp, R, Cp, o));
      GemmOpt v; v.n_valid = Rr;
      CK(gemm_nt(st, DT_F32, {{L.dsb, Mp, opp(e, fmt("fusl_%s.n", lv)), Mp, Mp}}, L.dvl, Cp, B, Cp, v)); }
    // -- spa_graph / graph_conv
    { const float scale = 1.0f / sqrtf((float)C);
      const std::string ln1 = fmt("gconv_feat_ln_spa_graph_%s", lv), ln2 = fmt("gconv_update_ln_spa_graph_%s", lv);
      CK(cmpc_gconv_post_bwd(dt, L.dX2, L.X2, L.rrow, L.U, L.sums2, pptr(e, ln2 + "/gamma"), L.dU, gptr(e, ln2 + "/gamma"), gptr(e, ln2 + "/beta"),
                             L.bs, B, N, Cp, C, st));
      CK(colsum(st, dt, L.dU, R, Cp, Cp, C, gptr(e, fmt("gconv_update_spa_graph_%s/biases", lv))));
      CK(gemm_tn(e, st, dt, L.G, Cp, Cp, L.dU, Cp, Cp, gptr(e, fmt("gconv_update_spa_graph_%s/DW", lv)), C, R, C, C, OFF0, d));
      GemmOpt o; o.n_valid = C;
      CK(gemm_nt(st, dt, {{L.dU, Cp, opp(e, fmt("gupd_%s.n", lv)), Cp, Cp}}, L.dG, Cp, R, Cp, o));
      // dX1 (already holding fusion's share) += dG*[G>0]; dY = LN backward
      CK(cmpc_gconv_pre_bwd(dt, L.dG, L.G, L.Y, L.sums1, pptr(e, ln1 + "/gamma"), L.dX1, 1, L.dY, gptr(e, ln1 + "/gamma"), gptr(e, ln1 + "/beta"),
                            L.bs, B, N, Cp, C, st));
      // Y = gw_w . Z,  Z = gw_v^T . X1
      TnOpt zt; zt.nb2 = B; zt.a_bs = (int64_t)N * Tp; zt.d_bs = (int64_t)N * Cp; zt.o_bs = (int64_t)Tp * Cp;
      if (!e->lowrank) {     // (lowrank: Z is the forward pass's)
          CK(gemm_tn(e, st, dt, L.gw_v_t, Tp, Tp, L.X1, Cp, Cp, L.Zf, Cp, N, T, C, OFF0, zt));
          CK(cmpc_cast(DT_F32, L.Zf, dt, L.Z, (int64_t)B * Tp * Cp, st));
      }
      GemmOpt gw; gw.batch = B; gw.sC = (int64_t)N * Tp; gw.c_f32 = 1;
      CK(gemm_nt(st, dt, {{L.dY, Cp, L.Z, Cp, Cp, (int64_t)N * Cp, (int64_t)Tp * Cp}}, L.dgw_w, Tp, N, Tp, gw));
      CK(gemm_tn(e, st, dt, L.gw_w_t, Tp, Tp, L.dY, Cp, Cp, L.dZf, Cp, N, T, C, OFF0, zt));
      CK(cmpc_cast(DT_F32, L.dZf, dt, L.dZ, (int64_t)B * Tp * Cp, st));
      if (!e->lowrank) {     // dZ^T, the GEMM operand of dX1 += gw_v . dZ
          TnOpt z2; z2.nb2 = B; z2.a_bs = (int64_t)N * Cp; z2.d_bs = (int64_t)N * Tp; z2.o_bs = (int64_t)Cp * Tp;
          CK(gemm_tn(e, st, dt, L.dY, Cp, Cp, L.gw_w_t, Tp, Tp, L.dZtf, Tp, N, C, T, OFF0, z2));
          CK(cmpc_cast(DT_F32, L.dZtf, dt, L.dZt, (int64_t)B * Cp * Tp, st));
      }
      CK(gemm_nt(st, dt, {{L.X1, Cp, L.dZ, Cp, Cp, (int64_t)N * Cp, (int64_t)Tp * Cp}}, L.dgw_v, Tp, N, Tp, gw));
      GemmOpt ax; ax.n_valid = C; ax.batch = B; ax.sC = (int64_t)N * Cp; ax.accumulate = 1;
      if (e->lowrank) CK(cmpc_lowrank_nn(dt, L.gw_v_t, Tp, (int64_t)N * Tp, L.dZ, Cp, (int64_t)Tp * Cp, L.dX1, Cp, (int64_t)N * Cp, N, Cp, C, T, B, 1.0f, 1, st));
      else CK(gemm_nt(st, dt, {{L.gw_v_t, Tp, L.dZt, Tp, Tp, (int64_t)N * Tp, (int64_t)Cp * Tp}}, L.dX1, Cp, N, Cp, ax));
      CK(cmpc_graph_softmax_bwd(dt, L.dgw_w, L.dgw_v, L.gw_w, L.gw_v, L.A0, L.pr, e->mask, L.dA0, L.dA0_t, L.dpr, L.gsc2, B, N, T, Tp, st));
      // A0 = scale * (X1 . PT^T) + k0s
      GemmOpt a0 = ax; a0.alpha = scale;
      if (e->lowrank) CK(cmpc_lowrank_nn(dt, L.dA0_t, Tp, (int64_t)N * Tp, L.PT, Cp, (int64_t)Tp * Cp, L.dX1, Cp, (int64_t)N * Cp, N, Cp, C, T, B, scale, 1, st));
      else CK(gemm_nt(st, dt, {{L.dA0_t, Tp, L.PTt, B * Tp, Tp, (int64_t)N * Tp, (int64_t)Tp}}, L.dX1, Cp, N, Cp, a0));
      TnOpt pt = zt; pt.alpha = scale;
      CK(gemm_tn(e, st, dt, L.dA0_t, Tp, Tp, L.X1, Cp, Cp, L.dPT, Cp, N, T, C, OFF0, pt));
      CK(colsum(st, DT_F32, L.dA0, R, Tp, Tp, T, nullptr, nullptr, nullptr, ACT_NONE, L.dk0s, Tp, N));
      // k0s = scale * Wd . b_t2 ; PT = Wd . W_t2^T
      CK(cmpc_wcolsum(DT_F32, L.Wd, L.dk0s, gptr(e, fmt("spa_graph_trans2_%s/biases", lv)), 0, 1, B * Tp, Cp, C, scale, st));
      CK(gemm_nt(st, DT_F32, {{L.dPT, Cp, opp(e, fmt("t2_%s.t", lv)), Cp, Cp}}, L.dWd, Cp, B * Tp, Cp, o));
      CK(gemm_tn(e, st, DT_F32, L.dPT, Cp, Cp, L.Wd, Cp, Cp, gptr(e, fmt("spa_graph_trans2_%s/DW", lv)), C, B * Tp, C, C, OFF0, d));
      CK(cmpc_rank1_update(DT_F32, L.dWd, L.dk0s, pptr(e, fmt("spa_graph_trans2_%s/biases", lv)), nullptr, nullptr, 0, scale, 0.0f, 1, B * Tp, Cp, C, st));
      // Wd = wf . W_w + b_w   (rows t < T of every sample; pad rows of dWd are zero)
      CK(colsum(st, DT_F32, L.dWd, B * Tp, Cp, Cp, C, gptr(e, fmt("words_trans_%s/biases", lv))));
      TnOpt ww; ww.nb2 = B; ww.a_bs = (int64_t)T * Cp; ww.d_bs = (int64_t)Tp * Cp; ww.o_bs = 0; ww.defer = true;
      CK(gemm_tn(e, st, DT_F32, e->wf, Cp, Cp, L.dWd, Cp, Cp, gptr(e, fmt("words_trans_%s/DW", lv)), C, T, C, C, OFF0, ww));
      GemmOpt dw; dw.n_valid = C; dw.batch = B; dw.sC = (int64_t)T * Cp;
      CK(gemm_nt(st, DT_F32, {{L.dWd, Cp, opp(e, fmt("wtrans_%s.n", lv)), Cp, Cp, (int64_t)Tp * Cp, 0}}, L.dwf, Cp, T, Cp, dw)); }
    // -- mutan
    { CK(cmpc_mutan_bwd(dt, L.P, L.g, L.X1, L.mut_rstd, L.dX1, L.dg, B, N, Cp, C, st));
      void* dP = L.P;                         // overwritten in place
      const int64_t base_w = poff(e, fmt("vis_trans_%s_head1/DW", lv));
      Offs ov, os;
      for (int hd = 0; hd < 5; ++hd) {
          const int64_t rel = poff(e, fmt("vis_trans_%s_head%d/DW", lv, hd + 1)) - base_w;
          ov.push_back({0, (int64_t)hd * Cp, rel}); os.push_back({0, (int64_t)hd * Cp, rel + (int64_t)C * C});
          if (!e->mutan_bias_row) CK(colsum(st, dt, (char*)dP + (size_t)hd * Cp * es, R, 5 * Cp, Cp, C, gptr(e, fmt("vis_trans_%s_head%d/biases", lv, hd + 1))));
      }
      float* gwv = gptr(e, fmt("vis_trans_%s_head1/DW", lv));
      CK(gemm_tn(e, st, dt, L.X0, Cp, Cp, dP, 5 * Cp, Cp, gwv, C, R, C, C, ov, d));
      // the grid rows of the five heads' DW; with the constant-1 channel the product's ninth row is the head's bias gradient (five column-sum
      // passes over dP = 130 MB per level less)
      if (e->mutan_bias_row) CK(gemm_tn(e, st, dt, e->spatial1, 64, 64, dP, 5 * Cp, Cp, gwv, C, R, 9, C, os, d));
      else CK(gemm_tn(e, st, dt, e->spatial, 64, 64, dP, 5 * Cp, Cp, gwv, C, R, 8, C, os, d));
      GemmOpt o; o.n_valid = C;
      CK(gemm_nt(st, dt, {{dP, 5 * Cp, opp(e, fmt("mutan_%s.n", lv)), 5 * Cp, 5 * Cp}}, L.dX0, Cp, R, Cp, o));
      const int64_t base_l = poff(e, fmt("lang_trans_%s_head1/DW", lv));
      Offs ol;
      for (int hd = 0; hd < 5; ++hd) {
          CK(colsum(st, DT_F32, L.dg + (size_t)hd * Cp, B, 5 * Cp, Cp, C, gptr(e, fmt("lang_trans_%s_head%d/biases", lv, hd + 1)),
                    L.g + (size_t)hd * Cp, L.dg + (size_t)hd * Cp, ACT_TANH));
          ol.push_back({0, (int64_t)hd * Cp, poff(e, fmt("lang_trans_%s_head%d/DW", lv, hd + 1)) - base_l});
      }
      CK(gemm_tn(e, st, DT_F32, e->vl, Cp, Cp, L.dg, 5 * Cp, Cp, gptr(e, fmt("lang_trans_%s_head1/DW", lv)), C, B, Rr, C, ol, d));
      GemmOpt v; v.n_valid = Rr; v.accumulate = 1;          // on top of fusion's share
      CK(gemm_nt(st, DT_F32, {{L.dg, 5 * Cp, opp(e, fmt("mlang_%s.n", lv)), 5 * Cp, 5 * Cp}}, L.dvl, Cp, B, Cp, v)); }
    // -- lateral
    CK(cmpc_l2norm_rows_bwd(dt, L.dX0, L.X0, L.lat_rstd, L.dV, R, Cp, C, 0, st));
    if (!e->v5) CK(colsum(st, dt, L.dV, R, Cp, Cp, C, gptr(e, fmt("%s_lateral/biases", lv))));
    else CK(colsum(st, dt, L.dV, R, Cp, Cp, C, gptr(e, fmt("%s_lateral/biases", lv)), L.X0t, L.dV, ACT_TANH));
    float* glw = gptr(e, fmt("%s_lateral/DW", lv));
    if (e->cfg.conv5) {      // conv5=True: the tap's own gradient, for the caller's backbone backward (CMPC_model.py:427-430)
        GemmOpt o; o.n_valid = L.cin;
        CK(gemm_nt(st, dt, {{L.dV, Cp, opp(e, fmt("lat_%s.n", lv)), Cp, Cp}}, L.dfeat, L.cin, R, L.cin, o));
    }
    if (e->v5 && e->cfg.hsv) CK(gemm_tn(e, st, dt, e->hsv, 64, 64, L.dV, Cp, Cp, glw + (size_t)L.cin * C, C, R, 3, C, OFF0, d));
    return gemm_tn(e, st, dt, L.feat, L.cin, L.cin, L.dV, Cp, Cp, glw, C, R, L.cin, C, OFF0, d);
}

// ------------------------------------------------------------------------------------------
// The video model's level (CMPC_video_mm_tgraph_allvec.py:368-402, "vid:", batch 1): laterals + Mutan on the Fr sampled frames (rows
// f * N + n), a temporal graph over per-frame language-attention pools, and -- for the middle frame -- a temporal context, CMPC_model's
// word graph and the fusion over [lateral | spatial graph | temporal context | language | grid].  Two folds keep C x C products off the
// maps, as in CMPC_model's word graph: tg_vtrans goes into the query (logit = X1 . (W_v lt); the constant b_v . lt cancels in the softmax
// over the nodes, so its exact bias gradient is 0), mm_trans into the node side of the affinity (PT = ct . W_m^T, k0 = ct . b_m).
// ------------------------------------------------------------------------------------------
int level_lang_fwd_video(E* e, hipStream_t st, int li) {
    LevelBuf& L = e->lv[li]; const char* lv = lvn(e, li);
    const int T = e->T, C = e->C, Cp = e->Cp, Tp = e->Tp, M = e->M, Mp = e->Mp, dt = e->dt, Fr = e->Fr, N = e->N;
    // Mutan gates from the entity+attribute vector (vid:370-371), tiled over the frames (vid:341)
    { GemmOpt o; o.bias = (const float*)opp(e, fmt("mlang_%s.b", lv)); o.act = ACT_TANH;
      CK(gemm_nt(st, DT_F32, {{e->vl, Cp, opp(e, fmt("mlang_%s.t", lv)), Cp, Cp}}, L.g1, 5 * Cp, 1, 5 * Cp, o));
      hipLaunchKernelGGL(tile_rows_kernel, dim3((Fr * 5 * Cp + 255) / 256), dim3(256), 0, st, L.g1, L.g, Fr, 5 * Cp);
      CK(cmpc_check_launch("tile_rows")); }
    // temporal pooling query: lt = conv(tg_ltrans)(ac_lang); kqv = W_v . lt (vid:464-472)
    { GemmOpt o; o.n_valid = C; o.bias = pptr(e, fmt("tg_ltrans_%s/biases", lv));
      CK(gemm_nt(st, DT_F32, {{e->ac, Cp, opp(e, fmt("tgl_%s.t", lv)), Cp, Cp}}, L.lt, Cp, 1, Cp, o));
      GemmOpt k; k.n_valid = C;
      CK(gemm_nt(st, DT_F32, {{L.lt, Cp, opp(e, fmt("tgv_%s.n", lv)), Cp, Cp}}, L.kqv, Cp, 1, Cp, k)); }
    // word graph of the middle frame: as CMPC_model's language side (B = 1)
    { const float scale = 1.0f / sqrtf((float)C);
      GemmOpt a; a.n_valid = C; a.bias = pptr(e, fmt("words_trans_%s/biases", lv));
      CK(gemm_nt(st, DT_F32, {{e->wf, Cp, opp(e, fmt("wtrans_%s.t", lv)), Cp, Cp}}, L.Wd, Cp, T, Cp, a));
      GemmOpt b; b.n_valid = C;
      CK(gemm_nt(st, DT_F32, {{L.Wd, Cp, opp(e, fmt("t2_%s.n", lv)), Cp, Cp}}, L.PTf, Cp, Tp, Cp, b));
      CK(cmpc_cast(DT_F32, L.PTf, dt, L.PT, (int64_t)Tp * Cp, st));
      if (!e->lowrank) CK(transpose_cast(st, L.PTf, dt, L.PTt, Tp, Cp));
      CK(cmpc_rowdot1(DT_F32, L.Wd, pptr(e, fmt("spa_graph_trans2_%s/biases", lv)), 0, L.k0s, 1, Tp, Cp, C, scale, st));
      hipLaunchKernelGGL(col_get_kernel, dim3((T + 255) / 256), dim3(256), 0, st, e->parse, e->NC, 2, L.pr, T);
      CK(cmpc_check_launch("col_get")); }
    // fusion: the tiled language vector (all but "unnecessary", vid:394-395) as a per-sample bias, the grid as a per-position bias
    { GemmOpt sv; sv.n_valid = M;
      CK(gemm_nt(st, DT_F32, {{e->nec, Cp, opp(e, fmt("fusl_%s.t", lv)), Cp, Cp}}, L.sb, Mp, 1, Mp, sv));
      GemmOpt pv; pv.n_valid = M; pv.c_f32 = 1;
      CK(gemm_nt(st, dt, {{e->spatial, 64, opp(e, fmt("fussp_%s.t", lv)), 64, 64}}, L.pb, Mp, N, Mp, pv)); }
    return CMPC_OK;
}

int level_fwd_video(E* e, hipStream_t st, int li, const float* target) {
    LevelBuf& L = e->lv[li]; const char* lv = lvn(e, li);
    const int N = e->N, T = e->T, C = e->C, Cp = e->Cp, Tp = e->Tp, M = e->M, Mp = e->Mp, dt = e->dt, Fr = e->Fr, RF = e->RF;
    const float scale = 1.0f / sqrtf((float)C);
    // -- laterals + Mutan on every sampled frame (vid:151-157,330-366)
    { GemmOpt o; o.n_valid = C; o.bias = pptr(e, fmt("%s_lateral/biases", lv));
      CK(gemm_nt(st, dt, {{L.feat, L.cin, opp(e, fmt("lat_%s.t", lv)), L.cin, L.cin}}, L.X0, Cp, RF, Cp, o));
      CK(cmpc_l2norm_rows_fwd(dt, L.X0, L.X0, L.lat_rstd, nullptr, RF, Cp, C, st)); }
    { const int ldk = Cp + 64; const std::string k = fmt("mutan_%s.t", lv);
      GemmOpt p; p.bias = (const float*)opp(e, fmt("mutan_%s.b", lv)); if (e->mutan_epilogue) p.act = ACT_TANH;
      CK(gemm_nt(st, dt, {{L.X0, Cp, opp(e, k), ldk, Cp}, {e->spatial, 64, opp(e, k, 0, Cp), ldk, 64}}, L.P, 5 * Cp, RF, 5 * Cp, p));
      CK(cmpc_mutan_fwd(dt, L.P, L.g, L.X1, L.mut_rstd, Fr, N, Cp, C, e->mutan_epilogue ? 1 : 0, st)); }
    // -- temporal graph (vid:458-503): per-frame attention pooling with the action vector, 5 x 5 adjacency, graph_conv, l2norm
    CK(cmpc_rowdot1(dt, L.X1, L.kqv, 0, L.tlog, Fr, N, Cp, C, scale, st));
    CK(cmpc_softmax_n_fwd(L.tlog, L.tatt, Fr, N, st));
    CK(cmpc_wcolsum(dt, L.X1, L.tatt, L.TG, Cp, Fr, N, Cp, C, 1.0f, st));
    { GemmOpt o; o.n_valid = C; o.bias = pptr(e, fmt("tg_query_%s/biases", lv));
      CK(gemm_nt(st, DT_F32, {{L.TG, Cp, opp(e, fmt("tgq_%s.t", lv)), Cp, Cp}}, L.q, Cp, Fr, Cp, o));
      o.bias = pptr(e, fmt("tg_key_%s/biases", lv));
      CK(gemm_nt(st, DT_F32, {{L.TG, Cp, opp(e, fmt("tgk_%s.t", lv)), Cp, Cp}}, L.k, Cp, Fr, Cp, o)); }
    hipLaunchKernelGGL(tgraph_fwd_kernel, dim3(1), dim3(256), 0, st, L.q, L.k, L.TG, L.adj, L.TY, Fr, Cp, C, scale);
    CK(cmpc_check_launch("tgraph_fwd"));
    { const std::string ln1 = fmt("gconv_feat_ln_temp_graph_%s", lv), ln2 = fmt("gconv_update_ln_temp_graph_%s", lv);
      CK(cmpc_sample_stats(DT_F32, L.TY, L.tsums1, 1, Fr, Cp, C, st));
      CK(cmpc_gconv_pre_fwd(DT_F32, L.TY, L.TG, L.tsums1, pptr(e, ln1 + "/gamma"), pptr(e, ln1 + "/beta"), L.TG1, 1, Fr, Cp, C, st));
      GemmOpt u; u.n_valid = C; u.bias = pptr(e, fmt("gconv_update_temp_graph_%s/biases", lv));
      CK(gemm_nt(st, DT_F32, {{L.TG1, Cp, opp(e, fmt("tgu_%s.t", lv)), Cp, Cp}}, L.TU, Cp, Fr, Cp, u));
      CK(cmpc_sample_stats(DT_F32, L.TU, L.tsums2, 1, Fr, Cp, C, st));
      CK(cmpc_gconv_post_fwd(DT_F32, L.TU, L.tsums2, pptr(e, ln2 + "/gamma"), pptr(e, ln2 + "/beta"), L.TGN, L.rrow_t, 1, Fr, Cp, C, st)); }
    // -- temporal context of the middle frame (vid:505-530): every pixel attends over the Fr graph nodes
    { GemmOpt o; o.n_valid = C; o.bias = pptr(e, fmt("ctx_trans_%s/biases", lv));
      CK(gemm_nt(st, DT_F32, {{L.TGN, Cp, opp(e, fmt("ctxt_%s.t", lv)), Cp, Cp}}, L.ctv, Cp, Fr, Cp, o));
      GemmOpt b; b.n_valid = C;
      CK(gemm_nt(st, DT_F32, {{L.ctv, Cp, opp(e, fmt("mmt_%s.n", lv)), Cp, Cp}}, L.ctPTf, Cp, Tp, Cp, b));
      CK(cmpc_cast(DT_F32, L.ctPTf, dt, L.ctPT, (int64_t)Tp * Cp, st));
      if (!e->lowrank) CK(transpose_cast(st, L.ctPTf, dt, L.ctPTt, Tp, Cp));
      CK(cmpc_rowdot1(DT_F32, L.ctv, pptr(e, fmt("mm_trans_%s/biases", lv)), 0, L.ctk0s, 1, Tp, Cp, C, scale, st));
      GemmOpt c; c.c_f32 = 1; c.alpha = scale; c.sbias = L.ctk0s; c.ld_sbias = Tp; c.rows_per_sample = N;
      CK(gemm_nt(st, dt, {{L.X1m, Cp, L.ctPT, Cp, Cp}}, L.ctA0, Tp, N, Tp, c));
      CK(cmpc_graph_softmax_fwd(dt, 0, L.ctA0, e->ones_t, e->ones_t, L.ctA, L.ctGv, L.ctA_t, L.ctGv_t, L.ctsc, 1, N, Fr, Tp, st));
      if (e->lowrank) {
          CK(cmpc_cast(DT_F32, L.TGN, dt, L.TGN16, (int64_t)Tp * Cp, st));
          CK(cmpc_lowrank_nn(dt, L.ctA_t, Tp, 0, L.TGN16, Cp, 0, L.GLO, Cp, 0, N, Cp, C, Fr, 1, 1.0f, 0, st));
      } else {
          CK(cmpc_cast(DT_F32, L.TGN, dt, L.TGN16, (int64_t)Tp * Cp, st));
          CK(transpose_cast(st, L.TGN, dt, L.TGNt, Tp, Cp));
          GemmOpt y; y.n_valid = C;
          CK(gemm_nt(st, dt, {{L.ctA_t, Tp, L.TGNt, Tp, Tp}}, L.GLO, Cp, N, Cp, y));
      }
      CK(cmpc_l2norm_rows_fwd(dt, L.GLO, L.CTX, L.ctx_rstd, nullptr, N, Cp, C, st)); }
    // -- spatial word graph on the middle frame's multimodal map (vid:388-390,435-456): CMPC_model's, B = 1
    { GemmOpt c; c.c_f32 = 1; c.alpha = scale; c.sbias = L.k0s; c.ld_sbias = Tp; c.rows_per_sample = N;
      CK(gemm_nt(st, dt, {{L.X1m, Cp, L.PT, Cp, Cp}}, L.A0, Tp, N, Tp, c));
      CK(cmpc_graph_softmax_fwd(dt, 0, L.A0, L.pr, e->mask, L.gw_w, L.gw_v, L.gw_w_t, L.gw_v_t, L.gsc, 1, N, T, Tp, st));
      if (e->lowrank) {
          TnOpt zt;
          CK(gemm_tn(e, st, dt, L.gw_v_t, Tp, Tp, L.X1m, Cp, Cp, L.Zf, Cp, N, T, C, OFF0, zt));
          CK(cmpc_cast(DT_F32, L.Zf, dt, L.Z, (int64_t)Tp * Cp, st));
          CK(cmpc_lowrank_nn(dt, L.gw_w_t, Tp, 0, L.Z, Cp, 0, L.Y, Cp, 0, N, Cp, C, T, 1, 1.0f, 0, st));
      } else {
          TnOpt z;
          CK(gemm_tn(e, st, dt, L.X1m, Cp, Cp, L.gw_v_t, Tp, Tp, L.Ztf, Tp, N, C, T, OFF0, z));
          CK(cmpc_cast(DT_F32, L.Ztf, dt, L.Zt, (int64_t)Cp * Tp, st));
          GemmOpt y; y.n_valid = C;
          CK(gemm_nt(st, dt, {{L.gw_w_t, Tp, L.Zt, Tp, Tp}}, L.Y, Cp, N, Cp, y));
      }
      CK(cmpc_sample_stats(dt, L.Y, L.sums1, 1, N, Cp, C, st));
      const std::string ln1 = fmt("gconv_feat_ln_spa_graph_%s", lv), ln2 = fmt("gconv_update_ln_spa_graph_%s", lv);
      CK(cmpc_gconv_pre_fwd(dt, L.Y, L.X1m, L.sums1, pptr(e, ln1 + "/gamma"), pptr(e, ln1 + "/beta"), L.G, 1, N, Cp, C, st));
      GemmOpt u; u.n_valid = C; u.bias = pptr(e, fmt("gconv_update_spa_graph_%s/biases", lv));
      CK(gemm_nt(st, dt, {{L.G, Cp, opp(e, fmt("gupd_%s.t", lv)), Cp, Cp}}, L.U, Cp, N, Cp, u));
      CK(cmpc_sample_stats(dt, L.U, L.sums2, 1, N, Cp, C, st));
      CK(cmpc_gconv_post_fwd(dt, L.U, L.sums2, pptr(e, ln2 + "/gamma"), pptr(e, ln2 + "/beta"), L.X2, L.rrow, 1, N, Cp, C, st)); }
    // -- fusion over [lateral of the middle frame | spatial graph | temporal context] + language and grid biases (vid:396-401)
    { const int ldk = 3 * Cp; const std::string k = fmt("fus_%s.t", lv);
      GemmOpt f; f.n_valid = M; f.bias = pptr(e, fmt("fusion_%s/biases", lv)); f.sbias = L.sb; f.ld_sbias = Mp; f.pbias = L.pb; f.ld_pbias = Mp;
      f.rows_per_sample = N; f.act = ACT_RELU;
      CK(gemm_nt(st, dt, {{L.X0m, Cp, opp(e, k), ldk, Cp}, {L.X2, Cp, opp(e, k, 0, Cp), ldk, Cp}, {L.CTX, Cp, opp(e, k, 0, 2 * Cp), ldk, Cp}}, L.F, Mp, N, Mp, f)); }
    CK(cmpc_score_conv_fwd(dt, L.F, pptr(e, fmt("score_%s/DW", lv)), pptr(e, fmt("score_%s/biases", lv)), L.score, 1, e->h, e->w, Mp, M, st));
    return cmpc_upsample_fwd(L.score, L.up, nullptr, target, L.loss, L.iu, L.iu + 1, 1, e->h, e->w, e->H, e->W, st);
}

// in: L.dfus (sum of the exchange modules' gradients of this level's fusion map); out: every parameter gradient of the level, L.dvl (d nec,
// the fusion's language bias), L.dea (d entity+attribute vector), L.dac (d action vector), L.dwf / L.dpr (word side of the spatial graph)
int level_bwd_video(E* e, hipStream_t st, int li, const float* target) {
    LevelBuf& L = e->lv[li]; const char* lv = lvn(e, li);
    const int N = e->N, T = e->T, C = e->C, Cp = e->Cp, Tp = e->Tp, M = e->M, Mp = e->Mp, dt = e->dt, Rr = e->RNN, Fr = e->Fr, RF = e->RF, es = e->esz;
    const float scale = 1.0f / sqrtf((float)C);
    TnOpt d; d.defer = true;
    GemmOpt oc; oc.n_valid = C;
    CK(cmpc_upsample_loss_bwd(L.up, target, L.dscore, e->cfg.loss_w[1 + li] * e->cfg.loss_scale, 1, e->h, e->w, e->H, e->W, st));
    CK(cmpc_score_conv_bwd(dt, L.dscore, L.F, pptr(e, fmt("score_%s/DW", lv)), L.dfus, 1, gptr(e, fmt("score_%s/DW", lv)),
                           gptr(e, fmt("score_%s/biases", lv)), 1, e->h, e->w, Mp, M, st));
    // -- fusion
    { CK(colsum(st, dt, L.dfus, N, Mp, Mp, M, gptr(e, fmt("fusion_%s/biases", lv)), L.F, L.dpre, ACT_RELU, L.dsb, Mp, N));
      float* gw = gptr(e, fmt("fusion_%s/DW", lv));
      CK(gemm_tn(e, st, dt, L.X0m, Cp, Cp, L.dpre, Mp, Mp, gw, M, N, C, M, OFF0, d));
      CK(gemm_tn(e, st, dt, L.X2, Cp, Cp, L.dpre, Mp, Mp, gw + (size_t)C * M, M, N, C, M, OFF0, d));
      CK(gemm_tn(e, st, dt, L.CTX, Cp, Cp, L.dpre, Mp, Mp, gw + (size_t)2 * C * M, M, N, C, M, OFF0, d));
      CK(gemm_tn(e, st, DT_F32, e->nec, Cp, Cp, L.dsb, Mp, Mp, gw + (size_t)3 * C * M, M, 1, Rr, M, OFF0, d));
      CK(gemm_tn(e, st, dt, e->spatial, 64, 64, L.dpre, Mp, Mp, gw + (size_t)(3 * C + Rr) * M, M, N, 8, M, OFF0, d));
      const std::string k = fmt("fus_%s.n", lv);
      CK(gemm_nt(st, dt, {{L.dpre, Mp, opp(e, k), Mp, Mp}}, L.dX0f, Cp, N, Cp, oc));
      CK(gemm_nt(st, dt, {{L.dpre, Mp, opp(e, k, Cp, 0), Mp, Mp}}, L.dX2, Cp, N, Cp, oc));
      CK(gemm_nt(st, dt, {{L.dpre, Mp, opp(e, k, 2 * Cp, 0), Mp, Mp}}, L.dCTX, Cp, N, Cp, oc));
      GemmOpt v; v.n_valid = Rr;
      CK(gemm_nt(st, DT_F32, {{L.dsb, Mp, opp(e, fmt("fusl_%s.n", lv)), Mp, Mp}}, L.dvl, Cp, 1, Cp, v)); }
    void* dX1m = (char*)L.dX1 + (size_t)(Fr / 2) * N * Cp * es;        // L.dX1 [RF, Cp] starts at zero (backward region): every producer accumulates
    // -- spatial word graph (CMPC_model's backward with B = 1; its node input is the middle frame of the multimodal map)
    { const std::string ln1 = fmt("gconv_feat_ln_spa_graph_%s", lv), ln2 = fmt("gconv_update_ln_spa_graph_%s", lv);
      CK(cmpc_gconv_post_bwd(dt, L.dX2, L.X2, L.rrow, L.U, L.sums2, pptr(e, ln2 + "/gamma"), L.dU, gptr(e, ln2 + "/gamma"), gptr(e, ln2 + "/beta"), L.bs, 1, N, Cp, C, st));
      CK(colsum(st, dt, L.dU, N, Cp, Cp, C, gptr(e, fmt("gconv_update_spa_graph_%s/biases", lv))));
      CK(gemm_tn(e, st, dt, L.G, Cp, Cp, L.dU, Cp, Cp, gptr(e, fmt("gconv_update_spa_graph_%s/DW", lv)), C, N, C, C, OFF0, d));
      CK(gemm_nt(st, dt, {{L.dU, Cp, opp(e, fmt("gupd_%s.n", lv)), Cp, Cp}}, L.dG, Cp, N, Cp, oc));
      CK(cmpc_gconv_pre_bwd(dt, L.dG, L.G, L.Y, L.sums1, pptr(e, ln1 + "/gamma"), dX1m, 1, L.dY, gptr(e, ln1 + "/gamma"), gptr(e, ln1 + "/beta"), L.bs, 1, N, Cp, C, st));
      TnOpt zt;
      if (!e->lowrank) {
          CK(gemm_tn(e, st, dt, L.gw_v_t, Tp, Tp, L.X1m, Cp, Cp, L.Zf, Cp, N, T, C, OFF0, zt));
          CK(cmpc_cast(DT_F32, L.Zf, dt, L.Z, (int64_t)Tp * Cp, st));
      }
      GemmOpt gw; gw.c_f32 = 1;
      CK(gemm_nt(st, dt, {{L.dY, Cp, L.Z, Cp, Cp}}, L.dgw_w, Tp, N, Tp, gw));
      CK(gemm_tn(e, st, dt, L.gw_w_t, Tp, Tp, L.dY, Cp, Cp, L.dZf, Cp, N, T, C, OFF0, zt));
      CK(cmpc_cast(DT_F32, L.dZf, dt, L.dZ, (int64_t)Tp * Cp, st));
      if (!e->lowrank) {
          TnOpt z2;
          CK(gemm_tn(e, st, dt, L.dY, Cp, Cp, L.gw_w_t, Tp, Tp, L.dZtf, Tp, N, C, T, OFF0, z2));
          CK(cmpc_cast(DT_F32, L.dZtf, dt, L.dZt, (int64_t)Cp * Tp, st));
      }
      CK(gemm_nt(st, dt, {{L.X1m, Cp, L.dZ, Cp, Cp}}, L.dgw_v, Tp, N, Tp, gw));
      GemmOpt ax; ax.n_valid = C; ax.accumulate = 1;
      if (e->lowrank) CK(cmpc_lowrank_nn(dt, L.gw_v_t, Tp, 0, L.dZ, Cp, 0, dX1m, Cp, 0, N, Cp, C, T, 1, 1.0f, 1, st));
      else CK(gemm_nt(st, dt, {{L.gw_v_t, Tp, L.dZt, Tp, Tp}}, dX1m, Cp, N, Cp, ax));
      CK(cmpc_graph_softmax_bwd(dt, L.dgw_w, L.dgw_v, L.gw_w, L.gw_v, L.A0, L.pr, e->mask, L.dA0, L.dA0_t, L.dpr, L.gsc2, 1, N, T, Tp, st));
      GemmOpt a0 = ax; a0.alpha = scale;
      if (e->lowrank) CK(cmpc_lowrank_nn(dt, L.dA0_t, Tp, 0, L.PT, Cp, 0, dX1m, Cp, 0, N, Cp, C, T, 1, scale, 1, st));
      else CK(gemm_nt(st, dt, {{L.dA0_t, Tp, L.PTt, Tp, Tp}}, dX1m, Cp, N, Cp, a0));
      TnOpt pt; pt.alpha = scale;
      CK(gemm_tn(e, st, dt, L.dA0_t, Tp, Tp, L.X1m, Cp, Cp, L.dPT, Cp, N, T, C, OFF0, pt));
      CK(colsum(st, DT_F32, L.dA0, N, Tp, Tp, T, nullptr, nullptr, nullptr, ACT_NONE, L.dk0s, Tp, N));
      CK(cmpc_wcolsum(DT_F32, L.Wd, L.dk0s, gptr(e, fmt("spa_graph_trans2_%s/biases", lv)), 0, 1, Tp, Cp, C, scale, st));
      CK(gemm_nt(st, DT_F32, {{L.dPT, Cp, opp(e, fmt("t2_%s.t", lv)), Cp, Cp}}, L.dWd, Cp, Tp, Cp, oc));
      CK(gemm_tn(e, st, DT_F32, L.dPT, Cp, Cp, L.Wd, Cp, Cp, gptr(e, fmt("spa_graph_trans2_%s/DW", lv)), C, Tp, C, C, OFF0, d));
      CK(cmpc_rank1_update(DT_F32, L.dWd, L.dk0s, pptr(e, fmt("spa_graph_trans2_%s/biases", lv)), nullptr, nullptr, 0, scale, 0.0f, 1, Tp, Cp, C, st));
      CK(colsum(st, DT_F32, L.dWd, Tp, Cp, Cp, C, gptr(e, fmt("words_trans_%s/biases", lv))));
      CK(gemm_tn(e, st, DT_F32, e->wf, Cp, Cp, L.dWd, Cp, Cp, gptr(e, fmt("words_trans_%s/DW", lv)), C, T, C, C, OFF0, d));
      CK(gemm_nt(st, DT_F32, {{L.dWd, Cp, opp(e, fmt("wtrans_%s.n", lv)), Cp, Cp}}, L.dwf, Cp, T, Cp, oc)); }
    // -- temporal context
    { CK(cmpc_l2norm_rows_bwd(dt, L.dCTX, L.CTX, L.ctx_rstd, L.dGLO, N, Cp, C, 0, st));
      GemmOpt gw; gw.c_f32 = 1;
      CK(gemm_nt(st, dt, {{L.dGLO, Cp, L.TGN16, Cp, Cp}}, L.dctA, Tp, N, Tp, gw));
      TnOpt zt;
      CK(gemm_tn(e, st, dt, L.ctA_t, Tp, Tp, L.dGLO, Cp, Cp, L.dTGN, Cp, N, Fr, C, OFF0, zt));                  // d tgraph via the attention-weighted sum
      CK(cmpc_graph_softmax_bwd(dt, L.dctA, e->zeros_nt, L.ctA, L.ctGv, L.ctA0, e->ones_t, e->ones_t, L.dctA0, L.dctA0_t, L.dprc, L.ctsc2, 1, N, Fr, Tp, st));
      GemmOpt a0; a0.n_valid = C; a0.accumulate = 1; a0.alpha = scale;
      if (e->lowrank) CK(cmpc_lowrank_nn(dt, L.dctA0_t, Tp, 0, L.ctPT, Cp, 0, dX1m, Cp, 0, N, Cp, C, Fr, 1, scale, 1, st));
      else CK(gemm_nt(st, dt, {{L.dctA0_t, Tp, L.ctPTt, Tp, Tp}}, dX1m, Cp, N, Cp, a0));
      TnOpt pt; pt.alpha = scale;
      CK(gemm_tn(e, st, dt, L.dctA0_t, Tp, Tp, L.X1m, Cp, Cp, L.dctPT, Cp, N, Fr, C, OFF0, pt));
      CK(colsum(st, DT_F32, L.dctA0, N, Tp, Tp, Fr, nullptr, nullptr, nullptr, ACT_NONE, L.dctk0s, Tp, N));
      CK(cmpc_wcolsum(DT_F32, L.ctv, L.dctk0s, gptr(e, fmt("mm_trans_%s/biases", lv)), 0, 1, Tp, Cp, C, scale, st));
      CK(gemm_nt(st, DT_F32, {{L.dctPT, Cp, opp(e, fmt("mmt_%s.t", lv)), Cp, Cp}}, L.dctv, Cp, Tp, Cp, oc));
      CK(gemm_tn(e, st, DT_F32, L.dctPT, Cp, Cp, L.ctv, Cp, Cp, gptr(e, fmt("mm_trans_%s/DW", lv)), C, Tp, C, C, OFF0, d));
      CK(cmpc_rank1_update(DT_F32, L.dctv, L.dctk0s, pptr(e, fmt("mm_trans_%s/biases", lv)), nullptr, nullptr, 0, scale, 0.0f, 1, Tp, Cp, C, st));
      // ct = tgraph . W_c + b_c on the Fr node rows (pad rows of dctv are zero)
      CK(colsum(st, DT_F32, L.dctv, Fr, Cp, Cp, C, gptr(e, fmt("ctx_trans_%s/biases", lv))));
      CK(gemm_tn(e, st, DT_F32, L.TGN, Cp, Cp, L.dctv, Cp, Cp, gptr(e, fmt("ctx_trans_%s/DW", lv)), C, Fr, C, C, OFF0, d));
      GemmOpt ag; ag.n_valid = C; ag.accumulate = 1;
      CK(gemm_nt(st, DT_F32, {{L.dctv, Cp, opp(e, fmt("ctxt_%s.n", lv)), Cp, Cp}}, L.dTGN, Cp, Fr, Cp, ag)); }
    // -- temporal graph
    { const std::string ln1 = fmt("gconv_feat_ln_temp_graph_%s", lv), ln2 = fmt("gconv_update_ln_temp_graph_%s", lv);
      CK(cmpc_gconv_post_bwd(DT_F32, L.dTGN, L.TGN, L.rrow_t, L.TU, L.tsums2, pptr(e, ln2 + "/gamma"), L.dTU, gptr(e, ln2 + "/gamma"), gptr(e, ln2 + "/beta"), L.tbs, 1, Fr, Cp, C, st));
      CK(colsum(st, DT_F32, L.dTU, Fr, Cp, Cp, C, gptr(e, fmt("gconv_update_temp_graph_%s/biases", lv))));
      CK(gemm_tn(e, st, DT_F32, L.TG1, Cp, Cp, L.dTU, Cp, Cp, gptr(e, fmt("gconv_update_temp_graph_%s/DW", lv)), C, Fr, C, C, OFF0, d));
      CK(gemm_nt(st, DT_F32, {{L.dTU, Cp, opp(e, fmt("tgu_%s.n", lv)), Cp, Cp}}, L.dTG1, Cp, Fr, Cp, oc));
      CK(cmpc_gconv_pre_bwd(DT_F32, L.dTG1, L.TG1, L.TY, L.tsums1, pptr(e, ln1 + "/gamma"), L.dTG, 0, L.dTY, gptr(e, ln1 + "/gamma"), gptr(e, ln1 + "/beta"), L.tbs, 1, Fr, Cp, C, st));
      hipLaunchKernelGGL(tgraph_bwd_kernel, dim3(1), dim3(256), 0, st, L.dTY, L.adj, L.TG, L.q, L.k, L.dTG, 1, L.dq, L.dk, Fr, Cp, C, scale);
      CK(cmpc_check_launch("tgraph_bwd"));
      GemmOpt ag; ag.n_valid = C; ag.accumulate = 1;
      CK(colsum(st, DT_F32, L.dq, Fr, Cp, Cp, C, gptr(e, fmt("tg_query_%s/biases", lv))));
      CK(gemm_tn(e, st, DT_F32, L.TG, Cp, Cp, L.dq, Cp, Cp, gptr(e, fmt("tg_query_%s/DW", lv)), C, Fr, C, C, OFF0, d));
      CK(gemm_nt(st, DT_F32, {{L.dq, Cp, opp(e, fmt("tgq_%s.n", lv)), Cp, Cp}}, L.dTG, Cp, Fr, Cp, ag));
      CK(colsum(st, DT_F32, L.dk, Fr, Cp, Cp, C, gptr(e, fmt("tg_key_%s/biases", lv))));
      CK(gemm_tn(e, st, DT_F32, L.TG, Cp, Cp, L.dk, Cp, Cp, gptr(e, fmt("tg_key_%s/DW", lv)), C, Fr, C, C, OFF0, d));
      CK(gemm_nt(st, DT_F32, {{L.dk, Cp, opp(e, fmt("tgk_%s.n", lv)), Cp, Cp}}, L.dTG, Cp, Fr, Cp, ag));
      // pooling: TG[f] = sum_n tatt[f, n] X1[f, n];  tlog = scale * X1 . kqv
      CK(cmpc_rowdot1(dt, L.X1, L.dTG, Cp, L.dtatt, Fr, N, Cp, C, 1.0f, st));
      CK(cmpc_softmax_n_bwd(L.dtatt, L.tatt, L.dtlog, Fr, N, st));
      CK(cmpc_rank1_update(dt, L.dX1, L.tatt, L.dTG, nullptr, nullptr, Cp, 1.0f, 0.0f, Fr, N, Cp, C, st));
      CK(cmpc_rank1_update(dt, L.dX1, L.dtlog, L.kqv, nullptr, nullptr, 0, scale, 0.0f, Fr, N, Cp, C, st));
      CK(cmpc_wcolsum(dt, L.X1, L.dtlog, L.dkqv, Cp, 1, RF, Cp, C, scale, st));
      CK(gemm_nt(st, DT_F32, {{L.dkqv, Cp, opp(e, fmt("tgv_%s.t", lv)), Cp, Cp}}, L.dlt, Cp, 1, Cp, oc));
      CK(gemm_tn(e, st, DT_F32, L.dkqv, Cp, Cp, L.lt, Cp, Cp, gptr(e, fmt("tg_vtrans_%s/DW", lv)), C, 1, C, C, OFF0, d));
      CK(colsum(st, DT_F32, L.dlt, 1, Cp, Cp, C, gptr(e, fmt("tg_ltrans_%s/biases", lv))));
      CK(gemm_tn(e, st, DT_F32, e->ac, Cp, Cp, L.dlt, Cp, Cp, gptr(e, fmt("tg_ltrans_%s/DW", lv)), C, 1, Rr, C, OFF0, d));
      GemmOpt v; v.n_valid = Rr;
      CK(gemm_nt(st, DT_F32, {{L.dlt, Cp, opp(e, fmt("tgl_%s.n", lv)), Cp, Cp}}, L.dac, Cp, 1, Cp, v)); }
    // -- Mutan on the Fr frames (dX1 complete), laterals
    { CK(cmpc_mutan_bwd(dt, L.P, L.g, L.X1, L.mut_rstd, L.dX1, L.dg, Fr, N, Cp, C, st));
      void* dP = L.P;
      const int64_t base_w = poff(e, fmt("vis_trans_%s_head1/DW", lv));
      Offs ov, os;
      for (int hd = 0; hd < 5; ++hd) {
          const int64_t rel = poff(e, fmt("vis_trans_%s_head%d/DW", lv, hd + 1)) - base_w;
          ov.push_back({0, (int64_t)hd * Cp, rel}); os.push_back({0, (int64_t)hd * Cp, rel + (int64_t)C * C});
          if (!e->mutan_bias_row) CK(colsum(st, dt, (char*)dP + (size_t)hd * Cp * es, RF, 5 * Cp, Cp, C, gptr(e, fmt("vis_trans_%s_head%d/biases", lv, hd + 1))));
      }
      float* gwv = gptr(e, fmt("vis_trans_%s_head1/DW", lv));
      CK(gemm_tn(e, st, dt, L.X0, Cp, Cp, dP, 5 * Cp, Cp, gwv, C, RF, C, C, ov, d));
      if (e->mutan_bias_row) CK(gemm_tn(e, st, dt, e->spatial1, 64, 64, dP, 5 * Cp, Cp, gwv, C, RF, 9, C, os, d));
      else CK(gemm_tn(e, st, dt, e->spatial, 64, 64, dP, 5 * Cp, Cp, gwv, C, RF, 8, C, os, d));
      CK(gemm_nt(st, dt, {{dP, 5 * Cp, opp(e, fmt("mutan_%s.n", lv)), 5 * Cp, 5 * Cp}}, L.dX0, Cp, RF, Cp, oc));
      // the gates were tiled over the frames: their gradient is the sum over the frames
      hipLaunchKernelGGL(sum_rows_kernel, dim3((5 * Cp + 255) / 256), dim3(256), 0, st, L.dg, L.g1 /* reused: d gates [5 Cp] */, Fr, 5 * Cp, 0);
      CK(cmpc_check_launch("sum_rows"));
      float* dg1 = L.g1;
      const int64_t base_l = poff(e, fmt("lang_trans_%s_head1/DW", lv));
      Offs ol;
      for (int hd = 0; hd < 5; ++hd) {
          CK(colsum(st, DT_F32, dg1 + (size_t)hd * Cp, 1, 5 * Cp, Cp, C, gptr(e, fmt("lang_trans_%s_head%d/biases", lv, hd + 1)),
                    L.g + (size_t)hd * Cp, dg1 + (size_t)hd * Cp, ACT_TANH));
          ol.push_back({0, (int64_t)hd * Cp, poff(e, fmt("lang_trans_%s_head%d/DW", lv, hd + 1)) - base_l});
      }
      CK(gemm_tn(e, st, DT_F32, e->vl, Cp, Cp, dg1, 5 * Cp, Cp, gptr(e, fmt("lang_trans_%s_head1/DW", lv)), C, 1, Rr, C, ol, d));
      GemmOpt v; v.n_valid = Rr;
      CK(gemm_nt(st, DT_F32, {{dg1, 5 * Cp, opp(e, fmt("mlang_%s.n", lv)), 5 * Cp, 5 * Cp}}, L.dea, Cp, 1, Cp, v)); }
    // the middle frame's lateral also fed the fusion conv
    CK(add_n(st, dt, (char*)L.dX0 + (size_t)(Fr / 2) * N * Cp * es, {L.dX0f}, true, (long)N * Cp));
    CK(cmpc_l2norm_rows_bwd(dt, L.dX0, L.X0, L.lat_rstd, L.dV, RF, Cp, C, 0, st));
    CK(colsum(st, dt, L.dV, RF, Cp, Cp, C, gptr(e, fmt("%s_lateral/biases", lv))));
    if (e->cfg.conv5) {      // finetune=True (vid:554-557): the tap's own gradient, for the caller's backbone backward
        GemmOpt o; o.n_valid = L.cin;
        CK(gemm_nt(st, dt, {{L.dV, Cp, opp(e, fmt("lat_%s.n", lv)), Cp, Cp}}, L.dfeat, L.cin, RF, L.cin, o));
    }
    return gemm_tn(e, st, dt, L.feat, L.cin, L.cin, L.dV, Cp, Cp, gptr(e, fmt("%s_lateral/DW", lv)), C, RF, L.cin, C, OFF0, d);
}

// ------------------------------------------------------------------------------------------
// stage: gated_exchange_module + l2_normalize, CMPC_model.py:194-259,271-284 (key folded into the query)
// ------------------------------------------------------------------------------------------
// the language side of an exchange module (query and folded key: functions of nec_lang only), run before the visual features exist
int exchange_lang_fwd(E* e, hipStream_t st, int xi) {
    ExgBuf& X = e->ex[xi]; const char* lv = exn(e, xi);
    const int B = e->B, Cp = e->Cp, M = e->M, Mp = e->Mp;
    GemmOpt q; q.n_valid = M; q.bias = pptr(e, fmt("lang_query_%sgv_f1/biases", lv));
    CK(gemm_nt(st, DT_F32, {{e->nec, Cp, opp(e, fmt("query_%s.t", lv)), Cp, Cp}}, X.q, Mp, B, Mp, q));
    GemmOpt k; k.n_valid = M;
    return gemm_nt(st, DT_F32, {{X.q, Mp, opp(e, fmt("key_%s.n", lv)), Mp, Mp}}, X.kq, Mp, B, Mp, k);
}
// f2 == nullptr: one gated branch (CMPCv5_BiLSTM_model.py:343-346), and tanh before the all-dims l2_normalize of gv_lang (v5:328-329)
int exchange_fwd(E* e, hipStream_t st, int xi, const void* feat, const void* f1, const void* f2) {
    ExgBuf& X = e->ex[xi]; const char* lv = exn(e, xi);
    const int B = e->B, N = e->N, R = e->R, Cp = e->Cp, M = e->M, Mp = e->Mp, dt = e->dt;
    const float s = 1.0f / sqrtf((float)M);
    const int nbr = f2 ? 2 : 1;
    CK(cmpc_rowdot1(dt, feat, X.kq, Mp, X.logits, B, N, Mp, M, s, st));
    CK(cmpc_softmax_n_fwd(X.logits, X.attn, B, N, st));
    CK(cmpc_wcolsum(dt, feat, X.attn, X.pooled, Mp, B, N, Mp, M, 1.0f, st));
    const int ldk = Mp + Cp; const std::string gvk = fmt("gv_%s.t", lv);
    GemmOpt g; g.n_valid = M; g.bias = pptr(e, fmt("gv_lang_%sgv_f1/biases", lv)); if (e->v5) g.act = ACT_TANH;
    CK(gemm_nt(st, DT_F32, {{X.pooled, Mp, opp(e, gvk), ldk, Mp}, {e->nec, Cp, opp(e, gvk, 0, Mp), ldk, Cp}}, X.gvpre, Mp, B, Mp, g));
    CK(cmpc_l2norm_all_fwd(X.gvpre, X.gv, X.rs1, B * Mp, st));
    const void* fx[2] = {f1, f2}; const char* fn[2] = {"f1", "f2"};
    NtJob tj[2];
    for (int i = 0; i < nbr; ++i) {
        GemmOpt a; a.n_valid = M; a.bias = pptr(e, fmt("lang_feat_%s_%s/biases", lv, fn[i])); a.act = ACT_SIGMOID;
        CK(gemm_nt(st, DT_F32, {{X.gv, Mp, opp(e, fmt("lfeat_%s_%s.t", lv, fn[i])), Mp, Mp}}, X.g[i], Mp, B, Mp, a));
        GemmOpt b; b.n_valid = M; b.bias = pptr(e, fmt("trans_feat_%s_%s/biases", lv, fn[i])); b.act = ACT_RELU;
        tj[i] = NtJob{Seg{fx[i], Mp, opp(e, fmt("tfeat_%s_%s.t", lv, fn[i])), Mp, Mp}, X.r[i], Mp, b};
    }
    if (nbr == 2) CK(gemm_nt_pair(st, dt, tj[0], tj[1], R, Mp));         // the two trans_feat 1x1 convs: one launch
    else CK(gemm_nt(st, dt, {tj[0].seg}, tj[0].C, tj[0].ldc, R, Mp, tj[0].o));
    return cmpc_exchange_combine_fwd(dt, feat, X.r[0], nbr == 2 ? X.r[1] : nullptr, X.g[0], nbr == 2 ? X.g[1] : nullptr, Mp, X.out, X.rstd, B, N, Mp, M, st);
}
// outputs: X.dfeat, X.dfs[0] (d f1), X.dfs[1] (d f2), X.dnec
int exchange_bwd(E* e, hipStream_t st, int xi, const void* dout, const void* feat, const void* f1, const void* f2) {
    ExgBuf& X = e->ex[xi]; const char* lv = exn(e, xi);
    const int B = e->B, N = e->N, R = e->R, Cp = e->Cp, M = e->M, Mp = e->Mp, dt = e->dt, Rr = e->RNN;
    const float s = 1.0f / sqrtf((float)M);
    const int nbr = f2 ? 2 : 1;
    TnOpt d; d.defer = true;
    CK(cmpc_exchange_combine_bwd(dt, dout, X.out, X.rstd, X.r[0], nbr == 2 ? X.r[1] : nullptr, X.g[0], nbr == 2 ? X.g[1] : nullptr, Mp, X.dfeat, 0,
                                 X.dp[0], nbr == 2 ? X.dp[1] : nullptr, X.dg[0], nbr == 2 ? X.dg[1] : nullptr,
                                 gptr(e, fmt("trans_feat_%s_f1/biases", lv)), nbr == 2 ? gptr(e, fmt("trans_feat_%s_f2/biases", lv)) : nullptr, B, N, Mp, M, st));
    const void* fx[2] = {f1, f2}; const char* fn[2] = {"f1", "f2"};
    NtJob dj[2];
    for (int i = 0; i < nbr; ++i) {
        CK(gemm_tn(e, st, dt, fx[i], Mp, Mp, X.dp[i], Mp, Mp, gptr(e, fmt("trans_feat_%s_%s/DW", lv, fn[i])), M, R, M, M, OFF0, d));
        GemmOpt o; o.n_valid = M;
        dj[i] = NtJob{Seg{X.dp[i], Mp, opp(e, fmt("tfeat_%s_%s.n", lv, fn[i])), Mp, Mp}, X.dfs[i], Mp, o};
        if (nbr == 2 && i == 1) CK(gemm_nt_pair(st, dt, dj[0], dj[1], R, Mp));       // d f1 and d f2: one launch
        if (nbr == 1) CK(gemm_nt(st, dt, {dj[0].seg}, dj[0].C, dj[0].ldc, R, Mp, dj[0].o));
        CK(colsum(st, DT_F32, X.dg[i], B, Mp, Mp, M, gptr(e, fmt("lang_feat_%s_%s/biases", lv, fn[i])), X.g[i], X.dg[i], ACT_SIGMOID));
        CK(gemm_tn(e, st, DT_F32, X.gv, Mp, Mp, X.dg[i], Mp, Mp, gptr(e, fmt("lang_feat_%s_%s/DW", lv, fn[i])), M, B, M, M, OFF0, d));
        GemmOpt a; a.n_valid = M; a.accumulate = i == 1;
        CK(gemm_nt(st, DT_F32, {{X.dg[i], Mp, opp(e, fmt("lfeat_%s_%s.n", lv, fn[i])), Mp, Mp}}, X.dgv, Mp, B, Mp, a));
    }
    CK(cmpc_l2norm_all_bwd(X.dgv, X.gv, X.rs1, X.dgvpre, B * Mp, st));
    if (!e->v5) CK(colsum(st, DT_F32, X.dgvpre, B, Mp, Mp, M, gptr(e, fmt("gv_lang_%sgv_f1/biases", lv))));
    else CK(colsum(st, DT_F32, X.dgvpre, B, Mp, Mp, M, gptr(e, fmt("gv_lang_%sgv_f1/biases", lv)), X.gvpre, X.dgvpre, ACT_TANH));     // X.gvpre holds tanh(...)
    float* gwg = gptr(e, fmt("gv_lang_%sgv_f1/DW", lv));
    CK(gemm_tn(e, st, DT_F32, X.pooled, Mp, Mp, X.dgvpre, Mp, Mp, gwg, M, B, M, M, OFF0, d));
    CK(gemm_tn(e, st, DT_F32, e->nec, Cp, Cp, X.dgvpre, Mp, Mp, gwg + (size_t)M * M, M, B, Rr, M, OFF0, d));
    const std::string gvn = fmt("gv_%s.n", lv);
    GemmOpt m; m.n_valid = M;
    CK(gemm_nt(st, DT_F32, {{X.dgvpre, Mp, opp(e, gvn), Mp, Mp}}, X.dpooled, Mp, B, Mp, m));
    GemmOpt r; r.n_valid = Rr;
    CK(gemm_nt(st, DT_F32, {{X.dgvpre, Mp, opp(e, gvn, Mp, 0), Mp, Mp}}, X.dnec, Cp, B, Cp, r));
    CK(cmpc_rowdot1(dt, feat, X.dpooled, Mp, X.dattn, B, N, Mp, M, 1.0f, st));
    CK(cmpc_softmax_n_bwd(X.dattn, X.attn, X.dlog, B, N, st));
    CK(cmpc_rank1_update(dt, X.dfeat, X.attn, X.dpooled, X.dlog, X.kq, Mp, 1.0f, s, B, N, Mp, M, st));
    CK(cmpc_wcolsum(dt, feat, X.dlog, X.dkq, Mp, B, N, Mp, M, s, st));
    CK(gemm_nt(st, DT_F32, {{X.dkq, Mp, opp(e, fmt("key_%s.t", lv)), Mp, Mp}}, X.dq, Mp, B, Mp, m));
    CK(gemm_tn(e, st, DT_F32, X.dkq, Mp, Mp, X.q, Mp, Mp, gptr(e, fmt("spa_graph_key_%sgv_f1/DW", lv)), M, B, M, M, OFF0, d));
    CK(colsum(st, DT_F32, X.dq, B, Mp, Mp, M, gptr(e, fmt("lang_query_%sgv_f1/biases", lv))));
    CK(gemm_tn(e, st, DT_F32, e->nec, Cp, Cp, X.dq, Mp, Mp, gptr(e, fmt("lang_query_%sgv_f1/DW", lv)), M, B, Rr, M, OFF0, d));
    GemmOpt n; n.n_valid = Rr; n.accumulate = 1;
    return gemm_nt(st, DT_F32, {{X.dq, Mp, opp(e, fmt("query_%s.n", lv)), Mp, Mp}}, X.dnec, Cp, B, Cp, n);
}

// ------------------------------------------------------------------------------------------
// stage: ConvLSTM over (exg3_2, exg4_2, exg5_2), util/cell.py:36-79 via CMPC_model.py:287-290
// ------------------------------------------------------------------------------------------
const char* LN_NAMES[5] = {"LayerNorm", "LayerNorm_1", "LayerNorm_2", "LayerNorm_3", "LayerNorm_4"};    // j, i, f, o, c
void clstm_ln(E* e, cmpc_convlstm_ln& ln, cmpc_convlstm_dln& dln) {
    for (int i = 0; i < 5; ++i) {
        const std::string p = std::string("rnn/conv_lstm_cell/") + LN_NAMES[i];
        ln.beta[i] = pptr(e, p + "/beta"); ln.gamma[i] = pptr(e, p + "/gamma");
        dln.dbeta[i] = gptr(e, p + "/beta"); dln.dgamma[i] = gptr(e, p + "/gamma");
    }
}
int clstm_fwd(E* e, hipStream_t st, hipEvent_t* x_ready) {
    const int B = e->B, N = e->N, R = e->R, M = e->M, Mp = e->Mp, dt = e->dt;
    cmpc_convlstm_ln ln; cmpc_convlstm_dln dln; clstm_ln(e, ln, dln);
    const std::string pre = "rnn/conv_lstm_cell/";
    const void* xs[3] = {e->ex[e->nex].out, e->ex[e->nex + 1].out, e->nex > 2 ? e->ex[e->nex + 2].out : nullptr};
    const void *hcur = nullptr, *ccur = nullptr;
    for (int s = 0; s < e->ncl; ++s) {
        ClstmStep& S = e->cl[s];
        if (x_ready && x_ready[s]) HCK(hipStreamWaitEvent(st, x_ready[s], 0));
        if (s == 0) CK(gemm_nt(st, dt, {{xs[s], Mp, opp(e, "clstm.t"), 2 * Mp, Mp}}, S.Yg, 4 * Mp, R, 4 * Mp));
        else CK(gemm_nt(st, dt, {{xs[s], Mp, opp(e, "clstm.t"), 2 * Mp, Mp}, {hcur, Mp, opp(e, "clstm.t", 0, Mp), 2 * Mp, Mp}}, S.Yg, 4 * Mp, R, 4 * Mp));
        CK(cmpc_convlstm_a(dt, S.Yg, ccur, pptr(e, pre + "W_ci"), pptr(e, pre + "W_cf"), S.sums, B, N, Mp, M, st));
        CK(cmpc_convlstm_b(dt, S.Yg, ccur, pptr(e, pre + "W_co"), &ln, S.sums, S.c_pre, B, N, Mp, M, st));
        CK(cmpc_convlstm_c(dt, S.Yg, S.c_pre, &ln, S.sums, S.c_new, S.h_new, B, N, Mp, M, st));
        hcur = S.h_new; ccur = S.c_new;
    }
    return CMPC_OK;
}
// in: e->dfused (gradient of the last h); out: cl[s].dx = gradient of exg_*_2
int clstm_bwd(E* e, hipStream_t st, hipEvent_t* dx_ready) {
    const int B = e->B, N = e->N, R = e->R, M = e->M, Mp = e->Mp, dt = e->dt;
    cmpc_convlstm_ln ln; cmpc_convlstm_dln dln; clstm_ln(e, ln, dln);
    const std::string pre = "rnn/conv_lstm_cell/";
    const void* xs[3] = {e->ex[e->nex].out, e->ex[e->nex + 1].out, e->nex > 2 ? e->ex[e->nex + 2].out : nullptr};
    float* gk = gptr(e, pre + "kernel");
    TnOpt d; d.defer = true;
    const void* dh = e->dfused; const void* dc = nullptr;
    for (int s = e->ncl - 1; s >= 0; --s) {
        ClstmStep& S = e->cl[s];
        const void* h_prev = s > 0 ? e->cl[s - 1].h_new : nullptr;
        const void* c_prev = s > 0 ? e->cl[s - 1].c_new : nullptr;
        CK(cmpc_convlstm_bwd(dt, dh, dc, S.Yg, c_prev, S.c_pre, pptr(e, pre + "W_ci"), pptr(e, pre + "W_cf"), pptr(e, pre + "W_co"), &ln, S.sums,
                             S.dYg, s > 0 ? S.dc_prev : nullptr, gptr(e, pre + "W_ci"), gptr(e, pre + "W_cf"), gptr(e, pre + "W_co"), &dln,
                             e->cl_scr, e->cl_bs, B, N, Mp, M, st));
        Offs ox, oh;
        for (int g = 0; g < 4; ++g) { ox.push_back({0, (int64_t)g * Mp, (int64_t)g * M}); oh.push_back({0, (int64_t)g * Mp, (int64_t)M * 4 * M + (int64_t)g * M}); }
        CK(gemm_tn(e, st, dt, xs[s], Mp, Mp, S.dYg, 4 * Mp, Mp, gk, 4 * M, R, M, M, ox, d));
        GemmOpt o; o.n_valid = M;
        // the input gradient of step s feeds the round-2 exchange module s, not the recurrence: with lanes it is computed on that
        // module's lane, so that the serial chain on `st` is gate backward -> dh product -> next step only
        hipStream_t xst = st;
        if (dx_ready && dx_ready[s] && e->cfg.n_lanes > 1) {
            hipEvent_t ev = next_event(e);
            HCK(hipEventRecord(ev, st));
            xst = e->lane[s];
            HCK(hipStreamWaitEvent(xst, ev, 0));
        }
        if (s > 0) {
            CK(gemm_tn(e, st, dt, h_prev, Mp, Mp, S.dYg, 4 * Mp, Mp, gk, 4 * M, R, M, M, oh, d));
            CK(gemm_nt(st, dt, {{S.dYg, 4 * Mp, opp(e, "clstm.n", Mp, 0), 4 * Mp, 4 * Mp}}, S.dh, Mp, R, Mp, o));
            dh = S.dh; dc = S.dc_prev;
        }
        CK(gemm_nt(xst, dt, {{S.dYg, 4 * Mp, opp(e, "clstm.n"), 4 * Mp, 4 * Mp}}, S.dx, Mp, R, Mp, o));
        if (dx_ready && dx_ready[s]) HCK(hipEventRecord(dx_ready[s], xst));
    }
    return CMPC_OK;
}

// ------------------------------------------------------------------------------------------
// stage: atrous_spatial_pyramid_pooling + decoder, CMPCv5_BiLSTM_model.py:190-251 (slim conv2d = convolution without bias +
// batch_norm(eps 1e-5) + relu under resnet_v2.resnet_arg_scope)
// ------------------------------------------------------------------------------------------
const float BN_EPS = 1e-5f;
int bn_fwd(E* e, hipStream_t st, BnLayer& L, void* y, int ldy, int Cy) {
    const std::string p = L.scope + "/BatchNorm";
    if (e->cfg.bn_train) CK(cmpc_bn_stats(L.dt, L.pre, L.ldpre, L.R, L.C, L.Cpad, BN_EPS, L.sums, L.mr, st));
    else CK(cmpc_bn_from_moving(L.mm, L.mv, L.C, L.Cpad, BN_EPS, L.mr, st));
    return cmpc_bn_apply_fwd(L.dt, L.pre, L.ldpre, L.mr, L.Cpad, pptr(e, p + "/gamma"), pptr(e, p + "/beta"), y, ldy, Cy, L.R, L.C, 1, st);
}
// dy / y: gradient wrt and value of the layer's (relu) output, possibly column blocks of wider maps; result in L.dpre
int bn_bwd(E* e, hipStream_t st, BnLayer& L, const void* dy, int lddy, const void* y, int ldy) {
    const std::string p = L.scope + "/BatchNorm";
    return cmpc_bn_bwd(L.dt, dy, lddy, y, ldy, L.pre, L.ldpre, L.mr, L.Cpad, pptr(e, p + "/gamma"), L.dpre, L.ldpre, gptr(e, p + "/gamma"), gptr(e, p + "/beta"),
                       L.means, L.R, L.C, 1, st);
}
// 3x3 'SAME' convolution with rate `dil` on [B, H, W] maps through the implicit-GEMM kernel; res: optional map added to the result (in place allowed)
int conv3x3(E* e, hipStream_t st, const void* X, int ldx, const void* Wt, int ldw, void* Y, int ldy, const void* res, int H, int W, int Cin, int Cout, int dil) {
    cmpc_conv_args a; memset(&a, 0, sizeof(a));
    a.dtype = e->dt; a.X = X; a.ldx = ldx; a.Wt = Wt; a.ldw = ldw; a.bias = e->zbias; a.res = res; a.Y = Y; a.ldy = ldy;
    a.B = e->B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ksize = 3; a.stride = 1; a.dil = dil; a.relu = 0; a.zeros = e->zero_page;
    return cmpc_conv_nhwc(&a, st);
}
// weight gradient of a 3x3 convolution: nine products out[t] += X(shifted by tap t)^T . dY, each a deferred gemm_tn with conv addressing.
// X [rows, ldx] with the input channels in `nseg` column segments (a_col, out_row, len); out = HWIO gradient [9][cin][cout]
int conv3x3_wgrad(E* e, hipStream_t st, const void* X, int ldx, int Ka, const void* dY, int lddy, int Nd, float* gw, int cin, int cout, int H, int W, int dil,
                  std::initializer_list<std::array<int, 3>> segs) {
    const int es = e->esz;
    for (int t = 0; t < 9; ++t) {
        for (const auto& sg : segs) {
            cmpc_gemm_tn_args a; memset(&a, 0, sizeof(a));
            a.dtype = e->dt; a.A = (const char*)X + (size_t)sg[0] * es; a.lda = ldx; a.Ka = Ka - sg[0];
            a.D = dY; a.ldd = lddy; a.Nd = Nd;
            a.out = gw + ((size_t)t * cin + sg[1]) * cout; a.ldo = cout;
            a.R = e->B * H * W; a.Kv = sg[2]; a.Nv = cout; a.nb = 1; a.nb2 = 1; a.rsplit = 1; a.alpha = 1.f; a.zeros = e->zero_page;
            a.conv_H = H; a.conv_W = W; a.conv_dy = (t / 3 - 1) * dil; a.conv_dx = (t % 3 - 1) * dil;
            e->deferred[bucket_of(e, a.out)].push_back(a);
        }
    }
    (void)st;
    return CMPC_OK;
}

// in: fused = last ConvLSTM output; out: e->score (= pred on the decoder's map), e->up, e->sigm, loss / IoU counters
int aspp_decoder_fwd(E* e, hipStream_t main, const float* target) {
    const int B = e->B, N = e->N, R = e->R, M = e->M, Mp = e->Mp, D = e->D, Dp = e->Dp, CATp = e->CATp, R2 = e->R2, dt = e->dt, es = e->esz;
    const void* X = e->cl[e->ncl - 1].h_new;
    hipStream_t st[3];
    CK(fork_lanes(e, main, st));
    // (a) 1x1 branch and the image-level features on main, the three atrous branches on the lanes (v5:234-246)
    for (int k = 1; k <= 3; ++k) {
        BnLayer& L = e->bn[BN_A0 + k];
        CK(conv3x3(e, st[k - 1], X, Mp, opp(e, fmt("aspp%d.t", k)), 9 * Mp, L.pre, Dp, nullptr, e->h, e->w, Mp, Dp, e->cfg.aspp_rates[k - 1]));
        CK(bn_fwd(e, st[k - 1], L, (char*)e->cat4 + (size_t)k * Dp * es, 4 * Dp, Dp));
    }
    { GemmOpt o; o.n_valid = D;
      CK(gemm_nt(main, dt, {{X, Mp, opp(e, "aspp0.t"), Mp, Mp}}, e->bn[BN_A0].pre, Dp, R, Dp, o));
      CK(bn_fwd(e, main, e->bn[BN_A0], e->cat4, 4 * Dp, Dp));
      CK(cmpc_wcolsum(dt, X, e->ones_n, e->pooled, Mp, B, N, Mp, M, 1.0f, main));                       // global average pooling (v5:242)
      CK(gemm_nt(main, DT_F32, {{e->pooled, Mp, opp(e, "aspp_img.t"), Mp, Mp}}, e->bn[BN_IMG].pre, Dp, B, Dp, o));
      CK(bn_fwd(e, main, e->bn[BN_IMG], e->img, Dp, Dp));
      // its bilinear upsampling from 1x1 is a constant map (v5:246): its share of conv_1x1_concat is a per-sample bias
      CK(gemm_nt(main, DT_F32, {{e->img, Dp, opp(e, "aspp_catl.t"), Dp, Dp}}, e->catsb, Dp, B, Dp, o)); }
    CK(join_lanes(e, main));
    { GemmOpt o; o.n_valid = D; o.sbias = e->catsb; o.ld_sbias = Dp; o.rows_per_sample = N;
      CK(gemm_nt(main, dt, {{e->cat4, 4 * Dp, opp(e, "aspp_cat.t"), 4 * Dp, 4 * Dp}}, e->bn[BN_CAT].pre, Dp, R, Dp, o));
      CK(bn_fwd(e, main, e->bn[BN_CAT], e->enc, Dp, Dp)); }
    CK(mark(e, "fwd:aspp_done", main));
    // decoder (v5:190-206): [upsampled encoder output | low-level features] -> 3x3 -> 3x3 -> 1x1 to one channel
    CK(cmpc_resize_bilinear_fwd(dt, e->enc, Dp, e->deccat, CATp, B, e->h, e->w, e->h2, e->w2, Dp, main));
    BnLayer &L1 = e->bn[BN_D1], &L2 = e->bn[BN_D2];
    CK(conv3x3(e, main, e->deccat, CATp, opp(e, "dec1.t"), 9 * CATp, L1.pre, Dp, nullptr, e->h2, e->w2, CATp, Dp, 1));
    CK(bn_fwd(e, main, L1, e->net1, Dp, Dp));
    CK(conv3x3(e, main, e->net1, Dp, opp(e, "dec2.t"), 9 * Dp, L2.pre, Dp, nullptr, e->h2, e->w2, Dp, Dp, 1));
    CK(bn_fwd(e, main, L2, e->net2, Dp, Dp));
    CK(cmpc_conv_to1_fwd(dt, e->net2, Dp, pptr(e, "decoder/upsampling_logits/conv_1x1/weights"), pptr(e, "decoder/upsampling_logits/conv_1x1/biases"),
                         e->score, R2, D, main));
    return cmpc_upsample_fwd(e->score, e->up, e->sigm, target, e->loss, e->iu, e->iu + B, B, e->h2, e->w2, e->H, e->W, main);
}
// the low-level branch of the decoder depends on the backbone only (v5:195-197): issued as soon as the taps are complete
int decoder_low_fwd(E* e, hipStream_t st) {
    BnLayer& L = e->bn[BN_LOW];
    GemmOpt o; o.n_valid = e->LOW;
    CK(gemm_nt(st, e->dt, {{e->c2_feed, e->C2, opp(e, "dec_low.t"), pad64(e->C2), e->C2}}, L.pre, 64, e->R2, 64, o));
    return bn_fwd(e, st, L, (char*)e->deccat + (size_t)e->Dp * e->esz, e->CATp, 64);
}
// out: e->dfused (gradient of the last ConvLSTM output) and every ASPP / decoder parameter gradient
int aspp_decoder_bwd(E* e, hipStream_t main, const float* target) {
    const int B = e->B, N = e->N, R = e->R, M = e->M, Mp = e->Mp, D = e->D, Dp = e->Dp, LOW = e->LOW, CATp = e->CATp, R2 = e->R2, dt = e->dt, es = e->esz;
    const void* X = e->cl[e->ncl - 1].h_new;
    TnOpt d; d.defer = true;
    auto gw = [&](int i) { return gptr(e, std::string(BN_SCOPES[i]) + "/weights"); };
    BnLayer &L1 = e->bn[BN_D1], &L2 = e->bn[BN_D2], &LL = e->bn[BN_LOW], &LC = e->bn[BN_CAT], &LI = e->bn[BN_IMG];
    CK(cmpc_upsample_loss_bwd(e->up, target, e->dscore, e->cfg.loss_w[0] * e->cfg.loss_scale / B, B, e->h2, e->w2, e->H, e->W, main));
    CK(cmpc_conv_to1_bwd(dt, e->dscore, e->net2, Dp, pptr(e, "decoder/upsampling_logits/conv_1x1/weights"), e->dnet2,
                         gptr(e, "decoder/upsampling_logits/conv_1x1/weights"), gptr(e, "decoder/upsampling_logits/conv_1x1/biases"), R2, D, main));
    CK(bn_bwd(e, main, L2, e->dnet2, Dp, e->net2, Dp));
    CK(conv3x3_wgrad(e, main, e->net1, Dp, Dp, L2.dpre, Dp, Dp, gw(BN_D2), D, D, e->h2, e->w2, 1, {{0, 0, D}}));
    CK(conv3x3(e, main, L2.dpre, Dp, opp(e, "dec2.n"), 9 * Dp, e->dnet1, Dp, nullptr, e->h2, e->w2, Dp, Dp, 1));
    CK(bn_bwd(e, main, L1, e->dnet1, Dp, e->net1, Dp));
    if (D == Dp) CK(conv3x3_wgrad(e, main, e->deccat, CATp, CATp, L1.dpre, Dp, Dp, gw(BN_D1), D + LOW, D, e->h2, e->w2, 1, {{0, 0, D + LOW}}));
    else CK(conv3x3_wgrad(e, main, e->deccat, CATp, CATp, L1.dpre, Dp, Dp, gw(BN_D1), D + LOW, D, e->h2, e->w2, 1, {{0, 0, D}, {Dp, D, LOW}}));
    CK(conv3x3(e, main, L1.dpre, Dp, opp(e, "dec1.n"), 9 * Dp, e->ddeccat, CATp, nullptr, e->h2, e->w2, Dp, CATp, 1));
    // low-level branch: only its weights (res2b_relu belongs to the frozen backbone)
    CK(bn_bwd(e, main, LL, (const char*)e->ddeccat + (size_t)Dp * es, CATp, (const char*)e->deccat + (size_t)Dp * es, CATp));
    CK(gemm_tn(e, main, dt, e->c2_feed, e->C2, e->C2, LL.dpre, 64, 64, gw(BN_LOW), LOW, R2, e->C2, LOW, OFF0, d));
    // encoder output
    CK(cmpc_resize_bilinear_bwd(dt, e->ddeccat, CATp, e->denc, Dp, B, e->h, e->w, e->h2, e->w2, Dp, main));
    CK(mark(e, "bwd:decoder_done", main));
    CK(bn_bwd(e, main, LC, e->denc, Dp, e->enc, Dp));
    CK(colsum(main, dt, LC.dpre, R, Dp, Dp, D, nullptr, nullptr, nullptr, ACT_NONE, e->dcatsb, Dp, N));           // gradient of the per-sample bias
    { Offs oc; for (int k = 0; k < 4; ++k) oc.push_back({(int64_t)k * Dp, 0, (int64_t)k * D * D});
      CK(gemm_tn(e, main, dt, e->cat4, 4 * Dp, Dp, LC.dpre, Dp, Dp, gw(BN_CAT), D, R, D, D, oc, d)); }
    CK(gemm_tn(e, main, DT_F32, e->img, Dp, Dp, e->dcatsb, Dp, Dp, gw(BN_CAT) + (size_t)4 * D * D, D, B, D, D, OFF0, d));
    { GemmOpt o; o.n_valid = D;
      CK(gemm_nt(main, DT_F32, {{e->dcatsb, Dp, opp(e, "aspp_catl.n"), Dp, Dp}}, e->dimg, Dp, B, Dp, o)); }
    CK(bn_bwd(e, main, LI, e->dimg, Dp, e->img, Dp));
    CK(gemm_tn(e, main, DT_F32, e->pooled, Mp, Mp, LI.dpre, Dp, Dp, gw(BN_IMG), D, B, M, D, OFF0, d));
    { GemmOpt o; o.n_valid = M;
      CK(gemm_nt(main, DT_F32, {{LI.dpre, Dp, opp(e, "aspp_img.n"), Dp, Dp}}, e->dpooled, Mp, B, Mp, o)); }
    CK(gemm_nt(main, dt, {{LC.dpre, Dp, opp(e, "aspp_cat.n"), Dp, Dp}}, e->dcat4, 4 * Dp, R, 4 * Dp));
    // branches: gradients of the four maps that read `fused` are summed into e->dfused (1x1 first, each atrous branch adds through the
    // convolution's residual input, the pooled path through a rank-1 update): a fixed order on one stream
    BnLayer& L0 = e->bn[BN_A0];
    CK(bn_bwd(e, main, L0, e->dcat4, 4 * Dp, e->cat4, 4 * Dp));
    CK(gemm_tn(e, main, dt, X, Mp, Mp, L0.dpre, Dp, Dp, gw(BN_A0), D, R, M, D, OFF0, d));
    { GemmOpt o; o.n_valid = M;
      CK(gemm_nt(main, dt, {{L0.dpre, Dp, opp(e, "aspp0.n"), Dp, Dp}}, e->dfused, Mp, R, Mp, o)); }
    for (int k = 1; k <= 3; ++k) {
        BnLayer& L = e->bn[BN_A0 + k];
        const int rate = e->cfg.aspp_rates[k - 1];
        CK(bn_bwd(e, main, L, (const char*)e->dcat4 + (size_t)k * Dp * es, 4 * Dp, (const char*)e->cat4 + (size_t)k * Dp * es, 4 * Dp));
        CK(conv3x3_wgrad(e, main, X, Mp, Mp, L.dpre, Dp, Dp, gw(BN_A0 + k), M, D, e->h, e->w, rate, {{0, 0, M}}));
        CK(conv3x3(e, main, L.dpre, Dp, opp(e, fmt("aspp%d.n", k)), 9 * Dp, e->dfused, Mp, e->dfused, e->h, e->w, Dp, Mp, rate));
    }
    return cmpc_rank1_update(dt, e->dfused, e->ones_n, e->dpooled, nullptr, nullptr, Mp, 1.0f, 0.0f, B, N, Mp, M, main);
}

// bucket b is complete: its deferred weight-gradient products in one grouped launch, its deferred bias / LayerNorm / peephole folds in
// another, then the event a data-parallel caller waits on before all-reducing the bucket
int flush_bucket(E* e, hipStream_t st, int b) {
    std::vector<cmpc_gemm_tn_args>& d = e->deferred[b];
    if (!d.empty()) {
        const int rc = cmpc_gemm_tn_grouped_cached(d.data(), (int)d.size(), e->tn_table[b], 4, &e->tn_victim[b], e->tn_table_bytes, e->tn_shadow[b], st);
        d.clear();
        CK(rc);
    }
    const float* lo[4]; const float* hi[4];
    int nr = 0;
    for (const E::Range& r : e->bucket[b]) { lo[nr] = e->grads + r.off; hi[nr] = e->grads + r.off + r.count; ++nr; }
    CK(cmpc_fold_flush_ranges(&e->fold, lo, hi, nr, e->fold_table[b], e->fold_shadow[b].data(), &e->fold_shadow_n[b], st));
    HCK(hipEventRecord(e->bucket_ev[b], st));
    return CMPC_OK;
}

int params_ready(E* e, hipStream_t st, int stage) {
    // the optimizer of the previous step may run on its own stream: stage 0 = Adam done + text operands repacked,
    // stage 1 = everything repacked
    if (!e->opt_pending) return CMPC_OK;
    HCK(hipStreamWaitEvent(st, stage == 0 ? e->ev_opt0 : e->ev_opt1, 0));
    if (stage == 1) e->opt_pending = false;
    return CMPC_OK;
}

int set_device(const E* e) {
    if (e->cfg.device < 0) { cmpc_set_error("this handle was created with device = -1 (planning only)"); return CMPC_EINVAL; }
    int cur = -1;
    HCK(hipGetDevice(&cur));
    if (cur != e->cfg.device) HCK(hipSetDevice(e->cfg.device));
    return CMPC_OK;
}

}  // namespace

// ==========================================================================================
extern "C" int cmpc_default_cfg(cmpc_cfg* c) {
    if (!c) { cmpc_set_error("default_cfg: null"); return CMPC_EINVAL; }
    memset(c, 0, sizeof(*c));
    c->batch_size = 1; c->num_steps = 20; c->vf_h = 40; c->vf_w = 40; c->H = 320; c->W = 320;
    c->vf_dim = 2048; c->c4_dim = 1024; c->c3_dim = 512;
    c->vocab_size = 12112; c->v_emb_dim = 1000; c->mlp_dim = 500; c->rnn_size = 1000; c->glove_dim = 300; c->parse_dim = 500;
    c->start_lr = 0.00025; c->end_lr = 0.00001; c->lr_power = 0.9; c->lr_decay_step = 800000; c->weight_decay = 0.0005f;
    c->loss_w[0] = 0.7f; c->loss_w[1] = c->loss_w[2] = c->loss_w[3] = 0.1f;
    c->dtype = DT_F16; c->n_lanes = 3; c->device = 0; c->loss_scale = 0.f;
    c->model = CMPC_MODEL_CMPC; c->hsv = 0; c->bn_train = 0; c->bn_decay = 0.9997f;
    c->c2_dim = 256; c->c2_h = c->H / 4; c->c2_w = c->W / 4; c->aspp_depth = 256; c->low_dim = 48;
    c->aspp_rates[0] = 6; c->aspp_rates[1] = 12; c->aspp_rates[2] = 18; c->sample_frames = 5;
    return CMPC_OK;
}
// the caller's graph-shaping fields (batch_size, num_steps, vf_h, vf_w, H, W, ...) are kept; only the model-specific ones are set
extern "C" int cmpc_default_cfg_model(cmpc_cfg* c, int model, int hsv) {
    if (!c || (model != CMPC_MODEL_CMPC && model != CMPC_MODEL_V5_BILSTM && model != CMPC_MODEL_VIDEO)) { cmpc_set_error("default_cfg_model: bad argument"); return CMPC_EINVAL; }
    c->sample_frames = 5;                             // CMPC_video_mm_tgraph_allvec.py:69
    c->conv5 = 0; c->freeze_bn = 0;
    c->model = model; c->hsv = (model == CMPC_MODEL_V5_BILSTM && hsv) ? 1 : 0;
    c->bn_decay = 0.9997f; c->c2_dim = 256; c->c2_h = c->H / 4; c->c2_w = c->W / 4; c->aspp_depth = 256; c->low_dim = 48;
    c->aspp_rates[0] = 6; c->aspp_rates[1] = 12; c->aspp_rates[2] = 18;
    if (model == CMPC_MODEL_V5_BILSTM) { c->loss_w[0] = 0.8f; c->loss_w[1] = 0.1f; c->loss_w[2] = 0.1f; c->loss_w[3] = 0.f; c->bn_train = 1; }     // v5:541-542
    else { c->loss_w[0] = 0.7f; c->loss_w[1] = c->loss_w[2] = c->loss_w[3] = 0.1f; c->bn_train = 0; }
    return CMPC_OK;
}

extern "C" int cmpc_destroy(cmpc_handle e) {
    if (!e) return CMPC_OK;
    if (e->cfg.device < 0) { delete e; return CMPC_OK; }
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    for (void* p : {(void*)e->params, (void*)e->grads, (void*)e->adam_m, (void*)e->adam_v, (void*)e->arena, (void*)e->descs_dev,
                    (void*)e->tile_prefix_dev, (void*)e->tile_desc_dev, (void*)e->segs_dev, (void*)e->ws, (void*)e->bn_state})
        if (p) (void)hipFree(p);
    for (hipStream_t s : e->own_lane) if (s) { cmpc_ws_release(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t ev : e->evpool) (void)hipEventDestroy(ev);
    if (e->ev_opt0) (void)hipEventDestroy(e->ev_opt0);
    if (e->ev_opt1) (void)hipEventDestroy(e->ev_opt1);
    for (hipEvent_t ev : e->bucket_ev) if (ev) (void)hipEventDestroy(ev);
    delete e;
    return CMPC_OK;
}

extern "C" int cmpc_create(const cmpc_cfg* c, cmpc_handle* out) {
    if (!c || !out) { cmpc_set_error("create: null argument"); return CMPC_EINVAL; }
    *out = nullptr;
    if (c->batch_size < 1 || c->num_steps < 1 || c->num_steps > 64 || c->vf_h < 1 || c->vf_w < 1 || c->H < c->vf_h || c->W < c->vf_w) {
        cmpc_set_error("create: need batch_size >= 1, 1 <= num_steps <= 64, H >= vf_h >= 1, W >= vf_w >= 1"); return CMPC_EINVAL;
    }
    if (c->dtype != DT_F32 && c->dtype != DT_BF16 && c->dtype != DT_F16) { cmpc_set_error("create: dtype must be 0 (f32), 1 (bf16) or 2 (f16)"); return CMPC_EINVAL; }
    if (c->loss_scale < 0.f) { cmpc_set_error("create: loss_scale must be >= 0 (0 = default)"); return CMPC_EINVAL; }
    if (c->rnn_size != c->v_emb_dim) { cmpc_set_error("create: rnn_size must equal v_emb_dim (the affinity contracts them, CMPC_model.py:384)"); return CMPC_EINVAL; }
    if (c->vf_dim % 64 || c->c4_dim % 64 || c->c3_dim % 64) { cmpc_set_error("create: vf_dim / c4_dim / c3_dim must be multiples of 64 (MFMA K tile)"); return CMPC_EINVAL; }
    if (c->v_emb_dim < 8 || c->mlp_dim < 8 || c->glove_dim < 1 || c->parse_dim < 4 || c->vocab_size < 1) {
        cmpc_set_error("create: need v_emb_dim >= 8, mlp_dim >= 8, glove_dim >= 1, parse_dim >= 4, vocab_size >= 1"); return CMPC_EINVAL;
    }
    if (pad64(c->v_emb_dim) > 2048 || pad64(c->mlp_dim) > 2048) { cmpc_set_error("create: v_emb_dim, mlp_dim <= 2048 (per-column registers of the map kernels)"); return CMPC_EINVAL; }
    if (c->n_lanes < 1 || c->n_lanes > 3) { cmpc_set_error("create: n_lanes must be 1, 2 or 3"); return CMPC_EINVAL; }
    if (c->model != CMPC_MODEL_CMPC && c->model != CMPC_MODEL_V5_BILSTM && c->model != CMPC_MODEL_VIDEO) {
        cmpc_set_error("create: model must be CMPC_MODEL_CMPC (0), CMPC_MODEL_V5_BILSTM (1) or CMPC_MODEL_VIDEO (2)"); return CMPC_EINVAL;
    }
    if (c->model == CMPC_MODEL_VIDEO && (c->batch_size != 1 || c->sample_frames < 1 || c->sample_frames > 8)) {
        cmpc_set_error("create (CMPC_video): the graph is only valid for batch_size = 1 (CMPC_video_mm_tgraph_allvec.py:323-324,379); 1 <= sample_frames <= 8"); return CMPC_EINVAL;
    }
    if (c->model == CMPC_MODEL_V5_BILSTM) {
        if (c->c2_dim < 64 || c->c2_dim % 64 || c->c2_h < c->vf_h || c->c2_w < c->vf_w || c->aspp_depth < 8 || c->aspp_depth % 4 || pad64(c->aspp_depth) > 512 ||
            c->low_dim < 4 || c->low_dim % 4 || c->low_dim > 64 || c->aspp_rates[0] < 1 || c->aspp_rates[1] < 1 || c->aspp_rates[2] < 1 || c->mlp_dim % 4 ||
            !(c->bn_decay >= 0.f && c->bn_decay <= 1.f)) {
            cmpc_set_error("create (CMPCv5_BiLSTM): need c2_dim a multiple of 64, c2_h >= vf_h, c2_w >= vf_w, aspp_depth a multiple of 4 with pad64 <= 512, "
                           "low_dim a multiple of 4 and <= 64, rates >= 1, mlp_dim a multiple of 4, 0 <= bn_decay <= 1"); return CMPC_EINVAL;
        }
    }
    const bool plan_only = c->device == -1;      // host-side planning only (manifest, operand plan, workspace size): no GPU needed
    if (!plan_only) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || c->device < 0 || c->device >= ndev) {
            cmpc_set_error("create: no HIP device %d (the CMPC head has no CPU path)", c->device); return CMPC_EHIP;
        }
        HCK(hipSetDevice(c->device));
    }
    E* e = new E();
    e->cfg = *c;
    // Static loss scaling, f16 storage only (range 6e-8 .. 65504).  The reference's loss is a SUM over the H x W pixels
    // (util/loss.py:6-16), so the upstream gradients are large, not small: a low-resolution logit receives up to
    // w * (H/h) * (W/w) / B (45 at B = 1).  The scale is the power of two that puts that maximum at <= 16; every gradient the
    // backward pass writes carries it and the optimizer divides it out again.  bf16 / f32 storage: 1.
    if (e->cfg.loss_scale == 0.f) {
        e->cfg.loss_scale = 1.f;
        if (c->dtype == DT_F16) {
            const double top = (double)c->loss_w[0] * ((double)c->H / c->vf_h) * ((double)c->W / c->vf_w) / c->batch_size;
            e->cfg.loss_scale = (float)exp2(floor(log2(16.0 / top)));
        }
    }
    e->B = c->batch_size; e->T = c->num_steps; e->h = c->vf_h; e->w = c->vf_w; e->N = e->h * e->w; e->R = e->B * e->N;
    e->H = c->H; e->W = c->W; e->V = c->vocab_size;
    e->C = c->v_emb_dim; e->Cp = pad64(e->C); e->M = c->mlp_dim; e->Mp = pad64(e->M); e->G = c->glove_dim; e->Gp = pad64(e->G);
    e->P = c->parse_dim; e->Pp = pad64(e->P); e->Tp = 64; e->RNN = c->rnn_size; e->dt = c->dtype; e->esz = c->dtype == DT_F32 ? 4 : 2;
    e->v5 = c->model == CMPC_MODEL_V5_BILSTM;
    e->vid = c->model == CMPC_MODEL_VIDEO;
    if (e->vid) { e->Fr = c->sample_frames; e->RF = e->Fr * e->N; e->NC = 5; }
    if (e->v5) {
        e->nlev = 2; e->nex = 2; e->ncl = 2; e->ndir = 2;
        e->D = c->aspp_depth; e->Dp = pad64(e->D); e->LOW = c->low_dim; e->CATp = e->Dp + 64; e->C2 = c->c2_dim; e->h2 = c->c2_h; e->w2 = c->c2_w; e->R2 = e->B * e->h2 * e->w2;
        const char* dn[2] = {"fw", "bw"};
        for (int d = 0; d < 2; ++d) {
            e->ldir[d].key = fmt("lstm_%s", dn[d]); e->ldir[d].pk = fmt("bidirectional_rnn/%s/lstm_cell/kernel", dn[d]); e->ldir[d].pb = fmt("bidirectional_rnn/%s/lstm_cell/bias", dn[d]);
        }
    } else if (e->vid) {
        e->ldir[0].key = "lstm"; e->ldir[0].pk = "RNN/multi_rnn_cell/cell_0/basic_lstm_cell/kernel"; e->ldir[0].pb = "RNN/multi_rnn_cell/cell_0/basic_lstm_cell/bias";
    } else { e->ldir[0].key = "lstm"; e->ldir[0].pk = "rnn/lstm_cell/kernel"; e->ldir[0].pb = "rnn/lstm_cell/bias"; }
    // the graph's T-deep products as streaming kernels (cmpc_lowrank_nn); decided before the workspace is planned (Z moves to the forward pass)
    e->lowrank = e->dt != DT_F32 && e->T <= 24 && e->Cp <= 1024 && ((e->Cp / 4) & (e->Cp / 4 - 1)) == 0;
    if (const char* v = getenv("CMPC_LOWRANK")) e->lowrank = e->lowrank && atoi(v) != 0;          // read once, at create
    build_manifest(e);
    plan_operands(e);
    if (plan_only) {
        Bump zf, zb, g;
        plan_workspace(e, zf, zb, g);
        e->zf_bytes = zf.off; e->zb_bytes = zb.off; e->ws_bytes = zf.off + zb.off + g.off;
        *out = e;
        return CMPC_OK;
    }
    auto fail = [&](int rc) { cmpc_destroy(e); return rc; };
#define ECK(x) do { const hipError_t _e = (x); if (_e != hipSuccess) { cmpc_set_error("create: %s: %s", #x, hipGetErrorString(_e)); return fail(CMPC_EHIP); } } while (0)
    const size_t pbytes = (size_t)e->total * sizeof(float);
    ECK(hipMalloc(&e->params, pbytes)); ECK(hipMalloc(&e->grads, pbytes)); ECK(hipMalloc(&e->adam_m, pbytes)); ECK(hipMalloc(&e->adam_v, pbytes));
    ECK(hipMemset(e->params, 0, pbytes)); ECK(hipMemset(e->grads, 0, pbytes)); ECK(hipMemset(e->adam_m, 0, pbytes)); ECK(hipMemset(e->adam_v, 0, pbytes));
    ECK(hipMalloc(&e->arena, e->arena_bytes)); ECK(hipMemset(e->arena, 0, e->arena_bytes));
    if (e->state_total > 0) {       // batch-norm moving statistics: mean 0, variance 1 (slim's initialisers)
        std::vector<float> init((size_t)e->state_total, 0.f);
        for (const auto& sp : e->state_specs)
            if (sp.first.size() > 15 && sp.first.compare(sp.first.size() - 15, 15, "moving_variance") == 0)
                for (int64_t i = 0; i < sp.second; ++i) init[(size_t)e->state_off.at(sp.first) + i] = 1.f;
        ECK(hipMalloc(&e->bn_state, (size_t)e->state_total * sizeof(float)));
        ECK(hipMemcpy(e->bn_state, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (upload_tables(e) != CMPC_OK) return fail(CMPC_EHIP);
    {   // workspace: measure, allocate, assign
        Bump zf, zb, g;
        plan_workspace(e, zf, zb, g);
        e->zf_bytes = zf.off; e->zb_bytes = zb.off; e->ws_bytes = zf.off + zb.off + g.off;
        ECK(hipMalloc(&e->ws, e->ws_bytes)); ECK(hipMemset(e->ws, 0, e->ws_bytes));
        Bump zf2, zb2, g2; zf2.base = e->ws; zb2.base = e->ws + e->zf_bytes; g2.base = e->ws + e->zf_bytes + e->zb_bytes;
        zf2.log = zb2.log = g2.log = &e->guards;
        plan_workspace(e, zf2, zb2, g2);
    }
    {   // spatial grid [B*N, 64]: generate_spatial_batch (util/processing_tools.py:5-17), float64 arithmetic, float32 values
        std::vector<float> sp((size_t)e->N * 64, 0.f);
        for (int y = 0; y < e->h; ++y)
            for (int x = 0; x < e->w; ++x) {
                const double xmin = (double)x / e->w * 2 - 1, xmax = (double)(x + 1) / e->w * 2 - 1;
                const double ymin = (double)y / e->h * 2 - 1, ymax = (double)(y + 1) / e->h * 2 - 1;
                float* r = &sp[(size_t)(y * e->w + x) * 64];
                r[0] = (float)xmin; r[1] = (float)ymin; r[2] = (float)xmax; r[3] = (float)ymax;
                r[4] = (float)((xmin + xmax) / 2); r[5] = (float)((ymin + ymax) / 2); r[6] = (float)(1.0 / e->w); r[7] = (float)(1.0 / e->h);
            }
        float* tmp = nullptr;
        const int reps = e->vid ? e->Fr : e->B;          // tf.tile(spatial, [sample_frames, 1, 1, 1]) for the per-frame Mutan of the video model (vid:333)
        ECK(hipMalloc(&tmp, (size_t)reps * e->N * 64 * sizeof(float)));
        for (int b = 0; b < reps; ++b) ECK(hipMemcpy(tmp + (size_t)b * e->N * 64, sp.data(), sp.size() * sizeof(float), hipMemcpyHostToDevice));
        if (e->dt == DT_F32) ECK(hipMemcpy(e->spatial, tmp, (size_t)reps * e->N * 64 * sizeof(float), hipMemcpyDeviceToDevice));
        else if (cmpc_cast(DT_F32, tmp, e->dt, e->spatial, (int64_t)reps * e->N * 64, nullptr) != CMPC_OK) { (void)hipFree(tmp); return fail(CMPC_EHIP); }
        ECK(hipDeviceSynchronize());
        // the same grid with a constant 1 in channel 8: out += A^T D then carries sum_rows(D) in row 8 (the bias gradient of a convolution
        // whose last input channels are the grid and whose bias follows its DW in the parameter layout)
        for (size_t i = 0; i < (size_t)e->N; ++i) sp[i * 64 + 8] = 1.0f;
        for (int b = 0; b < reps; ++b) ECK(hipMemcpy(tmp + (size_t)b * e->N * 64, sp.data(), sp.size() * sizeof(float), hipMemcpyHostToDevice));
        if (e->dt == DT_F32) ECK(hipMemcpy(e->spatial1, tmp, (size_t)reps * e->N * 64 * sizeof(float), hipMemcpyDeviceToDevice));
        else if (cmpc_cast(DT_F32, tmp, e->dt, e->spatial1, (int64_t)reps * e->N * 64, nullptr) != CMPC_OK) { (void)hipFree(tmp); return fail(CMPC_EHIP); }
        ECK(hipDeviceSynchronize());
        (void)hipFree(tmp);
        e->mutan_bias_row = !getenv("CMPC_NO_BIAS_ROW");
        for (int i = 0; i < e->nlev && e->mutan_bias_row; ++i)
            for (int hd = 1; hd <= 5; ++hd)
                if (poff(e, fmt("vis_trans_%s_head%d/biases", lvn(e, i), hd)) != poff(e, fmt("vis_trans_%s_head%d/DW", lvn(e, i), hd)) + (int64_t)(e->C + 8) * e->C) e->mutan_bias_row = false;
    }
    if (e->vid) {
        std::vector<float> on(64, 1.0f);
        ECK(hipMemcpy(e->ones_t, on.data(), on.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (e->v5) {
        std::vector<float> on((size_t)e->B * e->N, 1.0f / (float)e->N);       // tf.reduce_mean over the map (v5:242)
        ECK(hipMemcpy(e->ones_n, on.data(), on.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    for (int i = 0; i < 3; ++i) { ECK(hipStreamCreateWithFlags(&e->own_lane[i], hipStreamNonBlocking)); e->lane[i] = e->own_lane[i]; }
    if (c->n_lanes == 2) e->lane[2] = e->lane[0];
    e->evpool.resize(256);
    for (auto& ev : e->evpool) { ev = nullptr; ECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); }
    ECK(hipEventCreateWithFlags(&e->ev_opt0, hipEventDisableTiming));
    ECK(hipEventCreateWithFlags(&e->ev_opt1, hipEventDisableTiming));
    for (auto& d : e->deferred) d.reserve(128);
    if (const char* v = getenv("CMPC_WGRAD_OVERLAP")) e->wgrad_overlap = atoi(v) != 0;       // read once, at create
    if (const char* v = getenv("CMPC_MUTAN_EPILOGUE")) e->mutan_epilogue = atoi(v) != 0;
    // Opt-in (CMPC_LSTM_SEQ=1): measured SLOWER inside the multi-stream step (13.26 vs 10.71 ms per B=8 train step): its 256 spinning
    // workgroups cannot co-reside with the MFMA kernels of the other streams (2 waves x 256 VGPRs per SIMD fill the register file), so every
    // grid barrier waits for whole GEMM / persistent dW workgroups to retire.  It pays only where the recurrence has the GPU to itself.
    e->lstm_seq = false;
    if (const char* v = getenv("CMPC_LSTM_SEQ")) e->lstm_seq = atoi(v) != 0 && e->B <= 8 && e->Cp <= 1024;
    e->fold_descs.resize(e->fold.table_cap);
    for (int a = 0; a < E::NBK; ++a) { e->fold_shadow[a].resize(e->fold.table_cap); e->fold_shadow_n[a] = -1; }
    e->fold.descs = e->fold_descs.data(); e->fold.shadow = e->fold_shadow[E::NBK - 1].data();
    {   // bucket ranges from the manifest order (build_manifest)
        auto off = [&](const char* n) { return poff(e, n); };
        if (!e->v5) {
            const int64_t lat0 = off("c5_lateral/DW"), par0 = off("words_parse_1/DW"), l5 = off("vis_trans_c5_head1/DW"), l4 = off("vis_trans_c4_head1/DW"),
                          l3 = off("vis_trans_c3_head1/DW"), ex0 = off("spa_graph_key_c3gv_f1/DW");
            e->bucket[0] = {{ex0, e->total - ex0}};
            e->bucket[1] = {{l5, l4 - l5}};
            e->bucket[2] = {{l4, l3 - l4}};
            e->bucket[3] = {{l3, ex0 - l3}, {lat0, par0 - lat0}};
            e->bucket[4] = {{0, lat0}, {par0, l5 - par0}};
        } else {
            // CMPCv5_BiLSTM: 0 = exchange modules + ConvLSTM + ASPP + decoder, 1 = level c5, 2 = level c4 + score_c5 / score_c4 + laterals,
            // 3 = (empty), 4 = text encoder (both LSTMs, words_feat) + parser
            const int64_t lat0 = off("c5_lateral/DW"), par0 = off("words_parse_1/DW"), l5 = off("vis_trans_c5_head1/DW"), l4 = off("vis_trans_c4_head1/DW"),
                          ex0 = off("spa_graph_key_c4gv_f1/DW");
            e->bucket[0] = {{ex0, e->total - ex0}};
            e->bucket[1] = {{l5, l4 - l5}};
            e->bucket[2] = {{l4, ex0 - l4}, {lat0, par0 - lat0}};
            e->bucket[3] = {};
            e->bucket[4] = {{0, lat0}, {par0, l5 - par0}};
        }
        for (int b = 0; b < E::NBK; ++b) ECK(hipEventCreateWithFlags(&e->bucket_ev[b], hipEventDisableTiming));
        // the optimizer's work per bucket: runs of Adam segments and of pack tiles (both tables are in parameter / plan order)
        auto bucket_at = [&](int64_t o) { return bucket_of(e, e->grads + o); };
        for (int i = 0; i < e->nseg; ++i) {
            auto& v = e->bucket_segs[bucket_at(e->segs_host[i].off)];
            if (!v.empty() && v.back().second == i) v.back().second = i + 1; else v.push_back({i, i + 1});
        }
        for (int i = 0; i < e->ndesc; ++i) {
            auto& v = e->bucket_tiles[bucket_at(e->descs[i].src_off)];
            const int t0 = e->tile_prefix_host[i], t1 = e->tile_prefix_host[i + 1];
            if (!v.empty() && v.back().second == t0) v.back().second = t1; else v.push_back({t0, t1});
        }
    }
    e->fold.lo = e->grads; e->fold.hi = e->grads + e->total;
#undef ECK
    *out = e;
    return CMPC_OK;
}

extern "C" int cmpc_param_count(cmpc_handle e) { return e ? (int)e->specs.size() : 0; }
extern "C" int cmpc_param_info(cmpc_handle e, int i, const char** name, int64_t* offset, int* rank, int64_t shape[4]) {
    if (!e || i < 0 || i >= (int)e->specs.size()) { cmpc_set_error("param_info: bad index"); return CMPC_EINVAL; }
    const ParamSpec& s = e->specs[i];
    if (name) *name = s.name.c_str();
    if (offset) *offset = s.off;
    if (rank) *rank = s.rank;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = s.shape[k];
    return CMPC_OK;
}
extern "C" int cmpc_buffers(cmpc_handle e, float** params, float** grads, float** m, float** v, int64_t* total) {
    if (!e) { cmpc_set_error("buffers: null handle"); return CMPC_EINVAL; }
    if (params) *params = e->params;
    if (grads) *grads = e->grads;
    if (m) *m = e->adam_m;
    if (v) *v = e->adam_v;
    if (total) *total = e->total;
    return CMPC_OK;
}
static int find_param(cmpc_handle e, const char* name, int64_t count, const ParamSpec** out) {
    if (!e || !name) { cmpc_set_error("weights: null argument"); return CMPC_EINVAL; }
    auto it = e->pindex.find(name);
    if (it == e->pindex.end()) { cmpc_set_error("weights: no variable named %s", name); return CMPC_EINVAL; }
    const ParamSpec& s = e->specs[it->second];
    if (s.count != count) { cmpc_set_error("weights: %s has %lld elements, got %lld", name, (long long)s.count, (long long)count); return CMPC_EINVAL; }
    *out = &s;
    return CMPC_OK;
}
extern "C" int cmpc_set_weights(cmpc_handle e, const char* name, const float* src, int64_t count) {
    const ParamSpec* s = nullptr;
    CK(find_param(e, name, count, &s));
    CK(set_device(e));
    HCK(hipMemcpy(e->params + s->off, src, (size_t)count * sizeof(float), hipMemcpyHostToDevice));
    return CMPC_OK;
}
extern "C" int cmpc_get_weights(cmpc_handle e, const char* name, float* dst, int64_t count) {
    const ParamSpec* s = nullptr;
    CK(find_param(e, name, count, &s));
    CK(set_device(e));
    HCK(hipDeviceSynchronize());
    HCK(hipMemcpy(dst, e->params + s->off, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return CMPC_OK;
}
extern "C" int cmpc_pack(cmpc_handle e, void* stream) {
    if (!e) { cmpc_set_error("pack: null handle"); return CMPC_EINVAL; }
    CK(set_device(e));
    CK(params_ready(e, (hipStream_t)stream, 1));
    return cmpc_pack_weights_range(e->params, e->arena, e->descs_dev, e->tile_prefix_dev, e->tile_desc_dev, e->ndesc, 0, e->total_tiles, stream);
}
extern "C" int cmpc_state_count(cmpc_handle e) { return e ? (int)e->state_specs.size() : 0; }
extern "C" int cmpc_state_info(cmpc_handle e, int i, const char** name, int64_t* count) {
    if (!e || i < 0 || i >= (int)e->state_specs.size()) { cmpc_set_error("state_info: bad index"); return CMPC_EINVAL; }
    if (name) *name = e->state_specs[i].first.c_str();
    if (count) *count = e->state_specs[i].second;
    return CMPC_OK;
}
static int find_state(cmpc_handle e, const char* name, int64_t count, int64_t* off) {
    if (!e || !name) { cmpc_set_error("state: null argument"); return CMPC_EINVAL; }
    auto it = e->state_off.find(name);
    if (it == e->state_off.end()) { cmpc_set_error("state: no variable named %s", name); return CMPC_EINVAL; }
    for (const auto& sp : e->state_specs) if (sp.first == name && sp.second != count) { cmpc_set_error("state: %s has %lld elements, got %lld", name, (long long)sp.second, (long long)count); return CMPC_EINVAL; }
    *off = it->second;
    return CMPC_OK;
}
extern "C" int cmpc_get_state(cmpc_handle e, const char* name, float* dst, int64_t count) {
    int64_t off = 0;
    CK(find_state(e, name, count, &off));
    CK(set_device(e));
    HCK(hipDeviceSynchronize());
    HCK(hipMemcpy(dst, e->bn_state + off, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return CMPC_OK;
}
extern "C" int cmpc_set_state(cmpc_handle e, const char* name, const float* src, int64_t count) {
    int64_t off = 0;
    CK(find_state(e, name, count, &off));
    CK(set_device(e));
    HCK(hipDeviceSynchronize());
    HCK(hipMemcpy(e->bn_state + off, src, (size_t)count * sizeof(float), hipMemcpyHostToDevice));
    return CMPC_OK;
}
extern "C" int cmpc_get_step(cmpc_handle e, int64_t* step) { if (!e || !step) return CMPC_EINVAL; *step = e->step; return CMPC_OK; }
extern "C" int cmpc_set_step(cmpc_handle e, int64_t step) { if (!e || step < 0) return CMPC_EINVAL; e->step = step; return CMPC_OK; }

extern "C" int cmpc_tap_count(cmpc_handle e) { return e ? (int)e->taps.size() : 0; }
extern "C" int cmpc_tap_name(cmpc_handle e, int i, const char** name) {
    if (!e || i < 0 || i >= (int)e->taps.size() || !name) { cmpc_set_error("tap_name: bad index"); return CMPC_EINVAL; }
    *name = e->taps[i].name.c_str();
    return CMPC_OK;
}
extern "C" int cmpc_tap(cmpc_handle e, const char* name, void** ptr, int* dtype, int* rank, int64_t shape[4]) {
    if (!e || !name) { cmpc_set_error("tap: null argument"); return CMPC_EINVAL; }
    auto it = e->tapindex.find(name);
    if (it == e->tapindex.end()) { cmpc_set_error("tap: no intermediate named %s", name); return CMPC_EINVAL; }
    const Tap& t = e->taps[it->second];
    if (ptr) *ptr = t.ptr;
    if (dtype) *dtype = t.dt;
    if (rank) *rank = t.rank;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = t.shape[k];
    return CMPC_OK;
}
extern "C" int cmpc_plan_info(cmpc_handle e, const cmpc_pack_desc** descs, int* ndesc, int64_t* arena_bytes, int64_t* workspace_bytes, int* stage0_ndesc) {
    if (!e) { cmpc_set_error("plan_info: null handle"); return CMPC_EINVAL; }
    if (descs) *descs = e->descs.data();
    if (ndesc) *ndesc = (int)e->descs.size();
    if (arena_bytes) *arena_bytes = (int64_t)e->arena_bytes;
    if (workspace_bytes) *workspace_bytes = (int64_t)e->ws_bytes;
    if (stage0_ndesc) *stage0_ndesc = e->stage0_ndesc;
    return CMPC_OK;
}
extern "C" int cmpc_operand_info(cmpc_handle e, const char* key, int64_t* byte_off, int* dt, int* rows, int* ld) {
    if (!e || !key) { cmpc_set_error("operand_info: null argument"); return CMPC_EINVAL; }
    auto it = e->ops.find(key);
    if (it == e->ops.end()) { cmpc_set_error("operand_info: no operand named %s", key); return CMPC_EINVAL; }
    if (byte_off) *byte_off = (int64_t)it->second.off;
    if (dt) *dt = it->second.dt;
    if (rows) *rows = it->second.rows;
    if (ld) *ld = it->second.ld;
    return CMPC_OK;
}
extern "C" int cmpc_get_cfg(cmpc_handle e, cmpc_cfg* out) {
    if (!e || !out) { cmpc_set_error("get_cfg: null argument"); return CMPC_EINVAL; }
    *out = e->cfg;
    return CMPC_OK;
}
extern "C" int cmpc_grad_bucket_count(cmpc_handle e) { return e ? E::NBK : 0; }
extern "C" int cmpc_grad_bucket(cmpc_handle e, int b, int* nranges, int64_t offsets[4], int64_t counts[4]) {
    if (!e || b < 0 || b >= E::NBK || !nranges) { cmpc_set_error("grad_bucket: bad argument"); return CMPC_EINVAL; }
    *nranges = (int)e->bucket[b].size();
    for (int i = 0; i < *nranges; ++i) { if (offsets) offsets[i] = e->bucket[b][i].off; if (counts) counts[i] = e->bucket[b][i].count; }
    return CMPC_OK;
}
extern "C" int cmpc_grad_bucket_wait(cmpc_handle e, int b, void* stream) {
    if (!e || b < 0 || b >= E::NBK) { cmpc_set_error("grad_bucket_wait: bad argument"); return CMPC_EINVAL; }
    CK(set_device(e));
    HCK(hipStreamWaitEvent((hipStream_t)stream, e->bucket_ev[b], 0));
    return CMPC_OK;
}
extern "C" int cmpc_phase_marks(cmpc_handle e, int enable) {
    if (!e) { cmpc_set_error("phase_marks: null handle"); return CMPC_EINVAL; }
    for (auto& m : e->marks) (void)hipEventDestroy(m.second);
    e->marks.clear();
    e->marks_on = enable != 0;
    return CMPC_OK;
}
extern "C" int cmpc_phase_marks_read(cmpc_handle e, int index, const char** name, float* ms_since_first) {
    if (!e || index < 0 || index >= (int)e->marks.size()) return CMPC_EINVAL;       // also the end-of-list signal
    CK(set_device(e));
    HCK(hipEventSynchronize(e->marks[index].second));
    float ms = 0.f;
    HCK(hipEventElapsedTime(&ms, e->marks[0].second, e->marks[index].second));
    if (name) *name = e->marks[index].first.c_str();
    if (ms_since_first) *ms_since_first = ms;
    return CMPC_OK;
}
extern "C" int cmpc_set_lanes(cmpc_handle e, int n_lanes) {
    if (!e || n_lanes < 1 || n_lanes > 3) { cmpc_set_error("set_lanes: n_lanes must be 1, 2 or 3"); return CMPC_EINVAL; }
    e->cfg.n_lanes = n_lanes;
    e->lane[2] = n_lanes == 2 ? e->own_lane[0] : e->own_lane[2];
    return CMPC_OK;
}
extern "C" int cmpc_kernel_timing(cmpc_handle e, int enable) {
    if (!e) { cmpc_set_error("kernel_timing: null handle"); return CMPC_EINVAL; }
    for (hipEvent_t ev : e->tev) (void)hipEventDestroy(ev);
    e->tev.clear(); e->tflops.clear(); e->tbytes.clear(); e->tshape.clear();
    e->timing = enable != 0;
    return CMPC_OK;
}
extern "C" int cmpc_kernel_timing_read(cmpc_handle e, double* ms, double* flops, double* bytes, int64_t* launches) {
    if (!e) { cmpc_set_error("kernel_timing_read: null handle"); return CMPC_EINVAL; }
    CK(set_device(e));
    HCK(hipDeviceSynchronize());
    double t = 0, f = 0, b = 0;
    for (size_t i = 0; i < e->tflops.size(); ++i) {
        float dt_ms = 0.f;
        HCK(hipEventElapsedTime(&dt_ms, e->tev[2 * i], e->tev[2 * i + 1]));
        t += dt_ms; f += e->tflops[i]; b += e->tbytes[i];
    }
    if (getenv("CMPC_TIMING_DUMP")) {            // per-shape breakdown on stderr (diagnostic)
        std::map<std::array<int, 4>, std::array<double, 3>> agg;
        for (size_t i = 0; i < e->tflops.size(); ++i) {
            float d = 0.f; (void)hipEventElapsedTime(&d, e->tev[2 * i], e->tev[2 * i + 1]);
            auto& a = agg[e->tshape[i]]; a[0] += d; a[1] += e->tflops[i]; a[2] += 1;
        }
        for (auto& kv : agg)
            fprintf(stderr, "[gemm_nt] M=%6d N=%5d K=%5d segs=%d  launches %5.0f  total %8.3f ms  avg %7.1f us  %7.1f TFLOP/s\n", kv.first[0], kv.first[1], kv.first[2],
                    kv.first[3], kv.second[2], kv.second[0], 1e3 * kv.second[0] / kv.second[2], kv.second[1] / kv.second[0] / 1e9);
    }
    if (ms) *ms = t;
    if (flops) *flops = f;
    if (bytes) *bytes = b;
    if (launches) *launches = (int64_t)e->tflops.size();
    return CMPC_OK;
}
// CRC-32C (Castagnoli) of host bytes: the per-tensor and per-block checksum of TensorFlow checkpoints (tf_bundle.py); SSE4.2's crc32
// instruction where the host has it, a byte table otherwise.  crc: the value returned for the preceding bytes (0 to start).
namespace {
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(uint32_t c, const unsigned char* p, size_t n) {
    uint64_t c64 = c;
    while (n >= 8) { uint64_t v; memcpy(&v, p, 8); c64 = __builtin_ia32_crc32di(c64, v); p += 8; n -= 8; }
    c = (uint32_t)c64;
    while (n--) c = __builtin_ia32_crc32qi(c, *p++);
    return c;
}
uint32_t crc32c_sw(uint32_t c, const unsigned char* p, size_t n) {
    static uint32_t table[256];
    static const bool init = [] {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t r = i; for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (0x82F63B78u & (0u - (r & 1u))); table[i] = r; }
        return true;
    }();
    (void)init;
    while (n--) c = table[(c ^ *p++) & 0xff] ^ (c >> 8);
    return c;
}
}  // namespace
extern "C" uint32_t cmpc_crc32c(uint32_t crc, const void* data, size_t n) {
    const unsigned char* p = (const unsigned char*)data;
    const uint32_t c = ~crc;
    return ~(__builtin_cpu_supports("sse4.2") ? crc32c_hw(c, p, n) : crc32c_sw(c, p, n));
}

extern "C" int cmpc_launch_count(cmpc_handle e, int64_t* n) { if (!e || !n) return CMPC_EINVAL; *n = e->launches_step; return CMPC_OK; }

// ------------------------------------------------------------------------------------------
extern "C" int cmpc_forward(cmpc_handle e, const cmpc_feeds* f, const cmpc_fetches* fetch, void* stream) {
    if (!e || !f || !f->words || !f->seq_len || !f->c4 || !f->c5) { cmpc_set_error("forward: null handle / feed"); return CMPC_EINVAL; }
    if (!e->v5 && !f->c3) { cmpc_set_error("forward: CMPC_model needs the c3 tap"); return CMPC_EINVAL; }
    if (e->v5 && (!f->c2 || (e->cfg.hsv && !f->im))) { cmpc_set_error("forward: CMPCv5_BiLSTM needs the c2 tap (and the image feed for the HSV variant)"); return CMPC_EINVAL; }
    CK(set_device(e));
    hipStream_t main = (hipStream_t)stream;
    t_cur = e;
    e->l0 = g_cmpc_launches;
    e->seq_len_feed = f->seq_len; e->target_feed = f->target_fine; e->last_main = main;
    e->c2_feed = f->c2; e->im_feed = f->im;
    const int B = e->B, Cp = e->Cp, NL = e->nlev, NX = e->nex;
    e->have_target = f->target_fine != nullptr;
    for (auto& d : e->deferred) d.clear();
    e->lv[0].feat = f->c5; e->lv[1].feat = f->c4; e->lv[2].feat = f->c3;
    CK(params_ready(e, main, 0));
    CK(mark(e, "fwd:start", main));
    HCK(hipMemsetAsync(e->ws, 0, e->zf_bytes, main));                         // every accumulate-into buffer of the forward pass
    CK(text_fwd(e, main, f->words, f->seq_len));
    CK(parser_fwd(e, main));
    CK(cmpc_lang_pool_fwd(e->parse, e->wf, e->vl, e->vl_rstd, B, e->T, Cp, e->RNN, 2, 0, e->NC, main));      // valid_lang: entity + attribute (video: ea_lang)
    // nec_lang: + relation; the video model's valid_lang: all but "unnecessary" (vid:215-227), and its action vector (vid:203-213)
    CK(cmpc_lang_pool_fwd(e->parse, e->wf, e->nec, e->nec_rstd, B, e->T, Cp, e->RNN, e->vid ? 4 : 3, 0, e->NC, main));
    if (e->vid) CK(cmpc_lang_pool_fwd(e->parse, e->wf, e->ac, e->ac_rstd, B, e->T, Cp, e->RNN, 1, 3, e->NC, main));
    CK(params_ready(e, main, 1));
    CK(mark(e, "fwd:text_done", main));
    // Everything that is a function of the text alone (the levels' language operands, the exchange modules' queries) goes first, on
    // the lanes, while the backbone (caller's side stream) is still running; each lane then waits for the visual features itself.
    hipStream_t st[3];
    CK(fork_lanes(e, main, st));
    if (f->target_fine && e->cfg.n_lanes > 1) {
        // a training step: the backward pass's two clears (its accumulate-into region and the 304 MB gradient buffer, 40-60 us at the head of
        // cmpc_backward's serial chain) are issued here, on a lane, under the backbone -- nothing of the forward pass touches either, and the
        // previous step's optimizer (the last reader of the gradients) is complete: params_ready(.., 1) above
        HCK(hipMemsetAsync(e->ws + e->zf_bytes, 0, e->zb_bytes, st[2]));
        HCK(hipMemsetAsync(e->grads, 0, (size_t)e->total * sizeof(float), st[2]));
        e->bwd_zeroed = true;
    }
    for (int i = 0; i < NL; ++i) { CK(e->vid ? level_lang_fwd_video(e, st[i], i) : level_lang_fwd(e, st[i], i)); CK(exchange_lang_fwd(e, st[i], i)); CK(exchange_lang_fwd(e, st[i], NX + i)); }
    for (int i = 0; i < 3; ++i) {           // every lane waits for its own tap when the caller says when each is complete, else for all of them
        void* ev = f->feats_ready_lv[i] ? f->feats_ready_lv[i] : f->feats_ready;
        if (ev) HCK(hipStreamWaitEvent(st[i], (hipEvent_t)ev, 0));       // one lane: st[0..2] are the caller's stream, which then waits for all three
    }
    CK(mark(e, "fwd:feats_ready", st[0]));
    if (e->v5) {
        // hsv:120-126 (the image feed only) before the levels that read it; the decoder's low-level branch (backbone only) on the idle third lane
        if (e->cfg.hsv) {
            CK(cmpc_hsv_map(e->dt, f->im, e->hsv, 64, B, e->H, e->W, e->h, e->w, st[2]));
            if (e->cfg.n_lanes > 1) { hipEvent_t ev = next_event(e); HCK(hipEventRecord(ev, st[2])); HCK(hipStreamWaitEvent(st[0], ev, 0)); HCK(hipStreamWaitEvent(st[1], ev, 0)); }
        }
        CK(decoder_low_fwd(e, st[2]));
    }
    const char* lvm[3] = {"fwd:level_c5_done", "fwd:level_c4_done", "fwd:level_c3_done"};
    for (int i = 0; i < NL; ++i) { CK(e->vid ? level_fwd_video(e, st[i], i, f->target_fine) : level_fwd(e, st[i], i, f->target_fine)); CK(mark(e, lvm[i], st[i])); }
    CK(join_lanes(e, main));
    if (f->levels_done) HCK(hipEventRecord((hipEvent_t)f->levels_done, main));
    // gated_exchange_fusion_lstm_2times (CMPC_model.py:261-293: modules c3, c4, c5 = lv[2], lv[1], lv[0], each reading the other two;
    // CMPCv5_BiLSTM_model.py:349-388: modules c4, c5 = lv[1], lv[0], each reading the other one)
    const void* fz[3] = {e->lv[NL - 1].F, e->lv[NL - 2].F, NL > 2 ? e->lv[0].F : nullptr};
    const int o1[3] = {1, 0, 0}, o2[3] = {2, 2, 1};
    CK(fork_lanes(e, main, st));
    for (int i = 0; i < NX; ++i) CK(exchange_fwd(e, st[i], i, fz[i], fz[o1[i]], NX > 2 ? fz[o2[i]] : nullptr));
    CK(join_lanes(e, main));
    CK(mark(e, "fwd:exch1_done", main));
    const void* ez[3] = {e->ex[0].out, e->ex[1].out, NX > 2 ? e->ex[2].out : nullptr};
    CK(fork_lanes(e, main, st));
    hipEvent_t ex2_done[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < NX; ++i) {
        CK(exchange_fwd(e, st[i], NX + i, ez[i], ez[o1[i]], NX > 2 ? ez[o2[i]] : nullptr));
        if (e->cfg.n_lanes > 1) { ex2_done[i] = next_event(e); HCK(hipEventRecord(ex2_done[i], st[i])); }
    }
    if (e->v5 && e->cfg.n_lanes > 1) { hipEvent_t ev = next_event(e); HCK(hipEventRecord(ev, e->lane[2])); HCK(hipStreamWaitEvent(main, ev, 0)); }   // the low-level branch
    CK(clstm_fwd(e, main, ex2_done));            // ConvLSTM step s only waits for the round-2 module s that feeds it
    CK(mark(e, "fwd:clstm_done", main));
    const float* l5 = e->lv[0].loss; const float* l4 = e->lv[1].loss; const float* l3 = NL > 2 ? e->lv[2].loss : e->zeros_bt;
    if (!e->v5) {
        CK(cmpc_score_conv_fwd(e->dt, e->cl[2].h_new, pptr(e, "score/DW"), pptr(e, "score/biases"), e->score, B, e->h, e->w, e->Mp, e->M, main));
        CK(cmpc_upsample_fwd(e->score, e->up, e->sigm, f->target_fine, e->loss, e->iu, e->iu + B, B, e->h, e->w, e->H, e->W, main));
    } else CK(aspp_decoder_fwd(e, main, f->target_fine));
    if (e->have_target) {
        hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(64), 0, main, e->loss, l5, l4, l3, e->iu, e->iu + B, B,
                           e->cfg.loss_w[0], e->cfg.loss_w[1], e->cfg.loss_w[2], NL > 2 ? e->cfg.loss_w[3] : 0.f, e->scalars);
        CK(cmpc_check_launch("scalars"));
    }
    if (fetch) {
        const int ph = e->v5 ? e->h2 : e->h, pw = e->v5 ? e->w2 : e->w;
        if (fetch->pred) HCK(hipMemcpyAsync(fetch->pred, e->score, (size_t)B * ph * pw * 4, hipMemcpyDeviceToDevice, main));
        if (fetch->up) HCK(hipMemcpyAsync(fetch->up, e->up, (size_t)B * e->H * e->W * 4, hipMemcpyDeviceToDevice, main));
        if (fetch->sigm) HCK(hipMemcpyAsync(fetch->sigm, e->sigm, (size_t)B * e->H * e->W * 4, hipMemcpyDeviceToDevice, main));
    }
    CK(mark(e, "fwd:end", main));
    e->launches_step = g_cmpc_launches - e->l0;
    return CMPC_OK;
}

// ------------------------------------------------------------------------------------------
// backward: reverse stage order; a stage output consumed by several stages gets the SUM of their input gradients
// (add_n), exactly what tf.gradients' AddN nodes do for CMPC_model.py:447.
// ------------------------------------------------------------------------------------------
#include <chrono>
static double host_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HOSTPROF(tag) do { if (hp) { const double t_ = host_now(); fprintf(stderr, "[host] %-14s %7.3f ms\n", tag, t_ - hp_t); hp_t = t_; } } while (0)
// event (hipEvent_t, caller-owned; NULL = none) that every later cmpc_backward records once the pyramid levels' backward is complete: what
// follows is the grouped weight-gradient launch of the levels beside the text encoder's backward (room for independent work of the caller)
extern "C" int cmpc_debug_check_guards(cmpc_handle e) {
    if (!e) return -1;
    (void)hipDeviceSynchronize();
    int bad = 0;
    std::vector<unsigned char> buf(g_ws_guard);
    for (size_t i = 0; i < e->guards.size(); ++i) {
        if (hipMemcpy(buf.data(), e->guards[i].first, g_ws_guard, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        for (size_t k = 0; k < g_ws_guard; ++k)
            if (buf[k]) { fprintf(stderr, "[guard] allocation #%zu (%zu bytes, ws offset %zu): byte +%zu past its end is 0x%02x\n", i, e->guards[i].second,
                                  (size_t)(e->guards[i].first - e->ws) - ((e->guards[i].second + 255) / 256 * 256), k + ((e->guards[i].second + 255) / 256 * 256 - e->guards[i].second), buf[k]); ++bad; break; }
    }
    fprintf(stderr, "[guard] %zu allocations checked, %d violated\n", e->guards.size(), bad);
    return bad;
}

extern "C" int cmpc_set_bwd_levels_event(cmpc_handle e, void* event) {
    if (!e) { cmpc_set_error("set_bwd_levels_event: null handle"); return CMPC_EINVAL; }
    e->user_bwd_levels = (hipEvent_t)event;
    return CMPC_OK;
}

extern "C" int cmpc_backward(cmpc_handle e, void* stream) {
    if (!e) { cmpc_set_error("backward: null handle"); return CMPC_EINVAL; }
    static const bool hp = getenv("CMPC_HOST_PROFILE") != nullptr;
    double hp_t = host_now();
    if (!e->have_target || !e->seq_len_feed) { cmpc_set_error("backward: the last cmpc_forward had no target_fine"); return CMPC_EINVAL; }
    if (e->v5 && !e->cfg.bn_train) { cmpc_set_error("backward: CMPCv5_BiLSTM trains with batch statistics (cfg.bn_train = 1, mode == 'train')"); return CMPC_EINVAL; }
    CK(set_device(e));
    hipStream_t main = (hipStream_t)stream;
    t_cur = e;
    const int B = e->B, T = e->T, Cp = e->Cp, Mp = e->Mp, dt = e->dt, NL = e->nlev, NX = e->nex;
    const long nmap = (long)e->R * Mp;
    const float* target = e->target_feed;
    CK(mark(e, "bwd:start", main));
    if (!e->bwd_zeroed) {      // normally done during the forward pass (below the levels' language work, off the critical path)
        HCK(hipMemsetAsync(e->ws + e->zf_bytes, 0, e->zb_bytes, main));
        HCK(hipMemsetAsync(e->grads, 0, (size_t)e->total * sizeof(float), main));
    }
    e->bwd_zeroed = false;
    HOSTPROF("memsets");
    cmpc_fold_begin(&e->fold);
    struct FoldGuard { ~FoldGuard() { cmpc_fold_begin(nullptr); } } fold_guard;      // an early error return must not leave the collector on
    // final score head (CMPC_model) / ASPP + decoder (CMPCv5_BiLSTM), then the ConvLSTM
    if (!e->v5) {
        CK(cmpc_upsample_loss_bwd(e->up, target, e->dscore, e->cfg.loss_w[0] * e->cfg.loss_scale / B, B, e->h, e->w, e->H, e->W, main));
        CK(cmpc_score_conv_bwd(dt, e->dscore, e->cl[2].h_new, pptr(e, "score/DW"), e->dfused, 0, gptr(e, "score/DW"), gptr(e, "score/biases"),
                               B, e->h, e->w, Mp, e->M, main));
    } else {
        CK(aspp_decoder_bwd(e, main, target));
        // UPDATE_OPS of the train step (v5:575-577): the moving statistics follow this step's batch statistics
        for (BnLayer& L : e->bn) CK(cmpc_bn_update_moving(L.sums, L.R, e->cfg.bn_decay, L.mm, L.mv, L.C, L.Cpad, main));
        CK(mark(e, "bwd:aspp_done", main));
    }
    const int o1[3] = {1, 0, 0}, o2[3] = {2, 2, 1};
    const void* fz[3] = {e->lv[NL - 1].F, e->lv[NL - 2].F, NL > 2 ? e->lv[0].F : nullptr};
    const void* ez[3] = {e->ex[0].out, e->ex[1].out, NX > 2 ? e->ex[2].out : nullptr};
    hipStream_t st[3];
    // ConvLSTM backward (last step first, on main); the round-2 exchange module s starts on its lane as soon as step s has
    // produced its input gradient
    hipEvent_t dx_ready[3] = {nullptr, nullptr, nullptr};
    if (e->cfg.n_lanes > 1) for (int i = 0; i < NX; ++i) dx_ready[i] = next_event(e);
    CK(clstm_bwd(e, main, dx_ready));
    CK(mark(e, "bwd:clstm_done", main));
    HOSTPROF("clstm");
    for (int i = 0; i < NX; ++i) {
        st[i] = e->cfg.n_lanes > 1 ? e->lane[i] : main;
        if (dx_ready[i]) HCK(hipStreamWaitEvent(st[i], dx_ready[i], 0));
        CK(exchange_bwd(e, st[i], NX + i, e->cl[i].dx, ez[i], ez[o1[i]], NX > 2 ? ez[o2[i]] : nullptr));
    }
    CK(join_lanes(e, main));
    CK(mark(e, "bwd:exch2_done", main));
    HOSTPROF("exch2");
    // gradient of input j of a round = dfeat of module j + the f1 / f2 gradients of the modules that read it
    auto fan_in = [&](int base, int j, const void* (&src)[3]) {
        int n = 0;
        src[n++] = e->ex[base + j].dfeat;
        for (int i = 0; i < NX; ++i) {
            if (i == j) continue;
            if (o1[i] == j) src[n++] = e->ex[base + i].dfs[0];
            if (NX > 2 && o2[i] == j) src[n++] = e->ex[base + i].dfs[1];
        }
        return n;
    };
    auto add_fan = [&](hipStream_t s, void* dst, const void* (&src)[3], int n) {
        return n == 3 ? add_n(s, dt, dst, {src[0], src[1], src[2]}, false, nmap) : add_n(s, dt, dst, {src[0], src[1]}, false, nmap);
    };
    // exchange round 1
    CK(fork_lanes(e, main, st));
    for (int j = 0; j < NX; ++j) {
        const void* src[3];
        const int n = fan_in(NX, j, src);
        if (n != NX) { cmpc_set_error("backward: exchange fan-in"); return CMPC_EINVAL; }
        CK(add_fan(st[j], e->de1[j], src, n));
        CK(exchange_bwd(e, st[j], j, e->de1[j], fz[j], fz[o1[j]], NX > 2 ? fz[o2[j]] : nullptr));
    }
    CK(join_lanes(e, main));
    CK(mark(e, "bwd:exch1_done", main));
    HOSTPROF("exch1");
    // pyramid levels (lane i = level i = exchange input NL-1-i); the language-side sums run on main meanwhile, and so do
    // the weight-gradient products of the bucket that has just become final (exchange modules, ConvLSTM, final score / ASPP + decoder):
    // issued AFTER the fork, they run beside the levels instead of holding the lanes back
    CK(fork_lanes(e, main, st));
    if (NX > 2) CK(add_n(main, DT_F32, e->dnec, {e->ex[0].dnec, e->ex[1].dnec, e->ex[2].dnec, e->ex[3].dnec, e->ex[4].dnec, e->ex[5].dnec}, false, (long)B * Cp));
    else CK(add_n(main, DT_F32, e->dnec, {e->ex[0].dnec, e->ex[1].dnec, e->ex[2].dnec, e->ex[3].dnec}, false, (long)B * Cp));
    if (!e->vid) CK(cmpc_lang_pool_bwd(e->dnec, e->nec, e->nec_rstd, e->parse, e->wf, e->dparse, e->dwf, B, T, Cp, e->RNN, 3, 0, e->NC, main));
    CK(flush_bucket(e, main, 0));
    for (int i = 0; i < NL; ++i) {
        const void* src[3];
        const int n = fan_in(0, NL - 1 - i, src);
        CK(add_fan(st[i], e->lv[i].dfus, src, n));
        CK(e->vid ? level_bwd_video(e, st[i], i, target) : level_bwd(e, st[i], i, target));
        const char* lvb[3] = {"bwd:level_c5_done", "bwd:level_c4_done", "bwd:level_c3_done"};
        CK(mark(e, lvb[i], st[i]));
    }
    CK(join_lanes(e, main));
    if (e->user_bwd_levels) HCK(hipEventRecord(e->user_bwd_levels, main));
    HOSTPROF("levels");
    // Every weight-gradient product queued so far (levels, exchanges, ConvLSTM: ~1.7 ms of MFMA work) has its operands complete:
    // issue it on lane 0 now, beside the text encoder's backward (a serial chain of ~60 small launches on `main` that leaves
    // the chip nearly idle); the text encoder's own few products follow in a second, small launch.
    hipEvent_t wg_done = nullptr;
    hipStream_t wst = main;
    if (e->cfg.n_lanes > 1 && e->wgrad_overlap) {
        hipEvent_t ev = next_event(e);
        HCK(hipEventRecord(ev, main));
        HCK(hipStreamWaitEvent(e->lane[0], ev, 0));
        wst = e->lane[0];
    }
    for (int b = 1; b <= 3; ++b) CK(flush_bucket(e, wst, b));         // levels (+ score_cX, laterals)
    CK(mark(e, "bwd:dW_main_done", wst));
    if (wst != main) { wg_done = next_event(e); HCK(hipEventRecord(wg_done, wst)); }
    HOSTPROF("dW main");
    if (e->vid) {
        // video: the fusion's language bias is the exchange modules' vector (all but "unnecessary"), Mutan's the entity+attribute one, the
        // temporal pooling's the action one
        CK(add_n(main, DT_F32, e->dnec, {e->lv[0].dvl, e->lv[1].dvl, e->lv[2].dvl}, true, (long)B * Cp));
        CK(cmpc_lang_pool_bwd(e->dnec, e->nec, e->nec_rstd, e->parse, e->wf, e->dparse, e->dwf, B, T, Cp, e->RNN, 4, 0, e->NC, main));
        CK(add_n(main, DT_F32, e->dvl, {e->lv[0].dea, e->lv[1].dea, e->lv[2].dea}, false, (long)B * Cp));
        CK(add_n(main, DT_F32, e->dac, {e->lv[0].dac, e->lv[1].dac, e->lv[2].dac}, false, (long)B * Cp));
        CK(cmpc_lang_pool_bwd(e->dac, e->ac, e->ac_rstd, e->parse, e->wf, e->dparse, e->dwf, B, T, Cp, e->RNN, 1, 3, e->NC, main));
    } else if (NL > 2) CK(add_n(main, DT_F32, e->dvl, {e->lv[0].dvl, e->lv[1].dvl, e->lv[2].dvl}, false, (long)B * Cp));
    else CK(add_n(main, DT_F32, e->dvl, {e->lv[0].dvl, e->lv[1].dvl}, false, (long)B * Cp));
    CK(cmpc_lang_pool_bwd(e->dvl, e->vl, e->vl_rstd, e->parse, e->wf, e->dparse, e->dwf, B, T, Cp, e->RNN, 2, 0, e->NC, main));
    const float* dpr3 = NL > 2 ? e->lv[2].dpr : e->zeros_bt;
    hipLaunchKernelGGL(col_add3_kernel, dim3((B * T + 255) / 256), dim3(256), 0, main, e->dparse, e->NC, 2, e->lv[0].dpr, e->lv[1].dpr, dpr3, B * T);
    CK(cmpc_check_launch("col_add3"));
    if (NL > 2) CK(add_n(main, DT_F32, e->dwf, {e->lv[0].dwf, e->lv[1].dwf, e->lv[2].dwf}, true, (long)B * T * Cp));
    else CK(add_n(main, DT_F32, e->dwf, {e->lv[0].dwf, e->lv[1].dwf}, true, (long)B * T * Cp));
    CK(parser_bwd(e, main));
    CK(text_bwd(e, main, e->seq_len_feed));
    CK(mark(e, "bwd:text_done", main));
    HOSTPROF("text");
    if (wg_done) HCK(hipStreamWaitEvent(main, wg_done, 0));
    CK(flush_bucket(e, main, E::NBK - 1));     // text encoder + parser
    CK(cmpc_fold_flush(&e->fold, main));       // (nothing is left; ends the collection)
    CK(mark(e, "bwd:end", main));
    HOSTPROF("flush+fold");
    e->last_main = main;
    e->launches_step = g_cmpc_launches - e->l0;
    return CMPC_OK;
}

// tf.train.polynomial_decay + AdamOptimizer.apply_gradients (CMPC_model.py:450-478) + repack of the operands, for ONE gradient bucket.
// `st` first waits (on the device) for the event cmpc_backward recorded when the bucket became final, so the update of the exchange
// modules' and the levels' parameters runs beside the rest of the backward pass: none of those parameters (fp32 masters, packed
// operands) is read again after its bucket is final.  A data-parallel caller orders `st` after the bucket's all-reduce itself.
extern "C" int cmpc_optimizer_bucket(cmpc_handle e, int b, float gscale, void* stream, double* lr_used) {
    if (!e || b < 0 || b >= E::NBK) { cmpc_set_error("optimizer_bucket: bad argument"); return CMPC_EINVAL; }
    CK(set_device(e));
    hipStream_t st = (hipStream_t)stream;
    HCK(hipStreamWaitEvent(st, e->bucket_ev[b], 0));
    const cmpc_cfg& c = e->cfg;
    const double gs = (double)std::min<int64_t>(e->step, c.lr_decay_step);
    const double lr = ((double)c.start_lr - (double)c.end_lr) * pow(1.0 - gs / (double)c.lr_decay_step, (double)c.lr_power) + (double)c.end_lr;
    const double t = (double)(e->step + 1), b1 = 0.9, b2 = 0.999;
    const double lr_t = lr * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t));
    for (const auto& r : e->bucket_segs[b])
        CK(cmpc_adam_step(e->params, e->grads, e->adam_m, e->adam_v, e->segs_dev + r.first, r.second - r.first, (float)lr_t, (float)b1, (float)b2, 1e-8f,
                          gscale / e->cfg.loss_scale, e->nonfinite + b, st));
    for (const auto& r : e->bucket_tiles[b])
        CK(cmpc_pack_weights_range(e->params, e->arena, e->descs_dev, e->tile_prefix_dev, e->tile_desc_dev, e->ndesc, r.first, r.second, st));
    if (b == E::NBK - 1) {                         // the text encoder's bucket is the last to become final: the step is complete
        e->step += 1;
        HCK(hipEventRecord(e->ev_opt0, st));
        HCK(hipEventRecord(e->ev_opt1, st));
        CK(mark(e, "opt:end", st));
        e->opt_pending = true;
    }
    if (lr_used) *lr_used = lr;
    e->launches_step = g_cmpc_launches - e->l0;
    return CMPC_OK;
}

// the whole update on one stream: every bucket in order
extern "C" int cmpc_optimizer_step(cmpc_handle e, float gscale, void* stream, double* lr_used) {
    if (!e) { cmpc_set_error("optimizer_step: null handle"); return CMPC_EINVAL; }
    for (int b = 0; b < E::NBK; ++b) CK(cmpc_optimizer_bucket(e, b, gscale, stream, lr_used));
    return CMPC_OK;
}
