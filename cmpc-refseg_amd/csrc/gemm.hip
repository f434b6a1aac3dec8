// MFMA GEMMs for the CMPC head (gfx950 / CDNA4, wave64).
//
//  gemm_nt : C[M,N] = act(alpha * sum_s A_s[M,K_s] . Bt_s[N,K_s]^T + bias + sbias + pbias) (+C)
//            every 1x1 convolution of the reference (_conv, CMPC_model.py:412-417) in forward,
//            and the dX = dY . W^T products in backward.  Up to 3 K-segments share one
//            accumulator (the reference's channel concats at CMPC_model.py:339,238 and
//            util/cell.py:39 are never materialised).
//  gemm_tn : out[K,N] += sum_r A[r,K]^T . D[r,N]   (weight gradients; one writer per element: split reductions through slabs + a fold)
//
// Operands are T = bf16 (v_mfma_f32_16x16x32_bf16) or T = f32 (v_mfma_f32_16x16x4_f32, exact
// fp32 - the parity mode).  Both use the same byte-level LDS image: a lane's 16-byte chunk is
// one bf16 fragment (8 k-values) or four f32 k-values consumed by four MFMA steps.
#include "cmpc_common.h"
#include <algorithm>
#include <queue>
#include <vector>
#include "../../include/cmpc.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int BKB = 128;      // bytes of K per row per LDS stage (64 bf16 / 32 f32)

// LDS image of a [rows][128 B] tile: 16-B chunk c of row r lives at slot c ^ ((r>>1)&7).
// With 128-B rows two rows share a 256-B bank line, so 16 distinct rows reading the same
// logical chunk touch 16 distinct 16-B slots: conflict-free for ds_read_b128.
__device__ __forceinline__ int nt_lds_off(int row, int c) { return row * BKB + ((c ^ ((row >> 1) & 7)) << 4); }

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ f4 run(const uint4& a, const uint4& b, f4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    static __device__ __forceinline__ f4 run(const uint4& a, const uint4& b, f4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ f4 run(const uint4& a, const uint4& b, f4 acc) {
        // lane (r, q) holds k = 4*(chunk)+e, e = 0..3: MFMA step e pairs element e of both operands.
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
        return acc;
    }
};


// ---- shared epilogue: wave-private fp32 slab [WR][WC] in LDS -> 8 consecutive columns per lane ----
// (two 16-B LDS reads, 16-B bias loads, ONE 16-B bf16 store or two fp32 stores per lane and pass)
__device__ __forceinline__ int slab_col(int row, int col, int WC) { return col ^ ((((row >> 2) & 3) << 4) & (WC - 1)); }

// The activation is dispatched ONCE (template parameter): with a per-element switch on p.act hipcc emitted a scalar
// branch tree per element (8 per pass, 8 passes: ~5 us of the 24 us a 256x128 tile takes).  Everything that only
// depends on the lane's 8 columns (bias, the n_valid mask, tail flags) is hoisted out of the pass loop; row indices
// are 32-bit (a 64-bit division per pass was the other big cost); the pass loop is fully unrolled so that the LDS
// reads and the per-sample / accumulate loads of all passes are in flight together.
template <typename T, int WR, int WC, int ACT, int UNR>
__device__ __forceinline__ void gemm_nt_epilogue_act(const cmpc_gemm_nt_args& p, const float* slab, int lane, int row0, int col0, long bz) {
    constexpr int LPR = WC / 8, RPP = 64 / LPR, NPASS = WR / RPP;
    const int c8 = (lane % LPR) * 8, gn = col0 + c8, rl = lane / LPR;
    if (gn >= p.N) return;
    const bool full = gn + 8 <= p.N;               // N is only guaranteed to be a multiple of 4
    T* Ct = reinterpret_cast<T*>(p.C) + bz * p.sC;
    float* Cf = reinterpret_cast<float*>(p.C) + bz * p.sC;
    const bool out_f32 = p.c_f32 || sizeof(T) == 4;
    float bcol[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto addv = [&](float (&d)[8], const float* src) {
        const float4 t0 = *reinterpret_cast<const float4*>(src);
        d[0] += t0.x; d[1] += t0.y; d[2] += t0.z; d[3] += t0.w;
        if (full) { const float4 t1 = *reinterpret_cast<const float4*>(src + 4); d[4] += t1.x; d[5] += t1.y; d[6] += t1.z; d[7] += t1.w; }
    };
    if (p.bias) addv(bcol, p.bias + gn);
    bool ok[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ok[e] = gn + e < p.n_valid;     // pad columns stay exactly zero
    const unsigned rps = p.rows_per_sample > 0 ? (unsigned)p.rows_per_sample : 1u;
    const unsigned rbase = (unsigned)(bz * (long)p.M);            // batch * M < 2^31 for every caller
    const float alpha = p.alpha;
#pragma unroll UNR
    for (int pass = 0; pass < NPASS; ++pass) {
        const int row = pass * RPP + rl, gm = row0 + row;
        if (gm >= p.M) continue;
        const float4 va = *reinterpret_cast<const float4*>(slab + row * WC + slab_col(row, c8, WC));
        const float4 vb = *reinterpret_cast<const float4*>(slab + row * WC + slab_col(row, c8 + 4, WC));
        float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = bcol[e];
        if (p.sbias || p.pbias) {
            const unsigned idx = rbase + (unsigned)gm, smp = idx / rps;
            if (p.sbias) addv(bv, p.sbias + (long)smp * p.ld_sbias + gn);
            if (p.pbias) addv(bv, p.pbias + (long)(idx - smp * rps) * p.ld_pbias + gn);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = v[e] * alpha + bv[e];
            if (ACT == ACT_RELU) x = fmaxf(x, 0.0f);
            else if (ACT == ACT_TANH) x = cmpc_tanh(x);
            else if (ACT == ACT_SIGMOID) x = 1.0f / (1.0f + expf(-x));
            v[e] = ok[e] ? x : 0.0f;
        }
        const long off = (long)gm * p.ldc + gn;
        if (out_f32) {
            float* Cp = p.c_f32 ? Cf + off : reinterpret_cast<float*>(Ct) + off;
            if (p.accumulate) {
                const float4 o0 = *reinterpret_cast<const float4*>(Cp);
                v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w;
                if (full) { const float4 o1 = *reinterpret_cast<const float4*>(Cp + 4); v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w; }
            }
            *reinterpret_cast<float4*>(Cp) = make_float4(v[0], v[1], v[2], v[3]);
            if (full) *reinterpret_cast<float4*>(Cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            T* Cb = Ct + off;                      // 16-bit storage (bf16 or f16)
            if (full && ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0)) {
                if (p.accumulate) { float o[8]; ld8<T>(Cb, o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += o[e]; }
                st8<T>(Cb, v);
            } else {                               // ragged right edge / odd ldc: rare
                const int ne = full ? 8 : 4;
                for (int e = 0; e < ne; ++e) { float x = v[e]; if (p.accumulate) x += Elem<T>::ld(Cb + e); Elem<T>::st(Cb + e, x); }
            }
        }
    }
}

// UNR: how many passes are unrolled together (all of them when the accumulators are dead; 2 in the 256 x 256 kernel,
// where the second half's 64 accumulator registers are still live during the first half's epilogue)
template <typename T, int WR, int WC, int UNR = 64>
__device__ __forceinline__ void gemm_nt_epilogue(const cmpc_gemm_nt_args& p, const float* slab, int lane, int row0, int col0, long bz) {
    switch (p.act) {
        case ACT_RELU: gemm_nt_epilogue_act<T, WR, WC, ACT_RELU, UNR>(p, slab, lane, row0, col0, bz); break;
        case ACT_TANH: gemm_nt_epilogue_act<T, WR, WC, ACT_TANH, UNR>(p, slab, lane, row0, col0, bz); break;
        case ACT_SIGMOID: gemm_nt_epilogue_act<T, WR, WC, ACT_SIGMOID, UNR>(p, slab, lane, row0, col0, bz); break;
        default: gemm_nt_epilogue_act<T, WR, WC, ACT_NONE, UNR>(p, slab, lane, row0, col0, bz); break;
    }
}

// ------------------------------------------------------------------------------------------
// gemm_nt
// ------------------------------------------------------------------------------------------
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const cmpc_gemm_nt_args p) {
    constexpr int EPC = 16 / (int)sizeof(T);       // elements per 16-B chunk
    constexpr int BK = BKB / (int)sizeof(T);       // elements of K per stage
    constexpr int WAVES_N = (BN == 128 || BM == 64) ? 2 : 1;
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int TM = BM / WAVES_M / 16;          // 16x16 tiles per wave along M
    constexpr int TN = BN / WAVES_N / 16;
    constexpr int NA = BM * 8 / 256;               // A chunks per thread per stage
    constexpr int NB = BN * 8 / 256;
    constexpr int STAGE = (BM + BN) * BKB;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so
    // give each XCD a CONTIGUOUS run of tiles (n fastest): the 8 column tiles of one A row-panel
    // and the whole weight matrix then hit in that XCD's L2 (speed only, never correctness).
    const int gx = (p.N + BN - 1) / BN, nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int tix = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int m0 = (tix / gx) * BM, n0 = (tix % gx) * BN;
    const long bz = blockIdx.z;

    int ntile[3], ntot = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) { ntile[s] = (s < p.nseg) ? p.K[s] / BK : 0; ntot += ntile[s]; }

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[NA], rb[NB];

    auto gload = [&](int tile) {
        int s = 0, t = tile;
        if (t >= ntile[0]) { t -= ntile[0]; s = 1; if (t >= ntile[1]) { t -= ntile[1]; s = 2; } }
        const T* Ap = reinterpret_cast<const T*>(p.A[s]) + bz * p.sA[s];
        const T* Bp = reinterpret_cast<const T*>(p.Bt[s]) + bz * p.sB[s];
        const long lda = p.lda[s], ldb = p.ldb[s];
        const int k0 = t * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int id = tid + 256 * i, row = id >> 3, c = id & 7, gm = m0 + row;
            ra[i] = (gm < p.M) ? *reinterpret_cast<const uint4*>(Ap + gm * lda + k0 + c * EPC) : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int id = tid + 256 * i, row = id >> 3, c = id & 7, gn = n0 + row;
            rb[i] = (gn < p.N) ? *reinterpret_cast<const uint4*>(Bp + gn * ldb + k0 + c * EPC) : uint4{0u, 0u, 0u, 0u};
        }
    };
    auto lstore = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + BM * BKB;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int id = tid + 256 * i;
            *reinterpret_cast<uint4*>(sA + nt_lds_off(id >> 3, id & 7)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int id = tid + 256 * i;
            *reinterpret_cast<uint4*>(sB + nt_lds_off(id >> 3, id & 7)) = rb[i];
        }
    };

    if (ntot > 0) { gload(0); lstore(0); }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < ntot; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ntot) gload(kt + 1);
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + BM * BKB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const uint4*>(sA + nt_lds_off(wm * TM * 16 + i * 16 + fr, 4 * s + fq));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const uint4*>(sB + nt_lds_off(wn * TN * 16 + j * 16 + fr, 4 * s + fq));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < ntot) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> wave-private LDS slab -> row-contiguous vector stores ----
    constexpr int WR = TM * 16, WC = TN * 16;      // wave sub-tile
    float* slab = reinterpret_cast<float*>(smem) + wid * (WR * WC);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + fq * 4 + r, col = j * 16 + fr;
                slab[row * WC + (col ^ ((((row >> 2) & 3) << 4) & (WC - 1)))] = acc[i][j][r];
            }
    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): own wave's LDS writes landed
    __builtin_amdgcn_wave_barrier();

    gemm_nt_epilogue<T, WR, WC>(p, slab, lane, m0 + wm * WR, n0 + wn * WC, bz);
}

// ------------------------------------------------------------------------------------------
// gemm_nt v2: BM x 128 tile (BM = 256 or 128), 8 waves, THREE LDS stages filled by direct
// global->LDS loads (global_load_lds_dwordx4: no staging VGPRs, no ds_write), counted vmcnt so
// the loads of tile t+1 stay in flight across the barrier while tile t is multiplied.
// LDS image identical to v1 (linear destination, XOR applied to the per-lane SOURCE chunk).
// ------------------------------------------------------------------------------------------

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at the
// wave-uniform byte address lds_dst (+ lane*16).  Issued from inline asm so that hipcc does not
// put an s_waitcnt vmcnt(0) in front of every ds_read that might alias it (cdna_hip_programming
// guide 5.7): completion is counted by hand (wait_vmcnt<N> + s_barrier before the reads).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Same with a wave-uniform 64-bit base (SGPR pair) plus a 32-bit per-lane byte offset: half the address VGPRs.
__device__ __forceinline__ void glds16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}


// ------------------------------------------------------------------------------------------
// gemm_nt v4: v2's tile and LDS-DMA pipeline with
//  * fragment double buffering: the ds_reads of the NEXT half k-tile are issued before the 16 MFMAs
//    of the current one (phase trace of v2: 1.05 us per k-tile against 0.43 us of MFMA time, both
//    waves of a SIMD exposing their LDS-read latency twice per k-tile in lockstep);
//  * three k-tiles of loads in flight (the barrier that publishes tile t+1 also retires tile t);
//  * the last k-tile peeled out of the loop (see the loop);
//  * v2's epilogue (a register-direct epilogue with 8-byte stores was measured slower: 7.3 vs 5.5 us per tile).
// ------------------------------------------------------------------------------------------
// blk / nwg: this workgroup's index among the nwg workgroups of product p (a paired launch runs two products in one grid)
template <typename T, int BM>
__device__ __forceinline__ void gemm_nt_v4_body(const cmpc_gemm_nt_args& p, const int blk, const int nwg) {
    constexpr int BN = 128;
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int BK = BKB / (int)sizeof(T);
    constexpr int WAVES_N = 2, WAVES_M = 4;
    constexpr int TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
    constexpr int STAGE = (BM + BN) * BKB;
    constexpr int APW = BM / 8 / 8;
    constexpr int BPW = BN / 8 / 8;
    constexpr int LPT = APW + BPW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int gx = (p.N + BN - 1) / BN;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = blk & 7, xi = blk >> 3;
    const int tix = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int m0 = (tix / gx) * BM, n0 = (tix % gx) * BN;
    const long bz = blockIdx.z;

    int ntile[3], ntot = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) { ntile[s] = (s < p.nseg) ? p.K[s] / BK : 0; ntot += ntile[s]; }

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int r8 = lane >> 3, slot = lane & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // Issue cursor: per-lane row pointers of the CURRENT K-segment live in registers (recomputed only when the
    // segment changes), so that an issue is one 64-bit add per piece -- no kernarg (SMEM) loads and their
    // lgkmcnt(0) in the k-loop.
    const T* pa[APW];
    const T* pb[BPW];
    int iseg = 0, itile = 0, seg_left = ntile[0];
    auto load_seg = [&](const int sg) {      // sg is a literal at every call site: kernarg reads stay scalar loads
        const T* Ap = reinterpret_cast<const T*>(p.A[sg]) + bz * p.sA[sg];
        const T* Bp = reinterpret_cast<const T*>(p.Bt[sg]) + bz * p.sB[sg];
        const long lda = p.lda[sg], ldb = p.ldb[sg];
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int row = (wid * APW + j) * 8 + r8;
            const int gm = min(m0 + row, p.M - 1);           // rows past M: any valid address (never stored)
            pa[j] = Ap + gm * lda + (slot ^ ((row >> 1) & 7)) * EPC;
        }
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int row = (wid * BPW + j) * 8 + r8;
            const int gn = min(n0 + row, p.N - 1);
            pb[j] = Bp + gn * ldb + (slot ^ ((row >> 1) & 7)) * EPC;
        }
    };
    load_seg(0);
    auto issue_next = [&](int buf) {
        const int k0 = itile * BK;
        const uint32_t base = lds0 + buf * STAGE;
#ifndef V4_NO_LOADS          // (study builds: -DV4_NO_LOADS / -DV4_NO_MFMA isolate the two halves of the loop)
#pragma unroll
        for (int j = 0; j < APW; ++j)
            glds16(pa[j] + k0, __builtin_amdgcn_readfirstlane(base + (wid * APW + j) * 1024));
#pragma unroll
        for (int j = 0; j < BPW; ++j)
            glds16(pb[j] + k0, __builtin_amdgcn_readfirstlane(base + BM * BKB + (wid * BPW + j) * 1024));
#else
        (void)k0; (void)base;
#endif
        ++itile;
        if (--seg_left == 0) {
            itile = 0;
            ++iseg;
            if (iseg == 1 && p.nseg > 1) { seg_left = ntile[1]; load_seg(1); }
            else if (iseg == 2 && p.nseg > 2) { seg_left = ntile[2]; load_seg(2); }
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    auto read_frags = [&](int buf, int s, uint4 (&a)[TM], uint4 (&b)[TN]) {
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + BM * BKB;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[i] = *reinterpret_cast<const uint4*>(sA + nt_lds_off(wm * TM * 16 + i * 16 + fr, 4 * s + fq));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            b[j] = *reinterpret_cast<const uint4*>(sB + nt_lds_off(wn * TN * 16 + j * 16 + fr, 4 * s + fq));
    };
    auto mma_all = [&](const uint4 (&a)[TM], const uint4 (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
#ifndef V4_NO_MFMA
            for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
#else
            for (int j = 0; j < TN; ++j) acc[i][j][0] += __uint_as_float(a[i].x ^ b[j].x);
#endif
    };

    if (ntot > 0) issue_next(0);
    if (ntot > 1) issue_next(1);
    if (ntot > 2) issue_next(2);
    if (ntot > 2) wait_vmcnt<2 * LPT>(); else if (ntot > 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    uint4 a0[TM], b0[TN], a1[TM], b1[TN];
    if (ntot > 0) read_frags(0, 0, a0, b0);
    int cur = 0;
    // The last k-tile is peeled: a conditional skip inside the loop body makes hipcc merge the two paths'
    // LDS counters and wait for the NEXT half's reads in front of the first MFMA of every half.
    for (int kt = 0; kt + 1 < ntot; ++kt) {
        read_frags(cur, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);       // keep the reads of the next half IN FRONT of this half's MFMAs
        mma_all(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        const int nxt = (cur == 2) ? 0 : cur + 1;
        // every read of tile kt has landed in registers before the barrier lets another wave's DMA reuse its buffer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt + 2 < ntot) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < ntot) issue_next(cur);
        read_frags(nxt, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_all(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    if (ntot > 0) {
        read_frags(cur, 1, a1, b1);
        mma_all(a0, b0);
        mma_all(a1, b1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    constexpr int WR = TM * 16, WC = TN * 16;
    float* slab = reinterpret_cast<float*>(smem) + wid * (WR * WC);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + fq * 4 + r, col = j * 16 + fr;
                slab[row * WC + (col ^ (((row >> 2) & 3) << 4))] = acc[i][j][r];
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    gemm_nt_epilogue<T, WR, WC>(p, slab, lane, m0 + wm * WR, n0 + wn * WC, bz);
}

template <typename T, int BM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_v4_kernel(const cmpc_gemm_nt_args p) {
    gemm_nt_v4_body<T, BM>(p, blockIdx.x, gridDim.x);
}
// Two independent products of the same tile configuration in ONE grid (the two gated branches of an exchange module, forward and
// backward: 2 x 200 workgroups of a 12 800 x 512 product fill the chip where one leaves a fifth of it idle).  na % 8 == 0, so that
// both halves keep the XCD-aware tile order.
struct NtPair { cmpc_gemm_nt_args a, b; int na; };
template <typename T, int BM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_v4_pair_kernel(const NtPair pp) {
    const bool first = (int)blockIdx.x < pp.na;
    const cmpc_gemm_nt_args& p = first ? pp.a : pp.b;
    gemm_nt_v4_body<T, BM>(p, first ? blockIdx.x : blockIdx.x - pp.na, first ? pp.na : gridDim.x - pp.na);
}

// ------------------------------------------------------------------------------------------
// gemm_nt v5: 256 x 256 tile.  The phase trace of v2/v4 shows the main loop bound by what a CU can take in
// (about 50 GB/s of LDS-DMA per CU, 48 KB per 64-deep k-tile of a 256 x 128 tile), not by MFMA or LDS: a
// 256 x 256 tile needs 1.5x fewer operand bytes per flop.  K-step 32 (64-byte LDS rows) so that FOUR stages
// of 32 KiB fit (three k-tiles of loads in flight); 8 waves as 2 (M) x 4 (N), 128 x 64 per wave = 32 MFMA
// 16x16x32 per k-tile and wave from 12 ds_read_b128.  LDS image: row r at 64 r bytes, 16-B chunk c stored at
// chunk c ^ g((r >> 2) & 3), g = {0,3,2,1}: each 16-lane group of a ds_read_b128 (lane groups of
// MI355X_MICROARCH.md, LDS table) then covers all 64 banks once.  The XOR is applied to the per-lane SOURCE
// chunk of the LDS-DMA (linear destination).  Epilogue: the fp32 slab of v2 in two halves of 64 rows per wave.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int v5_g(int row) { return (4 - ((row >> 2) & 3)) & 3; }
__device__ __forceinline__ int v5_lds_off(int row, int c) { return row * 64 + ((c ^ v5_g(row)) << 4); }

#ifdef CMPC_V5_TRACE      // diagnostic build only (scripts/v5_trace.py): 100 MHz timestamps of workgroup phases
__device__ unsigned long long g_v5_trace[8 * 8192];
#define V5_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x + gridDim.x * blockIdx.z < 8192) g_v5_trace[(blockIdx.x + gridDim.x * blockIdx.z) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define V5_STAMP(i) do { } while (0)
#endif

template <typename T>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_v5_kernel(const cmpc_gemm_nt_args p) {
    constexpr int BM = 256, BN = 256, ROWB = 64;
    V5_STAMP(0);
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int BK = ROWB / (int)sizeof(T);              // 32 bf16
    constexpr int WAVES_N = 4;
    constexpr int TM = 8, TN = 4;                          // 128 x 64 per wave
    constexpr int STAGE = (BM + BN) * ROWB;                // 32 KiB
    constexpr int APW = BM / 16 / 8, BPW = BN / 16 / 8;    // 1-KiB pieces (16 rows x 64 B) per wave and stage: 2 + 2
    constexpr int LPT = APW + BPW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int gx = (p.N + BN - 1) / BN, nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int tix = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    // tile order: strips of 4 tile columns, row-major inside a strip.  The ~32 workgroups an XCD runs at a time then form an 8 x 4
    // block of tiles (12 operand panels per k-step through its L2) instead of 1.6 rows of a wide product (22 panels).
    const int gy = nwg / gx, strip = tix / (gy * 4), sw = min(4, gx - strip * 4), rem = tix - strip * gy * 4;
    const int m0 = (rem / sw) * BM, n0 = (strip * 4 + rem % sw) * BN;
    const long bz = blockIdx.z;

    int ntile[3], ntot = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) { ntile[s] = (s < p.nseg) ? p.K[s] / BK : 0; ntot += ntile[s]; }

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane >> 2, slot = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // issue cursor with the per-lane row pointers of the current K-segment in registers (as in v4)
    // wave-uniform segment bases (SGPR pairs) + 32-bit per-lane byte offsets (an operand of this path is < 4 GiB)
    const T* sA;
    const T* sB;
    uint32_t oa[APW], ob[BPW];
    int iseg = 0, itile = 0, seg_left = ntile[0];
    auto load_seg = [&](const int sg) {
        sA = reinterpret_cast<const T*>(p.A[sg]) + bz * p.sA[sg];
        sB = reinterpret_cast<const T*>(p.Bt[sg]) + bz * p.sB[sg];
        const int lda = p.lda[sg], ldb = p.ldb[sg];
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int row = (wid * APW + j) * 16 + r16;
            oa[j] = (uint32_t)((min(m0 + row, p.M - 1) * (long)lda + (slot ^ v5_g(row)) * EPC) * (long)sizeof(T));   // rows past M: any valid address
        }
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int row = (wid * BPW + j) * 16 + r16;
            ob[j] = (uint32_t)((min(n0 + row, p.N - 1) * (long)ldb + (slot ^ v5_g(row)) * EPC) * (long)sizeof(T));
        }
    };
    load_seg(0);
    auto issue_next = [&](int buf) {
        const int k0 = itile * BK;
        const uint32_t base = lds0 + buf * STAGE;
#pragma unroll
        for (int j = 0; j < APW; ++j)
            glds16s(sA + k0, oa[j], __builtin_amdgcn_readfirstlane(base + (wid * APW + j) * 1024));
#pragma unroll
        for (int j = 0; j < BPW; ++j)
            glds16s(sB + k0, ob[j], __builtin_amdgcn_readfirstlane(base + BM * ROWB + (wid * BPW + j) * 1024));
        ++itile;
        if (--seg_left == 0) {
            itile = 0;
            ++iseg;
            if (iseg == 1 && p.nseg > 1) { seg_left = ntile[1]; load_seg(1); }
            else if (iseg == 2 && p.nseg > 2) { seg_left = ntile[2]; load_seg(2); }
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    auto read_b = [&](int buf, uint4 (&b)[TN]) {
        const char* sB = smem + buf * STAGE + BM * ROWB;
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const uint4*>(sB + v5_lds_off(wn * (TN * 16) + j * 16 + fr, fq));
    };
    auto read_a = [&](int buf, int half, uint4 (&a)[4]) {
        const char* sA = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const uint4*>(sA + v5_lds_off(wm * (TM * 16) + (half * 4 + i) * 16 + fr, fq));
    };
    auto mma_half = [&](int half, const uint4 (&a)[4], const uint4 (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[half * 4 + i][j] = Mma<T>::run(a[i], b[j], acc[half * 4 + i][j]);
    };

    // Pipeline: three k-tiles of loads in flight (4 stages); per k-tile the wave's 32 MFMAs run as two halves of 16
    // with the LDS reads of the NEXT half issued in front of each (A rows 64-127 of this tile; then B and A rows 0-63 of
    // the next tile, which the barrier in between has just published).
    if (ntot > 0) issue_next(0);
    if (ntot > 1) issue_next(1);
    if (ntot > 2) issue_next(2);
    V5_STAMP(1);
    if (ntot > 2) wait_vmcnt<2 * LPT>(); else if (ntot > 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    V5_STAMP(2);
    uint4 alo[4], ahi[4], bb[TN];
    if (ntot > 0) { read_b(0, bb); read_a(0, 0, alo); }
    int cur = 0;
    for (int kt = 0; kt + 1 < ntot; ++kt) {
        if (kt + 3 < ntot) issue_next((cur + 3) & 3);           // buffer of tile kt-1: free since the previous barrier
        read_a(cur, 1, ahi);
        __builtin_amdgcn_sched_barrier(0);
        mma_half(0, alo, bb);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every read of tile kt is in registers before the barrier
        if (kt + 3 < ntot) wait_vmcnt<2 * LPT>(); else if (kt + 2 < ntot) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const int nxt = (cur + 1) & 3;
        read_a(nxt, 0, alo);                                   // next tile's first A half under this tile's second MFMA half
        __builtin_amdgcn_sched_barrier(0);
        mma_half(1, ahi, bb);
        __builtin_amdgcn_sched_barrier(0);
        read_b(nxt, bb);                                       // (B is single-buffered: 256 registers per wave is the budget)
        cur = nxt;
    }
    if (ntot > 0) {
        read_a(cur, 1, ahi);
        mma_half(0, alo, bb);
        mma_half(1, ahi, bb);
    }
    V5_STAMP(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    constexpr int WR = 64, WC = TN * 16;                   // slab = half of the wave's rows
    float* slab = reinterpret_cast<float*>(smem) + wid * (WR * WC);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i * 16 + fq * 4 + r, col = j * 16 + fr;
                    slab[row * WC + (col ^ (((row >> 2) & 3) << 4))] = acc[half * 4 + i][j][r];
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        gemm_nt_epilogue<T, WR, WC, 1>(p, slab, lane, m0 + wm * (TM * 16) + half * 64, n0 + wn * WC, bz);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads of this half are done before it is overwritten
        __builtin_amdgcn_wave_barrier();
    }
    V5_STAMP(4);
#ifdef CMPC_V5_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V5_STAMP(5);
#endif
}

// gemm_nt v3 = v2 with producer / consumer wave specialisation.
template <typename T, int BM>
__global__ __launch_bounds__(512) void gemm_nt_v3_kernel(const cmpc_gemm_nt_args p) {
    constexpr int BN = 128;
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int BK = BKB / (int)sizeof(T);
    // waves 0-3 (one per SIMD) only multiply, waves 4-7 (their SIMD partners) only issue LDS-DMA: the
    // ~100-cycle issue cost of each piece then runs beside the partner's MFMAs instead of in front of them
    constexpr int WAVES_N = 2, WAVES_M = 2;
    constexpr int TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
    constexpr int STAGE = (BM + BN) * BKB;
    constexpr int APW = BM / 8 / 4;                // A pieces (8 rows x 128 B = 1 KiB) per LOADER wave per stage
    constexpr int BPW = BN / 8 / 4;
    constexpr int LPT = APW + BPW;                 // LDS-DMA instructions per loader wave per tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wid >= 4;
    const int lw = wid & 3;                        // index within the role
    const int wm = lw / WAVES_N, wn = lw % WAVES_N;
    const int gx = (p.N + BN - 1) / BN, nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int tix = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int m0 = (tix / gx) * BM, n0 = (tix % gx) * BN;
    const long bz = blockIdx.z;

    int ntile[3], ntot = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) { ntile[s] = (s < p.nseg) ? p.K[s] / BK : 0; ntot += ntile[s]; }

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int r8 = lane >> 3, slot = lane & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int tile, int buf) {
        int s = 0, t = tile;
        if (t >= ntile[0]) { t -= ntile[0]; s = 1; if (t >= ntile[1]) { t -= ntile[1]; s = 2; } }
        const T* Ap = reinterpret_cast<const T*>(p.A[s]) + bz * p.sA[s];
        const T* Bp = reinterpret_cast<const T*>(p.Bt[s]) + bz * p.sB[s];
        const long lda = p.lda[s], ldb = p.ldb[s];
        const int k0 = t * BK;
        const uint32_t base = lds0 + buf * STAGE;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int blk = lw * APW + j, row = blk * 8 + r8;
            const int c = slot ^ ((row >> 1) & 7);
            const int gm = min(m0 + row, p.M - 1);           // rows past M: any valid address (never stored)
            glds16(Ap + gm * lda + k0 + c * EPC, __builtin_amdgcn_readfirstlane(base + blk * 1024));
        }
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = lw * BPW + j, row = blk * 8 + r8;
            const int c = slot ^ ((row >> 1) & 7);
            const int gn = min(n0 + row, p.N - 1);
            glds16(Bp + gn * ldb + k0 + c * EPC, __builtin_amdgcn_readfirstlane(base + BM * BKB + blk * 1024));
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    if (loader) {
        if (ntot > 0) issue(0, 0);
        if (ntot > 1) issue(1, 1);
        int cur = 0;
        for (int kt = 0; kt < ntot; ++kt) {
            if (kt + 1 < ntot) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();             // tile kt is in LDS; every consumer has finished tile kt-1
            if (kt + 2 < ntot) issue(kt + 2, cur == 0 ? 2 : cur - 1);
            cur = (cur == 2) ? 0 : cur + 1;
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
    int cur = 0;
    for (int kt = 0; kt < ntot; ++kt) {
        __builtin_amdgcn_s_barrier();
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + BM * BKB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const uint4*>(sA + nt_lds_off(wm * TM * 16 + i * 16 + fr, 4 * s + fq));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const uint4*>(sB + nt_lds_off(wn * TN * 16 + j * 16 + fr, 4 * s + fq));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
        }
        cur = (cur == 2) ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    constexpr int WR = TM * 16, WC = TN * 16;
    float* slab = reinterpret_cast<float*>(smem) + lw * (WR * WC);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + fq * 4 + r, col = j * 16 + fr;
                slab[row * WC + (col ^ (((row >> 2) & 3) << 4))] = acc[i][j][r];
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    gemm_nt_epilogue<T, WR, WC>(p, slab, lane, m0 + wm * WR, n0 + wn * WC, bz);
}


// ------------------------------------------------------------------------------------------
// conv_v3: NHWC convolution (1x1 or 3x3, stride 1/2, dilation d, TF 'SAME') as an implicit GEMM on
// the gemm_nt v4 pipeline: K walks (tap, Cin-slice); the weight side is a plain [Cout][taps*Cin]
// K-contiguous matrix, the activation side re-addresses each output pixel's row per tap and points
// out-of-image lanes at a page of zeros (LDS-DMA cannot zero-fill).  Epilogue: + folded-BN shift,
// + residual, ReLU  (deeplab_resnet/model.py bottlenecks; kaffe/tensorflow/network.py:105-188,260-270).
// ------------------------------------------------------------------------------------------
template <typename T, int BM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_v3_kernel(const cmpc_conv_args p) {
    constexpr int BN = 128;
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int BK = BKB / (int)sizeof(T);
    constexpr int WAVES_N = 2, WAVES_M = 4;
    constexpr int TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
    constexpr int STAGE = (BM + BN) * BKB;
    constexpr int APW = BM / 8 / 8, BPW = BN / 8 / 8, LPT = APW + BPW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int Ho = (p.H + p.stride - 1) / p.stride, Wo = (p.W + p.stride - 1) / p.stride;
    const int M = p.B * Ho * Wo, N = p.Cout;
    const int gx = (N + BN - 1) / BN, nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int tix = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int m0 = (tix / gx) * BM, n0 = (tix % gx) * BN;
    const int kpt = p.Cin / BK;                       // K-tiles per tap
    const int ntot = p.ksize * p.ksize * kpt;
    // TF SAME: total pad = max((out-1)*stride + (k-1)*dil + 1 - in, 0), before = total / 2
    const int padh = max((Ho - 1) * p.stride + (p.ksize - 1) * p.dil + 1 - p.H, 0) / 2;
    const int padw = max((Wo - 1) * p.stride + (p.ksize - 1) * p.dil + 1 - p.W, 0) / 2;

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int r8 = lane >> 3, slot = lane & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // this lane's output pixel for each of the wave's A pieces
    int pb[APW], py[APW], px[APW], pc[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int blk = wid * APW + j, row = blk * 8 + r8;
        const int gm = min(m0 + row, M - 1);
        pb[j] = gm / (Ho * Wo);
        const int rem = gm - pb[j] * (Ho * Wo);
        py[j] = (rem / Wo) * p.stride - padh;
        px[j] = (rem % Wo) * p.stride - padw;
        pc[j] = (slot ^ ((row >> 1) & 7)) * EPC;
    }
    const T* X = reinterpret_cast<const T*>(p.X);
    const T* Wt = reinterpret_cast<const T*>(p.Wt);
    const T* Z = reinterpret_cast<const T*>(p.zeros);
    // Issue cursor (tap, k0): the per-lane source pointers of the current tap live in registers and are recomputed
    // only when the tap changes (out-of-image taps point at the zero page and do not advance with k0).
    const T* pa[APW];
    const T* pw[BPW];
    int kadv[APW];
    int tap = 0, k0 = 0;
    auto load_tap = [&]() {
        const int dy = (tap / p.ksize) * p.dil, dx = (tap % p.ksize) * p.dil;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int yi = py[j] + dy, xi2 = px[j] + dx;
            const bool ok = yi >= 0 && yi < p.H && xi2 >= 0 && xi2 < p.W;
            pa[j] = ok ? X + ((long)(pb[j] * p.H + yi) * p.W + xi2) * p.ldx + pc[j] : Z + pc[j];
            kadv[j] = ok ? 1 : 0;
        }
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int row = (wid * BPW + j) * 8 + r8;
            const int gn = min(n0 + row, N - 1);
            pw[j] = Wt + (long)gn * p.ldw + tap * p.Cin + (slot ^ ((row >> 1) & 7)) * EPC;
        }
    };
    load_tap();
    auto issue_next = [&](int buf) {
        const uint32_t base = lds0 + buf * STAGE;
#pragma unroll
        for (int j = 0; j < APW; ++j)
            glds16(pa[j] + kadv[j] * k0, __builtin_amdgcn_readfirstlane(base + (wid * APW + j) * 1024));
#pragma unroll
        for (int j = 0; j < BPW; ++j)
            glds16(pw[j] + k0, __builtin_amdgcn_readfirstlane(base + BM * BKB + (wid * BPW + j) * 1024));
        k0 += BK;
        if (k0 == p.Cin) { k0 = 0; ++tap; if (tap < p.ksize * p.ksize) load_tap(); }
    };
    const int fr = lane & 15, fq = lane >> 4;
    auto read_frags = [&](int buf, int sidx, uint4 (&a)[TM], uint4 (&b)[TN]) {
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + BM * BKB;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[i] = *reinterpret_cast<const uint4*>(sA + nt_lds_off(wm * TM * 16 + i * 16 + fr, 4 * sidx + fq));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            b[j] = *reinterpret_cast<const uint4*>(sB + nt_lds_off(wn * TN * 16 + j * 16 + fr, 4 * sidx + fq));
    };
    auto mma_all = [&](const uint4 (&a)[TM], const uint4 (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
    };

    // same pipeline as gemm_nt_v4: three k-tiles of loads in flight, fragments of the next half k-tile read before
    // the MFMAs of the current one, last k-tile peeled
    if (ntot > 0) issue_next(0);
    if (ntot > 1) issue_next(1);
    if (ntot > 2) issue_next(2);
    if (ntot > 2) wait_vmcnt<2 * LPT>(); else if (ntot > 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    uint4 a0[TM], b0[TN], a1[TM], b1[TN];
    if (ntot > 0) read_frags(0, 0, a0, b0);
    int cur = 0;
    for (int kt = 0; kt + 1 < ntot; ++kt) {
        read_frags(cur, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma_all(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        const int nxt = (cur == 2) ? 0 : cur + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt + 2 < ntot) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < ntot) issue_next(cur);
        read_frags(nxt, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_all(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    if (ntot > 0) {
        read_frags(cur, 1, a1, b1);
        mma_all(a0, b0);
        mma_all(a1, b1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    constexpr int WR = TM * 16, WC = TN * 16;
    float* slab = reinterpret_cast<float*>(smem) + wid * (WR * WC);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + fq * 4 + r, col = j * 16 + fr;
                slab[row * WC + (col ^ (((row >> 2) & 3) << 4))] = acc[i][j][r];
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int LPR = WC / 8, RPP = 64 / LPR;      // 8 columns per lane (Cout % 8 == 0)
    T* Y = reinterpret_cast<T*>(p.Y);
    const T* Rs = reinterpret_cast<const T*>(p.res);
    const int c8 = (lane % LPR) * 8, gn = n0 + wn * WC + c8, rl = lane / LPR;
    if (gn >= N) return;
    float bv[8];
    ld8<float>(p.bias + gn, bv);                     // column terms once, not once per pass
    const bool relu = p.relu != 0;
#pragma unroll
    for (int pass = 0; pass < WR / RPP; ++pass) {    // fully unrolled: the residual loads of all passes go out together
        const int row = pass * RPP + rl;
        const int gm = m0 + wm * WR + row;
        if (gm >= M) continue;
        const float4 va = *reinterpret_cast<const float4*>(slab + row * WC + slab_col(row, c8, WC));
        const float4 vb = *reinterpret_cast<const float4*>(slab + row * WC + slab_col(row, c8 + 4, WC));
        float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        const long off = (long)gm * p.ldy + gn;
        float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (Rs) ld8<T>(Rs + off, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float x = v[e] + bv[e] + rv[e]; v[e] = relu ? fmaxf(x, 0.f) : x; }
        st8<T>(Y + off, v);
    }
}

// ------------------------------------------------------------------------------------------
// gemm_tn: out[k,n] += alpha * sum_r A[r,k] * D[r,n]
// LDS image [BR][128 elems] per operand, 32-B segments XOR-swizzled so that the transposing
// reads (ds_read_b64_tr_b16 for bf16, ds_read_b32 for f32) of 8 rows x 32 B are conflict-free.
// ------------------------------------------------------------------------------------------
template <typename T> struct TnCfg;
template <> struct TnCfg<bf16_t> { static constexpr int BR = 64; };
template <> struct TnCfg<f16_t> { static constexpr int BR = 64; };
template <> struct TnCfg<float> { static constexpr int BR = 32; };

__device__ __forceinline__ int tn_swz(int r, int byte, int rowb) {
    const int f = (r & 3) | (((r >> 3) & 1) << 2);
    return r * rowb + ((((byte >> 5) ^ f) << 5) | (byte & 31));
}

// How a workgroup delivers its tile -- always exactly ONE writer per destination element, so no atomics and a sum that does not
// depend on the order workgroups finish in:
//   TN_ACC   : accumulate into acc and return (the caller chains further products that add into the same output tile)
//   TN_RMW   : out[k, n] += alpha * acc        (the reduction is not split: this workgroup is the only writer of the tile)
//   TN_SLAB  : slab[k, n]  = alpha * acc       (split reduction: every part has its own [Kv][lds] fp32 slab, and a fold launch
//              sums the slabs of a tile in a fixed order into the output)
enum { TN_ACC = 0, TN_RMW = 1, TN_SLAB = 2 };
template <typename T>
__device__ __forceinline__ void gemm_tn_body(const cmpc_gemm_tn_args& p, const int bx, const int by, const int bz, f4 (&acc)[4][4],
                                             const int MODE, float* slab = nullptr, int lds = 0) {
    constexpr int BR = TnCfg<T>::BR;
    constexpr int ROWB = 128 * (int)sizeof(T);     // bytes per LDS row
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int CPR = ROWB / 16;                 // 16-B chunks per row
    constexpr int NCH = BR * CPR / 256;            // chunks per thread per operand
    constexpr int TILE = BR * ROWB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int ntn = (p.Nv + 127) / 128;
    const int k0 = (bx / ntn) * 128, n0 = (bx % ntn) * 128;
    const int b1 = bz % p.nb, b2 = bz / p.nb;
    const T* A = reinterpret_cast<const T*>(p.A) + p.a_off[b1] + (long)b2 * p.a_bs;
    const T* D = reinterpret_cast<const T*>(p.D) + p.d_off[b1] + (long)b2 * p.d_bs;
    float* out = p.out + p.o_off[b1] + (long)b2 * p.o_bs;

    // rows of this split
    const int per = (p.R + p.rsplit - 1) / p.rsplit;
    const int rbeg = by * per;
    const int rend = min(p.R, rbeg + per);
    const int nt = (rend > rbeg) ? (rend - rbeg + BR - 1) / BR : 0;

    uint4 ra[NCH], rd[NCH];
    auto gload = [&](int t) {
        const int r0 = rbeg + t * BR;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i, row = id / CPR, c = id % CPR, gr = r0 + row;
            const int ka = k0 + c * EPC, nd = n0 + c * EPC;
            const bool rv = gr < rend;
            // weight gradient of a k x k (atrous) convolution, one tap: rows are pixels (b, y, x) of [*, conv_H, conv_W] maps and the A row of
            // output pixel gr is the INPUT pixel (y + conv_dy, x + conv_dx) -- zero outside the image (TF 'SAME')
            long ga = gr; bool av = rv;
            if (p.conv_W > 0 && rv) {
                const int xx = gr % p.conv_W + p.conv_dx, yy = (gr / p.conv_W) % p.conv_H + p.conv_dy;
                av = yy >= 0 && yy < p.conv_H && xx >= 0 && xx < p.conv_W;
                ga = (long)gr + (long)p.conv_dy * p.conv_W + p.conv_dx;
            }
            ra[i] = (av && ka < p.Ka) ? *reinterpret_cast<const uint4*>(A + ga * p.lda + ka) : uint4{0u, 0u, 0u, 0u};
            rd[i] = (rv && nd < p.Nd) ? *reinterpret_cast<const uint4*>(D + (long)gr * p.ldd + nd) : uint4{0u, 0u, 0u, 0u};
        }
    };
    auto lstore = [&](int buf) {
        char* sA = smem + buf * 2 * TILE;
        char* sD = sA + TILE;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i, row = id / CPR, c = id % CPR;
            *reinterpret_cast<uint4*>(sA + tn_swz(row, c * 16, ROWB)) = ra[i];
            *reinterpret_cast<uint4*>(sD + tn_swz(row, c * 16, ROWB)) = rd[i];
        }
    };

    if (nt > 0) { gload(0); lstore(0); }
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) gload(t + 1);
        const char* sA = smem + cur * 2 * TILE;
        const char* sD = sA + TILE;
        if constexpr (sizeof(T) == 2) {
            // lane li = lane&15 of 16-lane group q supplies the address of block row (li>>2),
            // columns 4*(li&3)..+3, and receives column li of the block's 4 rows.
            const int q4 = fr >> 2, p4 = fr & 3;
#pragma unroll
            for (int rs = 0; rs < BR / 32; ++rs) {
                uint4 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = wm * 64 + i * 16 + 4 * p4;
                    const int r_lo = rs * 32 + 8 * fq + q4;
                    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s4*)(sA + tn_swz(r_lo, col * 2, ROWB)));
                    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s4*)(sA + tn_swz(r_lo + 4, col * 2, ROWB)));
                    const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    a[i] = __builtin_bit_cast(uint4, v);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = wn * 64 + j * 16 + 4 * p4;
                    const int r_lo = rs * 32 + 8 * fq + q4;
                    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s4*)(sD + tn_swz(r_lo, col * 2, ROWB)));
                    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s4*)(sD + tn_swz(r_lo + 4, col * 2, ROWB)));
                    const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    b[j] = __builtin_bit_cast(uint4, v);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
            }
        } else {
#pragma unroll 2
            for (int rs = 0; rs < BR / 4; ++rs) {
                float a[4], b[4];
                const int r = rs * 4 + fq;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = *reinterpret_cast<const float*>(sA + tn_swz(r, (wm * 64 + i * 16 + fr) * 4, ROWB));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    b[j] = *reinterpret_cast<const float*>(sD + tn_swz(r, (wn * 64 + j * 16 + fr) * 4, ROWB));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (t + 1 < nt) lstore(cur ^ 1);
        __syncthreads();
    }
    if (MODE == TN_ACC) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + wm * 64 + i * 16 + fq * 4 + r;
                const int n = n0 + wn * 64 + j * 16 + fr;
                if (k < p.Kv && n < p.Nv) {
                    if (MODE == TN_SLAB) slab[(long)k * lds + n] = acc[i][j][r] * p.alpha;
                    else out[(long)k * p.ldo + n] += acc[i][j][r] * p.alpha;
                }
            }
}
__device__ __forceinline__ void tn_zero(f4 (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
}


template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const cmpc_gemm_tn_args p, float* slabs, long slab_size, int lds) {
    f4 acc[4][4];
    tn_zero(acc);
    if (slabs) gemm_tn_body<T>(p, blockIdx.x, blockIdx.y, blockIdx.z, acc, TN_SLAB, slabs + ((long)blockIdx.z * p.rsplit + blockIdx.y) * slab_size, lds);
    else gemm_tn_body<T>(p, blockIdx.x, blockIdx.y, blockIdx.z, acc, TN_RMW);
}

// Grouped form: ALL deferred weight-gradient products of a backward pass in ONE persistent launch.  The
// descriptor table lives in device memory (written by tn_desc_upload_kernel, a few descriptors per launch through
// the kernel-argument segment); workgroup slot s walks the item list s, s+grid, ... (items = (product, output tile,
// reduction split, batch) sorted by decreasing length, so the tail of the launch is made of short items).  With
// every product in flight none needs a deep split of its reduction (few slabs to fold), and there is one launch tail
// instead of one per group.
// The grid is at most two workgroups per CU, so the launch does not sit in the workgroup dispatcher for its whole
// duration (a 4600-workgroup grid starves the small kernels of every other stream: 14 -> 51 ms per step).
struct TnGroupDesc {
    cmpc_gemm_tn_args a;
    int item_begin, tiles;
    int chain;          // > 0: the next `chain` descriptors add into the SAME output tiles (same shape, alpha): one workgroup accumulates them all
    int lds;            // slab row stride (slab mode)
    float* slabs;       // != NULL: split reduction, part (bz, by) writes slabs + (bz * rsplit + by) * slab_size; a fold launch follows
    long slab_size;
};
#define TN_UPLOAD 10
struct TnUploadArgs {
    int n, base;
    TnGroupDesc d[TN_UPLOAD];
};
static_assert(sizeof(TnUploadArgs) <= 4000, "descriptor batches travel through the kernel-argument segment (4 KB)");
__global__ void tn_desc_upload_kernel(const TnUploadArgs ua, TnGroupDesc* table) {
    constexpr int W = (int)(sizeof(TnGroupDesc) / sizeof(int));
    const int* src = reinterpret_cast<const int*>(&ua.d[0]);
    int* dst = reinterpret_cast<int*>(table + ua.base);
    for (int i = threadIdx.x; i < ua.n * W; i += blockDim.x) dst[i] = src[i];
}

__global__ __launch_bounds__(256, 2) void gemm_tn_grouped_kernel(const TnGroupDesc* __restrict__ table, int ndesc, int total, int* __restrict__ next_item) {
    // Items are dealt dynamically, longest first: a workgroup that finishes takes the next one (list scheduling).  The static deal s, s + grid, ...
    // gave the slots that drew a long item in the first round another one in the second (620 tiles of 200 steps on 512 slots: 400 steps on
    // slots 0..107 against an average of 242).  Which workgroup computes an item does not matter for the result: one writer per output / slab.
    __shared__ int s_item;
    for (int w_static = blockIdx.x;; w_static += gridDim.x) {
        if (next_item && threadIdx.x == 0) s_item = atomicAdd(next_item, 1);
        __syncthreads();
        const int w = next_item ? s_item : w_static;          // next_item == NULL: the static deal s, s + grid, ... (A/B switch CMPC_TN_STATIC)
        __syncthreads();
        if (w >= total) break;
        int lo = 0, hi = ndesc;                       // uniform binary search: scalar loads
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (table[mid].item_begin <= w) lo = mid; else hi = mid; }
        const TnGroupDesc& d = table[lo];
        const int local = w - d.item_begin;
        const int tiles = d.tiles, rs = d.a.rsplit;
        const int bx = local % tiles, rest = local / tiles, by = rest % rs, bz = rest / rs;
        f4 acc[4][4];
        tn_zero(acc);
        // a part of a split reduction writes its own slab; products chained behind d add into the same output tile: they are
        // accumulated in a fixed order and stored once (by the head, last).  ONE call site per dtype keeps the register budget.
        float* slab = d.slabs ? d.slabs + ((long)bz * rs + by) * d.slab_size : nullptr;
        for (int q = d.chain; q >= 0; --q) {
            const TnGroupDesc& c = table[lo + q];
            const int mode = slab ? TN_SLAB : (q > 0 ? TN_ACC : TN_RMW);
            if (c.a.dtype == DT_F32) gemm_tn_body<float>(c.a, bx, by, bz, acc, mode, slab, d.lds);
            else if (c.a.dtype == DT_BF16) gemm_tn_body<bf16_t>(c.a, bx, by, bz, acc, mode, slab, d.lds);
            else gemm_tn_body<f16_t>(c.a, bx, by, bz, acc, mode, slab, d.lds);
            __syncthreads();                          // the next product / item reuses the LDS stages
        }
    }
}


// ------------------------------------------------------------------------------------------
// gemm_nt for M <= 16 rows (the language side: [B, .] vectors against whole weight matrices).
// Weight-streaming: one wave per NC output columns, K split over the 64 lanes (float4 loads of
// the K-contiguous weight rows), the few A rows re-read from L1/L2; wave-shuffle reduction.
// ------------------------------------------------------------------------------------------
template <int MMAX>
__global__ __launch_bounds__(256) void gemm_nt_skinny_f32_kernel(const cmpc_gemm_nt_args p) {
    // one workgroup = NC output columns; its 4 waves split K (interleaved 1-KiB slices), so a
    // K = 1024 product is ONE round trip of 12 independent 16-B loads per lane.
    constexpr int NC = 4;
    __shared__ float red[4][MMAX * NC];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n0 = blockIdx.x * NC;
    float acc[MMAX][NC];
#pragma unroll
    for (int m = 0; m < MMAX; ++m)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[m][c] = 0.f;
    for (int s = 0; s < p.nseg; ++s) {
        const float* A = reinterpret_cast<const float*>(p.A[s]);
        const float* Bt = reinterpret_cast<const float*>(p.Bt[s]);
        const long lda = p.lda[s], ldb = p.ldb[s];
        const int K = p.K[s];
        for (int k = (wid * 64 + lane) * 4; k < K; k += 1024) {
            // Unconditional loads from clamped rows (results of rows >= M / columns >= N are never stored): with a
            // branch around each load hipcc waited for every one of them in turn -- 8 dependent L2 round trips, about
            // 8 of the 14 us this latency-bound kernel took.
            float4 w[NC], a[MMAX];
#pragma unroll
            for (int c = 0; c < NC; ++c)
                w[c] = *reinterpret_cast<const float4*>(Bt + (long)min(n0 + c, p.N - 1) * ldb + k);
#pragma unroll
            for (int m = 0; m < MMAX; ++m)
                a[m] = *reinterpret_cast<const float4*>(A + (long)min(m, p.M - 1) * lda + k);
#pragma unroll
            for (int m = 0; m < MMAX; ++m)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[m][c] += a[m].x * w[c].x + a[m].y * w[c].y + a[m].z * w[c].z + a[m].w * w[c].w;
        }
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
    const int m = t / NC, c = t % NC, gn = n0 + c;
    if (m >= p.M || gn >= p.N) return;
    // the optional terms are loaded unconditionally (absent ones from a dummy readable address, then dropped by a
    // select), so that the loads are ONE round trip instead of up to four dependent ones
    float* C = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + gn;
    const int rps = p.rows_per_sample > 0 ? p.rows_per_sample : 1;
    const float* pb0 = p.bias ? p.bias + gn : C;
    const float* pb1 = p.sbias ? p.sbias + (m / rps) * (long)p.ld_sbias + gn : C;
    const float* pb2 = p.pbias ? p.pbias + (m % rps) * (long)p.ld_pbias + gn : C;
    const float b0 = *pb0, b1 = *pb1, b2 = *pb2, old = *C;
    float x = 0.f;
    if (gn < p.n_valid) {
        x = (red[0][t] + red[1][t] + red[2][t] + red[3][t]) * p.alpha;
        x += (p.bias ? b0 : 0.f) + (p.sbias ? b1 : 0.f) + (p.pbias ? b2 : 0.f);
        x = act_apply(x, p.act);
    }
    if (p.accumulate) x += old;
    *C = x;
}

}  // namespace

// 16-bit operands (T = bf16_t or f16_t): the LDS-DMA MFMA pipelines.  Which pipeline wins where was measured with
// scripts/gemm_ksweep.py: 256 x 256 tiles (v5) for N >= 1024 (1.1-1.25x), fragment double buffering (v4) for short and
// medium K, producer / consumer wave specialisation (v3) for long K; 256-row tiles as soon as they fill 3/4 of the CUs.
template <typename T>
static int launch_nt16(const cmpc_gemm_nt_args* a, hipStream_t st) {
    const int gn = (a->N + 127) / 128;
    int ktot = 0;
    bool off32 = true;          // every operand addressable with v5's 32-bit byte offsets
    for (int s = 0; s < a->nseg; ++s) {
        ktot += a->K[s];
        off32 = off32 && ((long)a->M * a->lda[s] * 2 < (1L << 32)) && ((long)a->N * a->ldb[s] * 2 < (1L << 32));
    }
    if (a->N >= 1024 && a->M >= 2048 && off32) {
        static const bool attr = ((void)hipFuncSetAttribute((const void*)gemm_nt_v5_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 512 * 64), true);
        (void)attr;
        dim3 grid((unsigned)(((a->M + 255) / 256) * ((a->N + 255) / 256)), 1, a->batch);
        hipLaunchKernelGGL((gemm_nt_v5_kernel<T>), grid, dim3(512), 4 * 512 * 64, st, *a);
        return cmpc_check_launch("gemm_nt(v5)");
    }
    const bool big = (long)((a->M + 255) / 256) * gn * a->batch >= 192;       // one round of 256-row tiles beats two of 128-row ones
    const bool v4 = ktot < 2048 || (ktot < 4096 && a->N >= 1024);
    if (big) {
        constexpr int LDS = 3 * (256 + 128) * BKB;
        static const bool attr = ((void)hipFuncSetAttribute((const void*)gemm_nt_v4_kernel<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS),
                                  (void)hipFuncSetAttribute((const void*)gemm_nt_v3_kernel<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
        (void)attr;
        dim3 grid(((a->M + 255) / 256) * gn, 1, a->batch);
        if (v4) hipLaunchKernelGGL((gemm_nt_v4_kernel<T, 256>), grid, dim3(512), LDS, st, *a);
        else hipLaunchKernelGGL((gemm_nt_v3_kernel<T, 256>), grid, dim3(512), LDS, st, *a);
    } else {
        constexpr int LDS = 3 * (128 + 128) * BKB;
        static const bool attr = ((void)hipFuncSetAttribute((const void*)gemm_nt_v4_kernel<T, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS),
                                  (void)hipFuncSetAttribute((const void*)gemm_nt_v3_kernel<T, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
        (void)attr;
        dim3 grid(((a->M + 127) / 128) * gn, 1, a->batch);
        if (v4) hipLaunchKernelGGL((gemm_nt_v4_kernel<T, 128>), grid, dim3(512), LDS, st, *a);
        else hipLaunchKernelGGL((gemm_nt_v3_kernel<T, 128>), grid, dim3(512), LDS, st, *a);
    }
    return cmpc_check_launch("gemm_nt(v4/v3)");
}

template <typename T>
static int launch_nt_small(const cmpc_gemm_nt_args* a, hipStream_t st) {
    const int bn = (a->N % 128 == 0 || a->N > 64) ? 128 : 64;
    const long tiles128 = (long)((a->N + bn - 1) / bn) * ((a->M + 127) / 128) * a->batch;
    if (sizeof(T) == 4 && tiles128 < 192) {
        // fp32 MFMA runs at 1/16 of the 16-bit rate: small products need many small tiles to use the chip
        dim3 grid(((a->N + 63) / 64) * ((a->M + 63) / 64), 1, a->batch);
        hipLaunchKernelGGL((gemm_nt_kernel<T, 64, 64>), grid, dim3(256), 2 * (64 + 64) * BKB, st, *a);
        return cmpc_check_launch("gemm_nt(64x64)");
    }
    dim3 grid(((a->N + bn - 1) / bn) * ((a->M + 127) / 128), 1, a->batch);
    const size_t lds = 2 * (128 + bn) * BKB;
    if (bn == 128) hipLaunchKernelGGL((gemm_nt_kernel<T, 128, 128>), grid, dim3(256), lds, st, *a);
    else hipLaunchKernelGGL((gemm_nt_kernel<T, 128, 64>), grid, dim3(256), lds, st, *a);
    return cmpc_check_launch("gemm_nt");
}

#ifdef CMPC_V5_TRACE
extern "C" int cmpc_debug_v5_trace(unsigned long long* out, int n_words) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_v5_trace), sizeof(unsigned long long) * n_words) == hipSuccess ? 0 : -1;
}
#endif

static int nt_validate(const cmpc_gemm_nt_args* a) {
    if (!a || a->nseg < 1 || a->nseg > 3 || a->M <= 0 || a->N <= 0 || a->batch <= 0) { cmpc_set_error("gemm_nt: bad args"); return CMPC_EINVAL; }
    if (a->dtype != DT_F32 && a->dtype != DT_BF16 && a->dtype != DT_F16) { cmpc_set_error("gemm_nt: bad dtype"); return CMPC_EINVAL; }
    const int esz = a->dtype == DT_F32 ? 4 : 2;
    const int bk = BKB / esz;
    for (int s = 0; s < a->nseg; ++s) {
        if (a->K[s] <= 0 || a->K[s] % bk || (a->lda[s] * esz) % 16 || (a->ldb[s] * esz) % 16 || !a->A[s] || !a->Bt[s]) {
            cmpc_set_error("gemm_nt: segment %d: K=%d must be a multiple of %d and rows 16-B aligned", s, a->K[s], bk);
            return CMPC_EINVAL;
        }
    }
    if (a->N % 4 || a->ldc % 4 || !a->C) { cmpc_set_error("gemm_nt: N/ldc must be multiples of 4"); return CMPC_EINVAL; }
    return CMPC_OK;
}

// does this product dispatch to the 256 x 128 fragment-double-buffered pipeline (the one a paired launch exists for)?
static bool nt_is_v4_256(const cmpc_gemm_nt_args* a) {
    if (a->dtype == DT_F32 || !(a->N >= 128 && a->M >= 512) || a->batch != 1) return false;
    int ktot = 0; bool off32 = true;
    for (int s = 0; s < a->nseg; ++s) { ktot += a->K[s]; off32 = off32 && ((long)a->M * a->lda[s] * 2 < (1L << 32)) && ((long)a->N * a->ldb[s] * 2 < (1L << 32)); }
    if (a->N >= 1024 && a->M >= 2048 && off32) return false;                                  // v5
    const bool big = (long)((a->M + 255) / 256) * ((a->N + 127) / 128) * a->batch >= 192;
    return big && (ktot < 2048 || (ktot < 4096 && a->N >= 1024));
}

extern "C" int cmpc_gemm_nt_pair(const cmpc_gemm_nt_args* a, const cmpc_gemm_nt_args* b, void* stream) {
    int rc = nt_validate(a); if (rc != CMPC_OK) return rc;
    rc = nt_validate(b); if (rc != CMPC_OK) return rc;
    const int na = a ? ((a->M + 255) / 256) * ((a->N + 127) / 128) : 0;
    if (a->dtype == b->dtype && nt_is_v4_256(a) && nt_is_v4_256(b) && na % 8 == 0) {
        const int nb = ((b->M + 255) / 256) * ((b->N + 127) / 128);
        constexpr int LDS = 3 * (256 + 128) * BKB;
        NtPair pp; pp.a = *a; pp.b = *b; pp.na = na;
        if (a->dtype == DT_BF16) {
            static const bool attr = ((void)hipFuncSetAttribute((const void*)gemm_nt_v4_pair_kernel<bf16_t, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
            (void)attr;
            hipLaunchKernelGGL((gemm_nt_v4_pair_kernel<bf16_t, 256>), dim3(na + nb), dim3(512), LDS, (hipStream_t)stream, pp);
        } else {
            static const bool attr = ((void)hipFuncSetAttribute((const void*)gemm_nt_v4_pair_kernel<f16_t, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
            (void)attr;
            hipLaunchKernelGGL((gemm_nt_v4_pair_kernel<f16_t, 256>), dim3(na + nb), dim3(512), LDS, (hipStream_t)stream, pp);
        }
        return cmpc_check_launch("gemm_nt(pair)");
    }
    rc = cmpc_gemm_nt(a, stream);              // any other combination: two launches
    return rc != CMPC_OK ? rc : cmpc_gemm_nt(b, stream);
}

extern "C" int cmpc_gemm_nt(const cmpc_gemm_nt_args* a, void* stream) {
    if (!a || a->nseg < 1 || a->nseg > 3 || a->M <= 0 || a->N <= 0 || a->batch <= 0) {
        cmpc_set_error("gemm_nt: bad args"); return CMPC_EINVAL;
    }
    if (a->dtype != DT_F32 && a->dtype != DT_BF16 && a->dtype != DT_F16) { cmpc_set_error("gemm_nt: bad dtype"); return CMPC_EINVAL; }
    const int esz = a->dtype == DT_F32 ? 4 : 2;
    const int bk = BKB / esz;
    for (int s = 0; s < a->nseg; ++s) {
        if (a->K[s] <= 0 || a->K[s] % bk || (a->lda[s] * esz) % 16 || (a->ldb[s] * esz) % 16 || !a->A[s] || !a->Bt[s]) {
            cmpc_set_error("gemm_nt: segment %d: K=%d must be a multiple of %d and rows 16-B aligned", s, a->K[s], bk);
            return CMPC_EINVAL;
        }
    }
    if (a->N % 4 || a->ldc % 4 || !a->C) { cmpc_set_error("gemm_nt: N/ldc must be multiples of 4"); return CMPC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (a->dtype == DT_F32) {
        if (a->M <= 16 && a->batch == 1) {
            dim3 grid((a->N + 3) / 4);
            if (a->M <= 8) hipLaunchKernelGGL((gemm_nt_skinny_f32_kernel<8>), grid, dim3(256), 0, st, *a);
            else hipLaunchKernelGGL((gemm_nt_skinny_f32_kernel<16>), grid, dim3(256), 0, st, *a);
            return cmpc_check_launch("gemm_nt(skinny)");
        }
        return launch_nt_small<float>(a, st);
    }
    const bool pipe = a->N >= 128 && a->M >= 512;
    if (a->dtype == DT_BF16) return pipe ? launch_nt16<bf16_t>(a, st) : launch_nt_small<bf16_t>(a, st);
    return pipe ? launch_nt16<f16_t>(a, st) : launch_nt_small<f16_t>(a, st);
}


template <typename T>
static int launch_conv(const cmpc_conv_args* a, hipStream_t st) {
    const int Ho = (a->H + a->stride - 1) / a->stride, Wo = (a->W + a->stride - 1) / a->stride;
    const long M = (long)a->B * Ho * Wo;
    const int gn = (a->Cout + 127) / 128;
    if (((M + 255) / 256) * gn >= 192) {       // as in cmpc_gemm_nt
        constexpr int LDS = 3 * (256 + 128) * BKB;
        static const bool attr = ((void)hipFuncSetAttribute((const void*)conv_v3_kernel<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
        (void)attr;
        hipLaunchKernelGGL((conv_v3_kernel<T, 256>), dim3((unsigned)(((M + 255) / 256) * gn)), dim3(512), LDS, st, *a);
    } else {
        constexpr int LDS = 3 * (128 + 128) * BKB;
        static const bool attr = ((void)hipFuncSetAttribute((const void*)conv_v3_kernel<T, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
        (void)attr;
        hipLaunchKernelGGL((conv_v3_kernel<T, 128>), dim3((unsigned)(((M + 127) / 128) * gn)), dim3(512), LDS, st, *a);
    }
    return cmpc_check_launch("conv_nhwc");
}

extern "C" int cmpc_conv_nhwc(const cmpc_conv_args* a, void* stream) {
    if (!a || !a->X || !a->Wt || !a->Y || !a->bias || !a->zeros || a->B <= 0 || a->H <= 0 || a->W <= 0) { cmpc_set_error("conv_nhwc: bad args"); return CMPC_EINVAL; }
    const int esz = a->dtype == DT_F32 ? 4 : 2, bk = BKB / esz;
    if ((a->ksize != 1 && a->ksize != 3) || (a->stride != 1 && a->stride != 2) || a->dil < 1 || a->Cin % bk || a->Cout % 8 ||
        (a->ldx * esz) % 16 || (a->ldw * esz) % 16 || a->ldy % 8 || (a->res && a->ldy % 8)) {
        cmpc_set_error("conv_nhwc: need k in {1,3}, stride in {1,2}, Cin %% %d == 0, Cout %% 8 == 0, ldy %% 8 == 0, 16-B aligned rows", bk); return CMPC_EINVAL;
    }
    if (a->dtype == DT_F32) return launch_conv<float>(a, (hipStream_t)stream);
    if (a->dtype == DT_BF16) return launch_conv<bf16_t>(a, (hipStream_t)stream);
    if (a->dtype == DT_F16) return launch_conv<f16_t>(a, (hipStream_t)stream);
    cmpc_set_error("conv_nhwc: bad dtype"); return CMPC_EINVAL;
}


static int tn_validate(const cmpc_gemm_tn_args* a) {
    if (!a || a->R < 0 || a->Kv <= 0 || a->Nv <= 0 || a->nb < 1 || a->nb > 8 || a->nb2 < 1) {
        cmpc_set_error("gemm_tn: bad args"); return CMPC_EINVAL;
    }
    if (a->dtype != DT_F32 && a->dtype != DT_BF16 && a->dtype != DT_F16) { cmpc_set_error("gemm_tn: bad dtype"); return CMPC_EINVAL; }
    if (a->R == 0) return CMPC_OK;
    if (!a->A || !a->D || !a->out) { cmpc_set_error("gemm_tn: null operand"); return CMPC_EINVAL; }
    const int esz = a->dtype == DT_F32 ? 4 : 2;
    if ((a->lda * esz) % 16 || (a->ldd * esz) % 16 || (a->Ka * esz) % 16 || (a->Nd * esz) % 16) {
        cmpc_set_error("gemm_tn: rows must be 16-B aligned"); return CMPC_EINVAL;
    }
    for (int i = 0; i < a->nb; ++i)
        if ((a->a_off[i] * esz) % 16 || (a->d_off[i] * esz) % 16) { cmpc_set_error("gemm_tn: offsets must be 16-B aligned"); return CMPC_EINVAL; }
    if ((a->a_bs * esz) % 16 || (a->d_bs * esz) % 16) { cmpc_set_error("gemm_tn: batch strides must be 16-B aligned"); return CMPC_EINVAL; }
    return CMPC_OK;
}

// out region of batch entry (b1, b2) += sum over the rsplit slabs of that entry, rows k < Kv, columns n < Nv
static int tn_fold(const cmpc_gemm_tn_args& a, int rsplit, float* slabs, long slab_size, int lds, hipStream_t st) {
    // uniform outer stride (nb == 1): one fold over all nb2 entries; otherwise one per inner-batch offset
    for (int b1 = 0; b1 < a.nb; ++b1) {
        for (int b2 = 0; b2 < a.nb2; b2 += (a.nb == 1 ? a.nb2 : 1)) {
            const int nouter = a.nb == 1 ? a.nb2 : 1;
            const long bz = b1 + (long)a.nb * b2;
            const int rc = cmpc_reduce_parts_f32(slabs + bz * rsplit * slab_size, slab_size, nouter, rsplit, a.Kv, lds, a.Nv,
                                                 a.out + a.o_off[b1] + (long)b2 * a.o_bs, a.o_bs, a.ldo, 1, st);
            if (rc != CMPC_OK) return rc;
        }
    }
    return CMPC_OK;
}
#define CK_FOLD(x) do { const int rc_ = (x); if (rc_ != CMPC_OK) return rc_; } while (0)

// table_dev / shadow: optional persistent device table + host copy of what it holds.  With static shapes and a static workspace the
// descriptor table of a step is byte-identical to the previous step's: it is uploaded once and reused (no upload launches).
int cmpc_gemm_tn_grouped_cached(const void* args_, int n, void* const* tables_dev, int nslots, int* victim, size_t table_bytes,
                                std::vector<char>* shadows, hipStream_t st) {
    const cmpc_gemm_tn_args* args = (const cmpc_gemm_tn_args*)args_;
    if (n < 0 || (n > 0 && !args)) { cmpc_set_error("gemm_tn_grouped: bad args"); return CMPC_EINVAL; }
    // 1. expand: an outer batch that accumulates into ONE output (o_bs == 0) becomes a chain of single products
    std::vector<cmpc_gemm_tn_args> ex;
    for (int i = 0; i < n; ++i) {
        const int rc = tn_validate(&args[i]);
        if (rc != CMPC_OK) return rc;
        if (args[i].R == 0) continue;
        if (args[i].nb2 > 1 && args[i].o_bs == 0) {
            const int esz = args[i].dtype == DT_F32 ? 4 : 2;
            for (int b = 0; b < args[i].nb2; ++b) {
                cmpc_gemm_tn_args a = args[i];
                a.A = (const char*)a.A + (size_t)b * a.a_bs * esz; a.D = (const char*)a.D + (size_t)b * a.d_bs * esz;
                a.nb2 = 1; a.a_bs = a.d_bs = 0;
                ex.push_back(a);
            }
        } else ex.push_back(args[i]);
    }
    const int m = (int)ex.size();
    if (m == 0) return CMPC_OK;
    // 2. chains: products that add into the same output region (same tiles, same scale) are accumulated by ONE workgroup per tile,
    //    in list order, and stored once -- every output element has exactly one writer, so no atomics and a run-to-run identical sum
    auto same_out = [](const cmpc_gemm_tn_args& x, const cmpc_gemm_tn_args& y) {
        if (x.out != y.out || x.ldo != y.ldo || x.Kv != y.Kv || x.Nv != y.Nv || x.nb != y.nb || x.nb2 != y.nb2 || x.o_bs != y.o_bs || x.alpha != y.alpha) return false;
        for (int b = 0; b < x.nb; ++b) if (x.o_off[b] != y.o_off[b]) return false;
        return true;
    };
    std::vector<std::vector<int>> chains;
    for (int i = 0; i < m; ++i) {
        bool placed = false;
        for (auto& c : chains) if (same_out(ex[c[0]], ex[i])) { c.push_back(i); placed = true; break; }
        if (!placed) chains.push_back({i});
    }
    const bool cached = tables_dev && shadows && victim && nslots > 0 && (size_t)m * sizeof(TnGroupDesc) <= table_bytes;
    // 3. split the long reductions only as far as needed to fill the persistent grid twice (as many parts per product as the
    //    atomic version used); a chain whose parts number more than one writes per-part slabs that a fold launch sums in a fixed order
    const int slots = 512;
    long tiles_tot = 0;
    for (int i = 0; i < m; ++i) tiles_tot += (long)((ex[i].Kv + 127) / 128) * ((ex[i].Nv + 127) / 128) * ex[i].nb * ex[i].nb2;
    const int want = (int)((2 * slots + tiles_tot - 1) / tiles_tot);
    std::vector<int> rsplit(m, 1);
    std::vector<long> steps(m, 0);
    for (int i = 0; i < m; ++i) {
        const int br = ex[i].dtype == DT_F32 ? TnCfg<float>::BR : TnCfg<bf16_t>::BR;
        rsplit[i] = std::max(1, std::min(want, std::max(1, ex[i].R / (16 * br))));
        steps[i] = ((ex[i].R + rsplit[i] - 1) / rsplit[i] + br - 1) / br;
    }
    // 3b. balance: the kernel deals the items (sorted by decreasing length) to whichever of its `slots` persistent workgroups is free first,
    //     so the launch lasts as long as list scheduling makes it.  A level bucket of configs 1-3 is ~700 tiles of 200 steps on 512 slots: two
    //     rounds, 400 steps, against an average of 280.  Splitting EVERY product costs a slab per tile and part (measured slower); here
    //     the host simulates the schedule and doubles the split of one product at a time -- the one whose doubling shortens the launch most,
    //     each part of a split product priced at SLAB_COST extra steps (slab traffic + its share of the fold) -- until nothing gains 2 % any
    //     more.  This also covers config 4's outliers (131 072-row decoder reductions beside 32 768-row ones).
    if (!getenv("CMPC_TN_NO_BALANCE")) {
        constexpr long SLAB_COST = 6;
        auto n_tiles = [&](int i) { return (long)((ex[i].Kv + 127) / 128) * ((ex[i].Nv + 127) / 128) * ex[i].nb * ex[i].nb2; };
        auto brows = [&](int i) { return ex[i].dtype == DT_F32 ? TnCfg<float>::BR : TnCfg<bf16_t>::BR; };
        auto steps_of = [&](int i, int r) { return (long)(((ex[i].R + r - 1) / r + brows(i) - 1) / brows(i)); };
        // chains (several products accumulated by one workgroup per tile) are dealt as one item of the summed length and never split here
        std::vector<int> chain_of(m, -1);          // members of SHORT chains stay unsplit (they are walked by one workgroup)
        for (size_t c = 0; c < chains.size(); ++c) {
            long len = 0; for (int i : chains[c]) len += steps_of(i, rsplit[i]);
            if (chains[c].size() > 1 && len <= 64) for (int i : chains[c]) chain_of[i] = (int)c;
        }
        std::vector<std::pair<long, long>> items;       // (length, count)
        auto makespan = [&](const std::vector<int>& rs) {
            items.clear();
            for (size_t c = 0; c < chains.size(); ++c) {
                const auto& ch = chains[c];
                long len = 0, parts = 0;
                for (int i : ch) { len += steps_of(i, rs[i]); parts += rs[i]; }
                // step 4 below: one workgroup per tile walks a short unsplit chain; every other product of a chain is a slab item of its own
                if (ch.size() > 1 && parts == (long)ch.size() && len <= 64) { items.push_back({len, n_tiles(ch[0])}); continue; }
                for (int i : ch) items.push_back({steps_of(i, rs[i]) + ((ch.size() > 1 || rs[i] > 1) ? SLAB_COST : 0), n_tiles(i) * rs[i]});
            }
            std::stable_sort(items.begin(), items.end(), [](const std::pair<long, long>& x, const std::pair<long, long>& y) { return x.first > y.first; });
            // list scheduling as the kernel does it: the next item goes to the workgroup that is free first
            std::priority_queue<long, std::vector<long>, std::greater<long>> free_at;
            for (int k = 0; k < slots; ++k) free_at.push(0);
            long last = 0;
            for (const auto& it : items) for (long k = 0; k < it.second; ++k) { const long t = free_at.top() + it.first; free_at.pop(); free_at.push(t); last = std::max(last, t); }
            return last;
        };
        // outliers first (no single doubling shortens a launch that holds twenty 2048-step products): reductions longer than half a slot's
        // share of the launch, and than 256 steps, are cut to that length
        {
            long total_steps = 0;
            for (int i = 0; i < m; ++i) total_steps += n_tiles(i) * rsplit[i] * steps_of(i, rsplit[i]);
            const long cap = std::max<long>(256, total_steps / slots / 2);
            for (int i = 0; i < m; ++i) {
                if (chain_of[i] >= 0 || steps_of(i, rsplit[i]) <= cap) continue;
                const long full = (ex[i].R + brows(i) - 1) / brows(i);
                const int r = (int)std::min<long>((full + cap - 1) / cap, std::max(1, ex[i].R / (16 * brows(i))));
                if (r > rsplit[i]) rsplit[i] = r;
            }
        }
        long best = makespan(rsplit);
        for (int iter = 0; iter < 64; ++iter) {
            int pick = -1; long pick_ms = best;
            for (int i = 0; i < m; ++i) {
                if (chain_of[i] >= 0) continue;
                const int r2 = rsplit[i] * 2;
                if (r2 > std::max(1, ex[i].R / (4 * brows(i)))) continue;          // parts of at least 4 row blocks
                std::vector<int> t = rsplit; t[i] = r2;
                const long ms = makespan(t);
                if (ms < pick_ms) { pick_ms = ms; pick = i; }
            }
            if (pick < 0 || pick_ms * 100 > best * 98) break;
            rsplit[pick] *= 2; best = pick_ms;
        }
        for (int i = 0; i < m; ++i) steps[i] = steps_of(i, rsplit[i]);
        if (getenv("CMPC_TN_PLAN_DUMP")) {
            long tot = 0; for (int i = 0; i < m; ++i) tot += n_tiles(i) * rsplit[i] * steps[i];
            fprintf(stderr, "[tn plan] %d products, %ld steps, ideal %ld per slot, most loaded slot %ld:", m, tot, tot / slots, best);
            for (int i = 0; i < m; ++i) fprintf(stderr, " %ldx%dx%ld", n_tiles(i), rsplit[i], steps[i]);
            fprintf(stderr, "\n");
        }
    }
    struct Unit { std::vector<int> prods; bool slab; long len; };      // one table entry group: a chain stored directly, or ONE product in slab mode
    std::vector<Unit> units;
    struct Fold { int prod; float* slabs; long slab_size; int lds; };
    std::vector<Fold> folds;
    size_t slab_total = 0;
    for (const auto& c : chains) {
        int parts = 0; long len = 0;
        for (int i : c) { parts += rsplit[i]; len += steps[i]; }
        // a short chain of unsplit products (e.g. the B per-sample products of a [T, .] word-side gradient): one workgroup walks it
        const bool direct = (c.size() == 1 && rsplit[c[0]] == 1) || (parts == (int)c.size() && len <= 64);
        if (direct) { for (int i : c) rsplit[i] = 1; units.push_back(Unit{c, false, len}); continue; }
        for (int i : c) {
            const cmpc_gemm_tn_args& a = ex[i];
            const int lds = (a.Nv + 3) / 4 * 4;
            const long slab_size = (long)a.Kv * lds;
            folds.push_back(Fold{i, (float*)(uintptr_t)slab_total, slab_size, lds});        // offset for now: ONE block is carved below
            slab_total += ((size_t)slab_size * rsplit[i] * a.nb * a.nb2 * sizeof(float) + 255) / 256 * 256;
            units.push_back(Unit{{i}, true, steps[i]});
        }
    }
    // one scratch block for every slab of this launch (+ the descriptor table when the caller keeps none): cmpc_ws hands out the same
    // per-stream block on every call outside a fold collection, so separate calls would alias
    const size_t table_room = 256 + (cached ? 0 : ((size_t)m * sizeof(TnGroupDesc) + 255) / 256 * 256);     // 256: the item counter
    char* scratch = (char*)cmpc_ws(slab_total + table_room, st);
    if (!scratch) return CMPC_EHIP;
    for (Fold& f : folds) f.slabs = (float*)(scratch + table_room + (size_t)(uintptr_t)f.slabs);
    std::vector<int> idx(units.size());
    for (size_t u = 0; u < units.size(); ++u) idx[u] = (int)u;
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return units[x].len > units[y].len; });   // long items first
    std::vector<TnGroupDesc> sorted;
    sorted.reserve(m);
    int items = 0, fi = 0;
    std::vector<int> fold_of_unit(units.size(), -1);
    for (size_t u = 0; u < units.size(); ++u) if (units[u].slab) fold_of_unit[u] = fi++;
    for (int ui : idx) {
        const Unit& un = units[ui];
        const cmpc_gemm_tn_args& h = ex[un.prods[0]];
        const int tiles = ((h.Kv + 127) / 128) * ((h.Nv + 127) / 128);
        const int n_items = tiles * rsplit[un.prods[0]] * h.nb * h.nb2;
        for (size_t q = 0; q < un.prods.size(); ++q) {
            TnGroupDesc d;
            memset(&d, 0, sizeof(d));                  // padding bytes take part in the cache comparison
            d.a = ex[un.prods[q]];
            d.a.rsplit = rsplit[un.prods[q]];
            d.tiles = tiles;
            d.chain = q == 0 ? (int)un.prods.size() - 1 : 0;
            d.item_begin = q == 0 ? items : items + n_items;     // followers are never the target of the item search
            if (un.slab) { const Fold& f = folds[fold_of_unit[ui]]; d.slabs = f.slabs; d.slab_size = f.slab_size; d.lds = f.lds; }
            sorted.push_back(d);
        }
        items += n_items;
    }
    const size_t bytes = (size_t)m * sizeof(TnGroupDesc);
    TnGroupDesc* table = nullptr;
    std::vector<char>* shadow = nullptr;
    bool hit = false;
    if (cached) {
        for (int k = 0; k < nslots && !hit; ++k)
            if (shadows[k].size() == bytes && memcmp(shadows[k].data(), sorted.data(), bytes) == 0) { hit = true; table = (TnGroupDesc*)tables_dev[k]; }
        if (!hit) { const int k = *victim; *victim = (k + 1) % nslots; table = (TnGroupDesc*)tables_dev[k]; shadow = &shadows[k]; }
    } else {
        table = (TnGroupDesc*)(scratch + 256);
    }
    if (!table) return CMPC_EHIP;
    if (!hit) {
        for (int c0 = 0; c0 < m; c0 += TN_UPLOAD) {     // through the kernel-argument segment: no host synchronisation
            TnUploadArgs ua;
            ua.n = std::min(TN_UPLOAD, m - c0);
            ua.base = c0;
            for (int i = 0; i < ua.n; ++i) ua.d[i] = sorted[c0 + i];
            hipLaunchKernelGGL(tn_desc_upload_kernel, dim3(1), dim3(256), 0, st, ua, table);
        }
        if (shadow) shadow->assign((const char*)sorted.data(), (const char*)sorted.data() + bytes);
    }
    static const bool static_deal = getenv("CMPC_TN_STATIC") != nullptr;
    int* next_item = static_deal ? nullptr : (int*)scratch;                 // the launch's item counter: first bytes of its scratch block
    if (next_item && hipMemsetAsync(next_item, 0, sizeof(int), st) != hipSuccess) { cmpc_set_error("gemm_tn_grouped: memset"); return CMPC_EHIP; }
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(std::min(items, slots)), dim3(256), 2 * 2 * TnCfg<bf16_t>::BR * 128 * 2, st, table, m, items, next_item);
    if (cmpc_check_launch("gemm_tn_grouped") != CMPC_OK) return CMPC_EHIP;
    // fixed-order sums of the slabs into the outputs (recorded into the caller's deferred-fold list when one is active and the
    // output is a gradient: then they run with the bias / LayerNorm folds in ONE launch; otherwise launched here)
    for (const Fold& f : folds) CK_FOLD(tn_fold(ex[f.prod], rsplit[f.prod], f.slabs, f.slab_size, f.lds, st));
    return CMPC_OK;
}

extern "C" int cmpc_gemm_tn_grouped(const cmpc_gemm_tn_args* args, int n, void* stream) {
    return cmpc_gemm_tn_grouped_cached(args, n, nullptr, 0, nullptr, 0, nullptr, (hipStream_t)stream);
}

extern "C" int cmpc_gemm_tn(const cmpc_gemm_tn_args* a, void* stream) {
    const int rc = tn_validate(a);
    if (rc != CMPC_OK) return rc;
    if (a->rsplit < 1) { cmpc_set_error("gemm_tn: rsplit must be >= 1"); return CMPC_EINVAL; }
    if (a->R == 0) return CMPC_OK;              // empty reduction: out += 0
    dim3 grid(((a->Kv + 127) / 128) * ((a->Nv + 127) / 128), a->rsplit, a->nb * a->nb2);
    hipStream_t st = (hipStream_t)stream;
    // one writer per element: an unsplit reduction adds straight into `out`; a split one (or an outer batch that shares one
    // output, o_bs == 0) goes through per-part slabs and a fixed-order fold
    const bool shared_out = a->nb2 > 1 && a->o_bs == 0;
    float* slabs = nullptr; long slab_size = 0; int lds = 0;
    if (a->rsplit > 1 || shared_out) {
        lds = (a->Nv + 3) / 4 * 4; slab_size = (long)a->Kv * lds;
        slabs = (float*)cmpc_ws((size_t)slab_size * a->rsplit * a->nb * a->nb2 * sizeof(float), st);
        if (!slabs) return CMPC_EHIP;
    }
    if (a->dtype == DT_F32) hipLaunchKernelGGL((gemm_tn_kernel<float>), grid, dim3(256), 2 * 2 * TnCfg<float>::BR * 128 * 4, st, *a, slabs, slab_size, lds);
    else if (a->dtype == DT_BF16) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t>), grid, dim3(256), 2 * 2 * TnCfg<bf16_t>::BR * 128 * 2, st, *a, slabs, slab_size, lds);
    else hipLaunchKernelGGL((gemm_tn_kernel<f16_t>), grid, dim3(256), 2 * 2 * TnCfg<f16_t>::BR * 128 * 2, st, *a, slabs, slab_size, lds);
    if (slabs) {
        if (cmpc_check_launch("gemm_tn") != CMPC_OK) return CMPC_EHIP;
        if (shared_out) {      // every (b2, split) slab of inner entry b1 folds into the one shared region
            for (int b1 = 0; b1 < a->nb; ++b1) {
                // slabs are ordered [bz = b1 + nb * b2][by]: for nb == 1 the nb2 * rsplit slabs are contiguous
                if (a->nb != 1) { cmpc_set_error("gemm_tn: o_bs == 0 needs nb == 1"); return CMPC_EINVAL; }
                const int rc = cmpc_reduce_parts_f32(slabs, slab_size, 1, a->rsplit * a->nb2, a->Kv, lds, a->Nv, a->out + a->o_off[0], 0, a->ldo, 1, st);
                if (rc != CMPC_OK) return rc;
            }
            return CMPC_OK;
        }
        return tn_fold(*a, a->rsplit, slabs, slab_size, lds, st);
    }
    return cmpc_check_launch("gemm_tn");
}
