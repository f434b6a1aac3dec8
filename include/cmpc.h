/* cmpc.h -- C ABI of libcmpc_hip.so: the CMPC head (zigonk/CMPC-Refseg CMPC_model.build_graph /
 * train_op) as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference has no FFI: its only device boundary is sess.run(feed_dict) on a TF1 graph
 * (trainval_model.py:98-107, test.py:286-296).  Each entry point below replaces one stage of
 * that graph; the comment on each cites the reference lines (file:line under /root/reference)
 * whose arithmetic it performs.  The Python facade cmpc-refseg_amd/model.py chains them behind
 * LSTM_model's constructor / feed / fetch contract (CMPC_model.py:15-40,67-71,140-142).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all buffers
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *  - return value: 0 = ok, <0 = error (CMPC_EINVAL -1, CMPC_EHIP -2); cmpc_last_error() gives
 *    the message; nothing throws across the ABI
 *  - dt: 0 = float32, 1 = bfloat16, 2 = float16 (storage of feature maps and GEMM operands); statistics,
 *    accumulators, logits, losses and parameters are always float32 (sample sums: float64)
 *  - a "map" is a row-major [R = B*N, ld] matrix, rows r = b*N + n (n = y*w + x, NHWC order of the
 *    reference), C valid channels, ld >= C a multiple of 8; producers write pad columns as 0
 *  - not re-entrant per stream; one process per GPU for data-parallel runs
 */
#ifndef CMPC_H
#define CMPC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* cmpc_last_error(void);
/* Bumped whenever an exported signature or structure changes incompatibly; a binding must refuse a library whose version differs
 * (cmpc-refseg_amd/_lib.py does).  2: stat blocks (double[n][CMPC_STAT_PARTS][2]) replace double[B][2] sums, cmpc_mutan_fwd gained
 * pre_tanh, cmpc_adam_step gained `nonfinite`, cmpc_cfg gained `model`.  3: cmpc_feeds gained `levels_done`, n_lanes may be 2 */
#define CMPC_ABI_VERSION 3
int cmpc_abi_version(void);

/* ---- GEMMs: every _conv 1x1 (CMPC_model.py:412-417), tf.matmul (:173,187,226,235,362,384,400)
 *      and tf.nn.convolution 1x1 (util/cell.py:43) ---------------------------------------- */
typedef struct {
    int dtype;                 /* operand dtype (A, Bt, and C unless c_f32) */
    int nseg;                  /* 1..3 K-segments summed into one accumulator */
    const void* A[3];          /* [M, K_s] row-major, lda[s] */
    const void* Bt[3];         /* [N, K_s] row-major (the weight, output-channel major), ldb[s] */
    int K[3], lda[3], ldb[3];
    int64_t sA[3], sB[3];      /* batch strides in elements */
    void* C; int ldc; int64_t sC;
    int c_f32;                 /* 1: C is float32 regardless of dtype */
    int M, N, n_valid, batch;  /* columns >= n_valid are written as 0 */
    const float* bias;         /* [N] or NULL */
    const float* sbias; int ld_sbias;   /* per-sample bias  sbias[(row/rows_per_sample)*ld + n] */
    const float* pbias; int ld_pbias;   /* per-position bias pbias[(row%rows_per_sample)*ld + n] */
    int rows_per_sample;
    int act;                   /* 0 none, 1 relu, 2 tanh, 3 sigmoid */
    float alpha;               /* scales the accumulator before the biases */
    int accumulate;            /* C += result */
} cmpc_gemm_nt_args;
int cmpc_gemm_nt(const cmpc_gemm_nt_args* a, void* stream);
/* two independent products in one launch when both take the 256 x 128-tile 16-bit pipeline (otherwise: two launches) -- the two gated
 * branches of a gated_exchange_module (CMPC_model.py:245-259), forward and backward */
int cmpc_gemm_nt_pair(const cmpc_gemm_nt_args* a, const cmpc_gemm_nt_args* b, void* stream);

typedef struct {
    int dtype;
    const void* A; int lda; int Ka;     /* rows of lda elements; Ka = columns READABLE from A + a_off[i] */
    const void* D; int ldd; int Nd;     /* rows of ldd elements; Nd = columns READABLE from D + d_off[i] (block width when d_off selects a column block) */
    float* out; int ldo;                /* out[k, n] += alpha * sum_r A[r,k] D[r,n]  (one writer per element; split reductions go through slabs folded in a fixed order) */
    int R, Kv, Nv;                      /* reduction length, valid output rows / cols */
    int nb; int64_t a_off[8], d_off[8], o_off[8];  /* inner batch: element offsets */
    int nb2; int64_t a_bs, d_bs, o_bs;  /* outer batch: strides */
    int rsplit;                         /* workgroups along the reduction */
    float alpha;
    const void* zeros;                  /* optional: >= 256 B of device zeros (enables the LDS-DMA pipeline for bf16) */
    int conv_H, conv_W, conv_dy, conv_dx;   /* conv_W > 0: one tap of a convolution's weight gradient (CMPCv5_BiLSTM_model.py:203-204,235-237 via tf.gradients):
                                           rows are the pixels (b, y, x) of [*, conv_H, conv_W] maps and row r of A is read at pixel (y + conv_dy, x + conv_dx),
                                           as zeros outside the image; 0: plain product */
} cmpc_gemm_tn_args;
int cmpc_gemm_tn(const cmpc_gemm_tn_args* a, void* stream);
/* n independent products in as few launches as possible (weight gradients deferred to the end of the backward
 * pass: tf.gradients' dW ops of CMPC_model.py:447-456 have no consumer before the optimizer).  `rsplit` of
 * the entries is ignored: the split of each reduction is chosen for the group as a whole. */
int cmpc_gemm_tn_grouped(const cmpc_gemm_tn_args* args, int n, void* stream);

/* ---- backbone convolutions as implicit GEMM (deeplab_resnet/model.py:19-401; network.py:105-188 conv /
 *      atrous_conv 'SAME', :260-270 frozen batch_norm folded into Wt and bias, :194-201,233-235 relu / add).
 *      X [B,H,W,Cin] NHWC (row stride ldx), Wt [Cout][k*k*Cin] (tap-major, Cin fastest), Y / res [B,Ho,Wo,Cout];
 *      zeros: >= 256 bytes of device zeros (source for out-of-image taps). Needs ksize in {1,3}, stride in
 *      {1,2}, Cin % 64 (bf16) / 32 (f32) == 0, Cout % 8 == 0, ldy % 8 == 0, 16-B aligned rows. ------------- */
typedef struct {
    int dtype;
    const void* X; int ldx;
    const void* Wt; int ldw;
    const float* bias;
    const void* res;            /* optional residual added before the ReLU */
    void* Y; int ldy;
    int B, H, W, Cin, Cout, ksize, stride, dil, relu;
    const void* zeros;
} cmpc_conv_args;
int cmpc_conv_nhwc(const cmpc_conv_args* a, void* stream);

/* ---- utilities ---------------------------------------------------------------------------- */
int cmpc_cast(int src_dt, const void* src, int dst_dt, void* dst, int64_t n, void* stream);
/* act'(y) applied to a gradient, plus column sums:  dpre = dy * act'(y);  db[c] += sum_r dpre
 * (bias gradients of _conv, CMPC_model.py:416-417); dsb[b][c] += per-sample sums (optional).
 * dy / y / dpre are [R, ld] windows of matrices with row stride `stride`; y, dpre, db, dsb optional */
int cmpc_act_bwd(int dt, const void* dy, const void* y, void* dpre, int act, int R, int stride, int ld, int C,
                 float* db, float* dsb, int ld_dsb, int rows_per_sample, void* stream);
/* out[b, :] (+)= sum_n w[b,n] * x[b,n,:]   and   s[b,n] = scale * x[b,n,:] . v[b,:] */
int cmpc_wcolsum(int dt, const void* x, const float* w, float* out, int ld_out, int B, int N, int ld, int C,
                 float scale, void* stream);
int cmpc_rowdot1(int dt, const void* x, const float* v, int ld_v, float* s, int B, int N, int ld, int C,
                 float scale, void* stream);
/* x[b,n,:] += s1*w1[b,n]*v1[b,:] + s2*w2[b,n]*v2[b,:]  (w2/v2 optional) */
int cmpc_rank1_update(int dt, void* x, const float* w1, const float* v1, const float* w2, const float* v2,
                      int ld_v, float s1, float s2, int B, int N, int ld, int C, void* stream);
/* backbone epilogue (deeplab_resnet/model.py bottlenecks; network.py:260-270,194-201,233-235): in place
 * y = relu?(y + bias[c] (+ res)) on an NHWC conv output [R, C]; bias = the folded frozen batch-norm shift */
int cmpc_bias_act_res(int dt, void* y, const float* bias, const void* res, int relu, int64_t R, int C, void* stream);
/* y (+)= x elementwise on maps */
int cmpc_axpy(int dt, const void* x, void* y, float a, int64_t n, void* stream);

/* ---- tf.nn.l2_normalize(x, 3) (CMPC_model.py:109,111,113,272-284,324,408; eps 1e-12) ------- */
int cmpc_l2norm_rows_fwd(int dt, const void* x, void* y, float* rstd, float* nz_mask, int R, int ld, int C, void* stream);
int cmpc_l2norm_rows_bwd(int dt, const void* dy, const void* y, const float* rstd, void* dx, int R, int ld, int C,
                         int accumulate, void* stream);

/* ---- whole-sample statistics for tf.contrib.layers.layer_norm (CMPC_model.py:364,370;
 *      util/cell.py:53-66): {sum x, sum x^2} over (n, c<C) of every sample, float64, as a STAT BLOCK:
 *      double [n_stats][CMPC_STAT_PARTS][2], one pair per producing workgroup (unused slots zero); the kernels
 *      that consume a statistic add its pairs themselves in a fixed order, so no fold launch sits between
 *      producer and consumer.  Every `sums` / `bsums` argument below is such a block --------------------- */
#define CMPC_STAT_PARTS 128
int cmpc_sample_stats(int dt, const void* x, double* sums, int B, int N, int ld, int C, void* stream);

/* ---- mutan_fusion (CMPC_model.py:295-328): P [R, 5*ld] holds the five vis_trans pre-activations
 *      (GEMM incl. spatial channels and bias); g [B, 5*ld] = tanh(lang_trans); on return P holds
 *      tanh(P_h), X1 = l2norm(tanh(sum_h tanh(P_h) * g_h)).  pre_tanh != 0: P already holds the tanh
 *      values (the GEMM ran with ACT_TANH) and is only read ----------------------------------- */
int cmpc_mutan_fwd(int dt, void* P, const float* g, void* X1, float* rstd, int B, int N, int ld, int C, int pre_tanh, void* stream);
/* ---- short-reduction product of the cross-modal graph (CMPC_model.py:359-410: Y = gw_w . Z, dX1 += gw_v . dZ,
 *      dX1 += scale * dA0 . PT): C[b][m, n] (+)= alpha * sum_{k < Kv} A[b][m, k] * Bk[b][k, n], Kv <= 24 (the word axis),
 *      Bk k-major, 16-bit storage, fp32 accumulation; columns n >= n_valid contribute zero, columns k >= Kv of A are
 *      ignored.  A streaming kernel, not a GEMM ------------------------------------------------------------------- */
int cmpc_lowrank_nn(int dt, const void* A, int lda, int64_t sA, const void* Bk, int ldb, int64_t sB, void* C, int ldc, int64_t sC,
                    int M, int N, int n_valid, int Kv, int batch, float alpha, int accumulate, void* stream);
/* in: Th (tanh values), dX1; out: Th overwritten by dP_h, dg[b][5*ld] += column sums */
int cmpc_mutan_bwd(int dt, void* Th, const float* g, const void* X1, const float* rstd, const void* dX1,
                   float* dg, int B, int N, int ld, int C, void* stream);

/* ---- build_spa_graph softmaxes (CMPC_model.py:384-399): A0 [B,N,Tp] f32 = affinity/sqrt(C);
 *      gw_w = softmax_T(mask*pr*A0 + (1-mask)*FLT_MIN), gw_v = softmax_N(pr*A0)*mask.
 *      f32 copies are kept for backward, dt copies ([B,N,Tp], pad columns 0) feed the GEMMs -- */
/*      scratch: B * ceil(N/64) * 128 floats (per-chunk column statistics) */
/*      mask_after != 0: CMPCv5_BiLSTM_model.py:486-487 -- gw_w = mask * softmax_T(pr*A0): padded words stay in the softmax (their logit is 0) */
int cmpc_graph_softmax_fwd(int dt, int mask_after, const float* A0, const float* pr, const float* mask, float* gw_w, float* gw_v,
                           void* gw_w_t, void* gw_v_t, float* scratch, int B, int N, int T, int Tp, void* stream);
int cmpc_graph_softmax_bwd(int dt, const float* dgw_w, const float* dgw_v, const float* gw_w, const float* gw_v,
                           const float* A0, const float* pr, const float* mask, float* dA0, void* dA0_t, float* dpr,
                           float* scratch, int B, int N, int T, int Tp, void* stream);

/* ---- graph_conv (CMPC_model.py:359-374) around the two GEMMs ------------------------------- */
/* G = relu(X + LN(Y; gamma, beta)) */
int cmpc_gconv_pre_fwd(int dt, const void* Y, const void* X, const double* sums, const float* gamma, const float* beta,
                       void* G, int B, int N, int ld, int C, void* stream);
/* dX (+)= dG*[G>0]; dY = LN-backward; dgamma/dbeta += ; bsums: scratch stat block of B statistics */
int cmpc_gconv_pre_bwd(int dt, const void* dG, const void* G, const void* Y, const double* sums, const float* gamma,
                       void* dX, int accumulate_dX, void* dY, float* dgamma, float* dbeta, double* bsums,
                       int B, int N, int ld, int C, void* stream);
/* out = l2norm(relu(LN(U; gamma, beta)))  (CMPC_model.py:370-372,408) */
int cmpc_gconv_post_fwd(int dt, const void* U, const double* sums, const float* gamma, const float* beta,
                        void* out, float* rstd_row, int B, int N, int ld, int C, void* stream);
int cmpc_gconv_post_bwd(int dt, const void* dout, const void* out, const float* rstd_row, const void* U,
                        const double* sums, const float* gamma, void* dU, float* dgamma, float* dbeta, double* bsums,
                        int B, int N, int ld, int C, void* stream);

/* ---- gated_exchange_module (CMPC_model.py:245-259) ---------------------------------------- */
/* softmax over the N nodes of each sample (global_vec, :229) */
int cmpc_softmax_n_fwd(const float* logits, float* attn, int B, int N, void* stream);
int cmpc_softmax_n_bwd(const float* dattn, const float* attn, float* dlogits, int B, int N, void* stream);
/* tf.nn.l2_normalize(gv_lang) with no axis -> over the whole [B, M] tensor (:241) */
int cmpc_l2norm_all_fwd(const float* x, float* y, float* rstd1, int n, void* stream);
int cmpc_l2norm_all_bwd(const float* dy, const float* y, const float* rstd1, float* dx, int n, void* stream);
/* out = l2norm_rows(feat + r1*g1[b] + r2*g2[b])  (:256-258,272); r2 = g2 = NULL: one gated branch (CMPCv5_BiLSTM_model.py:343-346; dp2 / dg2 unused) */
int cmpc_exchange_combine_fwd(int dt, const void* feat, const void* r1, const void* r2, const float* g1, const float* g2,
                              int ld_g, void* out, float* rstd, int B, int N, int ld, int C, void* stream);
/* dfeat (+)= dE; dp1 = dE*g1*[r1>0]; dp2 likewise; dg1/dg2 [B][ld_g] += sum_n dE*r; db1/db2 [C] (optional) += column sums of dp1/dp2
 * over all B*N rows (the bias gradients of the trans_feat convolutions that produced r1/r2) */
int cmpc_exchange_combine_bwd(int dt, const void* dout, const void* out, const float* rstd, const void* r1, const void* r2,
                              const float* g1, const float* g2, int ld_g, void* dfeat, int accumulate_dfeat,
                              void* dp1, void* dp2, float* dg1, float* dg2, float* db1, float* db2, int B, int N, int ld, int C, void* stream);

/* ---- ConvLSTMCell.call (util/cell.py:36-79), one time step; Yg [R, 4*ld] = [x|h].kernel with
 *      gate blocks j,i,f,o; peepholes W_c* are [N, M] fp32 (row stride M); LayerNorm vectors
 *      in the order j,i,f,o,c; sums / bsums are stat blocks of 5*B statistics, [j,i,f,o,c][B] -- */
typedef struct { const float* beta[5]; const float* gamma[5]; } cmpc_convlstm_ln;
typedef struct { float* dbeta[5]; float* dgamma[5]; } cmpc_convlstm_dln;
/* A: i += W_ci*c_prev, f += W_cf*c_prev (skipped when c_prev NULL); sums[j,i,f] <- stats */
int cmpc_convlstm_a(int dt, void* Yg, const void* c_prev, const float* W_ci, const float* W_cf, double* sums,
                    int B, int N, int ld, int M, void* stream);
/* B: c_pre = c_prev*sig(LN f + 1) + sig(LN i)*tanh(LN j); o_pre = o + W_co*c_pre (in place);
 *    sums[o] <- stats(o_pre), sums[c] <- stats(c_pre) */
int cmpc_convlstm_b(int dt, void* Yg, const void* c_prev, const float* W_co, const cmpc_convlstm_ln* ln,
                    double* sums, void* c_pre, int B, int N, int ld, int M, void* stream);
/* C: c_new = LN(c_pre); h = sig(LN(o_pre)) * tanh(c_new) */
int cmpc_convlstm_c(int dt, const void* Yg, const void* c_pre, const cmpc_convlstm_ln* ln, const double* sums,
                    void* c_new, void* h, int B, int N, int ld, int M, void* stream);
/* backward of C,B,A in three passes.  dYg [R,4*ld] receives d(pre-LN gate inputs = GEMM output);
 * dc_prev the state gradient (untouched when c_prev NULL); LN / peephole gradients accumulate
 * (per-workgroup partial rows folded in a fixed order); scr [R, ld] dt scratch; bsums: scratch stat block of 5*B statistics */
int cmpc_convlstm_bwd(int dt, const void* dh, const void* dc_new, const void* Yg, const void* c_prev, const void* c_pre,
                      const float* W_ci, const float* W_cf, const float* W_co,
                      const cmpc_convlstm_ln* ln, const double* sums, void* dYg, void* dc_prev,
                      float* dW_ci, float* dW_cf, float* dW_co, const cmpc_convlstm_dln* dln, void* scr, double* bsums,
                      int B, int N, int ld, int M, void* stream);

/* ---- score heads, resize_bilinear, sigmoid, weighed_logistic_loss, mIoU
 *      (CMPC_model.py:128-142,440-447,486-490; util/loss.py:6-16) -------------------------- */
int cmpc_score_conv_fwd(int dt, const void* feat, const float* Wk, const float* bias, float* score,
                        int B, int h, int w, int ld, int M, void* stream);
int cmpc_score_conv_bwd(int dt, const float* dscore, const void* feat, const float* Wk, void* dfeat, int accumulate,
                        float* dWk, float* dbias, int B, int h, int w, int ld, int M, void* stream);
/* up = legacy bilinear(score); sigm = sigmoid(up) (optional); when target != NULL also
 * loss[b] += sum BCE, inter[b]/uni[b] += |pred&gt| / |pred|gt| with pred = up > 0 */
int cmpc_upsample_fwd(const float* score, float* up, float* sigm, const float* target, float* loss,
                      int* inter, int* uni, int B, int h, int w, int H, int W, void* stream);
/* dscore = sum over the bilinear footprint of wscale * (sigmoid(up) - target) */
int cmpc_upsample_loss_bwd(const float* up, const float* target, float* dscore, float wscale,
                           int B, int h, int w, int H, int W, void* stream);

/* ---- language side (CMPC_model.py:144-192,347-357) ---------------------------------------- */
int cmpc_embed_gather(const float* table, const int* words, float* out, int n_words, int G, int ld_out, int vocab, void* stream);
int cmpc_embed_scatter(const float* dout, int ld, const int* words, float* dtable, int n_words, int G, int vocab, void* stream);
/* tf LSTMCell step (gates i,j,f,o; forget_bias 1) with dynamic_rnn length masking.
 * gates [B, 4*ld] f32 pre-activations in, activated gates out; state f32 [B, ld] */
int cmpc_lstm_cell_fwd(float* gates, const float* c_prev, const float* h_prev, const int* seq_len, int t,
                       float* c_out, float* h_out, float* out_t, int ld_out, int B, int ld, int R, void* stream);
int cmpc_lstm_cell_bwd(const float* gates_act, const float* c_prev, const float* c_out, const int* seq_len, int t,
                       const float* dout_t, int ld_dout, float* dh, float* dc, float* dgates,
                       int B, int ld, int R, void* stream);
/* one launch per step of the backward recurrence (B <= 8): dh += dgates_t . W_h^T (Wn = the [ld][4 ld] operand rows of W_h),
 * then cmpc_lstm_cell_bwd of step t-1 on it (util dynamic_rnn / LSTMCell backward, CMPC_model.py:144-164 via tf.gradients) */
int cmpc_lstm_bwd_step(const float* dgates_t, const float* Wn, int ldw, const float* gates_act_tm1, const float* c_prev, const float* c_out,
                       const int* seq_len, int tm1, const float* dout_tm1, int ld_dout, float* dh, float* dc, float* dgates_tm1,
                       int B, int ld, int R, void* stream);
/* The whole recurrence of one direction in ONE launch (persistent workgroups, W_h in registers, a grid barrier per step; B <= 8, ld <= 1024,
 * otherwise CMPC_EINVAL: use the per-step entry points).  xg [T, B, 4 ld]: x-side pre-activations incl. bias; Wh [4 ld rows][ldw]: gate rows of W_h,
 * k contiguous; gates [T, B, 4 ld] receives the activated gates; h_all / c_all [(T+1), B, ld], slice 0 = initial state; outs [B, T, ld];
 * sync: 8 bytes of ZEROED device memory per launch (barrier counter + abort flag: non-zero afterwards = the launch was aborted by its watchdog).
 * cmpc_lstm_seq_bwd: dgates [T, B, 4 ld] from douts [B, T, ld]; Wn [ld rows][4 ld] = rows of W_h^T */
int cmpc_lstm_seq_fwd(const float* xg, const float* Wh, int ldw, const int* seq_len, float* gates, float* h_all, float* c_all, float* outs,
                      void* sync, int B, int T, int ld, int R, void* stream);
int cmpc_lstm_seq_bwd(const float* Wn, int ldw, const float* gates, const float* c_all, const int* seq_len, const float* douts, float* dgates,
                      void* sync, int B, int T, int ld, int R, void* stream);
/* softmax over the 4 parser classes times seq_mask (:352-353) */
/* ncls classes per word (4: CMPC_model.py:351; 5: CMPC_video_mm_tgraph_allvec.py:406); parse rows are ncls wide */
int cmpc_parse_softmax_fwd(const float* logits, int ld, const float* mask, float* parse, int n, int ncls, void* stream);
int cmpc_parse_softmax_bwd(const float* dparse, const float* parse, const float* mask, float* dlogits, int ld, int n, int ncls, void* stream);
/* valid_lang (classes 0,1) / nec_lang (0..2) / the video model's action vector (class 3 alone, vid:203-213):
 * v[b] = l2norm(sum_t (sum_{cls_lo <= k < cls_lo + ncls} parse[b,t,k]) wf[b,t,:]); pstride = classes per parse row */
int cmpc_lang_pool_fwd(const float* parse, const float* wf, float* v, float* rstd, int B, int T, int ld, int R, int ncls, int cls_lo, int pstride, void* stream);
int cmpc_lang_pool_bwd(const float* dv, const float* v, const float* rstd, const float* parse, const float* wf,
                       float* dparse, float* dwf, int B, int T, int ld, int R, int ncls, int cls_lo, int pstride, void* stream);

/* ---- CMPCv5_BiLSTM_model.py ("v5:") / CMPCv5_BiLSTM_HSV_model.py ("hsv:") stages that CMPC_model does not have ----------------
 * slim conv2d under resnet_v2.resnet_arg_scope (v5:192-193,229-230) = convolution without bias + batch_norm(decay, epsilon 1e-5,
 * scale=True) + relu.  The convolution is a GEMM / cmpc_conv_nhwc; these entry points are the batch_norm around it.  X [R, ldx] holds the
 * convolution output, C valid channels, Cpad >= C a multiple of 8 (row length of the statistics vectors).
 *   cmpc_bn_stats       training mode: per-channel batch statistics.  sums double[2][Cpad] = {sum x, sum x^2} (kept for the moving-average
 *                       update), mean_rstd float[2][Cpad] = {mean, 1/sqrt(biased var + eps)}
 *   cmpc_bn_from_moving inference mode: mean_rstd from moving_mean / moving_variance
 *   cmpc_bn_update_moving  UPDATE_OPS (v5:575-577): moving = decay * moving + (1 - decay) * batch, variance with Bessel's correction
 *   cmpc_bn_apply_fwd   y[r, c] = relu?(gamma[c] * (x - mean) * rstd + beta[c]) for c < C, 0 for C <= c < Cy (pad channels of the destination
 *                       block; y may be a column block of a wider map: row stride ldy)
 *   cmpc_bn_bwd         training-mode backward: g = dy * [y > 0]; dbeta += sum g; dgamma += sum g * xhat;
 *                       dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)); means_scratch float[2][Cpad] */
int cmpc_bn_stats(int dt, const void* x, int ldx, int R, int C, int Cpad, float eps, double* sums, float* mean_rstd, void* stream);
int cmpc_bn_from_moving(const float* moving_mean, const float* moving_var, int C, int Cpad, float eps, float* mean_rstd, void* stream);
int cmpc_bn_update_moving(const double* sums, int R, float decay, float* moving_mean, float* moving_var, int C, int Cpad, void* stream);
int cmpc_bn_apply_fwd(int dt, const void* x, int ldx, const float* mean_rstd, int Cpad, const float* gamma, const float* beta, void* y, int ldy, int Cy,
                      int R, int C, int relu, void* stream);
int cmpc_bn_bwd(int dt, const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx, const float* mean_rstd, int Cpad, const float* gamma,
                void* dx, int lddx, float* dgamma, float* dbeta, float* means_scratch, int R, int C, int relu, void* stream);
/* tf.image.resize_bilinear (legacy, align_corners=False) of a whole map [B, h, w, C] -> [B, H, W, C] (v5:201,246) and its gradient; C % 8 == 0 */
int cmpc_resize_bilinear_fwd(int dt, const void* x, int ldx, void* y, int ldy, int B, int h, int w, int H, int W, int C, void* stream);
int cmpc_resize_bilinear_bwd(int dt, const void* dy, int lddy, void* dx, int lddx, int B, int h, int w, int H, int W, int C, void* stream);
/* hsv:120-126: im f32 [B, H, W, 3] (BGR minus mean) -> + mean, reverse to RGB, tf.image.rgb_to_hsv, legacy bilinear to [h, w]: out [B*h*w, ld]
 * (3 channels, the rest of the row zero) */
int cmpc_hsv_map(int dt, const float* im, void* out, int ld, int B, int H, int W, int h, int w, void* stream);
/* array_ops.reverse_sequence of bidirectional_dynamic_rnn (v5:170-174) on [B, T, ld] float rows (not in place), and the time-major word ids of
 * the reversed sequences: out_tb[t * B + b] = words[b, t < len ? len - 1 - t : t] */
int cmpc_reverse_sequence(const float* in, float* out, const int* seq_len, int B, int T, int ld, void* stream);
int cmpc_reverse_words_tb(const int* words, const int* seq_len, int* out_tb, int B, int T, void* stream);
/* seq_mask of v5:181: mask[r] = (sum_c |a[r, c]| + |b[r, c]|) != 0 */
int cmpc_rows_nonzero2(const float* a, const float* b, float* mask, int rows, int ld, int C, void* stream);
/* the decoder's last 1x1 convolution to ONE channel (v5:205): out[r] = x[r, :C] . w + bias[0]; backward: dx[r, :] = d[r] * w, dw += sum_r d[r] x[r, :],
 * dbias += sum_r d[r] */
int cmpc_conv_to1_fwd(int dt, const void* x, int ld, const float* w, const float* bias, float* out, int R, int C, void* stream);
int cmpc_conv_to1_bwd(int dt, const float* d, const void* x, int ld, const float* w, void* dx, float* dw, float* dbias, int R, int C, void* stream);

/* ---- parameters: packing fp32 masters into padded GEMM operands; TF-Adam (CMPC_model.py:450-478)
 *      with L2 on 'DW' (:433,446, util/loss.py:28-32) and x2 on 'biases' (:464-465) folded in ---- */
typedef struct {
    int64_t src_off;           /* element offset into the fp32 master buffer; matrix [K, N], row stride ld_src */
    int ld_src;
    int64_t dst_off;           /* BYTE offset into the operand arena */
    int dst_dt;                /* dtype of the packed copy */
    int transpose;             /* 1: dst[n][k] (Bt for forward), 0: dst[k][n] (Bt for dX) */
    int rows, cols, ld_dst;    /* dst matrix extent (padded) */
    int nks; int ks_src[4], ks_len[4], ks_dst[4];   /* K segments: src row range -> dst k offset */
    int nns; int ns_src[5], ns_len[5], ns_dst[5];   /* N blocks:   src col range -> dst n offset */
} cmpc_pack_desc;
/* tile_prefix_dev[i] = number of 64(k) x 128(n) tiles of descriptors 0..i-1 (exclusive prefix sum, ndesc+1
 * entries; tile count = ceil(K/64)*ceil(N/128) of each padded block); total_tiles = tile_prefix[ndesc].
 * Every K/N segment boundary, ld_src and src_off must be a multiple of 4 (float4 reads). */
int cmpc_pack_weights(const float* master, void* arena, const cmpc_pack_desc* descs_dev, const int* tile_prefix_dev,
                      int ndesc, int total_tiles, void* stream);
/* the tiles [tile_begin, tile_end) only: lets the text encoder's operands (planned first) be published before the
 * rest, so that the next step's LSTM starts while the level weights are still being packed */
int cmpc_pack_weights_range(const float* master, void* arena, const cmpc_pack_desc* descs_dev, const int* tile_prefix_dev,
                            const int* tile_desc_dev /* optional: descriptor index of every tile */,
                            int ndesc, int tile_begin, int tile_end, void* stream);

typedef struct { int64_t off; int count; float wd; float gmult; } cmpc_adam_seg;
/* Overflow guard: an element whose gradient is inf / nan is left untouched (parameter and both moments) and counted in *nonfinite
 * (device int, optional) -- f16 storage saturates to inf, and one such batch must not poison the Adam state */
int cmpc_adam_step(float* params, const float* grads, float* m, float* v, const cmpc_adam_seg* segs_dev, int nseg,
                   float lr_t, float beta1, float beta2, float eps, float gscale, int* nonfinite, void* stream);

/* =====================================================================================================
 * Whole-path entry points (SURVEY.md 8b, row "C-ABI"): one handle = one LSTM_model graph
 * (CMPC_model.py:13-492) on one GPU.  The handle owns the parameters (fp32 masters, gradients, Adam moments),
 * the packed GEMM operands, every intermediate of a step (one static workspace: nothing is allocated after
 * cmpc_create), the statistics saved for backward, three lane streams and their events.  One call per
 * reference sess.run:
 *     sess.run([pred, up, sigm], feed)                 (test.py:286-296)          -> cmpc_forward
 *     sess.run([train, train_step, merged], feed)      (trainval_model.py:98-107) -> cmpc_forward (with target_fine),
 *                                                                                    cmpc_backward, [all-reduce of
 *                                                                                    cmpc_buffers().grads], cmpc_optimizer_step
 * No Python, no per-launch environment lookups: the host cost of a step is its hipLaunchKernel calls.
 * The handle is single-caller (not re-entrant); a step's calls must be issued in the order above.
 * ===================================================================================================== */
typedef struct cmpc_engine_s* cmpc_handle;

typedef struct {
    /* graph-shaping constructor arguments of LSTM_model (CMPC_model.py:15-40) */
    int batch_size, num_steps, vf_h, vf_w, H, W;
    int vf_dim, c4_dim, c3_dim;         /* channels of res5c / res4b22 / res3b3 (2048, 1024, 512: CMPC_model.py:108-112) */
    int vocab_size, v_emb_dim, mlp_dim, rnn_size, glove_dim, parse_dim;
    /* train_op() (CMPC_model.py:446-456) */
    double start_lr, end_lr, lr_power;  /* host-side arithmetic of tf.train.polynomial_decay, kept in double */
    int lr_decay_step;
    float weight_decay;
    float loss_w[4];                    /* weights of the BCE terms: final, c5, c4, c3 = 0.7, 0.1, 0.1, 0.1 (:444-445) */
    int dtype;                          /* storage of feature maps and visual GEMM operands: 0 f32 (exact-fp32 MFMA), 1 bf16,
                                           2 f16 (same MFMA rate as bf16, 8x smaller rounding: meets the 1e-4 mean-IoU bar) */
    float loss_scale;                   /* every gradient cmpc_backward writes is multiplied by it and cmpc_optimizer_step divides
                                           it out again; 0 = default (1; for f16 storage the power of two that keeps the largest upstream
                                           gradient, loss_w[0] * (H/vf_h) * (W/vf_w) / batch_size, at <= 16) */
    int n_lanes;                        /* 3: pyramid levels / exchange modules on three lane streams; 1: one stream */
    int device;                         /* HIP device ordinal; -1 = planning only (manifest, operand plan, workspace size:
                                           no GPU is touched; every compute entry point then returns CMPC_EINVAL) */
    /* which graph: get_model.get_segmentation_model(name) (get_model.py:15-17) */
    int model;                          /* CMPC_MODEL_CMPC (CMPC_model.py) or CMPC_MODEL_V5_BILSTM (CMPCv5_BiLSTM_model.py: BiLSTM encoder, levels c5 / c4,
                                           one gated branch, 2-step ConvLSTM, ASPP + DeepLabv3+ decoder on res2b_relu with slim batch-norm) */
    int hsv;                            /* model V5_BILSTM: CMPCv5_BiLSTM_HSV_model.py -- HSV of the image appended to the c5 / c4 taps (hsv:120-134) */
    int bn_train;                       /* model V5_BILSTM: batch_norm(is_training = mode == 'train') (v5:153-154): 1 = batch statistics (+ moving-average
                                           updates in the train step), 0 = moving statistics */
    float bn_decay;                     /* batch_norm_decay (v5:42), 0.9997 */
    int c2_dim, c2_h, c2_w;             /* res2b_relu tap (v5:88): channels (256) and map size (H/4 x W/4) */
    int aspp_depth, low_dim;            /* 256 (v5:208), 48 (v5:196) */
    int aspp_rates[3];                  /* 6, 12, 18: output_stride 16 (v5:153,225) */
    int sample_frames;                  /* model VIDEO: frames of the clip that go through the graph (5: indices 0, 4, 8, 12, 15 of 16; CMPC_video_mm_tgraph_allvec.py:69-70) */
    int freeze_bn;                      /* model V5_BILSTM: freeze_bn=True (v5:528-529): variables whose name contains 'beta' or 'gamma' (batch-norm AND layer-norm
                                           scales / offsets) are left out of the optimizer's variable list: their Adam segments get a zero gradient multiplier */
    int conv5;                          /* conv5=True (CMPC_model.py:427-430, v5:521-525; the video model's finetune=True, vid:554-557: res3 / res4 / res5 convolution
                                           weights are trained too): cmpc_backward also
                                           produces the gradients of the three backbone taps, d cost / d c5, c4, c3 (taps "dc5", "dc4", "dc3": [B*N, cin], cfg.dtype),
                                           which the caller's backbone backward consumes; 0 = the backbone is frozen and no such gradient is formed */
} cmpc_cfg;
#define CMPC_MODEL_CMPC 0
#define CMPC_MODEL_V5_BILSTM 1
#define CMPC_MODEL_VIDEO 2              /* CMPC_video/CMPC_video_mm_tgraph_allvec.py (BASELINE config 5): batch_size 1; feeds c3 / c4 / c5 are the taps of the
                                           sample_frames frames, [sample_frames, vf_h, vf_w, .]; words END-padded with seq_len = number of words (the host turns
                                           the reference's front padding + valid_idx around: the graph slices the pad steps away, vid:141-142) */
/* fills *cfg with the reference's defaults (CMPC_model.py:15-40), f16 storage (the 16-bit mode that meets the 1e-4 mean-IoU bar), 3 lanes, device 0 */
int cmpc_default_cfg(cmpc_cfg* cfg);
/* the same for a given model: CMPC_MODEL_V5_BILSTM sets loss_w = 0.8, 0.1, 0.1, 0 (v5:541-542), bn_train 1, bn_decay 0.9997, c2_dim 256, c2_h = H / 4,
 * c2_w = W / 4, aspp_depth 256, low_dim 48, rates 6 / 12 / 18 -- vf_h, vf_w, H, W must be set by the caller BEFORE (v5:52-53 does not derive them) */
int cmpc_default_cfg_model(cmpc_cfg* cfg, int model, int hsv);
int cmpc_create(const cmpc_cfg* cfg, cmpc_handle* out);
int cmpc_destroy(cmpc_handle h);
/* the configuration in effect (defaults resolved, e.g. loss_scale) */
int cmpc_get_cfg(cmpc_handle h, cmpc_cfg* out);

/* parameter manifest in the reference's variable order and names ("text_objseg/c5_lateral/DW", ...):
 * index 0..n-1 -> name, element offset into the flat buffers, rank and shape (<= 4 dims, HWIO for convolutions) */
int cmpc_param_count(cmpc_handle h);
int cmpc_param_info(cmpc_handle h, int index, const char** name, int64_t* offset, int* rank, int64_t shape[4]);
/* the flat fp32 device buffers (total elements incl. 16-B alignment gaps): masters, gradients (what a data-parallel
 * caller all-reduces between cmpc_backward and cmpc_optimizer_step; they carry the factor cfg.loss_scale), Adam m and v */
int cmpc_buffers(cmpc_handle h, float** params, float** grads, float** adam_m, float** adam_v, int64_t* total);
/* tf.train.Saver.restore / save by variable name (trainval_model.py:46-63,136-142): host pointers, `count` must
 * equal the variable's element count.  cmpc_set_weights does NOT repack; call cmpc_pack after the last one. */
int cmpc_set_weights(cmpc_handle h, const char* name, const float* host_src, int64_t count);
int cmpc_get_weights(cmpc_handle h, const char* name, float* host_dst, int64_t count);
/* masters -> padded GEMM operands (after cmpc_set_weights, or after the caller modified the master buffer) */
int cmpc_pack(cmpc_handle h, void* stream);
/* non-trainable variables of the graph (model V5_BILSTM: the batch-norm moving statistics `text_objseg/<scope>/BatchNorm/moving_mean|moving_variance`,
 * updated by the UPDATE_OPS of v5:575-577 inside cmpc_backward; none for CMPC_model): enumerate, read, write (host pointers, float32) */
int cmpc_state_count(cmpc_handle h);
int cmpc_state_info(cmpc_handle h, int index, const char** name, int64_t* count);
int cmpc_get_state(cmpc_handle h, const char* name, float* host_dst, int64_t count);
int cmpc_set_state(cmpc_handle h, const char* name, const float* host_src, int64_t count);
/* global_step (CMPC_model.py:450) get / set (checkpoint resume, trainval_model.py:82 -lastiter) */
int cmpc_get_step(cmpc_handle h, int64_t* step);
int cmpc_set_step(cmpc_handle h, int64_t step);

typedef struct {
    const int32_t* words;       /* [B, T] token ids, 0-padded at the end (CMPC_model.py:67; util/text_processing.py:55-67) */
    const int32_t* seq_len;     /* [B] (CMPC_model.py:71) */
    const void* c3;             /* res3b3_relu  [B, vf_h, vf_w, c3_dim] NHWC, cfg.dtype (CMPC_model.py:76) */
    const void* c4;             /* res4b22_relu [B, vf_h, vf_w, c4_dim] (:75) */
    const void* c5;             /* res5c_relu   [B, vf_h, vf_w, vf_dim] (:74) */
    const float* target_fine;   /* [B, H, W, 1] or NULL for inference (:69) */
    void* feats_ready;          /* optional hipEvent_t recorded (on any stream) once c3/c4/c5 are complete: the text encoder
                                   is enqueued first and only the pyramid levels wait for it; NULL = ordered on `stream` */
    const void* c2;             /* model V5_BILSTM: res2b_relu [B, c2_h, c2_w, c2_dim] NHWC, cfg.dtype (v5:88); c3 is unused there (may be NULL) */
    const float* im;            /* model V5_BILSTM with hsv: the image feed itself [B, H, W, 3] f32, BGR minus mean (v5:80; hsv:120-126) */
    void* feats_ready_lv[3];    /* optional, per pyramid level (c5, c4, c3 -- for V5_BILSTM slot 2 is the res2b tap): hipEvent_t recorded once THAT tap is
                                   complete; a level's lane then waits for its own tap only (c3 leaves the backbone after res3, c4 after res4), NULL
                                   entries fall back to feats_ready */
    void* levels_done;          /* optional hipEvent_t that cmpc_forward RECORDS once the pyramid levels' forward is complete: from there to the
                                   levels' backward the step is a serial chain of small launches (exchange modules, ConvLSTM, scores), the
                                   window a caller uses for independent heavy work -- the frozen backbone of the NEXT batch */
} cmpc_feeds;
typedef struct {                /* optional caller-owned device buffers the fetches are copied into (NULL = skip) */
    float* pred;                /* [B, vf_h, vf_w, 1] logits (CMPC_model.py:140) */
    float* up;                  /* [B, H, W, 1] logits (:141) */
    float* sigm;                /* [B, H, W, 1] (:142) */
} cmpc_fetches;
/* build_graph() on one batch (CMPC_model.py:89-142); with target_fine also the four BCE terms, cls_loss_all and the
 * in-graph mIoU (:438-447,486-490).  Everything is enqueued on `stream` and the handle's lane streams (which fork
 * from and join into `stream`); nothing synchronises the host. */
int cmpc_forward(cmpc_handle h, const cmpc_feeds* feeds, const cmpc_fetches* fetches, void* stream);
/* tf.gradients(cost) of the last cmpc_forward (which must have had target_fine) into the flat gradient buffer:
 * d cls_loss_all / d theta; the L2 term (:433,446) and the x2 on biases (:462-475) are applied by the optimizer. */
int cmpc_backward(cmpc_handle h, void* stream);
/* event (hipEvent_t, caller-owned; NULL = none) that every later cmpc_backward records once the pyramid levels' backward is complete */
int cmpc_set_bwd_levels_event(cmpc_handle h, void* event);
/* debugging (CMPC_WS_GUARD=<bytes> at create): number of workspace allocations whose trailing guard was written; details on stderr */
int cmpc_debug_check_guards(cmpc_handle h);
/* Gradient buckets for a data-parallel caller (one process per GPU; the reference has no distributed code).  The flat
 * gradient buffer becomes final in cmpc_grad_bucket_count() pieces, in this order, while cmpc_backward is still running:
 * the exchange modules + ConvLSTM + final score, the pyramid levels c5, c4, c3, the text encoder + parser.  Bucket b covers
 * `nranges` (<= 4) contiguous element ranges of cmpc_buffers().grads.  cmpc_grad_bucket_wait makes `stream` wait (on the device) until
 * bucket b of the most recent cmpc_backward is final: the caller then enqueues that bucket's all-reduce on `stream`, overlapping the
 * rest of the backward pass, and orders cmpc_optimizer_step after the last all-reduce. */
int cmpc_grad_bucket_count(cmpc_handle h);
int cmpc_grad_bucket(cmpc_handle h, int bucket, int* nranges, int64_t offsets[4], int64_t counts[4]);
int cmpc_grad_bucket_wait(cmpc_handle h, int bucket, void* stream);
/* TF-Adam with polynomial LR decay, L2 on 'DW', x2 on 'biases' (CMPC_model.py:450-478), then repack of the operands.
 * gscale multiplies the gradients first (1/world after a summing all-reduce).  May be enqueued on a stream of its
 * own: the next cmpc_forward waits (on the device) for the events this call records.  *lr_used = the step's LR. */
int cmpc_optimizer_step(cmpc_handle h, float gscale, void* stream, double* lr_used);
/* The same update for ONE gradient bucket (every bucket exactly once per step, the last bucket last): `stream` waits on the device until
 * the bucket is final, then runs Adam on the bucket's parameters and repacks the operands packed from them -- beside the rest of the
 * backward pass.  cmpc_optimizer_step = every bucket in order on one stream. */
int cmpc_optimizer_bucket(cmpc_handle h, int bucket, float gscale, void* stream, double* lr_used);

/* Named intermediate of the last forward / backward ("words_parse", "gw_w_c3", "up_c4", "scalars", ...: the fetches
 * of the reference's visualisers, test_visualize_graph.py:243-253, plus every stage output the parity tests compare).
 * dtype: 0 f32, 1 bf16, 2 f16, 3 int32, 4 f64.  The pointer stays valid for the handle's lifetime; its contents are those of
 * the most recent step once the work enqueued by that step has finished. */
int cmpc_tap(cmpc_handle h, const char* name, void** ptr, int* dtype, int* rank, int64_t shape[4]);
int cmpc_tap_count(cmpc_handle h);
int cmpc_tap_name(cmpc_handle h, int index, const char** name);
/* the handle's host-side plan: the pack descriptors (host array of ndesc entries, owned by the handle), the operand arena
 * and workspace sizes, how many leading descriptors belong to the text encoder / parser (packed and published first);
 * and one packed operand by key ("lstm.t", "mutan_c5.t", "fus_c3.n", ...: byte offset in the arena, dtype, rows, ld) */
int cmpc_plan_info(cmpc_handle h, const cmpc_pack_desc** descs, int* ndesc, int64_t* arena_bytes, int64_t* workspace_bytes, int* stage0_ndesc);
int cmpc_operand_info(cmpc_handle h, const char* key, int64_t* byte_off, int* dt, int* rows, int* ld);
/* Live timing of the dominant kernel family (the bf16 MFMA gemm_nt launches: every 1x1-conv and dX product of the head):
 * while enabled, every such launch of cmpc_forward / cmpc_backward is bracketed by a hipEvent pair on the stream it is
 * launched on.  cmpc_kernel_timing_read synchronises the device and returns the summed durations (ms), algorithmic FLOPs
 * (2 * rows * valid columns * valid K: padding not counted) and algorithmic bytes (A and C once per row, the weight once)
 * since it was enabled.  Use with n_lanes = 1: with lanes, an interval also contains other streams' kernels. */
/* Phase-boundary timestamps without a profiler (so the overlap between lanes is the real one): while enabled, every entry point
 * records a hipEvent at its stage boundaries on the stream that reaches them ("fwd:text_done", "bwd:level_c4_done", ...).
 * cmpc_phase_marks_read(index) -> name and milliseconds since the first mark; returns CMPC_EINVAL past the last mark. */
int cmpc_phase_marks(cmpc_handle h, int enable);
int cmpc_phase_marks_read(cmpc_handle h, int index, const char** name, float* ms_since_first);
int cmpc_set_lanes(cmpc_handle h, int n_lanes);      /* 1 or 3; takes effect from the next cmpc_forward */
int cmpc_kernel_timing(cmpc_handle h, int enable);
int cmpc_kernel_timing_read(cmpc_handle h, double* ms, double* flops, double* bytes, int64_t* launches);
/* Launch trace without a profiler (process-wide): while enabled, every launch of this library is followed by a hipEvent on `stream`, so that
 * with EVERYTHING on that one in-order stream (cmpc_set_lanes(h, 1), no side streams) the interval between consecutive events is the launch's
 * duration -- what `rocprofv3 --kernel-trace --stats` reports.  cmpc_launch_trace_read(i) -> per-name totals (name, summed ms, launches),
 * CMPC_EINVAL past the last name; index 0 ends the recording.  bench.py prices the HBM-bound stage kernels against their byte model with it. */
int cmpc_launch_trace(int enable, void* stream);
int cmpc_launch_trace_read(int index, const char** name, double* ms, int64_t* launches);
/* number of kernel launches / memsets the last forward+backward+optimizer_step issued (host-side counter) */
int cmpc_launch_count(cmpc_handle h, int64_t* n);

/* ---- Dense CRF post-processing of the evaluation script (test.py:309-322: pydensecrf DenseCRF2D(W, H, 2), unary -log(1-p) / -log(p),
 *      addPairwiseGaussian(sxy, compat), addPairwiseBilateral(sxy, srgb, rgbim, compat), inference(iters), argmax).  sigm [H*W] fp32
 *      probabilities and rgb [H*W*3] uint8 are device pointers; q_out [2][H*W] fp32 (label 0, label 1) and / or mask_out [H*W] uint8
 *      receive the result.  The Gaussian kernels are evaluated exactly inside a 4-sigma window where the library filters through a
 *      permutohedral lattice: same recursion and normalisation, parity-unpinned against the library (not installed, not in the
 *      reference tree) ------------------------------------------------------------------------------------------------------ */
int cmpc_dense_crf(const float* sigm, const unsigned char* rgb, int H, int W, float sxy_g, float compat_g, float sxy_b, float srgb,
                   float compat_b, int iters, float* q_out, unsigned char* mask_out, void* stream);

/* Host utility (no GPU): CRC-32C of `n` bytes, continuing from `crc` (0 to start) -- the checksum of TensorFlow's tensor-bundle
 * checkpoints (trainval_model.py:46-63 restores / saves them; cmpc-refseg_amd/tf_bundle.py reads and writes the format). */
uint32_t cmpc_crc32c(uint32_t crc, const void* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif
