/* deepards_hip.h -- C ABI of libdeepards_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the ONE hot path of hahnicity/deepards: cnn_linear / resnet18-1D /
 * densenet18-1D forward + backward over (sub_batch, 1, seq_len) windows.  The reference has no
 * FFI for this path: every op below replaces a stock torch.nn call reached from
 * deepards/models/{resnet,densenet,torch_cnn_linear_network}.py and train_ards_detector.py
 * (file:line cited per entry point).  The Python host in deepards_amd/ binds these with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless said otherwise; fp32 throughout
 *   - caller allocates outputs and workspaces; nothing is retained between calls
 *   - all launches go to `stream`, no hidden synchronisation, graph-capturable
 *   - return 0 on success, -1 on invalid arguments, >0 = hipError_t of a failed launch
 *   - activation layout "RLC": act[row][l][c] with channel pitch ld (floats, multiple of 4);
 *     a BatchNorm *window* = rows_per_window consecutive rows = Wn = rows_per_window*L positions
 */
#ifndef DEEPARDS_HIP_H
#define DEEPARDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* da_stream_t; /* == hipStream_t */

int da_version(void);

/* Activation storage type.  `da_act_t*` parameters below are RLC activation / activation-gradient tensors whose element
   type is float (default) or bf16 (after da_set_act_dtype(1): BASELINE's bf16 configs -- every tensor between two
   kernels of the breath block is then stored in bf16; statistics, sums, features, parameters and their gradients stay
   float, all arithmetic is fp32 or bf16-MFMA with fp32 accumulation).  Pitches (`ld*`) count ELEMENTS.  The setting is
   process-wide and not thread-safe: set it before the step is built.  Entry points that take `float*` activations
   (the fp32 conv kernels, concat / slice / dropout) return -1 while bf16 is selected. */
typedef void da_act_t;
int da_set_act_dtype(int bf16);
int da_get_act_dtype(void);
/* sizeof of {da_wgrad_job, da_conv_job, da_wgrad_reduce_desc, da_repack_desc, da_bn_running_desc, da_bn_pgrad_desc} as the
   library was built (ABI drift check for bindings) */
void da_abi_sizes(int* out);
int da_sizeof_wgrad_reduce_desc(void);
int da_sizeof_bn_running_desc(void);
int da_sizeof_bn_pgrad_desc(void);
/* address of hipGetLastError as bound by this library (loader sanity check: same HIP runtime as the host) */
const void* da_hip_runtime_symbol(void);

/* ---- Conv1d as implicit GEMM on the fp32 matrix cores ------------------------------------
 * replaces nn.Conv1d fwd + dgrad: resnet.py:5-8 (conv2x2), :16-19,:126-128 (BasicBlock convs,
 * 1x1 s2 downsample); densenet.py:25-32 (1x1 bottleneck, k3 growth), :75-76 (transition conv).
 * Y[row][j*dst_stride+dst_off][n] (+)= sum_t sum_c X[row][j*src_stride+src_off[t]][c] *
 * Wp[wtap[t]][n][c],  j in [0,Lm); reads outside [0,Lsrc) are zero.  Wp: packed [tap][N][C].
 * C % 32 == 0, N % 32 == 0, ntaps <= 3.  src_off / wtap are HOST int arrays. */
int da_conv_gemm(const float* x, const float* w, float* y, int rows, int Lm, int Lsrc, int ldx, int C,
                 int Ldst, int ldy, int N, int dst_stride, int dst_off, int src_stride, int ntaps,
                 const int* src_off, const int* wtap, int accumulate, da_stream_t stream);

/* nn.Conv1d weight gradient (autograd of the calls above; train_ards_detector.py:163).
 * dW[co][ci][k] (torch layout) (+)= sum_positions dY[row][j*dy_stride+dy_off][co] *
 * X[row][j*src_stride+src_off[k]][ci].  workspace: da_conv_wgrad_workspace() bytes. */
size_t da_conv_wgrad_workspace(int rows, int Lm, int N, int C, int ntaps);
int da_conv_wgrad(const float* dy, const float* x, float* dw, float* workspace, int rows, int Lm, int Ldy,
                  int lddy, int N, int Lx, int ldx, int C, int dy_stride, int dy_off, int src_stride, int ntaps,
                  const int* src_off, int accumulate, da_stream_t stream);

/* benchmark-only tuning knobs: key 0 = force conv tile id, key 1 = wgrad target blocks (0 = automatic) */
int da_debug_set(int key, int value);

/* n <= 4 independent da_conv_gemm problems (jobs: HOST array, N % 64 == 0, disjoint outputs) in ONE launch: the
   stride-2 conv + 1x1 downsample of a block (same input), the even / odd sub-problems of a stride-2 data gradient */
typedef struct {
  const float* x; const float* w; float* y;
  int rows, Lm, Lsrc, ldx, C, Ldst, ldy, N, dst_stride, dst_off, src_stride, ntaps; int src_off[3]; int wtap[3];
  int accumulate;
  const float* x2; const float* w2; int tap_split;   /* optional: taps >= tap_split read x2 / w2 (same shapes): two
                                                        convolutions adding into one output as one contraction */
} da_conv_job;
int da_conv_gemm_multi(const da_conv_job* jobs, int n, da_stream_t stream);
/* k3 stride-1 pad-1 conv (forward, or data gradient with the transposed taps) as Winograd F(2,3): y (+)= conv(x);
   u = da_wino_weights() taps [4][N][C].  x: [rows][L][ldx] (C channels), y: [rows][L][ldy] (N channels).
   replaces nn.Conv1d(k=3, s=1, p=1) forward / input-grad, reference models/resnet.py:5-8,27-38, models/densenet.py:25-32 */
int da_conv3_winograd(const float* x, const float* u, float* y, int rows, int L, int ldx, int C, int ldy, int N,
                      int accumulate, da_stream_t stream);
/* tuning / tests: 0 = the partly filled last round of tiles is NOT cut into split-K half tiles (1 = default);
   2 / 3 = da_conv3_winograd4 with a K step of 32 / 16 (default) channels */
int da_wino_debug_tail(int on);
/* tuning: pchunk > 0 = output pairs per split of the Winograd weight gradient (default 512); pchunk < 0 = -pchunk padded
   positions per split of the bf16 weight gradient (default 2048) */
int da_wino_debug_tapmod(int mod);   /* timing experiment only: F(4,3) taps read modulo `mod` output channels (0: off) */
int da_wino_debug_pchunk(int pchunk);
/* u[4][co][ci] (transpose = 0, forward) or u[4][ci][co] (transpose = 1, data gradient) from w[co][ci][3] */
int da_wino_weights(const float* w, float* u, int co, int ci, int transpose, da_stream_t stream);
/* Winograd F(4,3) variants (half the direct conv's MFMAs): u = 6 * N * C floats. */
int da_conv3_winograd4(const float* x, const float* u, float* y, int rows, int L, int ldx, int C, int ldy, int N,
                       int accumulate, da_stream_t stream);
int da_wino4_weights(const float* w, float* u, int co, int ci, int transpose, da_stream_t stream);

/* ---- bf16 matrix arithmetic for the k3 s1 p1 convs (BASELINE config C3) ----------------------
 * same nn.Conv1d calls as da_conv3_winograd (resnet.py:5-8,27-38): fp32 activations in and out, rounded to bf16
 * (nearest-even) on the way into LDS, bf16 taps wpk [3][N][C] from da_pack_conv3_bf16, v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation.  C % 32 == 0, N % 64 == 0.  wf [3][Co][Ci] forward taps, wd [3][Ci][Co] data-gradient taps
 * (reversed); either may be NULL. */
int da_conv3_bf16(const da_act_t* x, const void* wpk, da_act_t* y, int rows, int L, int ldx, int C, int ldy, int N,
                  int accumulate, da_stream_t stream);
/* da_conv3_bf16 with a BatchNorm (windows of R rows, R * L >= 130) folded into either end -- resnet.py:27-33 conv1 -> bn1 ->
 * relu -> conv2 without a pass for bn1.  stat_part != NULL: the statistics records of y AS STORED
 * (da_stat_records_floats(rows * L, N) floats, units = positions) from the epilogue.  in_pend != NULL: x is a raw conv output
 * with records in_pend; relu(gamma (x - mean) invstd + beta) is applied while x is staged and (mean, invstd) are published to
 * in_mean / in_invstd [rows / R][C] (the backward and the running statistics read them). */
int da_conv3_bf16_bn(const da_act_t* x, const void* wpk, da_act_t* y, int rows, int L, int ldx, int C, int ldy, int N, int R,
                     const float* in_pend, float* in_mean, float* in_invstd, const float* gamma, const float* beta, float eps,
                     float* stat_part, da_stream_t stream);
int da_pack_conv3_bf16(const float* w, void* wf, void* wd, int co, int ci, da_stream_t stream);
/* ---- fp32 convolutions on the bf16 matrix cores with PRE-SPLIT operands (conv_x3p.hip; conv arithmetic 'f32x3p', opt-in) ----
 * Every operand is split exactly into three bf16 terms and a product taken as six bf16 MFMA products (the dropped ones
 * are below one fp32 rounding); fp32 in / out / sums.  The "x3" activation format: an fp32 activation stored as its exact three-term bf16 split, per position C/16 groups of
 * [h 16 ch | m 16 ch | l 16 ch] bf16 (3 C bf16 = 6 C bytes per position, no pitch).  The BatchNorm / pool kernels in front
 * of a k3 s1 p1 conv store it (da_bn_fwd_x / da_bn_bwd_x / da_bn_relu_pool_fwd_x below), da_x3_split / da_x3_merge convert
 * fp32 <-> x3 for tests and boundaries.  da_conv3_x3p: same nn.Conv1d calls as da_conv3_bf16 (resnet.py:5-8,27-38), x in
 * x3 format, y fp32; wpk: the chunked split-bf16 pack -- (N/64) x (C/16) chunks of 18 KB laid out
 * [3 taps][2 halves of 32 outputs][3 terms][64 lanes][8 bf16] -- which da_repack_desc.points = 49 emits (Uf forward,
 * Ud data gradient; Co, Ci multiples of 64).  C % 16 == 0, N % 64 == 0. */
/* The stride-2 block entry on x3 operands (conv arithmetic 'f32x3p'), ONE launch each way:
 * forward   y1 = Conv1d(C, N, 3, stride 2, pad 1)(x) and yd = Conv1d(C, N, 1, stride 2)(x)   -- reference
 *           models/resnet.py:16-19 (conv3x3 stride 2), :27-29, :123-131 (downsample) on the same block input;
 * backward  dx = their two data gradients summed, every position written.
 * x3 / dy1_3 / dyd_3: x3 activations (da_x3_split); packs: da_repack_multi points 49 (forward / data-gradient side; the
 * 1x1 weights pack with K = 1 into tap 1 of the chunks).  Lin even; C, N multiples of 64. */
int da_conv_x3p_s2_fwd(const void* x3, const void* w1pk, const void* wdpk, float* y1, float* yd, int rows, int Lin, int C, int N,
                       da_stream_t stream);
int da_conv_x3p_s2_dgrad(const void* dy1_3, const void* w1pk, const void* dyd_3, const void* wdpk, float* dx, int rows, int Lout,
                         int N, int C, da_stream_t stream);
int da_conv3_x3p(const void* x, const void* wpk, float* y, int rows, int L, int C, int ldy, int N, int accumulate,
                 da_stream_t stream);
int da_x3_split(const float* x, int ld, void* out, size_t npos, int C, da_stream_t stream);
int da_x3_merge(const void* x, float* out, int ld, size_t npos, int C, da_stream_t stream);
/* da_conv_gemm_multi's contract with bf16 operands for the stride-2 block heads and 1x1 downsamples
 * (resnet.py:5-8,126-128), forward and data gradient: per job Lsrc == src_stride * Lm, source offsets within a span of
 * 2, x2 == NULL, w = bf16 [taps][N][C] (da_repack_desc.points = 16 emits them for K = 1 and K = 3); the jobs of a call
 * share src_stride (1 or 2); up to 4 per launch. */
int da_conv_bf16_multi(const da_conv_job* jobs, int n, da_stream_t stream);   /* jobs[].x / .y are da_act_t tensors here */
/* all weight-gradient GEMMs of a step in one launch per tile shape (jobs: HOST array); slabs only, reduce afterwards */
typedef struct {
  const float* dy; const float* x; float* workspace;
  int rows, Lm, Ldy, lddy, N, Lx, ldx, C, dy_stride, dy_off, src_stride, ntaps; int src_off[3];
  int winograd;   /* != 0: k3 s1 p1 job (N, C multiples of 64) in Winograd F(2,3) form; plan with winograd = 1;
                     16: bf16 operands; 49: split-bf16 (fp32-equivalent) products with dy AND x in the x3
                     format (see da_conv3_x3p; lddy == N, ldx == C), k3 s1 p1 only */
  /* dense-block operand forms of stride-1 jobs on the direct kernels (winograd == 0; see da_conv1x1_bn): xform = 1: X is
     relu(BatchNorm(x)) recomputed while staged from the statistics tables [rows * Lm / Wn][ldstat] (Wn >= 32);
     dy_half = 1: dY has Ldy = Lm / 2 positions per row, position j reads dy[j / 2] / 2 */
  int xform, dy_half, Wn, ldstat; const float* mean; const float* invstd; const float* gamma; const float* beta;
} da_wgrad_job;
int da_conv_wgrad_multi(const da_wgrad_job* jobs, int n, da_stream_t stream);  /* winograd == 16 jobs: dy / x are da_act_t tensors (the only kind accepted while bf16 is selected) */
/* the same call with the slab reductions chained: dws[i] = job i's gradient destination dW[co][ci][k] (or NULL); the
   reduction dws[i] (+)= sum of job i's slabs rides as the FIRST blocks of the NEXT launch of this call (memory-bound blocks
   beside matrix-bound ones, on slabs still in the Infinity Cache) and reduced[i] = 1; reduced[i] = 0: the caller still
   owes it (the jobs of the call's last launch, always) -- da_wgrad_reduce_multi / da_step_tail_multi.  The sums are the
   ones da_wgrad_reduce_multi forms, bit for bit.  (replaces nothing of its own in the reference: the reduction of
   loss.backward()'s conv weight gradients, train_ards_detector.py:161-173) */
int da_conv_wgrad_multi_reduce(const da_wgrad_job* jobs, int n, float* const* dws, int accumulate, int* reduced, int* splits, da_stream_t stream);
/* splits[i] (out): the slabs job i wrote -- what its reduction must be given.  A job's workspace here must hold TWICE
   da_conv_wgrad_plan's slabs: the winograd == 1 jobs of the launch's last, partly filled round run with half the pairs per
   split (twice the slabs); the dense-block jobs of a call with 8 or more of them share ONE launch, are planned as a batch
   and write fewer. */
/* deferred slab reduction: da_conv_wgrad with dw == NULL leaves da_conv_wgrad_splits() slabs in the workspace */
int da_conv_wgrad_splits(int rows, int Lm, int N, int C, int ntaps);
/* host only: out[4] = {tile_n, tile_c, splits, positions per split} the plan of da_conv_wgrad and
   da_conv_wgrad_multi for this shape; a job's workspace is splits * ntaps*N*C floats */
int da_conv_wgrad_plan(int rows, int Lm, int N, int C, int ntaps, int winograd, int* out);
typedef struct { const float* slab; float* dw; int splits, ntaps, N, C; } da_wgrad_reduce_desc;
int da_wgrad_reduce_multi(const da_wgrad_reduce_desc* descs, int n, int accumulate, da_stream_t stream);
/* The tail of a training step in ONE launch: the slab reductions, every BatchNorm's dgamma / dbeta fold
   (da_bn_param_grad_multi) and running-statistics update (da_bn_running_multi), and -- stem_partial != NULL -- the fold of
   the stem's weight-gradient partials that da_stem_bwd(dw = NULL) left behind (da_stem_bwd_partials says where).  n <= 32
   reductions and <= 24 BatchNorms of each kind share the launch, anything else runs as the separate calls. */
struct da_bn_pgrad_desc_;
struct da_bn_running_desc_;
int da_step_tail_multi(const da_wgrad_reduce_desc* descs, int n, const struct da_bn_pgrad_desc_* pg, int npg,
                       const struct da_bn_running_desc_* run, int nrun, const float* stem_partial, int stem_nblk, int stem_n,
                       float* stem_dw, int accumulate, da_stream_t stream);

/* torch [Co][Ci][K] -> Wf [K][Co][Ci] (forward) and Wd [K][Ci][Co] (data gradient). */
int da_repack_conv_weight(const float* W, float* Wf, float* Wd, int Co, int Ci, int K, da_stream_t stream);
/* Uf / Ud (K == 3 only, may be NULL): the Winograd taps of da_wino_weights (points 0 / 4) or da_wino4_weights
   (points == 6) for the forward / data gradient; points == 16: Uf / Ud point at bf16 buffers and receive the tap packs
   of da_pack_conv3_bf16 (K = 3) or [1][Co][Ci] / [1][Ci][Co] (K = 1); Co and Ci multiples of 32 */
typedef struct { const float* W; float* Wf; float* Wd; float* Uf; float* Ud; int Co, Ci, K; int points; } da_repack_desc;
int da_repack_multi(const da_repack_desc* descs, int n, da_stream_t stream);

/* ---- stem: Conv1d(1, C0, k7, s2, p3)  resnet.py:86-87,142 ; densenet.py:118-119 ----------- */
int da_stem_conv_fwd(const float* x, const float* w, da_act_t* y, int rows, int Lin, int C0, int ldy,
                     da_stream_t stream);
size_t da_stem_wgrad_workspace(int rows, int C0);
int da_stem_conv_wgrad(const da_act_t* dy, int lddy, const float* x, float* dw, float* workspace, int rows, int Lin,
                       int C0, int accumulate, da_stream_t stream);
/* The same for the other first convolutions the constructors can select: x [rows][Cin][Lin] (the NCL rows the dataset
 * hands over), w [C0][Cin][K], pad = K / 2, Lin % stride == 0 -> y [rows][Lin / stride][ldy].  Instantiated shapes
 * (Cin, K, stride): (1, 7, 2) the default; (2, 7, 2) / (3, 7, 2) DenseNet(only_fft / with_fft) conv0, densenet.py:109-119;
 * (1, 3, 1) ResNet(double_conv_first).conv1_alt, resnet.py:88-89,145.  Any other shape returns -1. */
int da_stem_conv_fwd_g(const float* x, const float* w, da_act_t* y, int rows, int Lin, int Cin, int K, int stride, int C0,
                       int ldy, da_stream_t stream);
size_t da_stem_wgrad_workspace_g(int rows, int C0, int Cin, int K);
int da_stem_conv_wgrad_g(const da_act_t* dy, int lddy, const float* x, float* dw, float* workspace, int rows, int Lin,
                         int Cin, int K, int stride, int C0, int accumulate, da_stream_t stream);

/* ---- window-grouped train-mode BatchNorm1d (+ReLU, +residual) ------------------------------
 * resnet.py:27-38,143,152 ; densenet.py:23-29,72-74,146 ; per-window statistics because
 * torch_cnn_linear_network.py:108-113 calls breath_block(x[i]) one window at a time. */
/* descriptors for the batched small kernels: HOST arrays of these are passed, 32 served per launch */
typedef struct da_bn_running_desc_ {
  const float* mean; const float* invstd; float* running_mean; float* running_var;
  long long* num_batches_tracked; int W, C, Wn; float eps, momentum;
} da_bn_running_desc;
typedef struct da_bn_pgrad_desc_ { const float* s1; const float* s2; float* dgamma; float* dbeta; int W, C; } da_bn_pgrad_desc;

/* two-stage statistics: P chunks of `chunk` positions per window so that W*C/32*P blocks fill the chip */
void da_bn_chunks(int W, int Wn, int C, int* P, int* chunk);
size_t da_bn_workspace(int W, int Wn, int C);      /* bytes of part[w][p][{mean,M2}][C] / backward scratch */
int da_bn_stats_partial(const da_act_t* x, int ld, int W, int Wn, int C, float* part, da_stream_t stream);
int da_bn_stats_merge(const float* part, int W, int Wn, int C, float eps, float* mean, float* invstd,
                      da_stream_t stream);
/* the reference's W sequential momentum-0.1 updates per BatchNorm in closed form; num_batches_tracked += W */
int da_bn_running_multi(const da_bn_running_desc* descs, int n, da_stream_t stream);
/* out = act(bn(x) (+res)); with part != NULL the chunk records are merged on the fly and mean/invstd WRITTEN */
int da_bn_apply(const da_act_t* x, int ldx, const da_act_t* res, int ldr, da_act_t* out, int ldo, int W, int Wn, int C,
                float* mean, float* invstd, const float* gamma, const float* beta, int relu, const float* part,
                float eps, da_stream_t stream);
/* statistics + normalisation in one call (mean/invstd [W][C] are OUTPUTS): one single-pass kernel when a window
   slab fits a block's registers (Wn <= 1280), else da_bn_stats_partial + da_bn_apply.  scratch: da_bn_workspace().
   replaces nn.BatchNorm1d(+ReLU)(+residual) forward, reference models/resnet.py:27-38, models/densenet.py:23-29 */
int da_bn_fwd(const da_act_t* x, int ldx, const da_act_t* res, int ldr, da_act_t* out, int ldo, int W, int Wn, int C,
              float* mean, float* invstd, const float* gamma, const float* beta, int relu, float eps, float* scratch,
              da_stream_t stream);
/* tests: on != 0 forces the two-stage kernels in da_bn_fwd / da_bn_bwd (both paths are checked against the oracle) */
int da_bn_debug_two_stage(int on);
/* tuning: blocks per launch the single-pass BatchNorm geometry aims for before its channel group stops shrinking
   (default 256 = one (window, 16-32 channel) slab per CU) */
int da_bn_debug_target_blocks(int blocks);
/* mask_mode 0: no ReLU; 1: ReLU, mask recomputed from bn(x); 2: ReLU, mask from `out` (residual).
 * scratch: da_bn_workspace() bytes.  ds: [2][W][C] per-window totals, always written.  dgamma/dbeta NULL:
 * fold ds later with da_bn_param_grad_multi. */
int da_bn_bwd(const da_act_t* dout, int ldd, const da_act_t* x, int ldx, const da_act_t* out, int ldo, da_act_t* dx, int lddx,
              da_act_t* gout, int ldg, int W, int Wn, int C, const float* mean, const float* invstd,
              const float* gamma, const float* beta, int mask_mode, float* scratch, float* ds, float* dgamma,
              float* dbeta, int accumulate, da_stream_t stream);
/* ReLU decisions as a bit mask, 64 bits per thread of the single-pass kernels (da_bn_mask_words() words; 0 = this shape
   takes the two-stage kernels, no mask form): da_bn_fwd_mask = da_bn_fwd(relu) + mask; da_bn_bwd_mask = da_bn_bwd of a
   ReLU'd BatchNorm(+residual) that reads the mask instead of the output tensor (mask_mode 2 reads `out` only for its sign) */
size_t da_bn_mask_words(int W, int Wn, int C);
int da_bn_fwd_mask(const da_act_t* x, int ldx, const da_act_t* res, int ldr, da_act_t* out, int ldo, int W, int Wn, int C,
                   float* mean, float* invstd, const float* gamma, const float* beta, float eps, float* scratch,
                   unsigned long long* mask, da_stream_t stream);
int da_bn_bwd_mask(const da_act_t* dout, int ldd, const da_act_t* x, int ldx, da_act_t* dx, int lddx, da_act_t* gout, int ldg, int W,
                   int Wn, int C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                   float* scratch, float* ds, float* dgamma, float* dbeta, int accumulate,
                   const unsigned long long* mask, da_stream_t stream);
/* The block-output BatchNorm of the LAST BasicBlock with the head's AvgPool1d(L) folded in (resnet.py:33-38 into
   :112,159-160): da_bn_fwd_mask(relu) whose output map is never stored -- flat[W * Wn / L][C] (always float) receives the
   average over the L positions of every row, bit for bit what da_head_fwd pools from the stored map -- and its backward,
   da_bn_bwd_mask whose dout is dflat[rows][ldd] (float), the gradient of those pooled features.  da_bn_pool_ok: single-pass
   geometry, L | Wn, Wn <= 160. */
int da_bn_pool_ok(int W, int Wn, int C, int L);
int da_bn_fwd_pool(const da_act_t* x, int ldx, const da_act_t* res, int ldr, float* flat, int W, int Wn, int C, int L, float* mean,
                   float* invstd, const float* gamma, const float* beta, float eps, unsigned long long* mask, da_stream_t stream);
int da_bn_bwd_pool(const float* dflat, int ldd, const da_act_t* x, int ldx, da_act_t* dx, int lddx, da_act_t* gout, int ldg, int W,
                   int Wn, int C, int L, const float* mean, const float* invstd, const float* gamma, const float* beta, float* ds,
                   const unsigned long long* mask, da_stream_t stream);
/* da_bn_bwd_mask / da_bn_bwd_pair with the upstream gradient as TWO terms, dout + dout2 (resnet.py:33-38 backward: the gradient
   of a block's output = the next block's data-gradient conv output + its identity branch; summed here, the conv needs no
   accumulating epilogue).  da_bn_two_ok: single-pass geometry of at most 512 threads.  ds only. */
int da_bn_two_ok(int W, int Wn, int C);
int da_bn_bwd_mask2(const da_act_t* dout, int ldd, const da_act_t* dout2, int ldd2, const da_act_t* x, int ldx, da_act_t* dx, int lddx,
                    da_act_t* gout, int ldg, int W, int Wn, int C, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, float* ds, const unsigned long long* mask, da_stream_t stream);
/* da_bn_bwd with dx = input gradient + add[pos][0:C] (pitch ldadd): a concatenation's pass-through gradient
   (densenet.py:41) joins in the same pass */
int da_bn_bwd_add(const da_act_t* dout, int ldd, const da_act_t* x, int ldx, const da_act_t* out, int ldo, da_act_t* dx, int lddx,
                  da_act_t* gout, int ldg, int W, int Wn, int C, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, int mask_mode, float* scratch, float* ds, float* dgamma, float* dbeta,
                  int accumulate, const da_act_t* add, int ldadd, da_stream_t stream);
/* ---- the dense block as one design (reference models/densenet.py:18-44 _DenseLayer, :46-66 _DenseBlock, :68-81 _Transition)
 * One pitched buffer [rows][L][Cb] per block: the growth conv writes its 32 new channels at their offset (with F.dropout in
 * its epilogue), per-(window, channel) statistics live in ONE pitched table per block ([W][ldstat] mean / invstd) and are
 * reused by every later norm1 and the transition norm, and h = relu(norm1(x)) is never stored: the 1x1 conv applies it while
 * staging (da_conv1x1_bn), its weight gradient recomputes it (da_wgrad_job.xform), the BatchNorm backward takes the ReLU
 * decision from the same fused multiply-add (da_bn_bwd_ss).  fp32 activations, single-pass BatchNorm geometry
 * (da_bn_mask_words() > 0) only; callers keep the per-op entry points above for every other shape. */
/* statistics only: mean / invstd of x[:, 0:C] per window, written into the pitched tables */
int da_bn_stats_fused(const float* x, int ldx, int W, int Wn, int C, float* mean, float* invstd, int ldstat, float eps,
                      da_stream_t stream);
/* out[:, 0:C] = max(fmaf(x, gamma invstd, beta - mean gamma invstd), 0): the un-stored activation, for tests / explainers */
int da_bn_relu_ss(const float* x, int ldx, float* out, int ldo, int W, int Wn, int C, const float* mean, const float* invstd,
                  int ldstat, const float* gamma, const float* beta, da_stream_t stream);
/* backward of relu(norm(x)) (relu = 1: ReLU decision from the fused multiply-add form, the forward of da_conv1x1_bn;
 * relu = 2: from the sign of the stored output `out`, the forward of da_bn_fwd) or norm(x) (relu = 0): statistics from the
 * pitched tables; half_dout: dout has Wn / 2 positions per window, g[p] = dout[p / 2] / 2 (a transition's AvgPool1d(2,2) in
 * front of its conv); dx = input gradient (+ add[:, 0:C]; dx may alias add), then with drop_p > 0 the dropout mask
 * (da_dropout's, seed / salt, contiguous [W Wn][drop_g]) on dx's channels [C - drop_g, C); ds [2][W][C] window sums;
 * hout != NULL (relu = 1): the activation relu(norm(x)) itself, which the forward never stored, is written there for the
 * weight gradient of the conv behind it (the residual blocks' bn1 under da_conv3_bf16_bn) */
int da_bn_bwd_ss(const da_act_t* dout, int ldd, const da_act_t* x, int ldx, const da_act_t* out, int ldo, da_act_t* dx, int lddx,
                 const da_act_t* add, int ldadd, int W, int Wn, int C, const float* mean, const float* invstd, int ldstat,
                 const float* gamma, const float* beta, int relu, int half_dout, const long long* drop_seed, unsigned drop_salt,
                 float drop_p, int drop_g, float* ds, da_act_t* hout, int ldh, da_stream_t stream);
/* y[m][0:N] (pitch ldy) = sum_c w[n][c] relu(norm(x))[m][c], the activation applied while x (first C channels, pitch ldx) is
 * staged; pool != 0: the transition form, (h[2m] + h[2m+1]) / 2 in front of the conv (Lin even, Lin / 2 outputs per row).
 * w [N][C] = the torch weight of the k = 1 conv as it lies.  N % 64 == 0, C % 32 == 0, R * Lout >= 64.
 * replaces norm1 -> relu1 -> conv1 (densenet.py:23-26) and norm -> relu -> conv -> pool (:72-79) */
int da_conv1x1_bn(const float* x, int ldx, const float* w, float* y, int ldy, int rows, int R, int Lin, int C, int N, int pool,
                  float* mean, float* invstd, int ldstat, const float* gamma, const float* beta, const float* pend, int pend_c0,
                  long pend_units, int pend_Wu, float eps, float* out_part, da_stream_t stream);
/* the growth conv on the 1x1 conv's OUTPUT x with relu(norm2(x)) applied while x is staged (densenet.py:27-32 norm2 -> relu2
 * -> conv2: the activation is never stored): x [rows][L][C] contiguous, C <= 128; statistics of x from the records in_pend
 * (da_conv1x1_bn out_part of the conv in front), published to in_mean / in_invstd [rows / R][C]; dropout / output records as
 * da_conv3_winograd_drop.  R * ceil(L / 2) >= 64. */
int da_conv3_winograd_bn(const float* x, const float* u, float* y, int rows, int L, int C, int ldy, int N, int R,
                         const float* in_pend, float* in_mean, float* in_invstd, const float* gamma, const float* beta,
                         float eps, const long long* drop_seed, unsigned drop_salt, float drop_p, float* stat_part,
                         da_stream_t stream);
/* Statistics records: a kernel that WRITES activation channels can hand their per-window BatchNorm statistics over from its
 * epilogue -- per 64-unit tile and window slot (a tile touches <= 2 windows) the count, mean and centred second moment of
 * each channel: floats [tiles][2][{mean, M2}][N] then counts [tiles][2] (da_stat_records_floats).  The consuming
 * da_conv1x1_bn (pend != NULL: channels [pend_c0, C) of its input, pend_units units in windows of pend_Wu >= 64) merges
 * them per window in tile order (Chan's update, as da_bn_stats_merge) and publishes mean / invstd to the tables, which
 * every later kernel reads.  Producers: da_conv3_winograd_drop (units = output PAIRS, rows * ceil(L / 2)) and
 * da_conv1x1_bn(out_part != NULL) (units = its output positions). */
size_t da_stat_records_floats(long units, int N);
/* da_conv3_winograd + F.dropout(p) in the epilogue (mask of da_dropout on the contiguous [rows L][N] tensor), y at pitch ldy:
 * a _DenseLayer's growth conv storing its new features at their channel offset (densenet.py:30-40) */
int da_conv3_winograd_drop(const float* x, const float* u, float* y, int rows, int L, int ldx, int C, int ldy, int N,
                           const long long* drop_seed, unsigned drop_salt, float drop_p, float* stat_part, int R,
                           da_stream_t stream);     /* stat_part != NULL: + the records of y for windows of R rows */

int da_bn_param_grad_multi(const da_bn_pgrad_desc* descs, int n, int accumulate, da_stream_t stream);
/* Two BatchNorms of ONE geometry (W, Wn, C) in one launch -- a stride-2 BasicBlock entry (resnet.py:27-29,33-38,126-128):
 * forward: bn1 + ReLU on conv1's output and the downsample's BatchNorm on the 1x1 conv's (d[0], d[1] as da_bn_fwd /
 * da_bn_fwd_mask take them); backward: the block-output BatchNorm (bn2) and the downsample's take the SAME masked gradient
 * dout * [out > 0] (mask = the ReLU bit mask of da_bn_fwd_mask), ds_i [2][W][C] for da_bn_param_grad_multi.
 * Single-pass geometry only (da_bn_mask_words(W, Wn, C) > 0, else -1: the caller issues the two calls). */
typedef struct {
  const da_act_t* x; int ldx; const da_act_t* res; int ldr; da_act_t* out; int ldo; float* mean; float* invstd;
  const float* gamma; const float* beta; int relu; unsigned long long* mask;
} da_bn_fwd_desc;
typedef struct {
  const da_act_t* x; int ldx; da_act_t* dx; int lddx; const float* mean; const float* invstd; const float* gamma;
  const float* beta; float* ds;
} da_bn_bwd_desc;
int da_bn_fwd_pair(const da_bn_fwd_desc* d, int W, int Wn, int C, float eps, da_stream_t stream);
int da_bn_bwd_pair(const da_act_t* dout, int ldd, const da_bn_bwd_desc* d, int W, int Wn, int C,
                   const unsigned long long* mask, da_stream_t stream);
int da_bn_bwd_pair2(const da_act_t* dout, int ldd, const da_act_t* dout2, int ldd2, const da_bn_bwd_desc* d, int W, int Wn, int C,
                    const unsigned long long* mask, da_stream_t stream);      /* upstream gradient dout + dout2 (da_bn_bwd_mask2) */
/* The same BatchNorm forward / backward (resnet.py:27-38) in front of an x3 consumer (conv arithmetic 'f32x3', see
 * da_conv3_x3p): float in, and res_x3 / out_x3 / dx_x3 say which of `res`, `out`, `dx` are in the x3 format (their pitches are
 * ignored).  Single-pass geometry only (da_bn_mask_words() > 0; -1 otherwise); mask may be NULL (no ReLU bit mask wanted /
 * mask_mode 0 or 1 in the backward). */
int da_bn_fwd_x(const float* x, int ldx, const void* res, int ldr, void* out, int ldo, int W, int Wn, int C, float* mean,
                float* invstd, const float* gamma, const float* beta, int relu, float eps, unsigned long long* mask,
                int res_x3, int out_x3, da_stream_t stream);
int da_bn_bwd_x(const float* dout, int ldd, const float* x, int ldx, void* dx, int lddx, float* gout, int ldg, int W, int Wn,
                int C, const float* mean, const float* invstd, const float* gamma, const float* beta, int mask_mode,
                float* ds, const unsigned long long* mask, int dx_x3, da_stream_t stream);

/* ---- pools ------------------------------------------------------------------------------
 * stem BN+ReLU+{Max,Avg}Pool1d(3,2,1): resnet.py:100-104,152-153 ; densenet.py:120-123 */
int da_bn_relu_pool_fwd(const da_act_t* y, int ldy, da_act_t* out, int ldo, int rows, int R, int Lin, int C,
                        const float* mean, const float* invstd, const float* gamma, const float* beta,
                        int pool_mode, da_stream_t stream);
/* ... with the pooled output stored in the x3 format (float activations): layer1's input under conv arithmetic 'f32x3' */
/* The recomputing default stem (conv k7 s2 p3 on ONE input channel -> BatchNorm -> ReLU -> pool(3,2,1): reference
 * models/resnet.py:86-87,100-104,141-153, models/densenet.py:118-124): the conv output is 36.7 MB at B = 64 and costs 7 FMAs
 * an element, so these entry points take the RAW rows xrows (rows, Lin) and the weights wt (C, 1, 7) and recompute it wherever
 * it is needed instead of storing and re-reading it.  Forward: da_stem_stats_partial (chunk records as da_bn_stats_partial's
 * on the stored map, bit for bit) -> da_bn_stats_merge -> da_stem_bn_relu_pool_fwd (the pooled map, float or x3: bit for bit
 * da_stem_conv_fwd + da_bn_relu_pool_fwd).  Backward: da_stem_bwd = da_pool_bwd + da_bn_bwd + da_stem_conv_wgrad in two passes
 * over dout: dw (C, 1, 7) and ds (2, W, C) for da_bn_param_grad_multi; workspace da_stem_bwd_workspace() bytes.  out / dout
 * in the activation storage type (with bf16 storage the conv output, being recomputed in fp32, is never rounded -- the
 * stored-map stem rounds it); Lin even, C a multiple of 32 (statistics) with 2 C <= 256. */
int da_stem_stats_partial(const float* xrows, const float* wt, int rows, int R, int Lin, int C, float* part, da_stream_t stream);
int da_stem_bn_relu_pool_fwd(const float* xrows, const float* wt, da_act_t* out, int ldo, int rows, int R, int Lin, int C,
                             const float* mean, const float* invstd, const float* gamma, const float* beta, int pool_mode,
                             int out_x3, da_stream_t stream);
size_t da_stem_bwd_workspace(int rows, int C);
/* da_stem_bwd with dw == NULL leaves its weight-gradient partials [nblk][C * 7] at workspace + *offset (floats): fold them
   with da_step_tail_multi (the step's tail launch) or da_stem_wgrad_reduce */
int da_stem_bwd_partials(int rows, int R, int C, size_t* offset, int* nblk);
int da_stem_wgrad_reduce(const float* partial, int nblk, int n, float* dw, int accumulate, da_stream_t stream);
int da_stem_bwd(const da_act_t* dout, int ldd, const float* xrows, const float* wt, int rows, int R, int Lin, int C,
                const float* mean, const float* invstd, const float* gamma, const float* beta, int pool_mode, float* ds,
                float* dw, int accumulate, float* workspace, da_stream_t stream);
int da_bn_relu_pool_fwd_x(const float* y, int ldy, void* out, int rows, int R, int Lin, int C, const float* mean,
                          const float* invstd, const float* gamma, const float* beta, int pool_mode, da_stream_t stream);
int da_pool_bwd(const da_act_t* dout, int ldd, const da_act_t* y, int ldy, da_act_t* dz, int lddz, int rows, int R, int Lin,
                int C, const float* mean, const float* invstd, const float* gamma, const float* beta, int pool_mode,
                da_stream_t stream);
/* AvgPool1d(k, stride k): densenet.py:79 (k=2) ; AvgPool1d(7,1) on L=7: resnet.py:112,159, densenet.py:167,183 */
int da_avgpool_fwd(const float* x, int ldx, float* out, int ldo, int rows, int Lin, int k, int C,
                   da_stream_t stream);
int da_avgpool_bwd(const float* dout, int ldd, float* dx, int lddx, int rows, int Lin, int k, int C,
                   da_stream_t stream);
/* AvgPool1d(k, stride 1) on a map longer than k followed by x.view(x.size(0), -1) (resnet.py:159-160,
 * densenet.py:183-184 when seq_len > 224): feat[row][c * Lout + j], Lout = Lin - k + 1. */
int da_avgpool_slide_fwd(const da_act_t* x, int ldx, float* feat, int rows, int Lin, int k, int C, da_stream_t stream);
int da_avgpool_slide_bwd(const float* dfeat, da_act_t* dx, int lddx, int rows, int Lin, int k, int C,
                         da_stream_t stream);
/* The activation -> feature boundary, AvgPool1d(L, stride 1) on an L-long map + view (resnet.py:112,159-160,
 * densenet.py:167,183-184): x in the activation storage type, feat[row][c] always float; and its backward. */
int da_global_avgpool_fwd(const da_act_t* x, int ldx, float* feat, int rows, int L, int C, da_stream_t stream);
int da_global_avgpool_bwd(const float* dfeat, da_act_t* dx, int lddx, int rows, int L, int C, da_stream_t stream);


/* ---- head + loss ---------------------------------------------------------------------------
 * linear_final on view(-1) of the (NB,F) block: torch_cnn_linear_network.py:102,110-112 ;
 * BCEWithLogitsLoss (mean): train_ards_detector.py:530,929-930 */
int da_linear2_fwd(const float* flat, const float* W, const float* bias, float* logits, int B, int K,
                   da_stream_t stream);
int da_bce_logits(const float* logits, const float* target, int n, float gscale, float* loss, float* dlogits,
                  da_stream_t stream);
int da_linear2_bwd(const float* dlogits, const float* flat, const float* W, float* dflat, float* dW, float* dbias,
                   int B, int K, int accumulate, da_stream_t stream);
/* The head chain of CNNLinearNetwork in two launches instead of six: da_head_fwd = AvgPool1d(L,1) + view -> flat and the
 * row groups' shares `part` [B][da_head_groups(R, F)][2] of linear_final's two dot products (finish != 0, forward-only callers:
 * also logits and the BCEWithLogitsLoss mean); da_head_bwd = logits / loss terms / dlogits from `part`, linear + pool
 * backward in one kernel, then dW / dbias and the loss mean (resnet.py:112,159-160 / densenet.py:167,183-184;
 * torch_cnn_linear_network.py:102,110-112; train_ards_detector.py:530).  x / dx: the breath block's last map
 * [B * R][L][ld] in the activation storage type. */
int da_head_groups(int R, int F);
int da_head_fwd(const da_act_t* x, int ldx, const float* W, const float* bias, const float* target, float* flat, float* part,
                float* logits, float* loss, int B, int R, int L, int F, int finish, da_stream_t stream);
int da_head_bwd(const float* part, const float* bias, const float* target, const float* flat, const float* W, da_act_t* dx,
                int lddx, float* logits, float* dlogits, float* terms, float* dW, float* dbias, float* loss, int B, int R, int L,
                int F, float gscale, int accumulate, da_stream_t stream);
/* the same two calls on features da_bn_fwd_pool pooled already: flat_in [B * R][F] (float) in the place of the map (a map of
   ONE position), dflat [B * R][F] (float) in the place of dx -- whatever the activation storage type */
int da_head_flat_fwd(const float* flat_in, const float* W, const float* bias, const float* target, float* flat, float* part,
                     float* logits, float* loss, int B, int R, int F, int finish, da_stream_t stream);
int da_head_flat_bwd(const float* part, const float* bias, const float* target, const float* flat, const float* W, float* dflat,
                     float* logits, float* dlogits, float* terms, float* dW, float* dbias, float* loss, int B, int R, int F,
                     float gscale, int accumulate, da_stream_t stream);


/* ---- optimiser: clamp hook + SGD(momentum .9, nesterov, weight decay) / Adam, fused ----------
 * train_ards_detector.py:474-476 (clamp), :419-421 (optimisers).  gscale = 1/world_size after the
 * gradient all-reduce (the clamp runs AFTER the reduce, SURVEY.md finding 7). */
int da_clamp_sgd_nesterov(float* p, const float* g, float* buf, size_t n, float lr, float momentum,
                          float weight_decay, float clip, float gscale, int first, da_stream_t stream);
int da_clamp_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                  float eps, int step, float clip, float gscale, da_stream_t stream);
/* the same with the step count in device memory (int64, incremented by the call): graph-replayable */
int da_clamp_adam_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                      long long* step, float clip, float gscale, da_stream_t stream);

/* ---- device-resident window store: batch gather fused with the (x-mu)/std normalisation -------------------
 * ARDSRawDataset.__getitem__ dataset.py:1343-1404 (index map :1349-1350, normalisation :1364,1379 in float64)
 * + the .float() cast of train_ards_detector.py:150-152; replaces DataLoader/collate/H2D per step. */
int da_gather_normalize(const double* tiles, const int64_t* idx, double mu, double stdv, float* out, int B,
                        int tile_elems, da_stream_t stream);
/* windows with C <= 4 channels, [N][NB][C][L]: the flow channel plus the real / imaginary channels of its spectrum
 * (ARDSRawDataset._perform_fft dataset.py:1330-1341, selected by --with-fft / --only-fft / --fft-real-only,
 * train_ards_detector.py:228-230); every channel has its own factors (dataset.py:627-649).  mu / stdv: HOST arrays [C]. */
int da_gather_normalize_ch(const double* tiles, const int64_t* idx, const double* mu, const double* stdv, float* out, int B,
                           int NB, int C, int L, da_stream_t stream);
/* ---- sibling heads of CNNLinearNetwork (torch_cnn_linear_network.py:7-89) ---------------------- */
/* CNNLinearComprToRF: lower median over the NB breath rows of each window (torch.median(outputs, dim=1)[0], :47);
   x [B*NB][ld], out [B][F], idx [B][F] = selected row (for the backward); NB <= 64.  The mean of CNNLinearToMean
   (:25) is da_avgpool_fwd over the breath axis; the per-breath / double linear heads (:67,:87) are da_linear2_*. */
int da_window_median_fwd(const float* x, int ld, int B, int NB, int F, float* out, int* idx, da_stream_t stream);
int da_window_median_bwd(const float* dout, const int* idx, int B, int NB, int F, float* dx, int ld, da_stream_t stream);

/* ---- LSTM head of CNNLSTMNetwork (torch_cnn_lstm_combo.py:6-50): nn.LSTM(F, H, 1 layer, batch_first), gates i,f,g,o ---
   the recurrence only: the input projection gx = x W_ih^T and the weight / input gradients are 1x1-conv GEMMs */
int da_lstm_fwd(const float* gx, const float* whh, const float* bih, const float* bhh, const float* h0, const float* c0,
                float* hs, float* cs, float* gates, float* hT, float* cT, int B, int T, int H, da_stream_t stream);
int da_lstm_bwd(const float* dh_all, const float* whh, const float* hs, const float* cs, const float* gates,
                const float* h0, const float* c0, float* dgates, float* dwhh_part, int B, int T, int H,
                da_stream_t stream);
/* out[n] (+)= column sums of m [rows][n] in a fixed order */
int da_reduce_rows(const float* m, int rows, int n, float* out, int accumulate, da_stream_t stream);

int da_gather_rows(const float* src, const int64_t* idx, float* out, int B, int width, da_stream_t stream);

/* ---- test epoch on the device: window predictions + per-patient vote table --------------------------------
 * CNNLinearModel._process_test_batch_results train_ards_detector.py:932-936 (outputs.argmax) and
 * DeepARDSResults.perform_patient_predictions metrics.py:572-604 (votes per pathology, pred_frac, majority). */
int da_vote_counts(const float* logits, const int64_t* group, int B, int n_groups, int* votes, int* pred,
                   da_stream_t stream);

/* _DenseLayer's dropout fused into its neighbours (densenet.py:38-41): concat with dropout on the new features, and
   the backward's slice of the concatenated gradient with the same mask (that of da_dropout on a contiguous tensor) */
int da_concat2_dropout(const float* a, int lda, int C1, const float* b, int ldb, int C2, float* out, int ldo, size_t npos,
                       const int64_t* seed, unsigned salt, float p, da_stream_t stream);
int da_slice_dropout(const float* src, int lds, int off, float* dst, int ldd, int C, size_t npos, const int64_t* seed,
                     unsigned salt, float p, da_stream_t stream);
/* ---- densenet helpers: torch.cat([x, new], 1) and F.dropout  densenet.py:36-40 -------------- */
int da_concat2(const float* a, int lda, int C1, const float* b, int ldb, int C2, float* out, int ldo, size_t npos,
               da_stream_t stream);
int da_slice_copy(const float* src, int lds, int off, float* dst, int ldd, int C, size_t npos, int accumulate,
                  da_stream_t stream);
int da_dropout(const float* x, float* y, size_t n, const int64_t* seed, unsigned salt, float p,
               da_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPARDS_HIP_H */
