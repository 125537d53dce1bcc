/* sfnative.h — C ABI of libsfnative.so: the MI355X (gfx950) native GRU-ODE future-state path of
 * StreamingFlow.  Every entry point takes raw device pointers, sizes and a hipStream_t (passed as
 * void*), enqueues work on that stream and returns 0 on success or a negative sf_status.  No
 * entry point allocates, synchronises or keeps global mutable state, so a caller may capture any
 * sequence of them into a hipGraph (sf_graph_* below).
 *
 * Process model: one process per GPU and one host thread launching on a device at a time (the only process-wide
 * state are first-launch flags per device for the kernels' dynamic-LDS attribute and the opt-in profiler / debug
 * stamps below); calls on different streams of one device may be enqueued from that thread and overlap.
 *
 * The reference (synsin0/StreamingFlow) is pure Python/PyTorch on this path — it has no FFI.  The
 * functions below are therefore the operator boundary a maintainer would bind (ctypes stub in
 * INTEGRATION.md); each cites the reference function it replaces (paths relative to the reference
 * root).
 *
 * Data layout.  Activations are fp32 NHWC: a tensor of n images is [n][H][W][C] contiguous, C a
 * multiple of 8.  The reference API is NCHW; sf_nchw_to_nhwc / sf_nhwc_to_nchw convert at the
 * module boundary.
 *
 * Packed convolution weights (sf_conv_w).  A reference Conv2d weight [cout][cin][kh][kw] is stored
 * as w[cout_pad][kh*kw*cin_pad] with k = (ky*kw + kx)*cin_pad + c, cin_pad = round_up(cin, 32),
 * cout_pad = round_up(cout, 16), zero filled.  cin = c0 + c1 is the channel concat of the (up to)
 * two input tensors a layer reads.  scale/bias hold the per-output-channel affine applied to the
 * accumulator (conv bias, eval-mode BatchNorm fold, or LayerNorm weight/bias), length cout_pad.
 * ConvTranspose2d(k3,s1,p1) layers are stored as the equivalent convolution (spatially flipped,
 * in/out swapped).  The last p_model convolution is stored with its output rows interleaved
 * (row 16T+4g+r  <->  r<2: loc channel 8T+2g+r, r>=2: raw-scale channel 8T+2g+r-2).
 */
#ifndef SFNATIVE_H
#define SFNATIVE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sf_status {
  SF_OK = 0,
  SF_ERR_INVALID = -1,    /* bad argument (null pointer, unsupported shape) */
  SF_ERR_WORKSPACE = -2,  /* workspace too small */
  SF_ERR_LAUNCH = -3,     /* HIP launch / runtime error */
  SF_ERR_UNSUPPORTED = -4
} sf_status;

enum { SF_ACT_NONE = 0, SF_ACT_LRELU = 1, SF_ACT_RELU = 2, SF_ACT_TANH = 3, SF_ACT_SIGMOID = 4, SF_ACT_GELU = 5 };
enum { SF_SOLVER_EULER = 0, SF_SOLVER_MIDPOINT = 1, SF_SOLVER_RK4 = 2 };
enum { SF_OP_JUMP = 0, SF_OP_STEP = 1 };
/* One coefficient record per ODE step, fp32, computed on the host in float64 and rounded once:
 * {dt, dt/2, dt/6, dt/3,  dt/2, dt/6,  dt/2, dt/3,  dt, dt/3,  dt/6, 0} (the last 8 = RK4 stage pairs) */
#define SF_COEF_STRIDE 12

typedef struct sf_conv_w {
  const float* w;
  const float* scale; /* may be NULL (= 1) */
  const float* bias;  /* may be NULL (= 0) */
  int32_t cout, cout_pad, c0, c1, cin_pad, kh, kw, dil, stride, pad, act;
  int32_t reserved;
  /* optional, NULL unless packed with SF_PACK_BF16X3: the same packed weights split into bf16 pieces for the opt-in
   * "bf16x3" math mode (a = hi + lo; a*b ~ hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulators).  Layers
   * whose struct carries it run the split-bf16 K loop where a kernel has one; results then differ from the exact-fp32
   * path by ~1e-5 (profiles/r03_bf16x3_*).  The default build of every module leaves it NULL: exact fp32. */
  const void* w_bf16x3;
  /* optional, NULL unless packed with SF_PACK_WINOGRAD and the layer qualifies (3x3, stride 1, pad 1, no dilation, cin a multiple of
   * 32, cout_pad a multiple of 64): the packed weights in Winograd F(2x2, 3x3) form, U = G g G^T as [cin/16][16][cout_pad][16].
   * Launches with enough pixels then run csrc/conv_wino.hip: exact fp32 arithmetic, 2.25x fewer multiplies, different rounding
   * (<= 7e-6 on the BEV outputs, profiles/r04_winograd_accuracy_study.json). */
  const float* w_wino;
} sf_conv_w;

/* ---- weight packing on the device (load time) ---------------------------------------------------------------------
 * sf_pack_conv packs ONE reference convolution into `blob` (device memory of at least sf_pack_conv_bytes) and fills
 * `out` with pointers into it:
 *   weight     Conv2d weight [cout][cin][kh][kw] fp32 (device); with SF_PACK_TRANSPOSED a ConvTranspose2d(k, s1, p=(k-1)/2)
 *              weight [cin][cout][kh][kw], stored as the equivalent convolution; with SF_PACK_FOLD_DUP a weight
 *              [cout][2*cin][kh][kw] of a layer that reads cat[s, s], folded to W[:, :cin] + W[:, cin:]
 *              (temporal_ode_bayes.py:148-161: gru_cell_2(s, s)); with SF_PACK_INTERLEAVE the output rows of the last
 *              p_model convolution are interleaved (loc, loc, raw, raw) for the sampling epilogue
 *   affine     either eval-mode BatchNorm2d buffers (bn_weight / bn_bias / bn_mean / bn_var, bn_eps: scale = w / sqrt(var + eps),
 *              bias = b - mean * scale + conv_bias * scale) or a plain per-channel `scale` (LayerNorm weight; NULL = 1) and
 *              `conv_bias` (NULL = 0)
 *   c0 + c1 = cin: channels read from the first / second input tensor; pad < 0 = "same" ((kh-1)*dil/2).
 * The composite structs below (sf_gru_w ... sf_deeplab_w) are assembled from packed convolutions by plain struct
 * assignment; [update ; reset] gate pairs are packed from the two weights stored one after the other (cout = 2*hidden). */
enum { SF_PACK_TRANSPOSED = 1, SF_PACK_FOLD_DUP = 2, SF_PACK_INTERLEAVE = 4, SF_PACK_BF16X3 = 8, SF_PACK_WINOGRAD = 16 };

/* conv-GRU cell: SpatialGRU.gru_cell (streamingflow/layers/temporal.py:44-57) */
typedef struct sf_gru_w {
  sf_conv_w gates;    /* [conv_update ; conv_reset] stacked on cout (2*hidden), sigmoid */
  sf_conv_w cand;     /* conv_state_tilde */
  sf_conv_w decoder;  /* 1x1 conv_decoder, w == NULL if absent */
} sf_gru_w;

/* DualGRUODECell / DualGRUCell (streamingflow/layers/temporal_ode_bayes.py:64-161 / 211-305) */
typedef struct sf_dual_w {
  sf_conv_w gates1, cand1;   /* gru_cell_1 on cat[x, s] */
  sf_conv_w gates2, cand2;   /* gru_cell_2 on cat[s, s]; gates2 has the duplicate input folded (cin = C) */
  sf_conv_w dec2;            /* conv_decoder_2 */
  sf_conv_w tg7, tgproj;     /* trusting_gate.0.layers.0 (7x7, LN in scale/bias), .projection.0 (1x1, GELU) */
  sf_conv_w tg1, tg3;        /* .layers.3 (1x1 + LN), .layers.6 (3x3 + LN) */
  const float* w_logit;      /* trusting_gate.1: [2][C] */
  int32_t C;
  /* optional (w == NULL: unused): the two input halves of gates1 packed on their own — gates1_x = W[:, :C] on x with the
   * bias and the sigmoid, gates1_s = W[:, C:] on s with neither.  In a rollout the s half is accumulated beside the
   * previous infer_state (the state is known five launches before x is) and added to the x half's sums. */
  sf_conv_w gates1_x, gates1_s;
  /* optional (w == NULL: unused), round 6: the two input halves of the trusting gate's 7x7 packed on their own — tg7_h = W[:, :C] on rnn_state1
   * with the LayerNorm weight / bias, tg7_r = W[:, C:] on rnn_state2 with neither.  In a rollout rnn_state2 = conv_decoder_2(h2) is a
   * function of the state alone: its half of the 7x7 (half of the longest launch of a step) runs on a forked stream beside the previous
   * infer_state and is added to the rnn_state1 half's sums. */
  sf_conv_w tg7_h, tg7_r;
} sf_dual_w;

/* ResBlock (streamingflow/layers/res_models.py:52-79) */
typedef struct sf_res_w {
  sf_conv_w conv1, conv2, proj; /* proj.w == NULL when in == out */
} sf_res_w;

/* ConvNet p_model + rsample (res_models.py:168-180, models/model_utils.py:60-109) */
typedef struct sf_pmodel_w {
  sf_res_w rb0, rb1;
  const float *se0_fc0, *se0_fc2, *se1_fc0, *se1_fc2; /* SELayer fc weights [2C/8][2C], [2C][2C/8] */
  sf_conv_w last;                                     /* interleaved rows, LeakyReLU */
  int32_t C;
} sf_pmodel_w;

/* SmallEncoder / SmallDecoder (res_models.py:82-147) */
typedef struct sf_encoder_w { sf_res_w blocks[5]; sf_conv_w last; int32_t C, F; } sf_encoder_w;
typedef struct sf_decoder_w { sf_conv_w first; sf_res_w blocks[5]; sf_conv_w last0, last1; int32_t C, F; } sf_decoder_w;

/* ConvNeXt Block (streamingflow/layers/convolutions.py:310-346) */
typedef struct sf_convnext_w {
  const float *dw_w /*[49][C]*/, *dw_b, *ln_w, *ln_b;
  sf_conv_w pw1, pw2; /* pw2: scale = gamma, bias = gamma*b2 */
  int32_t C;
} sf_convnext_w;

/* DeepLabHead / ASPP (convolutions.py:217-280) */
typedef struct sf_deeplab_w {
  sf_conv_w branch[4];   /* 1x1 and the three dilated 3x3, BN+ReLU */
  const float *pool_w /*[hid][C]*/, *pool_scale, *pool_bias; /* ASPPPooling conv + BN */
  const float* proj_pool_w; /* [hid][hid]: projection weights of the pooled branch */
  sf_conv_w project;     /* 1x1 over the 4 spatial branches (cin = 4*hid), BN+ReLU */
  sf_conv_w conv3, cls;  /* 3x3+BN+ReLU, final 1x1 (+bias) */
  int32_t C, hid;
} sf_deeplab_w;

/* Bottleneck (streamingflow/layers/convolutions.py:65-172 == beverse basic_modules.py:68-178) */
typedef struct sf_bottleneck_w {
  sf_conv_w down, conv, up, proj; /* BN folded; proj.w == NULL for the identity skip */
  int32_t downsample;
} sf_bottleneck_w;

/* Bottleblock (streamingflow/layers/convolutions.py:348-380), see sf_bottleblock_fwd */
typedef struct sf_bottle_w {
  sf_conv_w c7, c1, c3; /* layers.0 / .3 / .6 with the LayerNorm weight / bias in scale / bias */
  sf_conv_w proj;       /* projection.0; proj.w == NULL when in == out */
} sf_bottle_w;

int sf_version(void);
const char* sf_status_string(int status);
/* Persistent "flow" form of the single-latent rollout (csrc/conv_sp.hip: sp_flow_kernel): every launch group of sf_nnfo_rollout_* runs
 * as a phase of ONE resident launch ordered by tile-level dataflow — bitwise the results of the launch-per-layer form, ~5 % less time
 * per ODE step.  It needs every CU of an otherwise IDLE device (one workgroup per CU must be resident at once; waits are bounded, so a
 * device shared with another process or stream ends the launch with NaN results — see sf_flow_errors — instead of hanging): opt-in.  on: 1 / 0, -1 = the
 * SF_PERSIST environment variable (default 0).  Returns the previous setting.  Process-wide; set it before capturing a graph. */
int sf_set_flow_mode(int on);
/* Every dependency wait of the flow kernel is bounded (SF_FLOW_TIMEOUT polls).  A rollout in which one gave up does NOT return its
 * half-finished results: a tail kernel of the same call (captured with it in a graph) overwrites out_states / final_state with NaN.
 * sf_flow_errors synchronises `stream` and returns the number of timed-out waits of the calling thread's most recent persistent
 * rollout (0 = healthy), SF_ERR_INVALID if the thread's most recent rollout was not a persistent one.  The launch-per-layer form (the
 * default) has no such waits.
 * Lifetime and threads: the count lives in the WORKSPACE the caller passed to that rollout (the library owns no device memory), and
 * "most recent" means most recently ENQUEUED BY THIS THREAD — call sf_flow_errors from the thread that enqueued the rollout, before
 * that workspace is freed or reused, and not for a graph replay (a replay enqueues nothing: read the outputs — a rollout with a
 * timed-out wait has NaN in every element of out_states and final_state — as streamingflow_amd's checked mode does). */
int sf_flow_errors(void* stream);

/* ---- ABI guard ---------------------------------------------------------------------------------------------------------
 * The structs above are passed by pointer and sf_conv_w is embedded by value in every composite, so a host compiled
 * against an older header would hand the library mis-sized structs (round 3 grew sf_conv_w from 72 to 80 bytes and
 * sf_dual_w by two members).  SF_ABI_VERSION changes whenever a public struct changes layout.  A host calls
 * sf_abi_check_header() once after loading the library — it passes the sizes ITS compiler saw — and must not call
 * anything else unless it returns SF_OK; bindings without a C compiler (ctypes, cgo, JNI) compare their own struct sizes
 * with sf_abi_sizeof() the same way (streamingflow_amd/_lib.py does, INTEGRATION.md shows it).  Hosts must zero-initialise
 * the structs (optional members are "NULL = absent") and recompile when SF_ABI_VERSION changes. */
#define SF_ABI_VERSION 6
enum {
  SF_STRUCT_CONV_W = 0, SF_STRUCT_GRU_W, SF_STRUCT_DUAL_W, SF_STRUCT_RES_W, SF_STRUCT_PMODEL_W, SF_STRUCT_ENCODER_W,
  SF_STRUCT_DECODER_W, SF_STRUCT_CONVNEXT_W, SF_STRUCT_DEEPLAB_W, SF_STRUCT_BOTTLENECK_W, SF_STRUCT_BOTTLE_W, SF_STRUCT_COUNT
};
int sf_abi_version(void);                 /* the library's SF_ABI_VERSION */
size_t sf_abi_sizeof(int which);          /* sizeof(struct SF_STRUCT_<which>) as the library was compiled; 0 for an unknown id */
/* SF_OK when abi_version and all n = SF_STRUCT_COUNT sizes equal the library's, SF_ERR_INVALID otherwise */
int sf_abi_check(int abi_version, const size_t* struct_sizes, int n);
static inline int sf_abi_check_header(void) {
  const size_t sizes[SF_STRUCT_COUNT] = {sizeof(sf_conv_w),     sizeof(sf_gru_w),      sizeof(sf_dual_w),     sizeof(sf_res_w),
                                         sizeof(sf_pmodel_w),   sizeof(sf_encoder_w),  sizeof(sf_decoder_w),  sizeof(sf_convnext_w),
                                         sizeof(sf_deeplab_w),  sizeof(sf_bottleneck_w), sizeof(sf_bottle_w)};
  return sf_abi_check(SF_ABI_VERSION, sizes, SF_STRUCT_COUNT);
}

/* layout: [n][C][HW] <-> [n][HW][C] */
int sf_nchw_to_nhwc(const float* src, float* dst, int n, int C, int HW, void* stream);
int sf_nhwc_to_nchw(const float* src, float* dst, int n, int C, int HW, void* stream);
/* the same with the n images src_stride / dst_stride floats apart: frames read out of / written into a
 * [B][T][C][H][W] tensor (future_prediction_ode.py:36-49, :63-64) without an intermediate stack */
int sf_nchw_to_nhwc_strided(const float* src, size_t src_stride, float* dst, size_t dst_stride, int n, int C, int HW,
                            void* stream);
int sf_nhwc_to_nchw_strided(const float* src, size_t src_stride, float* dst, size_t dst_stride, int n, int C, int HW,
                            void* stream);

/* generic fused conv (test hook and building block): y = act(conv(cat[in0,in1])*scale + bias) + add */
int sf_conv2d_fwd(const sf_conv_w* w, const float* in0, const float* in1, const float* add, float* out,
                  int n_img, int Hin, int Win, int in_up, void* stream);

/* benchmarking aid: sf_conv2d_fwd enqueued `reps` times from C++; ws may be NULL */
int sf_conv2d_repeat(const sf_conv_w* w, const float* in0, const float* in1, const float* add, float* out,
                     int n_img, int Hin, int Win, int in_up, int reps, float* ws, size_t ws_bytes, void* stream);

/* SpatialGRU.gru_cell — temporal.py:44-57.  x [P][Cx], s [P][C] -> out [P][C] (P = n*H*W) */
int sf_gru_cell_fwd(const sf_gru_w* w, const float* x, const float* s, float* out, int n_img, int H, int W,
                    float* ws, size_t ws_bytes, void* stream);
size_t sf_gru_cell_ws_bytes(int C, int n_img, int H, int W);
/* SpatialGRUODECell.forward — temporal_ode_bayes.py:35-61 (defined, unused by the shipped model):
 * dh = u * (h~ - s), candidate = conv + BN + ReLU.  Same workspace as sf_gru_cell_fwd. */
int sf_gru_ode_cell_fwd(const sf_gru_w* w, const float* x, const float* s, float* out, int n_img, int H, int W,
                        float* ws, size_t ws_bytes, void* stream);

/* SpatialGRU.forward — temporal.py:26-42 on n_img samples at once.
 * x [T][n_img][H*W][Cx], state0 [n_img][H*W][C] -> out [T][n_img][H*W][Cx] */
int sf_spatial_gru_fwd(const sf_gru_w* w, const float* x, const float* state0, float* out, int T, int n_img, int H,
                       int W, float* ws, size_t ws_bytes, void* stream);
size_t sf_spatial_gru_ws_bytes(int C, int n_img, int H, int W);

/* The latent-space operators below take n_img >= 1 samples ([n_img][H][W][C] tensors) that are
 * processed as one pixel space: the reference handles one sample per call (its dual cell is only
 * well defined at batch 1); batching samples here is what fills the chip at a 50x50 latent. */

/* DualGRUODECell.forward (derivative != 0) / DualGRUCell.forward (derivative == 0) —
 * temporal_ode_bayes.py:92-131 / :239-275, fused with the integrator update:
 *   derivative: out = base + coef[0]*(cur - s); optional out2 = (acc2 ? out2 : base) + coef[1]*(cur - s)
 *   jump:       out = cur
 * coef points at device fp32 scalars (shared by all images).  out may not alias x/s/base. */
int sf_dual_cell_fwd(const sf_dual_w* w, const float* x, const float* s, float* out, int derivative,
                     const float* base, const float* coef, float* out2, int acc2, int n_img, int H, int W,
                     float* ws, size_t ws_bytes, void* stream);
size_t sf_dual_cell_ws_bytes(int C, int n_img, int H, int W);

/* The tail of both dual cells on given branch states — temporal_ode_bayes.py:122-131 / :266-275 and the loop body of
 * Dual_GRU.forward (layers/temporal.py:118-124): gate = softmax(trusting_gate(cat[r1, r2])); cur = r2*gate0 + r1*gate1;
 * derivative != 0: out = base + coef[0]*(cur - s), else out = cur (s, base, coef may then be NULL).
 * Workspace: sf_dual_cell_ws_bytes. */
int sf_trust_mix_fwd(const sf_dual_w* w, const float* r1, const float* r2, const float* s, float* out, int derivative,
                     const float* base, const float* coef, int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream);

/* Bottleblock.forward — convolutions.py:348-380 on cat[x0, x1] (x1 may be NULL): 7x7 + LN + GELU -> 1x1 + LN + GELU ->
 * 3x3 + LN + GELU, plus projection(x) (1x1 + GELU) or x itself when in == out (then x1 must be NULL).
 * Channel counts of the LayerNorm layers <= 64.  Weights: struct sf_bottle_w, above. */
int sf_bottleblock_fwd(const sf_bottle_w* w, const float* x0, const float* x1, float* out, int n_img, int H, int W,
                       float* ws, size_t ws_bytes, void* stream);
size_t sf_bottleblock_ws_bytes(int cin, int cout, int n_img, int H, int W);

/* NNFOwithBayesianJumps.infer_state — temporal_ode_bayes.py:463-477: p = loc + eps*(softplus(raw)+1e-8).
 * q_out (raw p_model output, [P][2C], reference channel order) may be NULL.  n_img <= 64. */
int sf_infer_state_fwd(const sf_pmodel_w* w, const float* s, const float* eps, float* p_out, float* q_out,
                       int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream);
size_t sf_infer_state_ws_bytes(int C, int n_img, int H, int W);

/* NNFOwithBayesianJumps.ode_step — temporal_ode_bayes.py:436-461 (euler, midpoint) and the
 * build-defined classical RK4.  coef: one device coefficient record (SF_COEF_STRIDE floats).
 * eps: [n_draws][n_img][H*W][C] with n_draws = 1 (euler), 2 (midpoint), 4 (rk4).  If impute == 0
 * the cell input is zeros (:442-443).  state_out / p_out must not alias state_in / p_in. */
int sf_ode_step_fwd(const sf_dual_w* gru_c, const sf_pmodel_w* pm, int solver, int impute, const float* state_in,
                    const float* p_in, const float* coef, const float* eps, float* state_out, float* p_out,
                    int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream);
size_t sf_ode_step_ws_bytes(int C, int n_img, int H, int W);

/* The observation/prediction loop of NNFOwithBayesianJumps.forward — temporal_ode_bayes.py:539-604,
 * driven by a schedule computed on the host (streamingflow_amd.schedule); all n_img samples share
 * the op sequence.  ops[2*i] = SF_OP_JUMP (ops[2*i+1] = observation index) or SF_OP_STEP
 * (ops[2*i+1] = step index).  coef: [n_steps][SF_COEF_STRIDE] (coef_per_image == 0) or
 * [n_steps][n_img][SF_COEF_STRIDE] (per-sample step sizes).  After op i (1-based count k = i+1)
 * the state is copied to out_states[t] for every target t with sel_nops[t] == k (:606-622).
 * hx_obs [n_obs][n_img][H*W][C]; eps [n_draws][n_img][H*W][C] in reference draw order;
 * out_states [n_targets][n_img][H*W][C]; final_state [n_img][H*W][C] (may be NULL). */
int sf_nnfo_rollout_fwd(const sf_dual_w* gru_c, const sf_dual_w* gru_obs, const sf_pmodel_w* pm, int solver,
                        int impute, const int32_t* ops, int n_ops, const float* hx_obs, const float* eps,
                        const float* coef, int coef_per_image, const int32_t* sel_nops, int n_targets,
                        float* out_states, float* final_state, int n_img, int H, int W, float* ws,
                        size_t ws_bytes, void* stream);
/* Throughput mode: the Gaussian noise of every infer_state is generated inside the sampling epilogue instead of being read
 * from an eps tensor (Philox4x32-10, Box-Muller; csrc/sf_math.h).  philox_state: device record {uint64 seed, uint64 offset};
 * the draw of an element is a function of (seed, offset, draw index, pixel, channel) only, so results do not depend on the
 * kernel / tile / batch, and a captured graph gives fresh noise when the host rewrites the 16-byte record between replays.
 * Same distribution as the reference's Normal.rsample (temporal_ode_bayes.py:474-476), different stream: parity tests use
 * the eps-fed entry points above. */
int sf_nnfo_rollout_philox_fwd(const sf_dual_w* gru_c, const sf_dual_w* gru_obs, const sf_pmodel_w* pm, int solver, int impute,
                               const int32_t* ops, int n_ops, const float* hx_obs, const uint64_t* philox_state, const float* coef,
                               int coef_per_image, const int32_t* sel_nops, int n_targets, float* out_states, float* final_state,
                               int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream);
int sf_infer_state_philox_fwd(const sf_pmodel_w* w, const float* s, const uint64_t* philox_state, int draw, float* p_out, float* q_out,
                              int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream);

size_t sf_nnfo_rollout_ws_bytes(int C, int n_img, int H, int W);

/* SmallEncoder.forward — res_models.py:98-109: [n][H][W][C] -> [n][H/4][W/4][C] */
int sf_small_encoder_fwd(const sf_encoder_w* w, const float* x, float* out, int n, int H, int W,
                         float* ws, size_t ws_bytes, void* stream);
size_t sf_small_encoder_ws_bytes(int C, int F, int n, int H, int W);
/* SmallDecoder.forward — res_models.py:134-147: [n][h][w][C] -> [n][4h][4w][C] */
int sf_small_decoder_fwd(const sf_decoder_w* w, const float* z, float* out, int n, int h, int wd,
                         float* ws, size_t ws_bytes, void* stream);
size_t sf_small_decoder_ws_bytes(int C, int F, int n, int h, int w);

/* ConvNeXt Block.forward — convolutions.py:333-346 */
int sf_convnext_block_fwd(const sf_convnext_w* w, const float* x, float* out, int n, int H, int W,
                          float* ws, size_t ws_bytes, void* stream);
size_t sf_convnext_block_ws_bytes(int C, int n, int H, int W);
/* DeepLabHead.forward — convolutions.py:272-280 */
int sf_deeplab_head_fwd(const sf_deeplab_w* w, const float* x, float* out, int n, int H, int W,
                        float* ws, size_t ws_bytes, void* stream);
size_t sf_deeplab_head_ws_bytes(int C, int hid, int n, int H, int W);
/* the same, the classifier writing the boundary's planar layout itself: image i as [cout][H][W] planes at
 * out + (i / group) * stride_major + (i % group) * stride_minor floats.  Frames (t, b) of a [T][B] run into the reference's
 * [B][T][C][H][W] result (future_prediction_ode.py:62-64): group = B, stride_major = C*H*W, stride_minor = T*C*H*W — no
 * transpose launches after the head.  Same workspace as sf_deeplab_head_fwd. */
int sf_deeplab_head_planar_fwd(const sf_deeplab_w* w, const float* x, float* out, int n, int H, int W, int group,
                               size_t stride_major, size_t stride_minor, float* ws, size_t ws_bytes, void* stream);

/* Bottleneck.forward — convolutions.py:164-172: [n][H][W][Cin] -> [n][Ho][Wo][Cout] */
int sf_bottleneck_fwd(const sf_bottleneck_w* w, const float* x, float* out, int n, int H, int W, float* ws,
                      size_t ws_bytes, void* stream);
size_t sf_bottleneck_ws_bytes(int Cin, int Cout, int n, int H, int W);
/* elementwise LogSigmoid, min(x, 0) - log1p(exp(-|x|)): the decoder of DistributionModule(method='BERNOULLI')
 * (streamingflow/models/distributions.py:29-33, :46-47) */
int sf_logsigmoid_fwd(const float* x, float* out, size_t n, void* stream);
/* last_conv of DistributionModule / SpatialDistributionModule — beverse motion_modules.py:34-46,74-88
 * (and streamingflow/models/distributions.py:35-49 with clamp == 0): [global avg-pool +] 1x1 conv + bias,
 * second half of the channels clamped to [lo, hi] */
int sf_dist_head_fwd(const sf_conv_w* w, const float* enc, float* out, int n, int H, int W, int global_pool,
                     int clamp, float lo, float hi, float* ws, size_t ws_bytes, void* stream);
size_t sf_dist_head_ws_bytes(int C, int n);

/* ---- N3 building blocks: BEV Decoder (streamingflow/models/decoder.py:8-140) ---------------------------
 * The decoder is a ResNet-18 U-Net of plain convolutions; its host mirror
 * (streamingflow_amd/models/decoder.py) drives these two entry points layer by layer.
 * sf_conv2d_ex_fwd: sf_conv2d_fwd with channel-sliced operands (in0/in1/add/out may be channel ranges
 * of wider NHWC tensors: *_cs = channel stride of the tensor, out_co = first output channel) and
 * `act_after_add` (torchvision BasicBlock: relu(bn2(conv2(.)) + identity)); ws/ws_bytes: optional
 * split-K scratch (sf_conv2d_ex_ws_bytes()), may be NULL. */
int sf_conv2d_ex_fwd(const sf_conv_w* w, const float* in0, int in0_cs, const float* in1, int in1_cs, const float* add,
                     int add_cs, int act_after_add, float* out, int out_cs, int out_co, int n_img, int Hin, int Win,
                     int in_up, float* ws, size_t ws_bytes, void* stream);
size_t sf_conv2d_ex_ws_bytes(void);
/* UpsamplingAdd — convolutions.py:204-215: out [n][2H][2W][C] = bilinear_x2(in, align_corners=False) + skip.
 * (The 1x1 conv + BatchNorm of the module commute with the interpolation and run before it, at the
 * low resolution: a quarter of the MACs.) */
int sf_upsample_bilinear2_add_fwd(const float* in, const float* skip, float* out, int n, int Hin, int Win, int C,
                                  void* stream);

/* TemporalBlock helpers (streamingflow/layers/temporal.py:394-432, :435-490): per-image channel means of
 * an NHWC tensor (the spatial part of PyramidSpatioTemporalPooling's AvgPool3d) and the broadcast of a
 * per-image channel vector over a channel slice (its bilinear upsampling of a 1x1 map is a constant). */
size_t sf_channel_mean_ws_bytes(int C, int n);
int sf_channel_mean_fwd(const float* x, float* out, int n, int HW, int C, float* ws, size_t ws_bytes, void* stream);
int sf_broadcast_channels_fwd(const float* vec, float* out, int n, int HW, int k, int out_cs, int out_co, void* stream);

/* ---- N1: camera lift-splat voxel pooling ---------------------------------------
 * Layouts: frustum point p = ((cam*D + d)*fH + h)*fW + w of batch element b, points of a call are
 * numbered b-major; BEV cell id = ((b*Z + z)*X + x)*Y + y; pooled output [n_cells][C] = [B][Z][X][Y][C]
 * (the reference kernel's own output layout, NHWC for Z == 1).  `lo`, `res` (3 floats) and `dim`
 * (X, Y, Z) are HOST arrays; lo = bev_start_position - bev_resolution / 2 evaluated in fp32. */

/* bev_pool_forward — mmdet3d/ops/bev_pool/src/bev_pool.cpp:26-49 + bev_pool_cuda.cu:20-42: x [n][c]
 * already in sorted order, geom_feats [n][4] = (x, y, z, b) int32, one interval per occupied cell;
 * out [b][d][h][w][c] is zero-filled, each interval is summed in the given order (fp32, sequential:
 * bit-identical to the reference kernel). */
int sf_bev_pool_fwd(const float* x, const int32_t* geom_feats, const int32_t* interval_lengths,
                    const int32_t* interval_starts, int n, int c, int n_intervals, int b, int d, int h, int w,
                    float* out, void* stream);

/* Quantise + filter + rank + sort of streamingflow.bev_pool (streamingflow.py:353-368) and
 * bev_pool() (bev_pool.py:85-93), without host round trips: geom [n_points][3] float ->
 * order [n_points] (point ids, points of a cell contiguous, ascending inside a cell; points outside
 * the grid last), cell_start [n_cells + 1] (CSR row starts; cell_start[n_cells] = number of kept
 * points), optional coords [n_points][4] = (x, y, z, b) per point in input order, -1 when outside. */
size_t sf_lift_index_ws_bytes(int n_points, int n_cells);
int sf_lift_index_fwd(const float* geom, int n_points, int n_batch, const float* lo, const float* res,
                      const int32_t* dim, int32_t* coords, int32_t* order, int32_t* cell_start, void* ws,
                      size_t ws_bytes, void* stream);
/* Same index from integer coordinates [n_points][4] = (x, y, z, b) — the input of mmdet3d's
 * bev_pool(feats, coords, B, D, H, W) (bev_pool.py:85-93; its D, H, W are Z, X, Y here). */
int sf_lift_index_coords_fwd(const int32_t* coords, int n_points, int B, int Z, int X, int Y, int32_t* order,
                             int32_t* cell_start, void* ws, size_t ws_bytes, void* stream);
/* Same index, geometry computed in the kernel from the camera rig instead of being read:
 * position = affine[b*n_cam + cam] (3x4 row major, device) applied to (us[w]*ds[d], vs[h]*ds[d], ds[d], 1)
 * — create_frustum (streamingflow.py:149-168), get_geometry (:277-292) and the ego-motion warps of
 * projection_to_birds_eye_view (:386-396) composed by the host into one affine per (frame, camera). */
int sf_lift_index_rig_fwd(const float* affine, const float* us, const float* vs, const float* ds, int n_batch,
                          int n_cam, int D, int fH, int fW, const float* lo, const float* res, const int32_t* dim,
                          int32_t* order, int32_t* cell_start, void* ws, size_t ws_bytes, void* stream);

/* Pooling proper (bev_pool_cuda.cu:20-42 over every cell, empty ones included) fused with the temporal
 * blend of projection_to_birds_eye_view (streamingflow.py:419): out = prev*discount + sum (prev NULL:
 * plain sum).  x [n_points][C] is the materialised depth (x) feature tensor in point order. */
int sf_lift_pool_fwd(const float* x, const int32_t* order, const int32_t* cell_start, int n_cells, int C,
                     const float* prev, float discount, float* out, void* stream);
/* Same with the outer product of encoder_forward (streamingflow.py:304-307) folded in:
 * x[p][c] = depth_prob[p] * feat[ray(p)][c]; feat [n_batch*n_cam*fHW][C], depth_prob [n_points]. */
int sf_lift_pool_fused_fwd(const float* feat, const float* depth_prob, int D, int fHW, const int32_t* order,
                           const int32_t* cell_start, int n_cells, int C, const float* prev, float discount,
                           float* out, void* stream);
/* depth.softmax(dim=1) of streamingflow.py:304 on [rows][D][fHW] */
int sf_depth_softmax_fwd(const float* logits, float* prob, int rows, int D, int fHW, void* stream);

/* ---- N2 (first half): LiDAR hard voxelisation ------------------------------------------------------
 * hard_voxelize — mmdet3d/ops/voxel/src/voxelization.h:63-85, deterministic GPU path
 * voxelization_cuda.cu:262-420 (same result as voxelization_cpu.cpp:45-102): points [num_points][F]
 * (x, y, z first), voxel_size / coors_range HOST arrays of 3 / 6 floats.  Outputs sized for
 * max_voxels and zero-filled as in voxelize.py:52-54: voxels [max_voxels][max_points][F] (may be NULL),
 * coors [max_voxels][3] = (x, y, z), num_points_per_voxel [max_voxels]; voxel_num: device int = the
 * reference's return value.  mean_feats (may be NULL) [max_voxels][F] = sum over the voxel's points /
 * their number — the reduction streamingflow.voxelize applies right after (streamingflow.py:190-195). */
/* Dynamic voxelisation (mmdet3d/ops/voxel/voxelize.py:46-49, the max_points == -1 / max_voxels == -1 branch of _Voxelization.forward ->
 * dynamic_voxelize; src/voxelization_cpu.cpp:8-43, src/voxelization_cuda.cu:25-60): coors [num_points][3] int32 = the (x, y, z) voxel of
 * every point, floor((p - min) / size) in fp32, or (-1, -1, -1) for a point outside the range on any axis. */
int sf_dynamic_voxelize_fwd(const float* points, int num_points, int num_features, const float* voxel_size, const float* coors_range,
                            int32_t* coors, void* stream);
size_t sf_hard_voxelize_ws_bytes(int num_points);
int sf_hard_voxelize_fwd(const float* points, int num_points, int num_features, const float* voxel_size,
                         const float* coors_range, int max_points, int max_voxels, float* voxels, int32_t* coors,
                         int32_t* num_points_per_voxel, float* mean_feats, int32_t* voxel_num, void* ws,
                         size_t ws_bytes, void* stream);

/* ---- N2 (second half): LiDAR SparseEncoder — sparse 3-D convolutions (mmdet3d/ops/spconv, spconv 1.x) ---------
 * Sites are rows of coords [n][4] = (batch, x, y, z) int32 in a grid `shape` = (X, Y, Z); ksize / stride /
 * padding are HOST arrays of 3 ints; kernel tap t = (kx*KY + ky)*KZ + kz.
 * sf_sparse_out_sites_fwd — output sites of a strided SparseConv3d (spconv/ops.py:19-33 output size; a
 *   site is active when some active input p and tap k satisfy p = o*stride - padding + k): sorted unique
 *   coordinates into out_coords (capacity `cap` rows), their number into the device int n_out.
 * sf_sparse_table_fwd — neighbour table nbr [n_out][ntaps]: input row read by output site j through tap t,
 *   or -1 (subm != 0: submanifold conv, out_coords == in_coords, offsets centred, stride/padding ignored).
 * sf_sparse_conv_fwd — the convolution (spconv/conv.py:114-214 + BatchNorm1d + ReLU of
 *   make_sparse_convmodule / SparseBasicBlock, sparse_block.py:88-107, :110-176):
 *   out[j] = act(scale * sum_t W_t^T . feats[nbr[j][t]] + bias) + add   (act_after_add: act(... + add));
 *   w packs [Cout][Cin][ntaps][1] (kh = ntaps, kw = 1); feats has n_in rows of feats_cs floats.
 * sf_sparse_to_dense_fwd — SparseConvTensor.dense() + permute/view of sparse_encoder.py:131-137 as NHWC:
 *   out [batch][X][Y][C*D], channel c*D + z (zero where no site is active). */
size_t sf_sparse_index_ws_bytes(int n_in, int ntaps);
int sf_sparse_out_sites_fwd(const int32_t* in_coords, int n_in, int batch, const int32_t* shape, const int32_t* ksize,
                            const int32_t* stride, const int32_t* padding, int32_t* out_coords, int cap, int32_t* n_out,
                            void* ws, size_t ws_bytes, void* stream);
int sf_sparse_table_fwd(const int32_t* in_coords, int n_in, const int32_t* out_coords, int n_out, int batch,
                        const int32_t* shape, const int32_t* ksize, const int32_t* stride, const int32_t* padding, int subm,
                        int32_t* nbr, void* ws, size_t ws_bytes, void* stream);
int sf_sparse_conv_fwd(const sf_conv_w* w, const float* feats, int feats_cs, int n_in, const int32_t* nbr, int n_out,
                       const float* add, int act_after_add, float* out, float* ws, size_t ws_bytes, void* stream);
/* sf_sparse_conv_masked_fwd — the same convolution with a TAP MASK (round 6): tile_mask64[i], i < ceil(n_out / 64), has bit t set when at
 *   least one of the output rows 64 i .. 64 i + 63 reads an input row through tap t (nbr[j][t] >= 0).  A pixel tile of the kernel then
 *   walks only the taps that are live for its rows; a skipped tap would have gathered zero rows, so the sums are bitwise those of
 *   sf_sparse_conv_fwd.  It pays when the caller keeps the output rows of a stage sorted by neighbour mask (rows with equal masks
 *   together: streamingflow_amd/models/sparse_encoder.py does; 38 % of the (site, tap) products of the shipped cloud have no input row and
 *   a tile of stored-order rows can drop < 1 % of them, a tile of mask-sorted rows 25 %).  NULL mask = sf_sparse_conv_fwd. */
int sf_sparse_conv_masked_fwd(const sf_conv_w* w, const float* feats, int feats_cs, int n_in, const int32_t* nbr, const uint32_t* tile_mask64,
                              int n_out, const float* add, int act_after_add, float* out, float* ws, size_t ws_bytes, void* stream);
int sf_sparse_to_dense_fwd(const float* feats, const int32_t* coords, int n, int C, int batch, int X, int Y, int D, float* out,
                           void* stream);

/* ---- N4: evaluation harness kernels (streamingflow/metrics.py, streamingflow/utils/instance.py) ------------
 * sf_confusion_fwd — out[b[i]*K + a[i]] += 1 over n int64 labels in [0, K) (zero-filled here); *bad != 0 when
 *   a label was outside the range.  IntersectionOverUnion.update (metrics.py:37: tp/fp/fn/support are sums of
 *   this matrix) and PanopticMetric.panoptic_metrics (:171-176: bincount of prediction + K * target).
 * sf_instance_centers_fwd — find_instance_centers (instance.py:80-92) on one [H][W] heat map: (row, col) of
 *   the thresholded 3x3 local maxima in row-major order (as torch.nonzero), count in the device int.
 * sf_group_pixels_fwd — group_pixels * foreground (instance.py:95-116, :136-137): offsets [2][H][W],
 *   foreground [H][W] bytes, instance [H][W] int64 = 1 + index of the nearest centre of (pixel + offset), 0 on background.
 * sf_instance_moments_fwd — F frames [F][H][W] of instance ids in one launch: per (frame, id in 1..max_id) pixel count,
 *   exact integer sums of (row, col) and — when flow [F][2][H][W] and warped_fx are given — sums of (row + flow0,
 *   col + flow1) in 2^-20 fixed point (int64).  Integer atomics only, so the sums do not depend on the arrival order:
 *   the masked means of instance.py:213-236 for every frame of a sequence at once.
 * sf_confusion_frames_fwd — sf_confusion_fwd per frame of F equally sized label-map pairs: out[F][K][K]. */
/* sf_warp_affine_fwd — warp_features (utils/geometry.py:196-236): F.affine_grid(theta [B][2][3], align_corners=False)
 *   + F.grid_sample(mode nearest | bilinear, padding zeros, align_corners=False) on NCHW maps x [B][C][H][W]. */
int sf_warp_affine_fwd(const float* x, const float* theta, int B, int C, int H, int W, int bilinear, float* out,
                       void* stream);
int sf_confusion_fwd(const int64_t* a, const int64_t* b, long n, int K, int64_t* out, int32_t* bad, void* stream);
size_t sf_instance_centers_ws_bytes(int H, int W);
int sf_instance_centers_fwd(const float* center, int H, int W, float conf_threshold, int32_t* centers, int cap,
                            int32_t* n_centers, void* ws, size_t ws_bytes, void* stream);
int sf_group_pixels_fwd(const int32_t* centers, int n_centers, const float* offsets, const uint8_t* foreground, int H,
                        int W, int64_t* instance, void* stream);
int sf_instance_moments_fwd(const int64_t* instance, const float* flow, int F, int H, int W, int max_id, int64_t* pos_sums,
                            int64_t* warped_fx, int32_t* counts, void* stream);
int sf_confusion_frames_fwd(const int64_t* a, const int64_t* b, long n_per_frame, int F, int K, int64_t* out, int32_t* bad,
                            void* stream);

/* hipGraph capture of whatever the caller enqueues between begin and end on `stream` (must not be
 * the legacy default stream). */
int sf_graph_begin(void* stream);
int sf_graph_end(void* stream, void** graph_exec_out);
int sf_graph_launch(void* graph_exec, void* stream);
int sf_graph_destroy(void* graph_exec);

/* hipEvent timing helper for bench.py (events recorded on the stream the kernels run on) */
int sf_event_create(void** ev);
int sf_event_record(void* ev, void* stream);
int sf_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int sf_event_destroy(void* ev);

/* Per-launch profiler for bench.py (off by default): when enabled every implicit-GEMM launch is
 * bracketed by hipEvents on its own stream.  sf_prof_collect fills SF_PROF_KEYS-entry arrays indexed by
 * kernel key = tile_config*8 + epilogue (calls, total ms, algorithmic flops, algorithmic bytes). */
#define SF_PROF_KEYS 168      /* part of SF_ABI_VERSION: sf_prof_collect fills this many entries of each of its four arrays (160 -> 168 was version 5 -> 6) */
int sf_prof_enable(int on);
int sf_prof_collect(int32_t* calls, double* ms, double* flops, double* bytes);

size_t sf_pack_conv_bytes(int cout, int cin, int kh, int kw, int flags);
int sf_pack_conv(const float* weight, const float* conv_bias, const float* scale, const float* bn_weight, const float* bn_bias,
                 const float* bn_mean, const float* bn_var, float bn_eps, int cout, int cin, int kh, int kw, int c0, int c1, int act,
                 int dil, int stride, int pad, int flags, void* blob, size_t blob_bytes, sf_conv_w* out, void* stream);

/* The BatchNorm fold of sf_pack_conv on its own (the same device code): scale[i] = w[i] / sqrt(var[i] + eps),
 * bias[i] = b[i] - mean[i] * scale[i] (+ conv_bias[i] * scale[i]), i < n.  For callers that stack or zero-pad the folded
 * vectors of several layers before packing them as one convolution (streamingflow/models/decoder.py heads,
 * streamingflow/layers/temporal.py blocks).  There is no second implementation of the fold on the host side. */
int sf_bn_fold(const float* conv_bias, const float* bn_weight, const float* bn_bias, const float* bn_mean, const float* bn_var,
               float bn_eps, int n, float* scale, float* bias, void* stream);

/* Diagnostic builds of the library only (hipcc -DSF_STAMP; the product build returns SF_ERR_UNSUPPORTED): `buf` is a
 * device buffer of 64 slots x 4096 workgroups x 8 uint64; every implicit-GEMM launch then takes the next slot (mod 64)
 * and wave 0 of each workgroup records s_memrealtime (100 MHz) at entry / prologue done / first chunk landed / K loop
 * done / split-K hand-off done / epilogue stores issued / stores drained.  NULL switches it off. */
int sf_debug_stamps(void* buf);
/* diagnostic: workgroups per CU the runtime grants the large LDS-DMA tiles with their dynamic LDS
 * (0: 128x128 fp32, 1: 128x128 bf16x3, 2: 64x128 fp32, 3: 64x128 bf16x3); -1 on error */
int sf_debug_occupancy(int which);

#ifdef __cplusplus
}
#endif
#endif /* SFNATIVE_H */
