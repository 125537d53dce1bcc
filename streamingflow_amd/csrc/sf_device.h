// Device-side problem descriptors shared by the HIP kernels and the host orchestration.
// gfx950 (MI355X / CDNA4) only.  Activations are NHWC fp32 ("[pixel][channel]"), weights are
// packed [cout_pad][taps * cin_pad] fp32 (tap-major, channel-minor, zero padded), see
// include/sfnative.h for the packing contract.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sf {

enum Act : int { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4, ACT_GELU = 5 };

// Epilogue families (template parameter of the conv kernel).
enum Epi : int {
  EPI_AFFINE = 0,   // y = act(acc*scale + bias) [+ add*add_scale]; optional per-16px channel sums
  EPI_BLEND  = 1,   // conv-GRU blend: h = acc + bias; y = (1-u)*s + u*h
  EPI_LNG    = 2,   // [LayerNorm over channels] -> GELU
  EPI_TRUST  = 3,   // LN -> GELU -> +skip -> 1x1(C->2) -> softmax -> mix -> Euler/RK update
  EPI_SAMPLE = 4,   // q = lrelu(acc + bias); p = loc + eps*(softplus(raw)+1e-8)
};

struct ConvProblem {
  // ---- operands -------------------------------------------------------------------------
  const float* in0;       // first input, [n_img*Hin*Win][in0_cs]
  const float* in1;       // optional second input (channel concat), [..][in1_cs]
  const float* gate;      // optional: in1 is multiplied by (1 - gate[pix][gate_co + c])
  const float* in_scale;  // optional per-(image, channel) scale on in0 (SE), [n_img][c0]
  const float* w;         // packed weights [cout_pad][ktot]
  const float* scale;     // per-cout scale (BN fold / LN weight), may be null (=1)
  const float* bias;      // per-cout bias (conv bias, BN fold / LN bias), may be null (=0)
  const float* add;       // optional residual added after the activation, [P][add_cs]
  const float* add_scale; // optional per-(image, channel) scale on `add` (SE)
  float* out;             // [P][out_cs], written at channel offset out_co
  float* out2;            // epilogue specific second output (may be null)
  float* chansum;         // optional [ceil(P/16)][cout] per-16-pixel channel sums of `out`
  // epilogue specific read-only tensors
  const float* e0; const float* e1; const float* e2; const float* e3; const float* e4; const float* e5;
  const float* coef;      // device scalars (dt coefficients), record of image i at coef + i*coef_stride
  // ---- geometry --------------------------------------------------------------------------
  int c0, c1;             // channels taken from in0 / in1
  int in0_cs, in1_cs, gate_cs, gate_co;
  int n_img, Hin, Win;    // physical input size per image
  int in_up;              // 1: input is nearest-upsampled x2 on the fly (logical = 2*Hin x 2*Win)
  int Hout, Wout;
  int KH, KW, dil, stride, pad;
  int cin_pad;            // per-tap padded channel count, multiple of 32
  int ktot;               // KH*KW*cin_pad
  int cout, cout_pad;     // real / padded (multiple of 16) output channels
  int act;
  int add_cs, out_cs, out_co, out2_cs;
  int bias_per_img;       // bias indexed [img][cout] (ASPP pooled branch folded into bias)
  int e0_cs, e1_cs;       // channel strides of e0/e1 where they are not C
  int mode;               // epilogue specific
  float eps;              // LayerNorm epsilon
  // cross-workgroup split-K (small pixel counts): blockIdx.z = K slice; partial tiles go to
  // `slab`, the last slice to arrive (ticket in `counters`) sums them in slice order + epilogue
  float* slab;            // [tiles][nsplit][waves][MT*NT*4][64]
  unsigned int* counters; // [tiles], zero before the launch; reset by the last arriver
  int nsplit;
  int coef_stride;        // floats between per-image coefficient records (0: shared)
  int clamp_from;         // AFFINE: output channels >= clamp_from are clamped to [clamp_lo, clamp_hi] (<0: off)
  float clamp_lo, clamp_hi;
  // AFFINE, LDS-staged kernels only: planar output (the boundary's [C][H][W] layout written by the last layer itself instead of a
  // transpose launch): element (image i, pixel p, channel c) goes to out + (i / pl_div) * pl_sa + (i % pl_div) * pl_sb + c * HW + p
  int out_planar, pl_div;
  size_t pl_sa, pl_sb;
  // AFFINE, Winograd kernel only (even Hout, Wout): `out` is the 2x2 max-pooled tensor [n][Hout/2][Wout/2][out_cs] — a thread of the
  // epilogue holds exactly one pooling window (its tile), so the pool costs three maxima and the full-size tensor is never written
  // (res_models.py:101-105: every encoder block is followed by MaxPool2d(2))
  // small-P kernel: ceil(2^32 / d) of the divisors its block decode and its loaders' pixel decode use (0: divide) — tiles of the
  // problem, pixel tiles, pixels per image, output width, 32-deep sub-chunks per tap, kernel width — and the chunks per K slice.
  // Filled by api.hip once the tile width and the split are chosen: an integer division is a ~25-instruction reciprocal sequence
  // through the vector unit, eight of them sat in front of the first DMA of every launch of a step
  unsigned sp_m_tiles, sp_m_npt, sp_m_hw, sp_m_w, sp_m_kcpt, sp_m_kw;
  int sp_cps, sp_bn;
  // small-P kernel, 64-pixel tiles: 1 = this 3x3 layer (one image, even H and W, w_wino present) runs in the Winograd F(2x2, 3x3) form;
  // its K slices then count 32-channel sub-chunks (sp_cps of them per slice) and sp_m_tw is ceil(2^32 / (Wout / 2)) (0: divide)
  int sp_wino;
  unsigned sp_m_tw;
  int pool2;
  // AFFINE, Winograd kernel only: `add` is a half-resolution tensor [n][Hout/2][Wout/2][add_cs] read with nearest x2 upsampling — the
  // four pixels of a tile share one source pixel (the identity skip of a residual block whose input is upsampled on read)
  int add_up;
  int gate_from;          // AFFINE with out2: output channels c >= gate_from are reset gates; out2[c - gate_from] = (1 - y) * e1[c - gate_from]
  // sparse (gather) convolution: the input row of output row p under kernel tap t is gather[p*KH + t]
  // (-1: inactive site); geometry is then n_img = 1, Hout = 1, Wout = number of output rows, KW = 1
  const int* gather;
  // ... and, optional (LDS-DMA kernel): one word per 64 consecutive output rows, bit t set when at least one of them has an input row under
  // tap t.  A pixel tile then walks only the taps that are live for it (the rows are kept sorted by neighbour mask so that whole taps
  // drop out of a tile: models/sparse_encoder.py); dropped taps would have gathered zero rows, the sums are bitwise the same
  const unsigned* tap_mask;
  int tap_mask_n;
  // small-P kernel, one image: the SE gate of the input (res_models.py:161-165) is computed in the consuming layer's
  // prologue from the per-tile channel sums the producer wrote: scale = sigmoid(fc2 relu(fc0 mean)), every workgroup
  // for itself; workgroup 0 also stores it to se_out (the residual of the next layer is scaled by it)
  // SAMPLE epilogue without an eps tensor (e0 == null): {seed, offset} record on the device + the draw index of this call
  const unsigned long long* philox;
  int draw;
  const float* se_sum;    // [se_nt][c0] per-tile channel sums (null: in_scale holds the gate, or no gate)
  const float* se_fc0;    // [se_cr][c0]
  const float* se_fc2;    // [c0][se_cr]
  float* se_out;          // [c0] (may be null)
  int se_nt, se_cr;
  float se_inv_hw;
  // small-P kernel, LNG epilogue: a following 1x1 + LayerNorm + GELU layer (the middle layer of the trusting-gate body,
  // convolutions.py:358-360) applied to the tile before it leaves the workgroup: out2 = GELU(LN(fuse_w . GELU(LN(conv)))).
  // fuse_w: packed [fuse_cout_pad <= 64][fuse_kpad <= 64] (K = this layer's cout); the first layer's own output is not stored
  const float* fuse_w;
  const float* fuse_scale;   // LayerNorm weight / bias of the fused layer
  const float* fuse_bias;
  float* fuse_out;           // [P][fuse_cout]
  int fuse_cout, fuse_cout_pad, fuse_kpad;
  // small-P kernel: a K-partial sum of this layer that an EARLIER launch left as a plain [P][acc_cs] tensor (the half of
  // a two-input layer whose operand was known early, computed beside other work): added to the accumulator sums before
  // the epilogue.  Fixed order (own K range summed first, then + acc_in): bitwise reproducible.
  const float* acc_in;
  int acc_cs;
  // opt-in math mode bf16x3: the packed weights split into bf16 pieces ([8 hi][8 lo] per aligned group of 8 K values, same
  // bytes and row pitch as `w`); use_w3 is set by the host when the launch runs on a kernel that has the split-bf16 K loop
  int use_w3;
  const void* w3;
  // Winograd F(2x2, 3x3): the packed weights transformed at pack time, U[cin_pad / 16][16 positions][cout_pad][16] (conv_wino.hip);
  // null: the layer runs in the direct form
  const float* w_wino;
  // split-K hand-off: 1 = the round-1 form as a known-good reference (SF_HANDOFF_FENCED=1): an agent-scope release fence
  // before the arrival ticket and an acquire fence behind it, on top of the sc1 stores / loads (tests/test_gpu_splitk_stress.py)
  int fenced;
};

#define SF_MAX_GROUP 4
struct ConvLaunch {
  ConvProblem p[SF_MAX_GROUP];
  int nprob;
  int xcd_shift;   // LDS-DMA kernel, large launches: log2 of the XCD tile chunk + 1 (0: workgroup b computes tile b)
  int stamp_slot;  // diagnostic builds (-DSF_STAMP): launch slot of the in-kernel time stamps
  // small-P kernel, compact 1-D grid: problem i owns the logical workgroups [wg_base[i], wg_base[i + 1]), K slice major,
  // then cout tile, then pixel tile (wg_base[nprob] = grid size; all zero: the 3-D grid of tiles x problems x slices)
  int wg_base[SF_MAX_GROUP + 1];
  // Winograd kernel: ceil(2^32 / d) of the block decode's divisors (0: d = 1) — workgroups per problem of a group (1: one problem),
  // cout blocks, tile-block columns, tile-block rows, tile columns per image
  unsigned wn_m[6];
};

// Persistent "flow" form of the small-P kernel (conv_sp.hip: sp_flow_kernel, SF_PERSIST=1): every launch group of a rollout —
// what one conv_sp_kernel launch was — is a PHASE of one persistent launch.  The phase and problem tables live in device memory
// (the caller's workspace, written by small writer kernels at the start of the call) and are read through the constant address
// space (scalar loads).  One 768-thread workgroup per CU stays resident and runs item `wg` of every phase in order.  Ordering
// is by dataflow, not by a grid barrier: finished tiles are counted per (phase, pixel tile) in `done`, and an item starts when
//   (1) every tile of phase q-2 is done (covers every input older than the previous phase and every buffer-reuse hazard:
//       api.hip's scratch aliasing has a reuse distance of >= 2 phases), and
//   (2) the tiles of phase q-1 under its halo are done (all of them for an SE gate, which is a global reduction).
#define SP_FLOW_MAX_TILES 80         // pixel tiles of one phase (one 50x50 latent on 32-pixel tiles: 79)
#define SP_PHASE_CONV 0
struct FlowPhase {
  int nprob, prob0;              // problems p[prob0 .. prob0 + nprob) of the flow
  int epi, scaled, nt;           // kernel variant of the phase (epilogue family, SE-scaled inputs, 16-pixel tiles per wave: 2 | 4)
  int n_wg;                      // workgroups that have an item in this phase (the others idle or copy)
  int wg_base[SF_MAX_GROUP + 1];
  int bn, n_ptiles;              // pixels per tile, pixel tiles
  // Counters (dwords of `done`, zero at the start of the flow).  Every counter that many workgroups poll sits on a line of its
  // own — 240 pollers on the three lines that held a phase's 80 tile counters serialised behind each other and behind the
  // finishers' atomics (MI355X_MICROARCH.md: one word saturates at ~88 accesses / us; replicas: hand-off table row 2):
  //   tile t:   done[tile_base + t * SP_FLOW_TILE_STRIDE] += 1 per finished (problem, cout tile) of pixel tile t  -> tile_expect
  //   total:    done[tot_base + r * SP_FLOW_TOT_STRIDE], r = 0..7, EVERY replica += 1 per finished item and per copying
  //             workgroup -> tot_expect; a waiter polls replica (wg & 7)
  int tile_base, tile_expect, tot_base, tot_expect;
  int halo_px;                   // reach of the phase's convolutions in linear pixels (pad * W + pad, the largest of its problems)
  int dep_full;                  // 1: wait for ALL of phase q-1 (an SE gate in the prologue is a global reduction)
  int prev_bn, prev_ntiles, prev_tile_base, prev_tile_expect, prev_tot_base, prev_tot_expect;      // phase q-1 (prev_ntiles == 0: none)
  int lag_tot_base, lag_tot_expect;                                                                 // phase q-2 (lag_tot_expect == 0: none)
  int copy_n4;                   // optional state copy-out riding in this phase (float4 count; 0: none): workgroups without an
  const float* copy_src;         //   item copy src -> dst after the phase's dependency wait (src is an output of phase q-1)
  float* copy_dst;
  int pad_[4];
};
static_assert(sizeof(FlowPhase) % 16 == 0, "table pieces are written 16 bytes at a time");
#define SP_FLOW_TILE_STRIDE 16       // dwords between tile counters (64 B)
#define SP_FLOW_TOT_STRIDE 32        // dwords between the replicas of a phase total (128 B)
#define SP_FLOW_PHASE_DWORDS (SP_FLOW_MAX_TILES * SP_FLOW_TILE_STRIDE + 8 * SP_FLOW_TOT_STRIDE)
struct SpFlow {
  int nphase;
  int timeout_polls;             // every spin is bounded: after this many polls the workgroup gives up, sets err[0] and runs on
  const FlowPhase* ph;           // device tables
  const ConvProblem* p;
  unsigned int* done;            // tile counters of the flow (zero at its start)
  unsigned int* err;             // [0]: number of timed-out waits (0 after a healthy run)
};
static_assert(sizeof(ConvProblem) % 8 == 0, "problems are stored back to back in the table");
// table writer: up to SP_WRITER_BYTES of a table per launch, passed by value (one or two launches per rollout)
#define SP_WRITER_BYTES 61440     /* measured on this runtime (tools/r04/kernarg_probe.hip): kernel arguments of 64 KB launch and capture fine */
struct FlowBlob { unsigned char b[SP_WRITER_BYTES]; };
// every struct that travels as a kernel argument, against the sizes the launch paths were measured with
static_assert(sizeof(ConvLaunch) <= 4096, "ConvLaunch is passed by value: one 4-KB kernarg page");
static_assert(sizeof(SpFlow) <= 64, "SpFlow is passed by value");
static_assert(sizeof(FlowBlob) + 64 <= 65536, "table writer: blob + the scalar arguments beside it stay below the probed 64 KB");

// Diagnostic builds only (-DSF_STAMP, tools/r02/stamps.py): wave 0 of every workgroup records s_memrealtime (100 MHz)
// at fixed points of the kernel into a debug buffer no other code reads.  The product build compiles none of it.
#ifdef SF_STAMP
#define SF_STAMP_WGS 4096
// g_sf_stamps is defined in conv_igemm.hip (no relocatable device code: only the kernels of that file are stamped)
#define SF_STAMP_AT(L, k)                                                                                             \
  do {                                                                                                                \
    if (g_sf_stamps && threadIdx.x == 0) {                                                                            \
      const size_t wg_ = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                      \
      if (wg_ < SF_STAMP_WGS) g_sf_stamps[((size_t)(L).stamp_slot * SF_STAMP_WGS + wg_) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                                                 \
  } while (0)
#define SF_STAMP_VAL_T(L, k, v, t)                                                                                    \
  do {                                                                                                                \
    if (g_sf_stamps && threadIdx.x == (t)) {                                                                          \
      const size_t wg_ = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                      \
      if (wg_ < SF_STAMP_WGS) g_sf_stamps[((size_t)(L).stamp_slot * SF_STAMP_WGS + wg_) * 16 + (k)] = (v);            \
    }                                                                                                                 \
  } while (0)
#define SF_STAMP_VAL(L, k, v)                                                                                         \
  do {                                                                                                                \
    if (g_sf_stamps && threadIdx.x == 0) {                                                                            \
      const size_t wg_ = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                      \
      if (wg_ < SF_STAMP_WGS) g_sf_stamps[((size_t)(L).stamp_slot * SF_STAMP_WGS + wg_) * 16 + (k)] = (v);            \
    }                                                                                                                 \
  } while (0)
#else
#define SF_STAMP_AT(L, k) do { } while (0)
#define SF_STAMP_VAL(L, k, v) do { } while (0)
#define SF_STAMP_VAL_T(L, k, v, t) do { } while (0)
#endif

}  // namespace sf
