// Host orchestration + C ABI (include/sfnative.h) of libsfnative.so.  gfx950 only.
// Every function only enqueues kernels on the caller's stream: no allocation, no sync, no
// global mutable state (graph-capture safe).
#include "sf_device.h"
#include "../../include/sfnative.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace sf {
hipError_t launch_conv(const ConvLaunch& L, int epi, int cfg, hipStream_t stream);
hipError_t launch_conv_direct(const ConvLaunch& L, int epi, int mt, int ks, hipStream_t stream);
hipError_t launch_conv_glds(const ConvLaunch& L, int epi, int tile, int variant, hipStream_t stream);
hipError_t set_stamp_buffer(unsigned long long* p);
int glds_occupancy(int which);
hipError_t set_stamp_buffer_sp(unsigned long long* p);
hipError_t set_stamp_buffer_wino(unsigned long long* p);
hipError_t launch_conv_sp(const ConvLaunch& L, int epi, bool scaled, int bn, hipStream_t stream);
hipError_t launch_conv_wino(const ConvLaunch& L, int epi, hipStream_t stream);
bool wino_takes(const ConvProblem& q, int epi);
bool wino_same_geometry(const ConvProblem& a, const ConvProblem& b);
int wino_variant(const ConvProblem& q);
double wino_tiles(const ConvProblem& q);
hipError_t launch_sp_flow(const SpFlow& F, int grid, bool b3, hipStream_t stream);
hipError_t launch_flow_write(const void* host_src, void* dev_dst, size_t bytes, hipStream_t stream);
bool sp_flow_has(int epi, bool scaled, int bn);
int sp_flow_capacity(bool b3);
hipError_t launch_convnext_mlp(const float* t, const float* x, float* out, const float* w1, const float* s1, const float* b1, const float* w2,
                               const float* s2, const float* b2, long P, hipStream_t stream);
hipError_t launch_transpose(const float* in, float* out, int n, int rows, int cols, hipStream_t s);
hipError_t launch_transpose_strided(const float* in, float* out, int n, int rows, int cols, size_t in_stride, size_t out_stride,
                                    hipStream_t s);
hipError_t launch_maxpool2(const float* in, float* out, int n, int Hin, int Win, int C, int ceil_pad, hipStream_t s);
hipError_t launch_mean_from_partials(const float* part, float* out, int n, int nslab, int C, int hw, hipStream_t s);
hipError_t launch_logsigmoid(const float* in, float* out, size_t n, hipStream_t s);
hipError_t launch_upsample2(const float* in, float* out, int n, int Hin, int Win, int C, hipStream_t s);
hipError_t launch_broadcast_channels(const float* vec, float* out, int n, int HW, int k, int out_cs, int out_co, hipStream_t s);
hipError_t launch_upsample_bilinear2_add(const float* in, const float* skip, float* out, int n, int Hin, int Win, int C,
                                         hipStream_t s);
hipError_t launch_se_fc(const float* chansum, int ntile, int C, int Cr, int hw, const float* fc0,
                        const float* fc2, float* scale, int n_img, hipStream_t s);
hipError_t launch_chan_partial(const float* in, float* part, int n, int HW, int C, int nslab, hipStream_t s);
hipError_t launch_dwconv7_ln(const float* in, float* out, const float* wdw, const float* bdw, const float* lnw,
                             const float* lnb, int n, int H, int W, int C, float eps, hipStream_t s);
hipError_t launch_aspp_pool(const float* in, float* part, float* bias_img, int n, int HW, int C, int hid,
                            const float* w1, const float* s1, const float* b1, const float* wp, const float* ps,
                            const float* pb, int nslab, hipStream_t s);
}  // namespace sf

using namespace sf;

namespace {

// Zero fill as a KERNEL (16-byte stores): inside a captured rollout these are kernel nodes of the graph like everything
// around them.  (With hipMemsetAsync nodes the replayed 46-step RK4 rollout differed from eager by ~3e-6 from the third
// replay on — tools/r02, profiles/README.md "graph memset nodes" — while eager never did.)
__global__ void zero_fill_kernel(float4* __restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// device-to-device copy as a kernel node, for the same reason (n floats, a multiple of 4; both pointers 16-byte aligned)
hipError_t copy_floats(const float* src, float* dst, size_t n, hipStream_t st) {
  const size_t n4 = n / 4;
  if (n4 == 0 || (n & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
    return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n4);
  return hipGetLastError();
}
// persistent flow: a dependency wait that timed out leaves err[0] != 0 and the rollout's results undefined — make that loud instead of
// silent: every output of the call becomes NaN.  A kernel, so a captured graph carries the check with it.
__global__ void flow_poison_kernel(const unsigned* __restrict__ err, float* __restrict__ a, size_t na, float* __restrict__ b, size_t nb) {
  if (__builtin_nontemporal_load(err) == 0) return;
  const float q = __builtin_nanf("");
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * blockDim.x) {
    if (i < na) a[i] = q;
    else b[i - na] = q;
  }
}
thread_local const unsigned* g_flow_err_last = nullptr;      // error word of this thread's most recent persistent rollout (sf_flow_errors)
hipError_t zero_fill(void* p, size_t bytes, hipStream_t st) {     // p 16-byte aligned, bytes a multiple of 16 (arena blocks are)
  const size_t n4 = bytes / 16;
  if (n4 == 0) return hipSuccess;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<float4*>(p), n4);
  return hipGetLastError();
}

#define SF_TRY(expr)                                   \
  do {                                                 \
    int _st = (expr);                                  \
    if (_st != SF_OK) return _st;                      \
  } while (0)
#define SF_HIP(expr)                                   \
  do {                                                 \
    if ((expr) != hipSuccess) return SF_ERR_LAUNCH;    \
  } while (0)

struct Arena {
  float* base;
  size_t cap, off;
  Arena(float* b, size_t bytes) : base(b), cap(bytes / sizeof(float)), off(0) {}
  float* take(size_t n) {
    size_t a = (n + 63) & ~size_t(63);
    if (!base || off + a > cap) { off = cap + 1; return nullptr; }
    float* p = base + off;
    off += a;
    return p;
  }
  bool ok() const { return off <= cap; }
};
inline size_t al(size_t n) { return (n + 63) & ~size_t(63); }

int large_p();   // pixels from which the 64x64 / 64x128 tiles are used (12288; SF_LARGE_P)
#define LARGE_P large_p()
constexpr int ASPP_SLABS = 64;

int pick_cfg(int P, int epi) {
  if (epi == EPI_LNG || epi == EPI_TRUST) return P >= LARGE_P ? 2 : 0;
  return P >= LARGE_P ? 1 : 0;
}

// Fill a problem from a packed layer + geometry.  Output spatial size follows the conv formula.
ConvProblem problem(const sf_conv_w& w, const float* in0, const float* in1, float* out, int n_img, int Hin,
                    int Win, int in_up = 0) {
  ConvProblem p;
  std::memset(&p, 0, sizeof(p));
  p.in0 = in0; p.in1 = in1; p.w = w.w; p.w3 = w.w_bf16x3; p.w_wino = w.w_wino; p.scale = w.scale; p.bias = w.bias; p.out = out;
  p.c0 = w.c0; p.c1 = w.c1; p.in0_cs = w.c0; p.in1_cs = w.c1;
  p.n_img = n_img; p.Hin = Hin; p.Win = Win; p.in_up = in_up;
  const int Hl = Hin << in_up, Wl = Win << in_up;
  p.KH = w.kh; p.KW = w.kw; p.dil = w.dil; p.stride = w.stride; p.pad = w.pad;
  p.Hout = (Hl + 2 * w.pad - w.dil * (w.kh - 1) - 1) / w.stride + 1;
  p.Wout = (Wl + 2 * w.pad - w.dil * (w.kw - 1) - 1) / w.stride + 1;
  p.cin_pad = w.cin_pad; p.ktot = w.kh * w.kw * w.cin_pad;
  p.cout = w.cout; p.cout_pad = w.cout_pad; p.act = w.act;
  p.add_cs = w.cout; p.out_cs = w.cout; p.out_co = 0; p.out2_cs = w.cout;
  p.eps = 1e-6f;
  p.clamp_from = -1;
  return p;
}

bool valid_w(const sf_conv_w& w) {
  return w.w && w.cout > 0 && (w.cout % 4) == 0 && (w.cout_pad % 16) == 0 && w.cout_pad >= w.cout &&
         (w.cin_pad % 32) == 0 && w.cin_pad >= w.c0 + w.c1 && (w.c0 % 4) == 0 && (w.c1 % 4) == 0 && w.kh > 0 &&
         w.kw > 0 && w.stride > 0 && w.dil > 0;
}

// ---- optional per-launch profiler (bench.py only; off by default, the only global state) ---------
struct ProfRec { int key; double flops, bytes; hipEvent_t a, b; };
struct Profiler {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
} g_prof;

// Tuning knobs, read once from the environment (experiments only; defaults are the shipped choice):
//   SF_DIRECT=0 disables the direct-fragment kernel, SF_DIRECT_MT / SF_DIRECT_KS force its tile
//   height / K-group count, SF_DIRECT_CPW sets the target chunks per wave.
struct Tune { int wino, wino_sp, wino_sp7, sp_short_tail, wsp_minsub, fork7, fork7_wgs, wino_min_p, flow_timeout, b3_small_tiles, wide64, seg_maxph, persist, fenced, b3, pipe, sp_fuse_1x1, mid_minch_ln, sp, sp_xcd, sp_split_wgs, sp_bn, sp_max_p, sp_wide_work, sp_fuse_se, direct, mt, ks, chunks_per_wave, split, split_target, split_min_chunks, split_from, mid_tiles, split_cfg, glds, glds_var, small_dma, large_p, narrow; };
const Tune& tune() {
  static const Tune t = [] {
    auto geti = [](const char* k, int d) { const char* v = std::getenv(k); return v ? std::atoi(v) : d; };
    Tune x;
    // 1: one latent: the launches of a rollout run as phases of ONE persistent flow kernel per cell boundary (conv_sp.hip: sp_flow_kernel;
    // workgroups flow from one layer's tile to the next on per-tile dependency counters, every wait bounded: sf_flow_errors).  Bitwise equal
    // to the launch-per-layer path in the same form of the layers, <= 1e-3 against the oracle (tests/test_gpu_persistent.py).  Round 6,
    // both measured in one session (profiles/r06_z_bench.json): 174.2 us per steady-state step against 141.5 for the launch path, whose
    // 3x3 / 7x7 layers run in the Winograd form the flow kernel does not have.  Off by default
    x.persist = geti("SF_PERSIST", 0);
    x.b3_small_tiles = geti("SF_B3_SMALL_TILES", 0);   // experiment: bf16x3 layers with 128-multiple cout on 64 x 128 tiles (3 workgroups per CU) instead of 128 x 128 (2)
    x.wide64 = geti("SF_WIDE64", 0);               // 1: 64-cout layers at >= 131072 pixels on 64 x 256 tiles (variant 10) instead of 64 x 128
    x.seg_maxph = geti("SF_SEG_MAXPH", 1 << 30);   // diagnostic: at most this many phases per persistent flow launch (1: every phase its own launch of the flow kernel)
    x.wino = geti("SF_WINO", 1);                   // layers packed with Winograd weights run conv_wino.hip from wino_min_p pixels (0: direct form everywhere)
    x.wino_min_p = geti("SF_WINO_MIN_P", 14000);       // measured (profiles/r04_zz_wino_min_p_sweep.txt, r04_zz_step_min_p_batched_latents.txt): 6 or more batched 50x50 latents
                                                     // and one 200x200 latent gain 6-11 % per ODE step, 5 latents / one 100x100 latent lose 4-7 %; 32 latents +1.6 % on the headline
    x.wino_sp = geti("SF_WINO_SP", 1);             // one latent (small-P kernel, launch path): its 3x3 layers run in the Winograd form too (conv_sp.hip; 0: direct form — the round-5 step)
    x.wino_sp7 = geti("SF_WINO_SP7", 1);           // ... and the trusting gate's 7x7 as nine Winograd 3x3 sub-kernels (144 instead of 196 products per 2x2 outputs; 0: direct form)
    x.sp_short_tail = geti("SF_SP_SHORT_TAIL", 1);  // ... short trailing problems of a group do not count against the 256-workgroup cap (see run())
    x.wsp_minsub = geti("SF_WSP_MINSUB", 1);       // ... a K slice of such a layer is at least this many 32-channel sub-chunks (measured: 1 -> 148.2 us per step, 2 -> 150.3)
    x.fork7 = geti("SF_FORK7", 0);                 // 1: one latent inside a rollout: conv_decoder_2 rides beside rb1.conv1 and the r2 half of the next cell's 7x7 runs on a forked stream beside the rest of infer_state.  Built, bitwise reproducible, oracle-tested (tests/test_gpu_persistent.py) and measured SLOWER: the 7x7 launch halves (38.5 -> 25.7 us) and the side launch does run beside the chain, but the two cross-queue dependencies per step cost more than they free — 159.5-161 us per step in a replayed hipGraph against 148.5 (profiles/r06_r_*).  Off by default
    x.fork7_wgs = geti("SF_FORK7_WGS", 88);        // ... workgroup budget of that side launch (the main stream's launches of the window have <= 160)
    x.flow_timeout = geti("SF_FLOW_TIMEOUT", 1 << 22);   // polls before a dependency wait of the flow kernel gives up (~1 us each: seconds); bring-up runs use a small value
    x.fenced = geti("SF_HANDOFF_FENCED", 0);       // 1: split-K hand-offs also run the agent-scope release / acquire fences of round 1 (known-good reference for the fence-free sc1 form; gfx950 only either way)
    x.b3 = geti("SF_BF16X3", 1);                   // layers packed with split-bf16 weights (opt-in at pack time) run the bf16x3 K loop where a kernel has one (0: exact fp32 even then)
    x.pipe = geti("SF_PIPE", 2);                   // one latent: branch 2 of the NEXT dual cell (gates2 -> cand2, functions of the state only) rides in the launches of infer_state, its conv_decoder_2 in the candidate launch (0: every cell on its own, 5 launches)
    x.sp = geti("SF_SP", 1);                       // small pixel counts: the loader / consumer kernel of conv_sp.hip (0: the round-1 kernels)
    x.sp_xcd = geti("SF_SP_XCD", 1);               // ... bit 0: compact 1-D grid (no idle workgroups: step 198 -> 195 us); bit 1: XCD-contiguous logical ids (measured: fabric traffic 156 -> 144 MB per step but 195 -> 203 us; tile-major 133 MB and 218 us — the round-robin spread of a layer's workgroups over the XCDs is the fast one)
    x.sp_split_wgs = geti("SF_SP_SPLIT_WGS", 240); // ... K ranges are split across about this many workgroups per launch
    x.sp_bn = geti("SF_SP_BN", 0);                 // ... pixels per tile (0: by the amount of work, see sp_bn)
    x.sp_max_p = geti("SF_SP_MAX_P", 4096);        // ... used below this many pixels (one 50x50 latent; measured: from two samples on the round-1 kernels are as fast or faster)
    x.sp_fuse_se = geti("SF_SP_FUSE_SE", 1);       // ... SE gates computed in the consuming layer's prologue (one sample)
    x.sp_wide_work = geti("SF_SP_WIDE_WORK", 1000);// ... 64-pixel tiles + split K from this many (64x64 tile) x (64-deep chunk) units per launch
    x.direct = geti("SF_DIRECT", 1);
    x.mt = geti("SF_DIRECT_MT", 0);
    x.ks = geti("SF_DIRECT_KS", 0);
    x.chunks_per_wave = geti("SF_DIRECT_CPW", 5);
    if (x.chunks_per_wave < 1) x.chunks_per_wave = 1;
    x.split = geti("SF_SPLIT", 1);                 // cross-workgroup split-K on 64x64 tiles (small P)
    x.split_target = geti("SF_SPLIT_WGS", 1024);   // round 2 (sc1 hand-off): 512 -> 1024, batch-8 step 754 -> 701 us    // aim for this many workgroups per launch
    x.split_min_chunks = geti("SF_SPLIT_MINCH", 2);
    x.mid_tiles = geti("SF_MID_TILES", 1300);      // round 2: 640 -> 1300 (the 200x200 latent splits its 7x7 too: 1364 -> 1311 us)
    x.glds = geti("SF_GLDS", 15);                  // LDS-DMA staging for large plain layers: bit 0 = 128-cout tiles, bit 1 = 64-cout tiles, bit 2 = LayerNorm-epilogue tiles, bit 3 = cross-workgroup split-K launches (0: register staging everywhere)
    x.glds_var = geti("SF_GLDS_VAR", -1);          // -1: shipped choice; 0..8: force a variant of launch_conv_glds (experiments)
    x.small_dma = geti("SF_SMALL_DMA", 1);         // >= 0: plain layers below LARGE_P run on the LDS-DMA kernel (32x32 tiles); bit 0: GRU candidates too (pre-gated state)
    x.narrow = geti("SF_NARROW", 9);             // tile variant for layers with <= 32 output channels (32 cout x 128 px; -1: the 64-row tiles)
    x.large_p = geti("SF_LARGE_P", 8192);      // measured: a 4-sample rollout (10000 px) is 18 % faster on the small-P kernels, 8 samples (20000 px) on the large tiles
    x.mid_minch_ln = geti("SF_MID_MINCH_LN", 1);   // LayerNorm-epilogue layers at mid P take the 64x64 tiles from this many K chunks (the 1x1 of the trusting gate: 4-sample step 415 -> 408 us; 8: the round-1 rule, 64x128 tiles for short K)
    x.sp_fuse_1x1 = geti("SF_SP_FUSE_1X1", 1);     // small-P kernel: the trusting gate's 1x1 layer runs inside the 7x7 layer's launch
    x.split_cfg = geti("SF_SPLIT_CFG", 4);         // tile config of the mid-P split-K launches without a LayerNorm epilogue (4 | 1)
    x.split_from = geti("SF_SPLIT_FROM", 100);     // only layers with at least this many K chunks (the 7x7)
    return x;
  }();
  return t;
}

int large_p() { return tune().large_p; }

// diagnostic builds (-DSF_STAMP): conv launches take consecutive slots (mod 64) of the stamp buffer
int g_stamp_slot = 0;
bool g_stamp_on = false;

// LDS-DMA kernel: a 32-deep K chunk must come from ONE source tensor (a single input, or two whose channel counts are
// multiples of the chunk depth); channels past cin are zero-filled by the range check either way
bool one_source_per_chunk(const ConvProblem& q) { return q.c1 == 0 || ((q.c0 % 32 == 0) && (q.c1 % 32 == 0)); }
// ... and an SE input scale is applied to the pixel fragments from a small LDS table: single input, <= 256 channels,
// a (<= 256-pixel) tile touching at most 4 images
bool scale_ok(const ConvProblem& q) { return !q.in_scale || (q.c1 == 0 && q.cin_pad <= 256 && (long)q.Hout * q.Wout >= 128); }

// Scratch for the cross-workgroup split-K path, carved from the caller's workspace by the
// top-level entry points (SplitScope) — thread-local pointer, no global allocation.
struct SplitCtx { float* slab; size_t slab_floats; unsigned* counters; int ncounters; };
thread_local SplitCtx* g_split = nullptr;
constexpr size_t SPLIT_SLAB_FLOATS = size_t(8) << 20;   // 32 MB: 2048 (tile, slice) pairs of 64x64 fp32
constexpr int SPLIT_COUNTERS = 4096;
constexpr size_t SPLIT_WS_FLOATS = SPLIT_SLAB_FLOATS + SPLIT_COUNTERS + 128;

// Small-P kernel (conv_sp.hip): every problem of the launch must be stageable by its loaders — no reset-gate multiply
// while staging (the gates launch pre-gates the state), no neighbour table, one source tensor per 32-deep sub-chunk,
// an SE input scale only on a single input of <= 256 channels (all problems or none), 32-bit DMA offsets
bool sp_takes(const ConvProblem* ps, int n, int epi) {
  if (!tune().sp) return false;
  int scaled = 0, unscalable = 0;
  for (int i = 0; i < n; ++i) {
    const ConvProblem& q = ps[i];
    const long Pi = (long)q.n_img * q.Hout * q.Wout;
    if (Pi >= tune().sp_max_p || q.gather || q.gate || q.out_planar || !one_source_per_chunk(q)) return false;
    if ((epi == EPI_LNG || epi == EPI_TRUST) && q.cout_pad > 64) return false;
    if (q.in_scale || q.se_sum) {
      ++scaled;
      if (q.c1 != 0 || q.cin_pad > 256 || (long)q.Hout * q.Wout < 32) return false;   // a 64-pixel tile touches <= 4 images
      if (q.se_sum && (q.n_img != 1 || q.c0 > 128 || q.c0 < 64 || q.se_cr < 1 || q.se_cr > 16 || q.c0 != q.cin_pad || q.se_nt > 20 * (512 / q.c0))) return false;
    } else if (q.c1 != 0 || q.cin_pad > 256 || (long)q.Hout * q.Wout < 32) {
      ++unscalable;      // could not ride in an SE-scaled launch (whose kernel multiplies every problem's input by a scale row: ones for this one)
    }
    const double span = (64.0 / ((double)q.Hout * q.Wout) + 2.0) * q.Hin * q.Win * 4.0;
    if (span * q.in0_cs >= 2147483648.0 || span * q.in1_cs >= 2147483648.0 || 4.0 * q.cout_pad * q.ktot >= 2147483648.0) return false;
  }
  // an SE-scaled launch may carry problems without a scale (round 6: conv_decoder_2 beside rb1.conv1) as long as each fits the scaled kernel's
  // staging — their rows of the scale table are ones
  if (scaled && (unscalable || (epi != EPI_AFFINE && epi != EPI_SAMPLE))) return false;
  return true;
}
// pixels per tile of the small-P kernel.  Measured on the conv launches of an Euler step at 50x50 (profiles/r02_*):
// 64-pixel tiles with the K range split across workgroups win where a launch has a lot of work (both gate / candidate
// pairs, the 128 -> 128 layers of p_model, the 7x7), 32-pixel tiles without a hand-off elsewhere
struct FlowBuilder;
extern thread_local FlowBuilder* g_seg;
// forked side launch of a rollout (fork_run below): > 0 while the side launch is being enqueued = its workgroup budget; g_fork_halves: the
// rollout uses the fork, so main-stream launches keep to the lower half of the split-K scratch
thread_local int g_fork_side = 0;
thread_local bool g_fork_halves = false;
// ... and which of its 3x3 layers take the Winograd F(2x2, 3x3) form there (conv_sp.hip, ConvProblem::sp_wino): one image with even
// sides (a 64-pixel tile is then 16 whole Winograd tiles and the tile counts of both forms agree), stride 1, pad 1, no dilation, inputs in
// whole 32-channel sub-chunks, cout in whole 64-row tiles, transformed weights packed; not inside a persistent flow (its tile-level
// dependencies are in linear pixels), not the LayerNorm layers (the 7x7 and its fused 1x1)
bool sp_wino_ok(const ConvProblem& q, int epi) {
  if (!tune().wino || !tune().wino_sp || g_seg || !q.w_wino) return false;
  // 3x3, or the trusting gate's 7x7 as nine 3x3 sub-kernels (SF_WINO_SP7=0: the 7x7 keeps the direct form)
  const bool k3 = q.KH == 3 && q.KW == 3 && q.pad == 1, k7 = q.KH == 7 && q.KW == 7 && q.pad == 3 && tune().wino_sp7 && epi == EPI_LNG;      // (the kernel carries the tap groups in its LayerNorm instantiation only)
  if (!(k3 || k7) || q.stride != 1 || q.dil != 1 || q.in_up || q.gather || q.gate || q.out_planar || q.pool2 || q.add_up) return false;
  if (q.fuse_w && epi != EPI_LNG) return false;
  if (q.n_img != 1 || (q.Hout & 1) || (q.Wout & 1) || q.Hin != q.Hout || q.Win != q.Wout || q.Wout < 4 || q.Hout < 4) return false;
  if ((q.c0 % 32) || (q.c1 % 32) || q.c0 + q.c1 != q.cin_pad || (q.cout_pad % 64)) return false;
  if (tune().b3 && q.w3) return false;
  return 4.0 * (k7 ? 9 : 1) * 16 * q.cout_pad * q.cin_pad < 2147483648.0;
}
// K units of a problem in the Winograd form: (tap group, 32-channel sub-chunk) pairs
int sp_wino_units(const ConvProblem& q) { return (q.KH == 7 ? 9 : 1) * (q.cin_pad / 32); }
int sp_bn(const ConvProblem* ps, int n, int epi) {
  if (tune().sp_bn) return tune().sp_bn;
  for (int i = 0; i < n; ++i)
    if (sp_wino_ok(ps[i], epi)) return 64;      // the Winograd form lives on the 64-pixel tiles
  double work = 0;
  for (int i = 0; i < n; ++i)
    work += (double)((ps[i].n_img * ps[i].Hout * ps[i].Wout + 63) / 64) * ((ps[i].cout_pad + 63) / 64) * ((ps[i].KH * ps[i].KW * (ps[i].cin_pad / 32) + 1) / 2);
  return work >= tune().sp_wide_work ? 64 : 32;
}
// pixels per SE partial-sum row the producing conv's epilogue writes (the SE gate kernel sums ceil(P / this) rows)
// (the producer's whole launch group decides its tile)
int chansum_tile_px(const ConvProblem* group, int n, int epi) { return sp_takes(group, n, epi) ? sp_bn(group, n, epi) : 16; }

// ---- persistent flow (one latent inside a rollout, SF_PERSIST=1): run() records its small-P launch groups as phases of ONE
// persistent launch (conv_sp.hip: sp_flow_kernel) instead of launching them.  The phase / problem tables are built on the host,
// written into the caller's workspace by small writer kernels (table pieces travel as kernel arguments: stateless and
// graph-capturable) and the flow kernel is launched when the rollout ends or something that is not a small-P launch intervenes.
constexpr size_t FLOW_TABLE_BYTES = size_t(1) << 20;       // phases + problems of one flow
constexpr int FLOW_DONE_COUNTERS = 1 << 20;                // counter dwords of one flow (SP_FLOW_PHASE_DWORDS = 1536 per phase: every polled counter on its own line)
constexpr size_t FLOW_WS_FLOATS = FLOW_TABLE_BYTES / 4 + FLOW_DONE_COUNTERS + 256;
struct FlowBuilder {
  std::vector<FlowPhase> phases;
  std::vector<ConvProblem> probs;
  hipStream_t st;
  unsigned char* table;          // device: [phases | problems]
  unsigned* done;                // device: tile counters (zeroed at the start of the rollout), done[-64 .. -1] = error words
  unsigned* err;
  int next_done = 0;
  int recorded = 0;              // phases recorded so far in this rollout (never reset: picks the split-K scratch half of the next phase)
  bool b3 = false;
  int grid = 0;
  int launches = 0;
  // a pending state copy-out (src = an output of the last recorded phase): rides in the next phase
  const float* copy_src = nullptr; float* copy_dst = nullptr; int copy_n4 = 0;
  FlowBuilder(unsigned char* t, unsigned* d, unsigned* e, hipStream_t s) : st(s), table(t), done(d), err(e) {}
  int problem_index(const ConvProblem& q) {      // identical problems (steady-state steps ping-pong between two sets) are stored once
    for (size_t i = probs.size(); i-- > 0 && probs.size() - i <= 64;)
      if (std::memcmp(&probs[i], &q, sizeof(q)) == 0) return (int)i;
    probs.push_back(q);
    return (int)probs.size() - 1;
  }
  int flush() {
    if (copy_n4 > 0 && !phases.empty()) {      // no later phase to ride in: a copy kernel behind the flow (kernel boundary = visibility)
      const int rc = launch();
      if (rc != SF_OK) return rc;
      if (copy_floats(copy_src, copy_dst, (size_t)copy_n4 * 4, st) != hipSuccess) return SF_ERR_LAUNCH;
      copy_n4 = 0;
      return SF_OK;
    }
    return launch();
  }
  int launch() {
    if (phases.empty()) return SF_OK;
    const size_t pb = phases.size() * sizeof(FlowPhase), qb = ((probs.size() * sizeof(ConvProblem)) + 15) & ~size_t(15);
    if (pb + qb > FLOW_TABLE_BYTES) return SF_ERR_WORKSPACE;
    // tables of the PREVIOUS flow launch of this call are dead once that launch has run: stream order
    std::vector<unsigned char> tmp(pb + qb, 0);
    std::memcpy(tmp.data(), phases.data(), pb);
    std::memcpy(tmp.data() + pb, probs.data(), probs.size() * sizeof(ConvProblem));
    if (launch_flow_write(tmp.data(), table, pb + qb, st) != hipSuccess) return SF_ERR_LAUNCH;
    SpFlow F;
    std::memset(&F, 0, sizeof(F));
    F.nphase = (int)phases.size();
    F.timeout_polls = tune().flow_timeout;      // ~1 us per poll round: seconds, then the wait gives up instead of hanging the GPU
    F.ph = reinterpret_cast<const FlowPhase*>(table);
    F.p = reinterpret_cast<const ConvProblem*>(table + pb);
    F.done = done;
    F.err = err;
    if (launch_sp_flow(F, grid, b3, st) != hipSuccess) return SF_ERR_LAUNCH;
    ++launches;
    phases.clear(); probs.clear();
    return SF_OK;
  }
  // record one launch group as a phase.  SF_ERR_UNSUPPORTED: the caller launches it the ordinary way (after flush()).
  int add(const ConvLaunch& L, int epi, bool scaled, int bn) {
    bool all3 = L.nprob > 0;
    for (int i = 0; i < L.nprob; ++i) all3 = all3 && L.p[i].w3 != nullptr && L.p[i].use_w3;
    const int cap = sp_flow_capacity(all3);
    if (L.wg_base[L.nprob] < 1 || L.wg_base[L.nprob] > cap || !sp_flow_has(epi, scaled, bn)) return SF_ERR_UNSUPPORTED;
    if (!phases.empty() && (all3 != b3 || (int)phases.size() >= tune().seg_maxph)) SF_TRY(flush());
    if ((phases.size() + 1) * sizeof(FlowPhase) + (probs.size() + L.nprob) * sizeof(ConvProblem) + 64 > FLOW_TABLE_BYTES ||
        next_done + SP_FLOW_PHASE_DWORDS > FLOW_DONE_COUNTERS) {
      SF_TRY(flush());
      if (next_done + SP_FLOW_PHASE_DWORDS > FLOW_DONE_COUNTERS) {      // counters used up: start over (stream order: the flow that used them is done)
        if (zero_fill(done, (size_t)FLOW_DONE_COUNTERS * sizeof(unsigned), st) != hipSuccess) return SF_ERR_LAUNCH;
        next_done = 0;
      }
    }
    b3 = all3;
    grid = cap;
    FlowPhase ph;
    std::memset(&ph, 0, sizeof(ph));
    // the problems of a phase sit next to each other in the table
    const int first = (int)probs.size();
    bool contiguous = true;
    int idx[SF_MAX_GROUP];
    for (int i = 0; i < L.nprob; ++i) { idx[i] = problem_index(L.p[i]); contiguous = contiguous && idx[i] == idx[0] + i; }
    if (!contiguous) {      // some were found earlier, some not: store the group again, in order
      probs.resize(first);
      for (int i = 0; i < L.nprob; ++i) probs.push_back(L.p[i]);
      idx[0] = first;
    }
    ph.nprob = L.nprob; ph.prob0 = idx[0];
    ph.epi = epi; ph.scaled = scaled ? 1 : 0; ph.nt = bn / 16;
    ph.n_wg = L.wg_base[L.nprob];
    for (int i = 0; i <= SF_MAX_GROUP; ++i) ph.wg_base[i] = L.wg_base[i];
    // every problem of a phase covers the same pixels (one latent): its tiles are counted per pixel tile
    const int Ptot = L.p[0].n_img * L.p[0].Hout * L.p[0].Wout;
    int expect = 0, halo = 0, full = 0;
    for (int i = 0; i < L.nprob; ++i) {
      const ConvProblem& q = L.p[i];
      if (q.n_img * q.Hout * q.Wout != Ptot || q.n_img != 1 || q.stride != 1 || q.in_up || q.Hin != q.Hout || q.Win != q.Wout) return SF_ERR_UNSUPPORTED;
      expect += (q.cout_pad + 63) / 64;
      const int ry = (q.KH - 1) / 2 * q.dil, rx = (q.KW - 1) / 2 * q.dil;
      const int h = ry * q.Win + rx;
      halo = h > halo ? h : halo;
      full = full || q.se_sum != nullptr;      // the SE gate of the prologue is a reduction over the whole producer
    }
    ph.bn = bn; ph.n_ptiles = (Ptot + bn - 1) / bn;
    if (ph.n_ptiles > SP_FLOW_MAX_TILES) return SF_ERR_UNSUPPORTED;
    ph.tile_base = next_done; ph.tile_expect = expect;
    ph.tot_base = next_done + SP_FLOW_MAX_TILES * SP_FLOW_TILE_STRIDE;
    next_done += SP_FLOW_PHASE_DWORDS;
    if (copy_n4 > 0) { ph.copy_n4 = copy_n4; ph.copy_src = copy_src; ph.copy_dst = copy_dst; copy_n4 = 0; }
    // finished items (every (problem, cout tile, pixel tile) once) + the workgroups that copy
    ph.tot_expect = expect * ph.n_ptiles + (ph.copy_n4 > 0 ? (grid - ph.n_wg > 0 ? grid - ph.n_wg : grid) : 0);
    ph.halo_px = halo; ph.dep_full = full;
    const size_t n = phases.size();
    if (n >= 1) {
      const FlowPhase& a = phases[n - 1];
      ph.prev_bn = a.bn; ph.prev_ntiles = a.n_ptiles; ph.prev_tile_base = a.tile_base; ph.prev_tile_expect = a.tile_expect;
      ph.prev_tot_base = a.tot_base; ph.prev_tot_expect = a.tot_expect;
    }
    if (n >= 2) {
      const FlowPhase& a = phases[n - 2];
      ph.lag_tot_base = a.tot_base; ph.lag_tot_expect = a.tot_expect;
    }
    phases.push_back(ph);
    ++recorded;
    return SF_OK;
  }
  int add_copy(const float* src, float* dst, size_t nfloats) {
    if ((nfloats & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || nfloats / 4 > 0x7fffffff || phases.empty() || copy_n4 > 0) return SF_ERR_UNSUPPORTED;
    copy_src = src; copy_dst = dst; copy_n4 = (int)(nfloats / 4);
    return SF_OK;
  }
};
thread_local FlowBuilder* g_seg = nullptr;
int g_flow_mode = -1;      // -1: SF_PERSIST from the environment (default 0); 0 / 1: set by sf_set_flow_mode
// anything that is not a small-P launch first sends the recorded phases on their way (stream order)
int seg_flush() { return g_seg ? g_seg->flush() : SF_OK; }

int run(const ConvProblem* ps, int n, int epi, hipStream_t st) {
  ConvLaunch L;
  std::memset(&L, 0, sizeof(L));
  if (n < 1 || n > SF_MAX_GROUP) return SF_ERR_INVALID;
  int P = 0;
  bool wide_ln = false;      // a LayerNorm / trust epilogue over 65..128 channels (hidden sizes above every shipped config)
  for (int i = 0; i < n; ++i) {
    L.p[i] = ps[i];
    L.p[i].fenced = tune().fenced;
    int Pi = ps[i].n_img * ps[i].Hout * ps[i].Wout;
    if (Pi > P) P = Pi;
    if ((epi == EPI_LNG || epi == EPI_TRUST) && ps[i].cout_pad > 128) return SF_ERR_UNSUPPORTED;
    wide_ln = wide_ln || ((epi == EPI_LNG || epi == EPI_TRUST) && ps[i].cout_pad > 64);
  }
  L.nprob = n;
  if (P <= 0) return SF_OK;
  for (int i = 0; i < n; ++i) {   // the staged kernels keep per-tap element offsets (relative to the tile's first image) in 32 bits
    const double e0 = 2.0 * ps[i].Hin * ps[i].Win * ps[i].in0_cs, e1 = 2.0 * ps[i].Hin * ps[i].Win * ps[i].in1_cs;
    if (!ps[i].gather && (e0 >= 2147483648.0 || e1 >= 2147483648.0)) return SF_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < n; ++i)      // planar output: the AFFINE epilogue of the LDS-staged kernels only (never silently ignored)
    if (ps[i].out_planar && (epi != EPI_AFFINE || ps[i].pl_div < 1 || ps[i].out2 || ps[i].chansum)) return SF_ERR_UNSUPPORTED;
  for (int i = 0; i < n; ++i)      // pooled output / half-size residual: the Winograd kernel only (res_block checks first; never silently ignored)
    if ((ps[i].pool2 || ps[i].add_up) && !(tune().wino && (double)ps[i].n_img * ps[i].Hout * ps[i].Wout >= tune().wino_min_p && wino_takes(ps[i], epi) &&
                         !(tune().b3 && ps[i].w3)))
      return SF_ERR_UNSUPPORTED;
  bool has_acc = false;      // K-partial inputs and the blend mode of the AFFINE epilogue: small-P kernel only (never silently ignored)
  for (int i = 0; i < n; ++i) has_acc = has_acc || ps[i].acc_in != nullptr || (epi == EPI_AFFINE && (ps[i].mode & 4));
  if (has_acc && (wide_ln || !sp_takes(ps, n, epi))) return SF_ERR_UNSUPPORTED;
  if (wide_ln) {
    SF_TRY(seg_flush());
    // all channels of a pixel must sit in one wave: 128 cout x 64 px tiles of the LDS-DMA kernel, whatever the pixel count
    for (int i = 0; i < n; ++i) {
      const ConvProblem& q = ps[i];
      const double span = (256.0 / ((double)q.Hout * q.Wout) + 2.0) * q.Hin * q.Win * 4.0;
      if (q.gate || q.gather || q.in_scale || q.se_sum || !one_source_per_chunk(q) || span * q.in0_cs >= 2147483648.0 ||
          span * q.in1_cs >= 2147483648.0 || 4.0 * q.cout_pad * q.ktot >= 2147483648.0)
        return SF_ERR_UNSUPPORTED;
    }
    if (tune().b3) {
      bool all3 = true;
      for (int i = 0; i < n; ++i) all3 = all3 && L.p[i].w3 != nullptr;
      for (int i = 0; i < n; ++i) L.p[i].use_w3 = all3 ? 1 : 0;
    }
    if (!g_prof.on) {
      SF_HIP(launch_conv_glds(L, epi, 5, 0, st));
      return SF_OK;
    }
    ProfRec r;      // key 5 = "dmaLN128x64" (streamingflow_amd/_lib.py KERNEL_NAMES)
    r.key = 5 * 8 + epi; r.flops = 0; r.bytes = 0;
    for (int i = 0; i < n; ++i) {
      const ConvProblem& q = ps[i];
      const double Pi = (double)q.n_img * q.Hout * q.Wout;
      const double K = (double)q.KH * q.KW * (q.c0 + q.c1);
      r.flops += 2.0 * Pi * q.cout * K;
      r.bytes += 4.0 * ((double)q.n_img * q.Hin * q.Win * (q.c0 + q.c1) + (double)q.cout * K + Pi * q.cout);
    }
    r.a = g_prof.get(); r.b = g_prof.get();
    SF_HIP(hipEventRecord(r.a, st));
    SF_HIP(launch_conv_glds(L, epi, 5, 0, st));
    SF_HIP(hipEventRecord(r.b, st));
    g_prof.recs.push_back(r);
    return SF_OK;
  }
  // Winograd F(2x2, 3x3) for the 3x3 / stride-1 layers of large launches (conv_wino.hip): 2.25x fewer MACs, exact fp32 arithmetic.
  // Groups are launched problem by problem (the kernel takes one); the profiler prices the launch at its EXECUTED FLOPs (key 16).
  if (tune().wino && P >= tune().wino_min_p && (epi == EPI_AFFINE || epi == EPI_BLEND || epi == EPI_LNG || epi == EPI_SAMPLE)) {      // (EPI_LNG: the 7x7 + LayerNorm layer as nine 3x3 tap groups; EPI_SAMPLE: the sampling layer)
    // the members of a group are independent layers: those the kernel takes run on it one by one, the others stay one group
    // (the ASPP group: three dilated 3x3 branches + the 1x1 branch)
    bool takes[SF_MAX_GROUP];
    int n_wino = 0;
    for (int i = 0; i < n; ++i) {
      takes[i] = (double)ps[i].n_img * ps[i].Hout * ps[i].Wout >= tune().wino_min_p && wino_takes(ps[i], epi) && !(tune().b3 && ps[i].w3);
      n_wino += takes[i] ? 1 : 0;
    }
    if (n_wino > 0) {
      SF_TRY(seg_flush());
      if (n_wino < n) {
        ConvProblem rest[SF_MAX_GROUP];
        int nr = 0;
        for (int i = 0; i < n; ++i)
          if (!takes[i]) rest[nr++] = ps[i];
        SF_TRY(run(rest, nr, epi, st));
      }
      // layers of identical geometry (the two branches of a dual cell) share ONE launch: their tails merge
      static const bool group_on = [] { const char* v = std::getenv("SF_WINO_GROUP"); return v ? std::atoi(v) != 0 : true; }();
      static const bool list_on = std::getenv("SF_WINO_LIST") != nullptr;
      if (group_on && !list_on && n_wino >= 2) {
        ConvLaunch WG;
        std::memset(&WG, 0, sizeof(WG));
        bool same = true;
        int first = -1;
        for (int i = 0; i < n; ++i) {
          if (!takes[i]) continue;
          if (first < 0) first = i;
          same = same && wino_same_geometry(ps[first], ps[i]);
          WG.p[WG.nprob++] = ps[i];
        }
        if (same && wino_tiles(ps[first]) * (ps[first].cout_pad / 64.0) * WG.nprob < 1.0e6) {      // (group decode by multiplication: < 2^32 / workgroups)
          WG.stamp_slot = g_stamp_slot;
          if (g_stamp_on) g_stamp_slot = (g_stamp_slot + 1) % 64;
          if (!g_prof.on) {
            // (the kernel's block decode multiplies by host-made reciprocals and refuses — hipErrorInvalidValue, nothing launched — a size whose
            // exactness check fails: the heuristic above is not that check, so a refused group runs as one launch per problem below)
            const hipError_t ge = launch_conv_wino(WG, epi, st);
            if (ge == hipSuccess) return SF_OK;
            if (ge != hipErrorInvalidValue) return SF_ERR_LAUNCH;
            (void)hipGetLastError();
            goto wino_one_by_one;
          }
          ProfRec r;
          const int wv = wino_variant(ps[first]);
          r.key = (16 + (wv == 4 ? 2 : wv)) * 8 + epi;
          r.flops = 0; r.bytes = 0;
          for (int i = 0; i < WG.nprob; ++i) {
            const ConvProblem& q = WG.p[i];
            r.flops += 2.0 * 16.0 * wino_tiles(q) * q.cout * (q.c0 + q.c1) * (q.KH == 7 ? 9 : 1);
            r.bytes += 4.0 * ((double)q.n_img * q.Hin * q.Win * (q.c0 + q.c1) + 16.0 * (q.KH == 7 ? 9 : 1) * q.cout * (q.c0 + q.c1) + (double)q.n_img * q.Hout * q.Wout * q.cout);
          }
          r.a = g_prof.get(); r.b = g_prof.get();
          SF_HIP(hipEventRecord(r.a, st));
          SF_HIP(launch_conv_wino(WG, epi, st));
          SF_HIP(hipEventRecord(r.b, st));
          g_prof.recs.push_back(r);
          return SF_OK;
        }
      }
    wino_one_by_one:
      for (int i = 0; i < n; ++i) {
        if (!takes[i]) continue;
        ConvLaunch W1;
        std::memset(&W1, 0, sizeof(W1));
        W1.p[0] = ps[i];
        W1.nprob = 1;
        W1.stamp_slot = g_stamp_slot;
        if (g_stamp_on) g_stamp_slot = (g_stamp_slot + 1) % 64;
        static const bool list = std::getenv("SF_WINO_LIST") != nullptr;      // debugging aid (tools/r05/wino_layers.py): every launch timed by itself
        if (list && !g_prof.on) {
          const ConvProblem& q = ps[i];
          hipEvent_t a, b;
          SF_HIP(hipEventCreate(&a)); SF_HIP(hipEventCreate(&b));
          SF_HIP(hipEventRecord(a, st));
          SF_HIP(launch_conv_wino(W1, epi, st));
          SF_HIP(hipEventRecord(b, st));
          SF_HIP(hipEventSynchronize(b));
          float ms = 0.f;
          SF_HIP(hipEventElapsedTime(&ms, a, b));
          (void)hipEventDestroy(a); (void)hipEventDestroy(b);
          std::fprintf(stderr, "[sf-wino] n=%d %dx%d c=%d+%d->%d epi=%d act=%d mode=%d add=%d add_scale=%d out2=%d in_scale=%d bias_img=%d clamp=%d dil=%d up=%d var=%d us=%.1f gflop=%.3f\n",
                       q.n_img, q.Hout, q.Wout, q.c0, q.c1, q.cout, epi, q.act, q.mode, q.add != nullptr, q.add_scale != nullptr, q.out2 != nullptr,
                       q.in_scale != nullptr, q.bias_per_img, q.clamp_from >= 0, q.dil, q.in_up, wino_variant(q), ms * 1e3,
                       2.0 * 16.0 * wino_tiles(q) * q.cout * (q.c0 + q.c1) * 1e-9);
          continue;
        }
        if (!g_prof.on) {
          SF_HIP(launch_conv_wino(W1, epi, st));
          continue;
        }
        const ConvProblem& q = ps[i];
        ProfRec r;
        const int wv = wino_variant(q);
        r.key = (16 + (wv == 4 ? 2 : wv)) * 8 + epi;   // _lib.KERNEL_NAMES: wino128x32t / wino64x64t / wino64x32t2 (also its form with concatenated images) / wino64x32t2dil
        const double tiles = wino_tiles(q);
        const double grp = q.KH == 7 ? 9.0 : 1.0;                   // tap groups of the 7x7 form
        r.flops = 2.0 * 16.0 * grp * tiles * q.cout * (q.c0 + q.c1);      // executed: 16 products per 2x2 outputs, (cin, cout) pair and tap group
        r.bytes = 4.0 * ((double)q.n_img * q.Hin * q.Win * (q.c0 + q.c1) + 16.0 * grp * q.cout * (q.c0 + q.c1) + (double)q.n_img * q.Hout * q.Wout * q.cout);
        r.a = g_prof.get(); r.b = g_prof.get();
        SF_HIP(hipEventRecord(r.a, st));
        SF_HIP(launch_conv_wino(W1, epi, st));
        SF_HIP(hipEventRecord(r.b, st));
        g_prof.recs.push_back(r);
      }
      return SF_OK;
    }
    static const bool why = std::getenv("SF_WINO_WHY") != nullptr;      // debugging aid: which large 3x3 launches keep the direct form
    if (why)
      for (int i = 0; i < n; ++i)
        if (ps[i].KH == 3 && !wino_takes(ps[i], epi))
          std::fprintf(stderr, "[sf] direct 3x3: n=%d/%d %dx%d c=%d+%d->%d stride=%d dil=%d up=%d gather=%d gate=%d in_scale=%d se_sum=%d nsplit=%d chansum=%d acc_in=%d fuse=%d mode=%d wino=%d\n",
                       i, n, ps[i].Hout, ps[i].Wout, ps[i].c0, ps[i].c1, ps[i].cout, ps[i].stride, ps[i].dil, ps[i].in_up, ps[i].gather != nullptr,
                       ps[i].gate != nullptr, ps[i].in_scale != nullptr, ps[i].se_sum != nullptr, ps[i].nsplit, ps[i].chansum != nullptr,
                       ps[i].acc_in != nullptr, ps[i].fuse_w != nullptr, ps[i].mode, ps[i].w_wino != nullptr);
  }
  if (sp_takes(ps, n, epi)) {
    // K ranges are split across workgroups (sc1 slab hand-off) so that the launch has about sp_split_wgs workgroups of
    // equal length: a 2500-pixel layer has only 40 tiles of 64 pixels per 64 output channels
    const int bn = sp_bn(ps, n, epi);
    bool wn_of[SF_MAX_GROUP];
    for (int i = 0; i < n; ++i) wn_of[i] = bn == 64 && sp_wino_ok(ps[i], epi);
    double work_total = 0;
    for (int i = 0; i < n; ++i) {
      const int tiles = ((ps[i].n_img * ps[i].Hout * ps[i].Wout + bn - 1) / bn) * ((ps[i].cout_pad + 63) / 64);
      // (a 32-channel sub-chunk of the Winograd form costs a 64-pixel tile what a 64-deep chunk of the direct form does: 64 MFMAs per wave)
      work_total += (double)tiles * (wn_of[i] ? sp_wino_units(ps[i]) : (ps[i].KH * ps[i].KW * (ps[i].cin_pad / 32) + 1) / 2);
    }
    const int wg_target = g_fork_side > 0 ? g_fork_side : tune().sp_split_wgs, wg_cap = g_fork_side > 0 ? g_fork_side : 256;
    const double per_wg = work_total / wg_target;      // chunks per workgroup at the target
    int ns_of[SF_MAX_GROUP], tiles_of[SF_MAX_GROUP], nch_of[SF_MAX_GROUP];
    int wgs = 0;
    const bool may_split = g_split && tune().split && bn == 64;
    for (int i = 0; i < n; ++i) {
      const ConvProblem& q = ps[i];
      tiles_of[i] = ((q.n_img * q.Hout * q.Wout + bn - 1) / bn) * ((q.cout_pad + 63) / 64);
      nch_of[i] = wn_of[i] ? sp_wino_units(q) : (q.KH * q.KW * (q.cin_pad / 32) + 1) / 2;
      int ns = (may_split && per_wg > 0) ? (int)(nch_of[i] / per_wg + 0.5) : 1;
      const int min_per = wn_of[i] ? (tune().wsp_minsub > 0 ? tune().wsp_minsub : 1) : 3;      // at least 3 chunks (direct) / wsp_minsub sub-chunks (Winograd) per slice
      if (ns > nch_of[i] / min_per) ns = nch_of[i] / min_per;
      if (ns > 8) ns = 8;
      if (ns < 1) ns = 1;
      ns_of[i] = ns;
      wgs += tiles_of[i] * ns;
    }
    // one workgroup owns a whole CU: a launch of more than 256 of them runs a second round for the few that are left
    // (measured: the 7x7 + projection launch at 280 workgroups took 54 us for 26 us of work per workgroup)
    // ... unless the ones that are left are SHORT and come last (round 6, SF_SP_SHORT_TAIL): problems whose workgroups do less than a third of the
    // longest one's chunks — the trusting gate's 1x1 projection beside its 7x7 — are not counted against the cap when they are the last
    // problems of the group: the dispatcher hands them to the few free CUs while the long ones run
    auto long_wgs = [&]() {
      if (!tune().sp_short_tail) return wgs;
      double cmax = 0;
      for (int i = 0; i < n; ++i) cmax = std::max(cmax, (double)nch_of[i] / ns_of[i]);
      int cnt = wgs;
      for (int i = n - 1; i >= 1; --i) {      // trailing problems only
        if ((double)nch_of[i] / ns_of[i] * 3.0 > cmax) break;
        cnt -= tiles_of[i] * ns_of[i];
      }
      return cnt;
    };
    while (may_split && long_wgs() > wg_cap) {
      int k = -1;
      for (int i = 0; i < n; ++i)
        if (ns_of[i] > 1 && (k < 0 || tiles_of[i] * ns_of[i] > tiles_of[k] * ns_of[k])) k = i;
      if (k < 0) break;
      wgs -= tiles_of[k];
      --ns_of[k];
    }
    // inside a persistent flow consecutive phases overlap in time (a slice of phase q+1 may start while a last arriver of phase q
    // still reads its slabs): slabs and tickets alternate between the two halves of the scratch by phase parity (phase q+2 starts
    // only when phase q is complete)
    // (a running count of recorded phases, NOT the index inside the flow under construction: add() may flush first and the phase then
    // opens a new flow — with the index, the phase before the flush and the one after it could land on the same half)
    // (the same halves keep a launch on the forked side stream — fork_run: the r2 half of the next cell's 7x7 beside infer_state — apart from
    // the launches of the main stream that run at the same time: g_fork_halves is set for the whole rollout, g_fork_side inside fork_run)
    const bool halves = g_seg || g_fork_halves;
    const int parity = g_seg ? (g_seg->recorded & 1) : (g_fork_side ? 1 : 0);
    const size_t slab_lim = g_split ? (halves ? (parity + 1) * (g_split->slab_floats / 2) : g_split->slab_floats) : 0;
    const int cnt_lim = g_split ? (halves ? (parity + 1) * (g_split->ncounters / 2) : g_split->ncounters) : 0;
    size_t slab_off = (halves && g_split) ? parity * (g_split->slab_floats / 2) : 0;
    int cnt_off = (halves && g_split) ? parity * (g_split->ncounters / 2) : 0;
    for (int i = 0; i < n && may_split; ++i) {
      ConvProblem& q = L.p[i];
      int ns = ns_of[i];
      if (ns < 2) continue;
      const int cps = (nch_of[i] + ns - 1) / ns;
      ns = (nch_of[i] + cps - 1) / cps;      // every slice non-empty
      if (ns < 2) continue;
      const size_t per = (size_t)64 * bn;
      if (slab_off + (size_t)tiles_of[i] * ns * per > slab_lim || cnt_off + tiles_of[i] > cnt_lim) continue;
      q.nsplit = ns;
      q.slab = g_split->slab + slab_off;
      q.counters = g_split->counters + cnt_off;
      slab_off += (size_t)tiles_of[i] * ns * per;
      cnt_off += tiles_of[i];
    }
    for (int i = 0; i < n; ++i) {
      ConvProblem& q = L.p[i];
      q.sp_wino = wn_of[i] ? 1 : 0;
      if (wn_of[i]) {      // K slices in 32-channel sub-chunks (also when no reciprocals are made below)
        const int ns = q.nsplit > 1 ? q.nsplit : 1;
        q.sp_cps = (sp_wino_units(q) + ns - 1) / ns;
      }
    }
    L.stamp_slot = g_stamp_slot;
    if (g_stamp_on) g_stamp_slot = (g_stamp_slot + 1) % 64;
    // compact 1-D grid: problem i owns logical workgroups [wg_base[i], wg_base[i + 1]); bit 0 of sp_xcd: compact grid,
    // bit 1: XCD-contiguous logical ids
    if (tune().sp_xcd & 1) {
      int base = 0;
      for (int i = 0; i < n; ++i) {
        L.wg_base[i] = base;
        base += tiles_of[i] * (L.p[i].nsplit > 1 ? L.p[i].nsplit : 1);
      }
      for (int i = n; i <= SF_MAX_GROUP; ++i) L.wg_base[i] = base;
      L.xcd_shift = (tune().sp_xcd >> 1) & 1;
    }
    // reciprocals of the divisors of the kernel's block decode and pixel decode (conv_sp.hip: sp_mdiv; exact while dividend x
    // divisor < 2^32: a launch has < 2^16 workgroups and < 2^13 pixels)
    {
      static const bool magic_on = [] { const char* v = std::getenv("SF_SP_MAGIC"); return v ? std::atoi(v) != 0 : true; }();
      auto magic = [](long d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); };
      for (int i = 0; i < n && magic_on; ++i) {
        ConvProblem& q = L.p[i];
        const long Pi = (long)q.n_img * q.Hout * q.Wout, n_pt = (Pi + bn - 1) / bn, n_mt = (q.cout_pad + 63) / 64, tiles = n_pt * n_mt;
        const int ns = q.nsplit > 1 ? q.nsplit : 1, kcpt = q.cin_pad >> 5, nch_all = wn_of[i] ? sp_wino_units(q) : (q.KH * q.KW * kcpt + 1) >> 1;
        if (tiles * ns * tiles >= 0x100000000L || (Pi + 64) * q.Hout * q.Wout >= 0x100000000L) continue;
        q.sp_m_tw = magic(q.Wout / 2);
        // d = 1 has no reciprocal (0 = "divide"): dividing by one is what the fallback does
        q.sp_m_tiles = magic(tiles); q.sp_m_npt = magic(n_pt); q.sp_m_hw = magic((long)q.Hout * q.Wout); q.sp_m_w = magic(q.Wout);
        q.sp_m_kcpt = magic(kcpt); q.sp_m_kw = magic(q.KW);
        q.sp_cps = (nch_all + ns - 1) / ns;
        q.sp_bn = bn;
      }
    }
    bool scaled = false;
    for (int i = 0; i < n; ++i) scaled = scaled || (ps[i].in_scale != nullptr) || (ps[i].se_sum != nullptr);
    if (tune().b3) {      // opt-in math mode: every problem of the launch was packed with split-bf16 weights
      bool all3 = true;
      for (int i = 0; i < n; ++i) all3 = all3 && L.p[i].w3 != nullptr;
      for (int i = 0; i < n; ++i) L.p[i].use_w3 = all3 ? 1 : 0;
    }
    if (g_seg && !g_prof.on && (tune().sp_xcd & 1)) {      // inside a rollout: a phase of the persistent flow (diagnostic stamps: slot = phase index mod 64)
      const int rc = g_seg->add(L, epi, scaled, bn);
      if (rc == SF_OK) return SF_OK;
      if (rc != SF_ERR_UNSUPPORTED) return rc;
      SF_TRY(g_seg->flush());                                               // does not fit the flow kernel: an ordinary launch
    } else {
      SF_TRY(seg_flush());
    }
    if (!g_prof.on) {
      SF_HIP(launch_conv_sp(L, epi, scaled, bn, st));
      return SF_OK;
    }
    ProfRec r;
    r.key = 14 * 8 + epi; r.flops = 0; r.bytes = 0;
    for (int i = 0; i < n; ++i) {
      const ConvProblem& q = ps[i];
      const double Pi = (double)q.n_img * q.Hout * q.Wout;
      const double K = (double)q.KH * q.KW * (q.c0 + q.c1);
      r.flops += 2.0 * Pi * q.cout * K;
      r.bytes += 4.0 * ((double)q.n_img * q.Hin * q.Win * (q.c0 + q.c1) + (double)q.cout * K + Pi * q.cout);
    }
    r.a = g_prof.get(); r.b = g_prof.get();
    SF_HIP(hipEventRecord(r.a, st));
    SF_HIP(launch_conv_sp(L, epi, scaled, bn, st));
    SF_HIP(hipEventRecord(r.b, st));
    g_prof.recs.push_back(r);
    return SF_OK;
  }
  SF_TRY(seg_flush());
  int cfg = pick_cfg(P, epi);
  bool gathered = false;
  for (int i = 0; i < n; ++i) gathered = gathered || (ps[i].gather != nullptr);
  if (gathered) {            // sparse convolution: only the LDS-staged kernels read the neighbour table
    if (epi != EPI_AFFINE) return SF_ERR_UNSUPPORTED;
    if (cfg == 0) cfg = 1;
  }
  if (cfg == 1 && (epi == EPI_AFFINE || epi == EPI_BLEND)) {
    // measured (profiles/r01_e_sweep_large_tiles.txt): 128x128 tiles with 8 waves win 7-10 % once
    // there are >= ~1000 of them and cout is a multiple of 128; 64x64 wins everywhere else
    bool big = true;
    for (int i = 0; i < n; ++i)
      big = big && (ps[i].cout_pad % 128 == 0) && ((long)ps[i].n_img * ps[i].Hout * ps[i].Wout >= 131072);
    if (big && !(tune().b3_small_tiles && ps[0].w3 != nullptr)) cfg = 9;
  }
  // small pixel counts: direct-fragment kernel (no LDS staging), see conv_igemm.hip
  int mt = 0, ks = 1;
  if (cfg == 0 && tune().direct && !gathered) {
    const bool ln = (epi == EPI_LNG || epi == EPI_TRUST);
    int cp_max = 0, cp_gcd = 0, chunks = 0;
    bool ok = true;
    for (int i = 0; i < n; ++i) {
      const ConvProblem& q = ps[i];
      ok = ok && (q.c0 % 8 == 0) && (q.c1 % 8 == 0);
      cp_max = q.cout_pad > cp_max ? q.cout_pad : cp_max;
      int a = q.cout_pad, b = cp_gcd;
      while (b) { int t = a % b; a = b; b = t; }
      cp_gcd = a;
      int nc = q.KH * q.KW * (q.cin_pad / 32);
      chunks = nc > chunks ? nc : chunks;
    }
    if (ln) mt = (cp_max == 16 || cp_max == 32 || cp_max == 64) ? cp_max / 16 : 0;
    else mt = tune().mt ? tune().mt : ((cp_gcd % 32 == 0) ? 2 : 1)   /* measured: 32-row tiles (twice the workgroups) beat 64-row ones */;
    ks = tune().ks ? tune().ks : (chunks + tune().chunks_per_wave - 1) / tune().chunks_per_wave;
    ks = ks < 1 ? 1 : (ks > 8 ? 8 : ks);
    if (ok && mt) cfg = 3;
  }
  // ... or, preferred: 64x64 tiles with the K range split across workgroups (4x fewer weight
  // re-reads than 16-pixel tiles, and enough workgroups for all 256 CUs)
  int chunks_max = 0;
  for (int i = 0; i < n; ++i) {
    const int nc = ps[i].KH * ps[i].KW * (ps[i].cin_pad / 32);
    chunks_max = nc > chunks_max ? nc : chunks_max;
  }
  // measured (profiles/r01_d_sweep_convs.txt): the slab publish + ticket + acquire costs ~8 us, so
  // it only pays for the long-K layers (7x7: 196 chunks, 68 -> 44 us)
  int tiles_total = 0;
  for (int i = 0; i < n; ++i) {
    const int Pi = ps[i].n_img * ps[i].Hout * ps[i].Wout;
    tiles_total += ((Pi + 63) / 64) * ((ps[i].cout_pad + 63) / 64);
  }
  const bool split_small = (cfg == 0 || cfg == 3) && chunks_max >= tune().split_from;
  // a few batched samples (P = 8k..40k pixels): the large-tile kernels would have too few tiles,
  // each walking the whole K range; split the K range instead (also keeps LayerNorm layers on 64x64)
  const bool split_mid = (cfg == 1 || cfg == 2) && tiles_total < tune().mid_tiles && chunks_max >= (cfg == 2 ? tune().mid_minch_ln : 8);
  if ((split_small || split_mid) && tune().split && g_split) {
    // workgroup budget shared in proportion to each problem's work (tiles x chunks)
    double work_total = 0;
    for (int i = 0; i < n; ++i) {
      const int Pi = ps[i].n_img * ps[i].Hout * ps[i].Wout;
      work_total += (double)((Pi + 63) / 64) * ((ps[i].cout_pad + 63) / 64) * ps[i].KH * ps[i].KW * (ps[i].cin_pad / 32);
    }
    (void)tiles_total;
    size_t slab_off = 0;
    int cnt_off = 0;
    bool fits = true;
    for (int i = 0; i < n; ++i) {
      ConvProblem& q = L.p[i];
      const int Pi = q.n_img * q.Hout * q.Wout;
      const int tiles = ((Pi + 63) / 64) * ((q.cout_pad + 63) / 64);
      const int nc = q.KH * q.KW * (q.cin_pad / 32);
      int ns = (int)(tune().split_target * (double)nc / work_total + 0.5);
      if (ns > nc / tune().split_min_chunks) ns = nc / tune().split_min_chunks;
      if (ns > 16) ns = 16;
      if (ns < 1) ns = 1;
      const int cps = (nc + ns - 1) / ns;
      ns = (nc + cps - 1) / cps;      // every slice non-empty
      q.nsplit = ns;
      q.slab = g_split->slab + slab_off;
      q.counters = g_split->counters + cnt_off;
      slab_off += (size_t)tiles * ns * 4096;
      cnt_off += tiles;
      if (slab_off > g_split->slab_floats || cnt_off > g_split->ncounters) fits = false;
    }
    if (fits) cfg = (split_mid && tune().split_cfg == 1 && (epi == EPI_AFFINE || epi == EPI_BLEND || epi == EPI_SAMPLE)) ? 1 : 4;
    else for (int i = 0; i < n; ++i) { L.p[i].nsplit = 0; L.p[i].slab = nullptr; L.p[i].counters = nullptr; }
  }
  // Plain large layers (no reset gate / SE scale / neighbour table / split-K): LDS-DMA staging with the barrier in the
  // middle of the MFMA stream (conv_glds_kernel).  Measured (profiles/r01_v_*): 128x128 tiles +2 %, 64-cout layers
  // +7 % on 64x128 tiles once there are >= 1024 of them, +3 % on 64x64 tiles below that.
  int glds_tile = -1, glds_var = 4;
  if (tune().glds && (((cfg == 1 || cfg == 9) && (epi == EPI_AFFINE || epi == EPI_BLEND || epi == EPI_SAMPLE)) ||
                      (cfg == 2 && (epi == EPI_LNG || epi == EPI_TRUST) && (tune().glds & 4)))) {
    bool ok = true;
    long pmin = 1L << 40;
    for (int i = 0; i < n; ++i) {
      const ConvProblem& q = L.p[i];
      ok = ok && !q.gate && scale_ok(q) && q.nsplit <= 1;
      ok = ok && one_source_per_chunk(q);
      // 32-bit byte offsets: over the images a (<= 256-pixel) tile can touch (sparse: over all feature rows), and over the packed weights
      const double span = q.gather ? 4.0 * q.Win : (256.0 / ((double)q.Hout * q.Wout) + 2.0) * q.Hin * q.Win * 4.0;   // bytes per channel stride unit
      ok = ok && span * q.in0_cs < 2147483648.0 && span * q.in1_cs < 2147483648.0 && 4.0 * q.cout_pad * q.ktot < 2147483648.0;
      const long Pi = (long)q.n_img * q.Hout * q.Wout;
      pmin = Pi < pmin ? Pi : pmin;
    }
    if (ok && cfg == 9 && (tune().glds & 1)) glds_tile = 0;
    if (ok && cfg == 1 && (tune().glds & 2)) {
      glds_tile = 1; glds_var = pmin >= 131072 ? (tune().wide64 ? 10 : 6) : 4;
      bool narrow = true;      // every problem has at most 32 output channels: half of a 64-row tile would multiply zeros
      for (int i = 0; i < n; ++i) narrow = narrow && L.p[i].cout_pad <= 32;
      if (narrow && tune().narrow >= 0) glds_var = tune().narrow;
    }
    bool scaled = false;
    for (int i = 0; i < n; ++i) scaled = scaled || (L.p[i].in_scale != nullptr);
    if (ok && cfg == 2 && !scaled) glds_tile = 2;
    if (scaled && glds_tile == 2) glds_tile = -1;
  }
  // small pixel counts: the same kernel on 32x32 tiles beats the direct-fragment kernel by 5-15 % per plain layer
  // (profiles/r01_v_sweep_glds_wide_tiles.txt; single-sample rollout 5.10 -> 4.64 ms with the pre-gated candidates).
  // SF_SMALL_DMA=-1 switches it off, 0 keeps the reset gate in the candidate's staging.  LayerNorm epilogues on
  // 64x32 tiles were slower than the direct kernel and stay there.
  if ((tune().glds & 8) && cfg == 4) {   // cross-workgroup split-K launches (the slab hand-off is shared); SE-scaled layers on the SCALE instantiation
    bool ok = true;
    int nscaled = 0;
    for (int i = 0; i < n; ++i) nscaled += L.p[i].in_scale != nullptr;
    ok = (nscaled == 0 && epi != EPI_SAMPLE) || (nscaled == n && (epi == EPI_AFFINE || epi == EPI_SAMPLE));
    for (int i = 0; i < n; ++i) {
      const ConvProblem& q = L.p[i];
      const double span = (256.0 / ((double)q.Hout * q.Wout) + 2.0) * q.Hin * q.Win * 4.0;
      ok = ok && !q.gate && scale_ok(q) && !q.gather && one_source_per_chunk(q) && span * q.in0_cs < 2147483648.0 &&
           span * q.in1_cs < 2147483648.0 && 4.0 * q.cout_pad * q.ktot < 2147483648.0;
    }
    if (ok) glds_tile = 4;
  }
  if (tune().small_dma >= 0 && (cfg == 0 || cfg == 3) && (epi == EPI_AFFINE || epi == EPI_BLEND || epi == EPI_SAMPLE)) {
    bool ok = true;
    for (int i = 0; i < n; ++i) {
      const ConvProblem& q = L.p[i];
      ok = ok && !q.gate && scale_ok(q) && !q.gather && q.nsplit <= 1 && one_source_per_chunk(q);
    }
    if (ok) { glds_tile = 3; glds_var = tune().small_dma; }
    if (tune().glds_var >= 0) glds_var = tune().glds_var;
  }
  L.stamp_slot = g_stamp_slot;
  if (g_stamp_on) g_stamp_slot = (g_stamp_slot + 1) % 64;
  if (glds_tile >= 0 && tune().b3) {      // opt-in math mode: every problem of the launch was packed with split-bf16 weights
    bool all3 = true;
    for (int i = 0; i < n; ++i) all3 = all3 && L.p[i].w3 != nullptr;
    for (int i = 0; i < n; ++i) L.p[i].use_w3 = all3 ? 1 : 0;
  }
  auto launch = [&]() -> hipError_t {
    if (glds_tile >= 0) return launch_conv_glds(L, epi, glds_tile, glds_var, st);
    return cfg == 3 ? launch_conv_direct(L, epi, mt, ks, st) : launch_conv(L, epi, cfg, st);
  };
  if (!g_prof.on) {
    SF_HIP(launch());
    return SF_OK;
  }
  ProfRec r;
  r.key = (glds_tile < 0 ? cfg : glds_tile == 4 ? 15 : glds_tile == 0 ? 10 : glds_tile == 2 ? 13 : (glds_var == 6 || glds_var == 10) ? 12 : 11) * 8 + epi; r.flops = 0; r.bytes = 0;
  for (int i = 0; i < n; ++i) {
    const ConvProblem& q = ps[i];
    const double Pi = (double)q.n_img * q.Hout * q.Wout;
    const double K = (double)q.KH * q.KW * (q.c0 + q.c1);
    r.flops += 2.0 * Pi * q.cout * K;
    // algorithmic bytes: each input pixel and each weight once, output once
    r.bytes += 4.0 * ((double)q.n_img * q.Hin * q.Win * (q.c0 + q.c1) + (double)q.cout * K + Pi * q.cout);
  }
  r.a = g_prof.get(); r.b = g_prof.get();
  SF_HIP(hipEventRecord(r.a, st));
  SF_HIP(launch());
  SF_HIP(hipEventRecord(r.b, st));
  g_prof.recs.push_back(r);
  return SF_OK;
}
int run1(const ConvProblem& p, int epi, hipStream_t st) { return run(&p, 1, epi, st); }

// RAII: carve + zero the split-K scratch for the duration of one top-level call
struct SplitScope {
  SplitCtx ctx;
  SplitCtx* prev;
  bool active = false;
  SplitScope(Arena& A, hipStream_t st) : prev(g_split) {
    if (!tune().split) return;
    ctx.slab = A.take(SPLIT_SLAB_FLOATS);
    ctx.counters = reinterpret_cast<unsigned*>(A.take(SPLIT_COUNTERS));
    ctx.slab_floats = SPLIT_SLAB_FLOATS;
    ctx.ncounters = SPLIT_COUNTERS;
    if (!A.ok() || !ctx.slab || !ctx.counters) return;
    if (zero_fill(ctx.counters, SPLIT_COUNTERS * sizeof(unsigned), st) != hipSuccess) return;
    g_split = &ctx;
    active = true;
  }
  ~SplitScope() { g_split = prev; }
};

// ---- modules ---------------------------------------------------------------------------------

// conv-GRU cell (temporal.py:44-57): gates -> blend.  gates buffer g: [P][2C] = [u | r]
// rs: optional [P][C] scratch.  Large pixel counts: the gates epilogue also writes (1 - r) * s there and the candidate
// convolution reads cat[x, rs] as a plain layer (LDS-DMA staging); otherwise the gate is applied while staging.
bool pregate(long P, const sf_conv_w& cand) {
  const bool dma = P >= LARGE_P ? tune().glds != 0 : (tune().small_dma >= 0 && (tune().small_dma & 1));
  return dma && (cand.c0 % 32 == 0) && (cand.c1 % 32 == 0);
}
int gru_cell(const sf_gru_w& w, const float* x, const float* s, float* out, float* g, float* rs, int n, int H, int W,
             hipStream_t st, int ode_derivative = 0) {
  const int C = w.cand.cout;
  const bool pre = rs && pregate((long)n * H * W, w.cand);
  ConvProblem a = problem(w.gates, x, s, g, n, H, W);
  if (pre) { a.out2 = rs; a.out2_cs = C; a.e1 = s; a.e1_cs = C; a.gate_from = C; }
  SF_TRY(run1(a, EPI_AFFINE, st));
  ConvProblem b = problem(w.cand, x, pre ? rs : s, out, n, H, W);
  if (!pre) { b.gate = g; b.gate_cs = 2 * C; b.gate_co = C; }
  b.e0 = g; b.e0_cs = 2 * C; b.e1 = s; b.e1_cs = C;
  b.mode = ode_derivative ? 1 : 0;
  return run1(b, EPI_BLEND, st);
}

size_t dual_ws_floats(int C, int P) { return 2 * al((size_t)P * 2 * C) + 8 * al((size_t)P * C); }

// scratch of one dual cell (temporal_ode_bayes.py:92-131 / :239-275)
struct CellBufs {
  float *g1, *g2, *h1, *h2, *r2, *t1, *sk, *t2, *rs1, *rs2;
  bool take(Arena& A, int C, int P) {
    g1 = A.take((size_t)P * 2 * C); g2 = A.take((size_t)P * 2 * C);
    h1 = A.take((size_t)P * C); h2 = A.take((size_t)P * C); r2 = A.take((size_t)P * C);
    t1 = A.take((size_t)P * C); sk = A.take((size_t)P * C); t2 = A.take((size_t)P * C);
    rs1 = A.take((size_t)P * C); rs2 = A.take((size_t)P * C);
    return A.ok();
  }
};
void cell_gate_problems(const sf_dual_w& w, const float* x, const float* s, const CellBufs& b, bool pre, int B, int H, int W,
                        ConvProblem& g1, ConvProblem& g2) {
  const int C = w.C;
  g1 = problem(w.gates1, x, s, b.g1, B, H, W);
  g2 = problem(w.gates2, s, nullptr, b.g2, B, H, W);   // cell 2 sees cat[s,s]: duplicate input folded into the packed weights
  if (pre) {
    g1.out2 = b.rs1; g1.out2_cs = C; g1.e1 = s; g1.e1_cs = C; g1.gate_from = C;
    g2.out2 = b.rs2; g2.out2_cs = C; g2.e1 = s; g2.e1_cs = C; g2.gate_from = C;
  }
}
void cell_cand_problems(const sf_dual_w& w, const float* x, const float* s, const CellBufs& b, bool pre, int B, int H, int W,
                        ConvProblem& c1, ConvProblem& c2) {
  const int C = w.C;
  c1 = problem(w.cand1, x, pre ? b.rs1 : s, b.h1, B, H, W);
  c1.e0 = b.g1; c1.e0_cs = 2 * C; c1.e1 = s; c1.e1_cs = C;
  c2 = problem(w.cand2, s, pre ? b.rs2 : s, b.h2, B, H, W);
  c2.e0 = b.g2; c2.e0_cs = 2 * C; c2.e1 = s; c2.e1_cs = C;
  if (!pre) {
    c1.gate = b.g1; c1.gate_cs = 2 * C; c1.gate_co = C;
    c2.gate = b.g2; c2.gate_cs = 2 * C; c2.gate_co = C;
  }
}
// Small-P kernel: a 1x1 + LayerNorm + GELU layer `q` (weights w1) that follows the LayerNorm layer ps[0] is applied to ps[0]'s
// tile before it leaves the workgroup (conv_sp.hip, ConvProblem::fuse_*); ps[0]'s own output is then not stored.  Returns
// whether the launch of ps carries q.
bool fuse_following_1x1(ConvProblem* ps, int n, const ConvProblem& q, const sf_conv_w& w1, float* out) {
  if (!tune().sp_fuse_1x1 || !sp_takes(ps, n, EPI_LNG) || !sp_takes(&q, 1, EPI_LNG)) return false;
  if (w1.kh != 1 || w1.kw != 1 || w1.c1 != 0 || w1.cin_pad > 64 || w1.cout_pad > 64 || ps[0].cout_pad > 64 || !w1.scale || !w1.bias ||
      w1.c0 != ps[0].cout || q.add)
    return false;
  ps[0].out = nullptr;
  ps[0].fuse_w = w1.w; ps[0].fuse_scale = w1.scale; ps[0].fuse_bias = w1.bias; ps[0].fuse_out = out;
  ps[0].fuse_cout = w1.cout; ps[0].fuse_cout_pad = w1.cout_pad; ps[0].fuse_kpad = w1.cin_pad;
  return true;
}

int fork_join(hipStream_t st);
// trusting gate + mix + integrator update   (:124-131, convolutions.py:348-380)
int cell_tail(const sf_dual_w& w, const float* s, float* out, int derivative, const float* base, const float* coef,
              int coef_stride, float* out2, int acc2, const CellBufs& b, int B, int H, int W, hipStream_t st, const float* acc7 = nullptr) {
  ConvProblem ps[2];
  if (acc7) {      // the r2 half of the 7x7 was computed on the forked stream: wait for it, then the h1 half + its sums
    SF_TRY(fork_join(st));
    ps[0] = problem(w.tg7_h, b.h1, nullptr, b.t1, B, H, W);
    ps[0].acc_in = acc7; ps[0].acc_cs = w.C;
  } else {
    ps[0] = problem(w.tg7, b.h1, b.r2, b.t1, B, H, W);
  }
  ps[0].mode = 1;                                                            // 7x7 + LN + GELU
  ps[1] = problem(w.tgproj, b.h1, b.r2, b.sk, B, H, W); ps[1].mode = 0;   // 1x1 projection + GELU
  ConvProblem q = problem(w.tg1, b.t1, nullptr, b.t2, B, H, W); q.mode = 1;
  // one latent on the small-P kernel: the 1x1 + LN + GELU layer is applied to the 7x7 layer's tile before it leaves the
  // workgroup (one launch and one 0.64-MB round trip fewer per cell evaluation)
  const bool fuse_1x1 = fuse_following_1x1(ps, 2, q, w.tg1, b.t2);
  // batched latents: the 7x7 runs on the Winograd kernel (nine 3x3 tap groups, conv_wino.hip GRP = 9) by itself — the 1x1 projection, which only
  // reads h1 / r2, then shares the launch of the 1x1 + LN layer behind it instead of the 7x7's
  const bool wino7 = !fuse_1x1 && !acc7 && tune().wino && (double)B * H * W >= tune().wino_min_p && wino_takes(ps[0], EPI_LNG) && !(tune().b3 && ps[0].w3);
  if (wino7) {
    SF_TRY(run1(ps[0], EPI_LNG, st));
    ConvProblem g2[2] = {q, ps[1]};
    SF_TRY(run(g2, 2, EPI_LNG, st));
  } else {
    SF_TRY(run(ps, 2, EPI_LNG, st));
    if (!fuse_1x1) SF_TRY(run1(q, EPI_LNG, st));
  }
  ConvProblem f = problem(w.tg3, b.t2, nullptr, out, B, H, W);
  f.e0 = b.sk; f.e1 = w.w_logit; f.e2 = b.r2; f.e3 = b.h1; f.e4 = s; f.e5 = base ? base : s;
  f.coef = coef; f.coef_stride = coef_stride; f.out2 = out2; f.mode = (derivative ? 1 : 0) | (acc2 ? 2 : 0);
  if (derivative && !coef) return SF_ERR_INVALID;
  return run1(f, EPI_TRUST, st);
}

// Branch 2 of a dual cell — gru_cell_2(s, s) and conv_decoder_2 (temporal_ode_bayes.py:112-119 / :256-263) — is a function of
// the state alone.  In a rollout the state of cell j+1 is the output of cell j, so while infer_state(state) runs (five
// launches that only the NEXT cell's branch 1 waits for) branch 2 of the next cell is computed beside it: its gate and
// candidate convolutions ride in infer_state's first two launches (`Side`), its decoder convolution in the next cell's
// candidate launch.  One latent on the small-P kernel only (every launch there leaves CUs idle); same arithmetic, same
// summation orders as the cell on its own.
struct Carry {
  float *g2, *rs2, *h2;      // gates2 output [P][2C], (1 - r2) * s [P][C], blended hidden state of cell 2 [P][C]
  float* g1s;                // state half of gates1, raw sums [P][2C] (null: not carried)
  // round 6: rnn_state2 = conv_decoder_2(h2) [P][C] computed beside rb1.conv1, and the raw sums [P][C] of the r2 HALF of the trusting gate's
  // 7x7 (K = 49 x C of its 49 x 2C) computed on a forked stream beside the rest of infer_state and the next cell's first two launches
  // (null: not carried — the cell computes both itself)
  float *r2, *acc7;
};
struct Side {                // extra problems for infer_state's launches (AFFINE): [0] with conv1 + projection, [1], [2] with conv2
  ConvProblem p[3];
  int n;                     // 2 or 3
  ConvProblem dec2, s7;      // has_r2: conv_decoder_2 beside rb1.conv1, then the r2 half of the 7x7 on the forked stream
  bool has_r2;
};

// ---- fork / join around one side launch (one latent inside a rollout).  A launch of a step's chain rarely has more than 160 workgroups of
// 768 threads (one per CU): a third of the chip idles while the serial chain runs.  The r2 half of the next cell's 7x7 — half of the
// longest launch of a step, and a function of the state alone — is enqueued on a second stream between two events: eager, that is a
// second hardware queue; captured, a parallel branch of the hipGraph.  The library owns one non-blocking side stream and two events per
// host thread, created on first use (a creation that fails — e.g. inside a capture that forbids it — switches the fork off: the cell then
// computes the whole 7x7 itself).  Results do not depend on timing: the side launch writes its own tensor and its own half of the split-K
// scratch, and the main stream waits for it before the 7x7's launch.
struct Fork {
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int state = 0;             // 0: not tried, 1: ready, -1: unavailable
  bool pending = false;
};
thread_local Fork g_fork;
bool fork_ready() {
  Fork& f = g_fork;
  if (f.state == 0) {
    f.state = -1;
    if (hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&f.ev_fork, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&f.ev_join, hipEventDisableTiming) == hipSuccess)
      f.state = 1;
    else
      (void)hipGetLastError();
  }
  return f.state == 1;
}
int fork_run(const ConvProblem& q, int epi, int wgs, hipStream_t st) {
  Fork& f = g_fork;
  if (!fork_ready() || f.pending) return SF_ERR_UNSUPPORTED;
  SF_HIP(hipEventRecord(f.ev_fork, st));
  SF_HIP(hipStreamWaitEvent(f.side, f.ev_fork, 0));
  g_fork_side = wgs;
  const int rc = run(&q, 1, epi, f.side);
  g_fork_side = 0;
  SF_HIP(hipEventRecord(f.ev_join, f.side));      // (also after a failed launch: the side stream must rejoin a capture)
  f.pending = true;
  return rc;
}
int fork_join(hipStream_t st) {
  Fork& f = g_fork;
  if (!f.pending) return SF_OK;
  f.pending = false;
  SF_HIP(hipStreamWaitEvent(st, f.ev_join, 0));
  return SF_OK;
}
bool carry_ok(const sf_dual_w& w, const sf_pmodel_w& pm, int B, int H, int W) {
  const long P = (long)B * H * W;
  // One latent only.  (Measured with the blend mode / acc_in also in the large-tile AFFINE epilogues: per steady-state step 8 samples
  // 684 -> 669 us, one 200x200 latent 1200 -> 1200, 32 samples 2178 -> 2197 — and the two extra branches cost every AFFINE launch
  // of the batched forward 0.9 % (255.5 -> 257.8 ms): not worth the headline, reverted; profiles/README.md round-3 log.)
  if (!(tune().pipe && B == 1 && P < tune().sp_max_p && tune().sp && pregate(P, w.cand1) && pregate(P, w.cand2))) return false;
  if (!pm.rb0.proj.w) return false;
  // The launch GROUPS the carried form issues must all be ones the small-P kernel takes (its AFFINE epilogue alone has the blend
  // mode and acc_in; run() refuses them elsewhere and the rollout would fail instead of falling back — ADVICE r3).  Built here with
  // the pointers that decide sp_takes set to a non-null placeholder: sp_takes never dereferences them.
  float* const ph = reinterpret_cast<float*>(uintptr_t(64));
  auto prob = [&](const sf_conv_w& c, bool two_inputs) { return problem(c, ph, two_inputs ? ph : nullptr, ph, B, H, W); };
  ConvProblem g[3];
  // infer_state launch 1: rb0.conv1 + rb0.proj + gates2 (with the pre-gated state as second output)
  g[0] = prob(pm.rb0.conv1, false); g[1] = prob(pm.rb0.proj, false); g[2] = prob(w.gates2, false); g[2].out2 = ph; g[2].e1 = ph;
  if (!sp_takes(g, 3, EPI_AFFINE)) return false;
  // launch 2: rb0.conv2 (+ residual, channel sums) + cand2 (blend mode) [+ the state half of gates1]
  g[0] = prob(pm.rb0.conv2, false); g[0].add = ph; g[0].chansum = ph;
  g[1] = prob(w.cand2, true); g[1].e0 = ph; g[1].e1 = ph; g[1].mode = 4;
  int n2 = 2;
  if (w.gates1_x.w && w.gates1_s.w && tune().pipe >= 2) { g[2] = prob(w.gates1_s, false); n2 = 3; }
  if (!sp_takes(g, n2, EPI_AFFINE)) return false;
  // the next cell: gates1 (x half with acc_in when the state half is carried), then cand1 (blend mode) + conv_decoder_2
  g[0] = n2 == 3 ? prob(w.gates1_x, false) : prob(w.gates1, true);
  g[0].out2 = ph; g[0].e1 = ph; if (n2 == 3) g[0].acc_in = ph;
  if (!sp_takes(g, 1, EPI_AFFINE)) return false;
  g[0] = prob(w.cand1, true); g[0].e0 = ph; g[0].e1 = ph; g[0].mode = 4;
  g[1] = prob(w.dec2, false);
  return sp_takes(g, 2, EPI_AFFINE);
}
// ... and whether conv_decoder_2 / the r2 half of the 7x7 can be carried too (the launch groups this adds must be ones the small-P kernel takes)
bool fork7_ok(const sf_dual_w& w, const sf_pmodel_w& pm, int B, int H, int W) {
  if (!tune().fork7 || g_seg || !w.tg7_h.w || !w.tg7_r.w || !w.gates1_x.w || !w.gates1_s.w || tune().pipe < 2 || !g_split || !fork_ready()) return false;
  float* const ph = reinterpret_cast<float*>(uintptr_t(64));
  ConvProblem g[2];
  g[0] = problem(pm.rb1.conv1, ph, nullptr, ph, B, H, W); g[0].se_sum = ph; g[0].se_fc0 = ph; g[0].se_fc2 = ph; g[0].se_cr = 2 * w.C / 8; g[0].se_nt = 1;
  g[1] = problem(w.dec2, ph, nullptr, ph, B, H, W);
  if (!sp_takes(g, 2, EPI_AFFINE)) return false;
  g[0] = problem(w.tg7_r, ph, nullptr, ph, B, H, W);
  if (!sp_takes(g, 1, EPI_AFFINE)) return false;
  g[0] = problem(w.tg7_h, ph, nullptr, ph, B, H, W); g[0].acc_in = ph; g[0].mode = 1;
  g[1] = problem(w.tgproj, ph, ph, ph, B, H, W);
  return sp_takes(g, 2, EPI_LNG);
}
// the two side problems of cell `w` on state `s` (= the output of the cell before it)
void side_problems(const sf_dual_w& w, const float* s, const Carry& c, int B, int H, int W, Side& sd) {
  const int C = w.C;
  ConvProblem& g2 = sd.p[0];
  g2 = problem(w.gates2, s, nullptr, c.g2, B, H, W);
  g2.out2 = c.rs2; g2.out2_cs = C; g2.e1 = s; g2.e1_cs = C; g2.gate_from = C;
  ConvProblem& c2 = sd.p[1];
  c2 = problem(w.cand2, s, c.rs2, c.h2, B, H, W);
  c2.e0 = c.g2; c2.e0_cs = 2 * C; c2.e1 = s; c2.e1_cs = C;
  c2.mode = 4;               // blend inside an AFFINE launch
  sd.n = 2;
  if (c.g1s && w.gates1_x.w && w.gates1_s.w && tune().pipe >= 2) {      // gates1 = sigmoid(W_x x + [W_s s] + b): the bracket now, raw
    sd.p[2] = problem(w.gates1_s, s, nullptr, c.g1s, B, H, W);
    sd.n = 3;
  }
  sd.has_r2 = c.r2 != nullptr && c.acc7 != nullptr;
  if (sd.has_r2) {
    sd.dec2 = problem(w.dec2, c.h2, nullptr, c.r2, B, H, W);        // rnn_state2 = conv_decoder_2(h2)
    sd.s7 = problem(w.tg7_r, c.r2, nullptr, c.acc7, B, H, W);       // raw sums of the r2 half of trusting_gate.0.layers.0
  }
}

// B images (samples) are processed as one pixel space; coef_stride = floats between the
// coefficient records of consecutive images (0: one record shared by all).  `carry`: gates2 / cand2 of this cell were
// computed beside the previous infer_state (see Carry).
int dual_cell(const sf_dual_w& w, const float* x, const float* s, float* out, int derivative, const float* base,
              const float* coef, int coef_stride, float* out2, int acc2, int B, int H, int W, Arena& A, hipStream_t st,
              const Carry* carry = nullptr) {
  const int C = w.C, P = B * H * W;
  CellBufs b;
  if (!b.take(A, C, P)) return SF_ERR_WORKSPACE;
  const bool pre = pregate(P, w.cand1) && pregate(P, w.cand2);
  ConvProblem ps[2];
  if (carry) {
    if (!pre) return SF_ERR_INVALID;
    ConvProblem unused;
    cell_gate_problems(w, x, s, b, pre, B, H, W, ps[0], unused);
    if (carry->g1s) {                                               // x half only; the state half is carried
      ConvProblem gx = problem(w.gates1_x, x, nullptr, b.g1, B, H, W);
      gx.out2 = ps[0].out2; gx.out2_cs = ps[0].out2_cs; gx.e1 = ps[0].e1; gx.e1_cs = ps[0].e1_cs; gx.gate_from = ps[0].gate_from;
      gx.acc_in = carry->g1s; gx.acc_cs = 2 * C;
      ps[0] = gx;
    }
    SF_TRY(run(ps, 1, EPI_AFFINE, st));                             // gates of cell 1
    cell_cand_problems(w, x, s, b, pre, B, H, W, ps[0], unused);
    ps[0].mode = 4;                                                 // candidate of cell 1 + blend ...
    if (carry->r2 && carry->acc7) {                                 // rnn_state2 and the r2 half of the 7x7 are carried too
      SF_TRY(run(ps, 1, EPI_AFFINE, st));
      b.r2 = carry->r2;
      return cell_tail(w, s, out, derivative, base, coef, coef_stride, out2, acc2, b, B, H, W, st, carry->acc7);
    }
    ps[1] = problem(w.dec2, carry->h2, nullptr, b.r2, B, H, W);     // ... beside rnn_state2 = conv_decoder_2(h2)
    SF_TRY(run(ps, 2, EPI_AFFINE, st));
    return cell_tail(w, s, out, derivative, base, coef, coef_stride, out2, acc2, b, B, H, W, st);
  }
  cell_gate_problems(w, x, s, b, pre, B, H, W, ps[0], ps[1]);     // gates of both cells in one launch
  SF_TRY(run(ps, 2, EPI_AFFINE, st));
  cell_cand_problems(w, x, s, b, pre, B, H, W, ps[0], ps[1]);     // candidates + blend
  SF_TRY(run(ps, 2, EPI_BLEND, st));
  SF_TRY(run1(problem(w.dec2, b.h2, nullptr, b.r2, B, H, W), EPI_AFFINE, st));   // rnn_state2 = conv_decoder_2(h2)
  return cell_tail(w, s, out, derivative, base, coef, coef_stride, out2, acc2, b, B, H, W, st);
}

constexpr int SE_SLABS = 64;
size_t infer_ws_floats(int C, int P) {
  int nt = (P + 15) / 16;
  if (nt < SE_SLABS * 64) nt = SE_SLABS * 64;   // room for [B<=64][SE_SLABS] slab sums as well
  return al((size_t)P * C) + 4 * al((size_t)P * 2 * C) + 2 * al((size_t)nt * 2 * C) + 3 * al((size_t)64 * 2 * C);
}

// temporal_ode_bayes.py:463-477 on B images.  SE channel means: B == 1 uses the per-16-pixel sums
// the producing conv's epilogue writes; B > 1 (16-pixel tiles straddle images) sums per-image
// slabs with one extra small launch.  Both are fixed-order.
int infer_state(const sf_pmodel_w& w, const float* s, const float* eps, float* p_out, float* q_out, int B, int H, int W,
                Arena& A, hipStream_t st, const unsigned long long* philox = nullptr, int draw = 0, const Side* side = nullptr) {
  if (!eps && !philox) return SF_ERR_INVALID;
  const int C = w.C, C2 = 2 * C, HW = H * W, P = B * HW;
  if (B < 1 || B > 64) return SF_ERR_UNSUPPORTED;
  // SE channel means.  One sample: the producing conv's epilogue writes per-tile channel sums ("rows"); a small latent sums
  // them in the consuming layer's prologue (small-P kernel) or in the gate kernel; a large one (thousands of rows) first
  // reduces the rows to SE_SLABS slab sums (two short launches instead of a pass over the whole tensor: 22 -> ~10 us per
  // gate at 200x200).  Several samples (tiles straddle images): per-image slab sums of the tensor.  All fixed-order.
  const bool tiles = (B == 1);
  const bool two_level = tiles && P >= 8192;
  // rows of per-tile channel sums each SE producer writes (its kernel's pixel tile)
  ConvProblem probe1[3] = {problem(w.rb0.conv2, s, nullptr, nullptr, B, H, W), ConvProblem(), ConvProblem()}, probe2 = problem(w.rb1.conv2, s, nullptr, nullptr, B, H, W);
  if (side) { probe1[1] = side->p[1]; probe1[2] = side->p[2]; }      // they ride in rb0.conv2's launch
  const int tpx1 = chansum_tile_px(probe1, side ? side->n : 1, EPI_AFFINE), tpx2 = chansum_tile_px(&probe2, 1, EPI_AFFINE);
  const int nt1 = tiles ? (P + tpx1 - 1) / tpx1 : SE_SLABS, nt2 = tiles ? (P + tpx2 - 1) / tpx2 : SE_SLABS;
  float* a = A.take((size_t)P * C);
  float* pr = A.take((size_t)P * C2);
  float* y1 = A.take((size_t)P * C2);
  float* b = A.take((size_t)P * C2);
  float* y2 = A.take((size_t)P * C2);
  float* cs1 = A.take(tiles ? (size_t)nt1 * C2 : (size_t)B * SE_SLABS * C2);
  float* cs2 = A.take(tiles ? (size_t)nt2 * C2 : (size_t)B * SE_SLABS * C2);
  float* sc1 = A.take((size_t)B * C2);
  float* sc2 = A.take((size_t)B * C2);
  float* slabs = A.take((size_t)SE_SLABS * C2);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  if (!w.rb0.proj.w || w.rb1.proj.w) return SF_ERR_INVALID;
  auto gate = [&](const float* y, const float* rows, int nrows, const float* fc0, const float* fc2, float* scale) -> int {
    SF_TRY(seg_flush());
    if (!tiles) {
      SF_HIP(launch_chan_partial(y, const_cast<float*>(rows), B, HW, C2, SE_SLABS, st));
      SF_HIP(launch_se_fc(rows, SE_SLABS, C2, C2 / 8, HW, fc0, fc2, scale, B, st));
    } else if (two_level) {
      SF_HIP(launch_chan_partial(rows, slabs, 1, nrows, C2, SE_SLABS, st));      // the [nrows][C2] sums as a one-image "tensor"
      SF_HIP(launch_se_fc(slabs, SE_SLABS, C2, C2 / 8, HW, fc0, fc2, scale, 1, st));
    } else {
      SF_HIP(launch_se_fc(rows, nrows, C2, C2 / 8, HW, fc0, fc2, scale, B, st));
    }
    return SF_OK;
  };
  ConvProblem ps[3];
  ps[0] = problem(w.rb0.conv1, s, nullptr, a, B, H, W);
  ps[1] = problem(w.rb0.proj, s, nullptr, pr, B, H, W);
  if (side) ps[2] = side->p[0];
  SF_TRY(run(ps, side ? 3 : 2, EPI_AFFINE, st));
  ConvProblem c2 = problem(w.rb0.conv2, a, nullptr, y1, B, H, W);
  c2.add = pr; c2.chansum = tiles ? cs1 : nullptr;
  ps[0] = c2;
  if (side) { ps[1] = side->p[1]; ps[2] = side->p[2]; }
  SF_TRY(run(ps, side ? side->n : 1, EPI_AFFINE, st));
  // SE gates.  One sample on the small-P kernel: the consuming layer computes the gate in its prologue from the
  // producer's per-tile channel sums (two launches fewer per infer_state); otherwise the gate kernel
  ConvProblem c3 = problem(w.rb1.conv1, y1, nullptr, b, B, H, W);
  c3.se_sum = cs1; c3.se_fc0 = w.se0_fc0; c3.se_fc2 = w.se0_fc2; c3.se_out = sc1; c3.se_nt = nt1; c3.se_cr = C2 / 8;
  c3.se_inv_hw = 1.f / (float)HW;
  const bool fuse_se = tiles && tune().sp_fuse_se && sp_takes(&c3, 1, EPI_AFFINE);
  if (!fuse_se) {
    c3.se_sum = nullptr;
    SF_TRY(gate(y1, cs1, nt1, w.se0_fc0, w.se0_fc2, sc1));
    c3.in_scale = sc1;
  }
  if (side && side->has_r2 && fuse_se) {      // rnn_state2 = conv_decoder_2(h2) of the next cell rides here; the r2 half of its 7x7 forks off behind it
    ps[0] = c3; ps[1] = side->dec2;
    SF_TRY(run(ps, 2, EPI_AFFINE, st));
    SF_TRY(fork_run(side->s7, EPI_AFFINE, tune().fork7_wgs, st));
  } else if (side && side->has_r2) {
    return SF_ERR_UNSUPPORTED;                 // (fork7_ok promised the fused SE gate)
  } else {
    SF_TRY(run1(c3, EPI_AFFINE, st));
  }
  ConvProblem c4 = problem(w.rb1.conv2, b, nullptr, y2, B, H, W);
  c4.add = y1; c4.add_scale = sc1; c4.chansum = tiles ? cs2 : nullptr;
  SF_TRY(run1(c4, EPI_AFFINE, st));
  ConvProblem c5 = problem(w.last, y2, nullptr, p_out, B, H, W);
  c5.e0 = eps; c5.out2 = q_out; c5.philox = philox; c5.draw = draw;
  if (fuse_se) {
    c5.se_sum = cs2; c5.se_fc0 = w.se1_fc0; c5.se_fc2 = w.se1_fc2; c5.se_out = nullptr; c5.se_nt = nt2; c5.se_cr = C2 / 8;
    c5.se_inv_hw = 1.f / (float)HW;
  }
  if (!fuse_se || !sp_takes(&c5, 1, EPI_SAMPLE)) {
    c5.se_sum = nullptr;
    SF_TRY(gate(y2, cs2, nt2, w.se1_fc0, w.se1_fc2, sc2));
    c5.in_scale = sc2;
  }
  return run1(c5, EPI_SAMPLE, st);
}

// temporal_ode_bayes.py:436-461 (+ build-defined RK4).  `zeros`: [P][C] zero tensor (IMPUTE=False).
// Scratch: k (P*C), pk (P*C), acc (P*C) + cell/infer workspaces.
int ode_step(const sf_dual_w& gc, const sf_pmodel_w& pm, int solver, int impute, const float* s_in,
             const float* p_in, const float* coef, int coef_stride, const float* eps, float* s_out, float* p_out,
             const float* zeros, int skip_dead_infer, int B, int H, int W, Arena& A0, hipStream_t st) {
  const int C = gc.C;
  const size_t PC = (size_t)B * H * W * C;
  const float* x = impute ? p_in : zeros;
  auto cell = [&](const float* xx, const float* ss, float* out, const float* base, const float* cf, float* out2,
                  int acc2) {
    Arena A = A0;
    return dual_cell(gc, xx, ss, out, 1, base, cf, coef_stride, out2, acc2, B, H, W, A, st);
  };
  auto infer = [&](const float* ss, int draw, float* po) {
    Arena A = A0;
    return infer_state(pm, ss, eps + (size_t)draw * PC, po, nullptr, B, H, W, A, st);
  };
  if (solver == SF_SOLVER_EULER) {
    SF_TRY(cell(x, s_in, s_out, s_in, coef + 0, nullptr, 0));
    if (!(skip_dead_infer && !impute)) SF_TRY(infer(s_out, 0, p_out));
    return SF_OK;
  }
  float* k = A0.take(PC);
  float* pk = A0.take(PC);
  float* acc = A0.take(PC);
  if (!A0.ok()) return SF_ERR_WORKSPACE;
  if (solver == SF_SOLVER_MIDPOINT) {
    SF_TRY(cell(x, s_in, k, s_in, coef + 1, nullptr, 0));              // k = s + dt/2 f(p, s)
    SF_TRY(infer(k, 0, pk));            // :452 — evaluated even when IMPUTE is off (only :443's input is zeroed)
    SF_TRY(cell(pk, k, s_out, s_in, coef + 0, nullptr, 0));            // s' = s + dt f(pk, k)
    if (!(skip_dead_infer && !impute)) SF_TRY(infer(s_out, 1, p_out));
    return SF_OK;
  }
  if (solver == SF_SOLVER_RK4) {
    // coef = {dt, dt/2, dt/6, dt/3}.  acc = s + dt/6 k1 + dt/3 k2 + dt/3 k3 ; s' = acc + dt/6 k4
    float* s2 = k;
    float* s3 = A0.take(PC);
    if (!A0.ok()) return SF_ERR_WORKSPACE;
    // stage coefficient pairs {out coef, out2 coef} follow the four scalars in the coef record:
    // {dt/2, dt/6}, {dt/2, dt/3}, {dt, dt/3}, {dt/6, 0}   (see SF_COEF_STRIDE in sfnative.h)
    const float* rk = coef + 4;
    SF_TRY(cell(x, s_in, s2, s_in, rk + 0, acc, 0));
    SF_TRY(infer(s2, 0, pk));           // stage inputs always come from infer_state, as in midpoint
    SF_TRY(cell(pk, s2, s3, s_in, rk + 2, acc, 1));
    SF_TRY(infer(s3, 1, pk));
    SF_TRY(cell(pk, s3, s2, s_in, rk + 4, acc, 1));   // s4 reuses s2
    SF_TRY(infer(s2, 2, pk));
    SF_TRY(cell(pk, s2, s_out, acc, rk + 6, nullptr, 0));
    if (!(skip_dead_infer && !impute)) SF_TRY(infer(s_out, 3, p_out));
    return SF_OK;
  }
  return SF_ERR_INVALID;
}

size_t ode_step_ws_floats(int C, int P) {
  size_t cellw = dual_ws_floats(C, P), inf = infer_ws_floats(C, P);
  return 4 * al((size_t)P * C) + (cellw > inf ? cellw : inf);
}

// ResBlock (res_models.py:74-79) on n images; t: [P][cin] scratch, pr: [P][cout] scratch (if proj)
// `pooled` (optional): MaxPool2d(2) of the block's output, [n][H/2][W/2][C]; written by the last convolution's own epilogue where the
// Winograd kernel runs it (`out` is then not written at all), by a pooling launch after it otherwise
int res_block(const sf_res_w& w, const float* x, float* out, float* t, float* pr, int n, int Hin, int Win, int in_up,
              hipStream_t st, float* pooled = nullptr) {
  ConvProblem c2 = problem(w.conv2, t, nullptr, out, n, Hin << in_up, Win << in_up);
  c2.add = w.proj.w ? pr : x;
  if (!w.proj.w && in_up) {
    // identity skip of an input that is upsampled on read: the Winograd epilogue reads the half-size tensor itself (one source pixel
    // per tile); anywhere else the caller has to materialise the upsampled input (SF_ERR_UNSUPPORTED before anything is launched)
    static const bool fuse = [] { const char* v = std::getenv("SF_UPSAMPLE_FUSED"); return v ? std::atoi(v) != 0 : true; }();
    c2.add_up = 1;
    if (!(fuse && tune().wino && (double)c2.n_img * c2.Hout * c2.Wout >= tune().wino_min_p && wino_takes(c2, EPI_AFFINE) && !(tune().b3 && c2.w3)))
      return SF_ERR_UNSUPPORTED;
  }
  ConvProblem ps[2];
  ps[0] = problem(w.conv1, x, nullptr, t, n, Hin, Win, in_up);
  int np = 1;
  if (w.proj.w) { ps[1] = problem(w.proj, x, nullptr, pr, n, Hin, Win, in_up); np = 2; }
  SF_TRY(run(ps, np, EPI_AFFINE, st));
  if (pooled) {
    static const bool fuse = [] { const char* v = std::getenv("SF_POOL_FUSED"); return v ? std::atoi(v) != 0 : true; }();
    ConvProblem cp = c2;
    cp.pool2 = 1; cp.out = pooled;
    if (fuse && tune().wino && (double)cp.n_img * cp.Hout * cp.Wout >= tune().wino_min_p && wino_takes(cp, EPI_AFFINE) && !(tune().b3 && cp.w3))
      return run1(cp, EPI_AFFINE, st);
    SF_TRY(run1(c2, EPI_AFFINE, st));
    SF_HIP(launch_maxpool2(out, pooled, n, c2.Hout, c2.Wout, c2.cout, 0, st));
    return SF_OK;
  }
  return run1(c2, EPI_AFFINE, st);
}

}  // namespace

// =================================================================================================
extern "C" {

int sf_version(void) { return 110; }

// ABI guard (include/sfnative.h): the sizes this translation unit was compiled with
int sf_abi_version(void) { return SF_ABI_VERSION; }
size_t sf_abi_sizeof(int which) {
  switch (which) {
    case SF_STRUCT_CONV_W: return sizeof(sf_conv_w);
    case SF_STRUCT_GRU_W: return sizeof(sf_gru_w);
    case SF_STRUCT_DUAL_W: return sizeof(sf_dual_w);
    case SF_STRUCT_RES_W: return sizeof(sf_res_w);
    case SF_STRUCT_PMODEL_W: return sizeof(sf_pmodel_w);
    case SF_STRUCT_ENCODER_W: return sizeof(sf_encoder_w);
    case SF_STRUCT_DECODER_W: return sizeof(sf_decoder_w);
    case SF_STRUCT_CONVNEXT_W: return sizeof(sf_convnext_w);
    case SF_STRUCT_DEEPLAB_W: return sizeof(sf_deeplab_w);
    case SF_STRUCT_BOTTLENECK_W: return sizeof(sf_bottleneck_w);
    case SF_STRUCT_BOTTLE_W: return sizeof(sf_bottle_w);
  }
  return 0;
}
int sf_abi_check(int abi_version, const size_t* struct_sizes, int n) {
  if (abi_version != SF_ABI_VERSION || !struct_sizes || n != SF_STRUCT_COUNT) return SF_ERR_INVALID;
  for (int i = 0; i < n; ++i)
    if (struct_sizes[i] != sf_abi_sizeof(i)) return SF_ERR_INVALID;
  return SF_OK;
}

const char* sf_status_string(int s) {
  switch (s) {
    case SF_OK: return "ok";
    case SF_ERR_INVALID: return "invalid argument";
    case SF_ERR_WORKSPACE: return "workspace too small";
    case SF_ERR_LAUNCH: return "HIP launch/runtime error";
    case SF_ERR_UNSUPPORTED: return "unsupported shape";
  }
  return "unknown";
}

int sf_nchw_to_nhwc(const float* src, float* dst, int n, int C, int HW, void* stream) {
  if (!src || !dst) return SF_ERR_INVALID;
  SF_HIP(launch_transpose(src, dst, n, C, HW, (hipStream_t)stream));
  return SF_OK;
}
int sf_nhwc_to_nchw(const float* src, float* dst, int n, int C, int HW, void* stream) {
  if (!src || !dst) return SF_ERR_INVALID;
  SF_HIP(launch_transpose(src, dst, n, HW, C, (hipStream_t)stream));
  return SF_OK;
}

/* the same with the n images `src_stride` / `dst_stride` floats apart (>= C*HW): frames picked out of / written into a
 * larger [B][T][C][H][W] tensor without an intermediate copy */
int sf_nchw_to_nhwc_strided(const float* src, size_t src_stride, float* dst, size_t dst_stride, int n, int C, int HW, void* stream) {
  if (!src || !dst || src_stride < (size_t)C * HW || dst_stride < (size_t)C * HW) return SF_ERR_INVALID;
  SF_HIP(launch_transpose_strided(src, dst, n, C, HW, src_stride, dst_stride, (hipStream_t)stream));
  return SF_OK;
}
int sf_nhwc_to_nchw_strided(const float* src, size_t src_stride, float* dst, size_t dst_stride, int n, int C, int HW, void* stream) {
  if (!src || !dst || src_stride < (size_t)C * HW || dst_stride < (size_t)C * HW) return SF_ERR_INVALID;
  SF_HIP(launch_transpose_strided(src, dst, n, HW, C, src_stride, dst_stride, (hipStream_t)stream));
  return SF_OK;
}

int sf_conv2d_fwd(const sf_conv_w* w, const float* in0, const float* in1, const float* add, float* out, int n_img,
                  int Hin, int Win, int in_up, void* stream) {
  if (!w || !valid_w(*w) || !in0 || !out || (w->c1 > 0 && !in1)) return SF_ERR_INVALID;
  ConvProblem p = problem(*w, in0, in1, out, n_img, Hin, Win, in_up);
  p.add = add;
  return run1(p, EPI_AFFINE, (hipStream_t)stream);
}

// The same with channel-sliced operands (a layer reading / writing a channel range of a wider NHWC
// tensor, e.g. the stacked heads of the BEV decoder), the residual added before the activation
// (ResNet BasicBlock) and optional split-K scratch for mid-sized pixel counts.
int sf_conv2d_ex_fwd(const sf_conv_w* w, const float* in0, int in0_cs, const float* in1, int in1_cs, const float* add, int add_cs,
                     int act_after_add, float* out, int out_cs, int out_co, int n_img, int Hin, int Win, int in_up, float* ws,
                     size_t ws_bytes, void* stream) {
  if (!w || !valid_w(*w) || !in0 || !out || (w->c1 > 0 && !in1)) return SF_ERR_INVALID;
  if (in0_cs < w->c0 || (w->c1 > 0 && in1_cs < w->c1) || out_cs < out_co + w->cout || (in0_cs % 4) || (out_cs % 4) || (out_co % 4) ||
      (add && (add_cs < w->cout || (add_cs % 4))))
    return SF_ERR_INVALID;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, (hipStream_t)stream);
  ConvProblem p = problem(*w, in0, in1, out, n_img, Hin, Win, in_up);
  p.in0_cs = in0_cs;
  if (w->c1 > 0) p.in1_cs = in1_cs;
  p.add = add;
  if (add) p.add_cs = add_cs;
  p.out_cs = out_cs;
  p.out_co = out_co;
  if (act_after_add) p.mode |= 2;
  return run1(p, EPI_AFFINE, (hipStream_t)stream);
}
size_t sf_conv2d_ex_ws_bytes(void) { return SPLIT_WS_FLOATS * sizeof(float); }

/* UpsamplingAdd tail — convolutions.py:208,215: out = bilinear_x2(in) + skip (skip may be NULL) */
int sf_upsample_bilinear2_add_fwd(const float* in, const float* skip, float* out, int n, int Hin, int Win, int C, void* stream) {
  if (!in || !out || n < 1 || Hin < 1 || Win < 1 || C < 4 || (C % 4)) return SF_ERR_INVALID;
  SF_HIP(launch_upsample_bilinear2_add(in, skip, out, n, Hin, Win, C, (hipStream_t)stream));
  return SF_OK;
}

/* sparse (gather) convolution: out[j] = act(scale * sum_t W_t . feats[nbr[j][t]] + bias) (+ add), see sfnative.h */
int sf_sparse_conv_fwd(const sf_conv_w* w, const float* feats, int feats_cs, int n_in, const int32_t* nbr, int n_out,
                       const float* add, int act_after_add, float* out, float* ws, size_t ws_bytes, void* stream) {
  if (!w || !valid_w(*w) || !feats || !nbr || !out || n_out < 0 || n_in < 1 || w->kw != 1 || w->c1 != 0 || feats_cs < w->c0 ||
      (feats_cs % 4))
    return SF_ERR_INVALID;
  if (n_out == 0) return SF_OK;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, (hipStream_t)stream);
  ConvProblem p = problem(*w, feats, nullptr, out, 1, 1, n_in, 0);   // the input "image" is the n_in feature rows
  p.in0_cs = feats_cs;
  p.Hout = 1; p.Wout = n_out;
  p.gather = nbr;
  p.add = add;
  if (act_after_add) p.mode |= 2;
  return run1(p, EPI_AFFINE, (hipStream_t)stream);
}

/* the same with a tap mask: tile_mask64[i] bit t = some row of 64 i .. 64 i + 63 has an input row under tap t (see sfnative.h) */
int sf_sparse_conv_masked_fwd(const sf_conv_w* w, const float* feats, int feats_cs, int n_in, const int32_t* nbr, const uint32_t* tile_mask64, int n_out,
                              const float* add, int act_after_add, float* out, float* ws, size_t ws_bytes, void* stream) {
  if (!w || !valid_w(*w) || !feats || !nbr || !out || n_out < 0 || n_in < 1 || w->kw != 1 || w->c1 != 0 || feats_cs < w->c0 ||
      (feats_cs % 4))
    return SF_ERR_INVALID;
  if (n_out == 0) return SF_OK;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, (hipStream_t)stream);
  ConvProblem p = problem(*w, feats, nullptr, out, 1, 1, n_in, 0);
  p.in0_cs = feats_cs;
  p.Hout = 1; p.Wout = n_out;
  p.gather = nbr;
  p.tap_mask = tile_mask64; p.tap_mask_n = tile_mask64 ? (n_out + 63) / 64 : 0;
  p.add = add;
  if (act_after_add) p.mode |= 2;
  return run1(p, EPI_AFFINE, (hipStream_t)stream);
}

/* per-image channel means of an NHWC tensor (fixed-order two-level sum: reproducible) */
size_t sf_channel_mean_ws_bytes(int C, int n) { return al((size_t)n * 64 * C) * sizeof(float); }
int sf_channel_mean_fwd(const float* x, float* out, int n, int HW, int C, float* ws, size_t ws_bytes, void* stream) {
  if (!x || !out || n < 1 || HW < 1 || C < 4 || (C % 4) || C > 1024) return SF_ERR_INVALID;
  Arena A(ws, ws_bytes);
  float* part = A.take((size_t)n * 64 * C);
  if (!A.ok() || !part) return SF_ERR_WORKSPACE;
  SF_HIP(launch_chan_partial(x, part, n, HW, C, 64, (hipStream_t)stream));
  SF_HIP(launch_mean_from_partials(part, out, n, 64, C, HW, (hipStream_t)stream));
  return SF_OK;
}
/* elementwise LogSigmoid (DistributionModule(method='BERNOULLI'), streamingflow/models/distributions.py:33, :47) */
int sf_logsigmoid_fwd(const float* x, float* out, size_t n, void* stream) {
  if (!x || !out) return SF_ERR_INVALID;
  if (n == 0) return SF_OK;
  SF_HIP(launch_logsigmoid(x, out, n, (hipStream_t)stream));
  return SF_OK;
}
/* out[img][pixel][out_co .. out_co+k) = vec[img][0..k) for every pixel (k, out_cs, out_co multiples of 4) */
int sf_broadcast_channels_fwd(const float* vec, float* out, int n, int HW, int k, int out_cs, int out_co, void* stream) {
  if (!vec || !out || n < 1 || HW < 1 || k < 4 || (k % 4) || (out_cs % 4) || (out_co % 4) || out_co + k > out_cs) return SF_ERR_INVALID;
  SF_HIP(launch_broadcast_channels(vec, out, n, HW, k, out_cs, out_co, (hipStream_t)stream));
  return SF_OK;
}

// benchmarking aid: the same fused conv enqueued `reps` times back to back from C++ (no per-launch
// host-language overhead); ws may be NULL (then no cross-workgroup split-K scratch is available)
int sf_conv2d_repeat(const sf_conv_w* w, const float* in0, const float* in1, const float* add, float* out, int n_img,
                     int Hin, int Win, int in_up, int reps, float* ws, size_t ws_bytes, void* stream) {
  if (!w || !valid_w(*w) || !in0 || !out || (w->c1 > 0 && !in1)) return SF_ERR_INVALID;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, (hipStream_t)stream);
  ConvProblem p = problem(*w, in0, in1, out, n_img, Hin, Win, in_up);
  p.add = add;
  for (int i = 0; i < reps; ++i) SF_TRY(run1(p, EPI_AFFINE, (hipStream_t)stream));
  return SF_OK;
}

size_t sf_gru_cell_ws_bytes(int C, int n_img, int H, int W) {
  return (al((size_t)n_img * H * W * 2 * C) + al((size_t)n_img * H * W * C)) * sizeof(float);
}
int sf_gru_cell_fwd(const sf_gru_w* w, const float* x, const float* s, float* out, int n_img, int H, int W, float* ws,
                    size_t ws_bytes, void* stream) {
  if (!w || !x || !s || !out || !valid_w(w->gates) || !valid_w(w->cand)) return SF_ERR_INVALID;
  Arena A(ws, ws_bytes);
  float* g = A.take((size_t)n_img * H * W * 2 * w->cand.cout);
  float* rs = A.take((size_t)n_img * H * W * w->cand.cout);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  return gru_cell(*w, x, s, out, g, rs, n_img, H, W, (hipStream_t)stream, 0);
}
// SpatialGRUODECell.forward — temporal_ode_bayes.py:35-61: dh = u * (h~ - s)
int sf_gru_ode_cell_fwd(const sf_gru_w* w, const float* x, const float* s, float* out, int n_img, int H, int W, float* ws,
                        size_t ws_bytes, void* stream) {
  if (!w || !x || !s || !out || !valid_w(w->gates) || !valid_w(w->cand)) return SF_ERR_INVALID;
  Arena A(ws, ws_bytes);
  float* g = A.take((size_t)n_img * H * W * 2 * w->cand.cout);
  float* rs = A.take((size_t)n_img * H * W * w->cand.cout);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  return gru_cell(*w, x, s, out, g, rs, n_img, H, W, (hipStream_t)stream, 1);
}

size_t sf_spatial_gru_ws_bytes(int C, int n_img, int H, int W) {
  return (al((size_t)n_img * H * W * 2 * C) + 3 * al((size_t)n_img * H * W * C)) * sizeof(float);
}
int sf_spatial_gru_fwd(const sf_gru_w* w, const float* x, const float* state0, float* out, int T, int n_img, int H,
                       int W, float* ws, size_t ws_bytes, void* stream) {
  if (!w || !x || !state0 || !out || !valid_w(w->gates) || !valid_w(w->cand)) return SF_ERR_INVALID;
  const bool has_dec = w->decoder.w != nullptr;
  if (has_dec && !valid_w(w->decoder)) return SF_ERR_INVALID;
  const int C = w->cand.cout, Cx = w->gates.c0;
  const size_t P = (size_t)n_img * H * W;
  Arena A(ws, ws_bytes);
  float* g = A.take(P * 2 * C);
  float* sa = A.take(P * C);
  float* sb = A.take(P * C);
  float* rs = A.take(P * C);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const float* cur = state0;
  for (int t = 0; t < T; ++t) {   // temporal.py:35-39
    // without a decoder (BEVerse SpatialGRU, basic_modules.py:225-284) the states ARE the output
    float* nxt = has_dec ? ((t & 1) ? sb : sa) : out + t * P * C;
    SF_TRY(gru_cell(*w, x + t * P * Cx, cur, nxt, g, rs, n_img, H, W, st));
    if (has_dec)
      SF_TRY(run1(problem(w->decoder, nxt, nullptr, out + t * P * w->decoder.cout, n_img, H, W), EPI_AFFINE, st));
    cur = nxt;
  }
  return SF_OK;
}

size_t sf_dual_cell_ws_bytes(int C, int n_img, int H, int W) {
  return (dual_ws_floats(C, n_img * H * W) + SPLIT_WS_FLOATS) * sizeof(float);
}
int sf_dual_cell_fwd(const sf_dual_w* w, const float* x, const float* s, float* out, int derivative, const float* base,
                     const float* coef, float* out2, int acc2, int n_img, int H, int W, float* ws, size_t ws_bytes,
                     void* stream) {
  if (!w || !x || !s || !out || w->C <= 0 || (w->C % 8) || n_img < 1) return SF_ERR_INVALID;
  if (w->C > 128) return SF_ERR_UNSUPPORTED;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, (hipStream_t)stream);
  return dual_cell(*w, x, s, out, derivative, base, coef, 0, out2, acc2, n_img, H, W, A, (hipStream_t)stream);
}

// trusting gate + mix of two given branch states (the tail of both dual cells; the recurrent Dual_GRU calls it per step)
int sf_trust_mix_fwd(const sf_dual_w* w, const float* r1, const float* r2, const float* s, float* out, int derivative,
                     const float* base, const float* coef, int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream) {
  if (!w || !r1 || !r2 || !out || w->C <= 0 || (w->C % 8) || n_img < 1) return SF_ERR_INVALID;
  if (derivative && (!s || !coef)) return SF_ERR_INVALID;
  if (w->C > 128) return SF_ERR_UNSUPPORTED;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, (hipStream_t)stream);
  CellBufs b;
  if (!b.take(A, w->C, n_img * H * W)) return SF_ERR_WORKSPACE;
  b.h1 = const_cast<float*>(r1);      // read only from here on
  b.r2 = const_cast<float*>(r2);
  return cell_tail(*w, s ? s : r1, out, derivative, base, coef, 0, nullptr, 0, b, n_img, H, W, (hipStream_t)stream);
}

// Bottleblock (convolutions.py:348-380) on cat[x0, x1]: out = layers(x) + (projection(x) | x)
size_t sf_bottleblock_ws_bytes(int cin, int cout, int n_img, int H, int W) {
  const size_t P = (size_t)n_img * H * W;
  return (2 * al(P * (size_t)(cin / 2 > 0 ? cin / 2 : 1)) + al(P * (size_t)cout) + SPLIT_WS_FLOATS) * sizeof(float);
}
int sf_bottleblock_fwd(const sf_bottle_w* w, const float* x0, const float* x1, float* out, int n_img, int H, int W, float* ws,
                       size_t ws_bytes, void* stream) {
  if (!w || !x0 || !out || n_img < 1 || !valid_w(w->c7) || !valid_w(w->c1) || !valid_w(w->c3)) return SF_ERR_INVALID;
  const bool has_proj = w->proj.w != nullptr;
  if (!has_proj && (x1 || w->c7.c1 != 0 || w->c3.cout != w->c7.c0)) return SF_ERR_INVALID;      // the residual is x itself
  if (w->c7.cout_pad > 128 || w->c3.cout_pad > 128) return SF_ERR_UNSUPPORTED;                    // LayerNorm epilogue: all channels of a pixel in one wave
  hipStream_t st = (hipStream_t)stream;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, st);
  const size_t P = (size_t)n_img * H * W;
  float* t1 = A.take(P * w->c7.cout);
  float* t2 = A.take(P * w->c1.cout);
  float* sk = has_proj ? A.take(P * w->proj.cout) : nullptr;
  if (!A.ok()) return SF_ERR_WORKSPACE;
  ConvProblem ps[2];
  ps[0] = problem(w->c7, x0, x1, t1, n_img, H, W); ps[0].mode = 1;        // 7x7 + LN + GELU
  ConvProblem q = problem(w->c1, t1, nullptr, t2, n_img, H, W); q.mode = 1;
  if (has_proj) { ps[1] = problem(w->proj, x0, x1, sk, n_img, H, W); ps[1].mode = 0; }   // 1x1 + GELU
  const bool fused = fuse_following_1x1(ps, has_proj ? 2 : 1, q, w->c1, t2);
  SF_TRY(run(ps, has_proj ? 2 : 1, EPI_LNG, st));
  if (!fused) SF_TRY(run1(q, EPI_LNG, st));
  ConvProblem f = problem(w->c3, t2, nullptr, out, n_img, H, W); f.mode = 1;
  f.add = has_proj ? sk : x0;
  f.add_cs = has_proj ? w->proj.cout : w->c7.c0;
  return run1(f, EPI_LNG, st);
}

size_t sf_infer_state_ws_bytes(int C, int n_img, int H, int W) {
  return (infer_ws_floats(C, n_img * H * W) + SPLIT_WS_FLOATS) * sizeof(float);
}
int sf_infer_state_fwd(const sf_pmodel_w* w, const float* s, const float* eps, float* p_out, float* q_out, int n_img,
                       int H, int W, float* ws, size_t ws_bytes, void* stream) {
  if (!w || !s || !eps || !p_out || w->C <= 0 || (w->C % 8) || n_img < 1) return SF_ERR_INVALID;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, (hipStream_t)stream);
  return infer_state(*w, s, eps, p_out, q_out, n_img, H, W, A, (hipStream_t)stream);
}

size_t sf_ode_step_ws_bytes(int C, int n_img, int H, int W) {
  const int P = n_img * H * W;
  return (ode_step_ws_floats(C, P) + al((size_t)P * C) + SPLIT_WS_FLOATS) * sizeof(float);
}
int sf_ode_step_fwd(const sf_dual_w* gru_c, const sf_pmodel_w* pm, int solver, int impute, const float* state_in,
                    const float* p_in, const float* coef, const float* eps, float* state_out, float* p_out, int n_img,
                    int H, int W, float* ws, size_t ws_bytes, void* stream) {
  if (!gru_c || !pm || !state_in || !p_in || !coef || !eps || !state_out || !p_out || n_img < 1) return SF_ERR_INVALID;
  if (gru_c->C > 128 || (gru_c->C % 8)) return SF_ERR_UNSUPPORTED;
  Arena A(ws, ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  SplitScope sp(A, st);
  const size_t PC = (size_t)n_img * H * W * gru_c->C;
  float* zeros = A.take(PC);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  if (!impute) SF_HIP(zero_fill(zeros, al(PC) * sizeof(float), st));
  return ode_step(*gru_c, *pm, solver, impute, state_in, p_in, coef, 0, eps, state_out, p_out, zeros, 0, n_img, H, W, A, st);
}

// ---- rollout as a list of stages -----------------------------------------------------------------------------------------
// A stage = one dual cell (ODE derivative + integrator update, or a Bayesian jump) followed by an optional infer_state.
// (Measured and rejected, profiles/README.md "two-stream rollout": running gru_cell_2 / conv_decoder_2 of stage j+1 on a
// second stream beside stage j's infer_state — they only need the state — made the 18-op rollout 11 % slower: every
// small-P workgroup owns a whole CU, so the two chains take turns instead of overlapping, and the split launches lose
// the grouping of the two gate / candidate convolutions.)
struct Stage {
  const sf_dual_w* w;
  const float* x; const float* s; float* out;
  int derivative; const float* base; const float* coef; float* out2; int acc2;
  bool infer_after; int draw; float* p_out;
  int op_end;      // index of the op this stage completes (-1: an inner solver stage)
};

size_t rollout_ws_floats(int C, int P) {
  const size_t cellw = dual_ws_floats(C, P), inf = infer_ws_floats(C, P);
  // the persistent flow's tables and counters (5 MB) only where the flow form is switched on at the time of the query: the size query and
  // the call see the same setting (a workspace sized without them makes a flow-mode call fail with SF_ERR_WORKSPACE, not overrun)
  const bool flow = g_flow_mode < 0 ? tune().persist != 0 : g_flow_mode != 0;
  return (cellw > inf ? cellw : inf) + 17 * al((size_t)P * C) + SPLIT_WS_FLOATS + (flow ? FLOW_WS_FLOATS : 0) + 256;
}

int run_stages(const std::vector<Stage>& stages, const sf_pmodel_w& pm, const float* eps, const unsigned long long* philox, int coef_stride,
               const int32_t* sel_nops, int n_targets, float* out_states, int B, int H, int W, Arena& A, hipStream_t st, const unsigned** flow_err) {
  const size_t PC = (size_t)B * H * W * pm.C;
  // buffers of the carried branch 2 (outside the per-stage arenas: written during one stage's infer_state, read by the next cell)
  Carry cb, cnow;
  cb.g2 = A.take(2 * PC); cb.rs2 = A.take(PC); cb.h2 = A.take(PC); cb.g1s = A.take(2 * PC);
  cb.r2 = A.take(PC); cb.acc7 = A.take(PC);
  cnow = cb;
  // one latent: the launch groups of the stages become phases of ONE persistent flow launch (conv_sp.hip: sp_flow_kernel)
  const bool persist = (g_flow_mode < 0 ? tune().persist : g_flow_mode) && B == 1 && (long)B * H * W < tune().sp_max_p && tune().sp && g_split != nullptr;
  unsigned char* table = persist ? reinterpret_cast<unsigned char*>(A.take(FLOW_TABLE_BYTES / 4)) : nullptr;
  unsigned* done = persist ? reinterpret_cast<unsigned*>(A.take(FLOW_DONE_COUNTERS + 64)) : nullptr;
  if (!A.ok()) return SF_ERR_WORKSPACE;
  FlowBuilder seg(table, done, done ? done + FLOW_DONE_COUNTERS : nullptr, st);
  struct SegScope {      // g_seg is set for the duration of this function only
    bool on;
    SegScope(FlowBuilder* b, bool enable) : on(enable) { if (on) g_seg = b; }
    ~SegScope() { if (on) g_seg = nullptr; }
  } seg_scope(&seg, persist);
  if (persist) SF_HIP(zero_fill(done, (FLOW_DONE_COUNTERS + 64) * sizeof(unsigned), st));
  *flow_err = persist ? done + FLOW_DONE_COUNTERS : nullptr;
  struct HalvesScope {      // the forked side launch and the main stream's launches keep to their own halves of the split-K scratch
    bool was;
    HalvesScope(bool on) : was(g_fork_halves) { g_fork_halves = on; }
    ~HalvesScope() { g_fork_halves = was; }
  } halves_scope(!persist && B == 1 && tune().fork7 != 0);
  bool carried = false;
  for (size_t j = 0; j < stages.size(); ++j) {
    const Stage& g = stages[j];
    Arena Ac = A;
    SF_TRY(dual_cell(*g.w, g.x, g.s, g.out, g.derivative, g.base, g.coef, coef_stride, g.out2, g.acc2, B, H, W, Ac, st, carried ? &cnow : nullptr));
    carried = false;
    if (g.op_end >= 0)
      for (int t = 0; t < n_targets; ++t)
        if (sel_nops[t] == g.op_end + 1) {
          if (!(g_seg && g_seg->add_copy(g.out, out_states + (size_t)t * PC, PC) == SF_OK)) {
            SF_TRY(seg_flush());
            SF_HIP(copy_floats(g.out, out_states + (size_t)t * PC, PC, st));
          }
        }
    if (g.infer_after) {
      Arena Ai = A;
      Side sd = {};
      const Stage* nx = j + 1 < stages.size() ? &stages[j + 1] : nullptr;
      const bool pipe = nx && nx->s == g.out && carry_ok(*nx->w, pm, B, H, W);
      if (pipe) {
        Carry cuse = cb;
        if (!fork7_ok(*nx->w, pm, B, H, W)) { cuse.r2 = nullptr; cuse.acc7 = nullptr; }
        side_problems(*nx->w, g.out, cuse, B, H, W, sd);
        cnow = cuse;
        if (sd.n < 3) cnow.g1s = nullptr;
      }
      SF_TRY(infer_state(pm, g.out, eps ? eps + (size_t)g.draw * PC : nullptr, g.p_out, nullptr, B, H, W, Ai, st, philox, g.draw, pipe ? &sd : nullptr));
      carried = pipe;
    }
  }
  SF_TRY(seg_flush());
  SF_TRY(fork_join(st));      // (every forked launch is joined by the cell that follows it; this one only after an error path left one pending)
  return SF_OK;
}

size_t sf_nnfo_rollout_ws_bytes(int C, int n_img, int H, int W) { return rollout_ws_floats(C, n_img * H * W) * sizeof(float); }
static int rollout_core(const sf_dual_w* gru_c, const sf_dual_w* gru_obs, const sf_pmodel_w* pm, int solver, int impute,
                        const int32_t* ops, int n_ops, const float* hx_obs, const float* eps, const unsigned long long* philox,
                        const float* coef, int coef_per_image, const int32_t* sel_nops, int n_targets, float* out_states,
                        float* final_state, int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream) {
  if (!gru_c || !gru_obs || !pm || !ops || !hx_obs || (!eps && !philox) || !sel_nops || !out_states || n_img < 1) return SF_ERR_INVALID;
  if (gru_c->C > 128 || (gru_c->C % 8)) return SF_ERR_UNSUPPORTED;
  if (solver != SF_SOLVER_EULER && solver != SF_SOLVER_MIDPOINT && solver != SF_SOLVER_RK4) return SF_ERR_INVALID;
  const int C = gru_c->C, B = n_img;
  const size_t PC = (size_t)B * H * W * C;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, (hipStream_t)stream);
  float* zeros = A.take(PC);
  float* sbuf[2] = {A.take(PC), A.take(PC)};
  float* pbuf[2] = {A.take(PC), A.take(PC)};
  float* k = A.take(PC);       // solver stages (midpoint / RK4)
  float* pk = A.take(PC);
  float* acc = A.take(PC);
  float* s3 = A.take(PC);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  SF_HIP(zero_fill(zeros, al(PC) * sizeof(float), st));
  SF_HIP(zero_fill(sbuf[0], al(PC) * sizeof(float), st));   // state = zeros  (temporal_ode_bayes.py:507)
  SF_HIP(zero_fill(pbuf[0], al(PC) * sizeof(float), st));   // input: overwritten by the first jump (:565,574)
  int si = 0, pi = 0, draw = 0;
  const int cstride = coef_per_image ? SF_COEF_STRIDE : 0;
  const size_t step_stride = (size_t)SF_COEF_STRIDE * (coef_per_image ? B : 1);
  std::vector<Stage> stages;
  for (int i = 0; i < n_ops; ++i) {
    const int kind = ops[2 * i], arg = ops[2 * i + 1];
    // the imputed input this op leaves behind is only read by a following ODE step (:446-455); a jump ignores it
    // (:327-344) and then overwrites it (:574), and nothing reads it after the last op: those infer_state passes
    // are skipped (their noise draw keeps its index)
    const bool next_is_step = (i + 1 < n_ops) && ops[2 * (i + 1)] == SF_OP_STEP;
    const bool need_p = impute && next_is_step;
    float* s = sbuf[si]; float* s2 = sbuf[si ^ 1];
    float* p = pbuf[pi]; float* p2 = pbuf[pi ^ 1];
    if (kind == SF_OP_JUMP) {   // :562-574
      stages.push_back(Stage{gru_obs, hx_obs + (size_t)arg * PC, s, s2, 0, nullptr, nullptr, nullptr, 0, need_p, draw, p, i});
      draw += 1;
      si ^= 1;
    } else if (kind == SF_OP_STEP) {
      if (!coef) return SF_ERR_INVALID;
      const float* cf = coef + (size_t)arg * step_stride;
      const float* x = impute ? p : zeros;
      if (solver == SF_SOLVER_EULER) {
        stages.push_back(Stage{gru_c, x, s, s2, 1, s, cf + 0, nullptr, 0, need_p, draw, p2, i});
        draw += 1;
      } else if (solver == SF_SOLVER_MIDPOINT) {
        // k = s + dt/2 f(p, s); pk = infer(k) (:452 — evaluated even when IMPUTE is off); s' = s + dt f(pk, k)
        stages.push_back(Stage{gru_c, x, s, k, 1, s, cf + 1, nullptr, 0, true, draw, pk, -1});
        stages.push_back(Stage{gru_c, pk, k, s2, 1, s, cf + 0, nullptr, 0, need_p, draw + 1, p2, i});
        draw += 2;
      } else {
        // RK4 (build-defined): acc = s + dt/6 k1 + dt/3 k2 + dt/3 k3 ; s' = acc + dt/6 k4.  Stage coefficient pairs
        // {out coef, out2 coef} follow the four scalars in the coef record (SF_COEF_STRIDE in sfnative.h); stage inputs
        // always come from infer_state, as in midpoint
        const float* rk = cf + 4;
        stages.push_back(Stage{gru_c, x, s, k, 1, s, rk + 0, acc, 0, true, draw, pk, -1});
        stages.push_back(Stage{gru_c, pk, k, s3, 1, s, rk + 2, acc, 1, true, draw + 1, pk, -1});
        stages.push_back(Stage{gru_c, pk, s3, k, 1, s, rk + 4, acc, 1, true, draw + 2, pk, -1});
        stages.push_back(Stage{gru_c, pk, k, s2, 1, acc, rk + 6, nullptr, 0, need_p, draw + 3, p2, i});
        draw += 4;
      }
      si ^= 1;
      pi ^= 1;
    } else {
      return SF_ERR_INVALID;
    }
  }
  const unsigned* flow_err = nullptr;
  SF_TRY(run_stages(stages, *pm, eps, philox, cstride, sel_nops, n_targets, out_states, B, H, W, A, st, &flow_err));
  if (final_state) SF_HIP(copy_floats(sbuf[si], final_state, PC, st));
  if (flow_err) {      // the flow kernel's bounded waits: a timeout must not pass as a result
    hipLaunchKernelGGL(flow_poison_kernel, dim3(256), dim3(256), 0, st, flow_err, out_states, (size_t)n_targets * PC, final_state, final_state ? PC : 0);
    SF_HIP(hipGetLastError());
    g_flow_err_last = flow_err;
  } else {
    g_flow_err_last = nullptr;      // this thread's most recent rollout has no bounded waits: sf_flow_errors says so instead of reporting an older one
  }
  return SF_OK;
}
int sf_nnfo_rollout_fwd(const sf_dual_w* gru_c, const sf_dual_w* gru_obs, const sf_pmodel_w* pm, int solver, int impute,
                        const int32_t* ops, int n_ops, const float* hx_obs, const float* eps, const float* coef,
                        int coef_per_image, const int32_t* sel_nops, int n_targets, float* out_states,
                        float* final_state, int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream) {
  if (!eps) return SF_ERR_INVALID;
  return rollout_core(gru_c, gru_obs, pm, solver, impute, ops, n_ops, hx_obs, eps, nullptr, coef, coef_per_image, sel_nops, n_targets,
                      out_states, final_state, n_img, H, W, ws, ws_bytes, stream);
}
// the same with the Gaussian noise of infer_state generated in the sampling epilogue (Philox4x32-10, csrc/sf_math.h)
int sf_nnfo_rollout_philox_fwd(const sf_dual_w* gru_c, const sf_dual_w* gru_obs, const sf_pmodel_w* pm, int solver, int impute,
                               const int32_t* ops, int n_ops, const float* hx_obs, const uint64_t* philox_state, const float* coef,
                               int coef_per_image, const int32_t* sel_nops, int n_targets, float* out_states, float* final_state,
                               int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream) {
  if (!philox_state) return SF_ERR_INVALID;
  return rollout_core(gru_c, gru_obs, pm, solver, impute, ops, n_ops, hx_obs, nullptr, reinterpret_cast<const unsigned long long*>(philox_state),
                      coef, coef_per_image, sel_nops, n_targets, out_states, final_state, n_img, H, W, ws, ws_bytes, stream);
}
int sf_infer_state_philox_fwd(const sf_pmodel_w* w, const float* s, const uint64_t* philox_state, int draw, float* p_out, float* q_out,
                              int n_img, int H, int W, float* ws, size_t ws_bytes, void* stream) {
  if (!w || !s || !philox_state || !p_out || w->C <= 0 || (w->C % 8) || n_img < 1 || draw < 0) return SF_ERR_INVALID;
  Arena A(ws, ws_bytes);
  SplitScope sp(A, (hipStream_t)stream);
  return infer_state(*w, s, nullptr, p_out, q_out, n_img, H, W, A, (hipStream_t)stream, reinterpret_cast<const unsigned long long*>(philox_state), draw);
}

// ---- SmallEncoder / SmallDecoder ------------------------------------------------------------------
size_t sf_small_encoder_ws_bytes(int C, int F, int n, int H, int W) {
  size_t P0 = (size_t)n * H * W;
  int Cm = C > F ? C : F;
  // level 0: t (C), pr (F), o0 (F), pooled (F@1/4); level 1 at P0/4: t (F), pr (2F), o (2F), pooled@1/16;
  // level 2 at P0/16: t (4F)... be generous: 4 buffers of the largest tensor per level
  size_t lvl0 = 3 * al(P0 * Cm), lvl1 = 4 * al(P0 / 4 * 2 * F + 64), lvl2 = 6 * al(P0 / 16 * 4 * F + 64);
  return (lvl0 + lvl1 + lvl2) * sizeof(float);
}
int sf_small_encoder_fwd(const sf_encoder_w* w, const float* x, float* out, int n, int H, int W, float* ws,
                         size_t ws_bytes, void* stream) {
  if (!w || !x || !out) return SF_ERR_INVALID;
  const int C = w->C, F = w->F;
  hipStream_t st = (hipStream_t)stream;
  Arena A(ws, ws_bytes);
  const size_t P0 = (size_t)n * H * W;
  const int H1 = H / 2, W1 = W / 2, H2 = H1 / 2, W2 = W1 / 2;
  const size_t P1 = (size_t)n * H1 * W1, P2 = (size_t)n * H2 * W2;
  const int Cm = C > F ? C : F;
  float* t0 = A.take(P0 * Cm); float* pr0 = A.take(P0 * Cm); float* o0 = A.take(P0 * Cm);
  float* q1 = A.take(P1 * F); float* t1 = A.take(P1 * F); float* pr1 = A.take(P1 * 2 * F); float* o1 = A.take(P1 * 2 * F);
  float* q2 = A.take(P2 * 2 * F); float* t2 = A.take(P2 * 4 * F); float* pr2 = A.take(P2 * 4 * F);
  float* o2 = A.take(P2 * 4 * F); float* o3 = A.take(P2 * 4 * F); float* o4 = A.take(P2 * 4 * F);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  SF_TRY(res_block(w->blocks[0], x, o0, t0, pr0, n, H, W, 0, st, q1));        // res_models.py:101-105: block, MaxPool2d(2)
  SF_TRY(res_block(w->blocks[1], q1, o1, t1, pr1, n, H1, W1, 0, st, q2));
  SF_TRY(res_block(w->blocks[2], q2, o2, t2, pr2, n, H2, W2, 0, st));
  SF_TRY(res_block(w->blocks[3], o2, o3, t2, pr2, n, H2, W2, 0, st));
  SF_TRY(res_block(w->blocks[4], o3, o4, t2, pr2, n, H2, W2, 0, st));
  return run1(problem(w->last, o4, nullptr, out, n, H2, W2), EPI_AFFINE, st);   // :106 conv+BN+tanh
}

size_t sf_small_decoder_ws_bytes(int C, int F, int n, int h, int w) {
  size_t P = (size_t)n * h * w;
  return (4 * al(P * 4 * F) + 3 * al(4 * P * 2 * F) + 4 * al(16 * P * (size_t)(F > C ? F : C))) * sizeof(float);
}
int sf_small_decoder_fwd(const sf_decoder_w* w, const float* z, float* out, int n, int h, int wd, float* ws,
                         size_t ws_bytes, void* stream) {
  if (!w || !z || !out) return SF_ERR_INVALID;
  const int F = w->F, C = w->C;
  const int Cm = F > C ? F : C;
  hipStream_t st = (hipStream_t)stream;
  Arena A(ws, ws_bytes);
  const size_t P = (size_t)n * h * wd;
  float* a0 = A.take(P * 4 * F); float* a1 = A.take(P * 4 * F); float* a2 = A.take(P * 4 * F); float* a3 = A.take(P * 4 * F);
  float* b0 = A.take(4 * P * 2 * F); float* b1 = A.take(4 * P * 2 * F); float* b2 = A.take(4 * P * 2 * F);
  float* c0 = A.take(16 * P * Cm); float* c1 = A.take(16 * P * Cm); float* c2 = A.take(16 * P * Cm); float* c3 = A.take(16 * P * Cm);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  SF_TRY(run1(problem(w->first, z, nullptr, a0, n, h, wd), EPI_AFFINE, st));           // res_models.py:136
  SF_TRY(res_block(w->blocks[0], a0, a1, a2, a3, n, h, wd, 0, st));                     // 4F -> 2F
  SF_TRY(res_block(w->blocks[1], a1, a0, a2, a3, n, h, wd, 0, st));                     // 2F -> 2F
  SF_TRY(res_block(w->blocks[2], a0, a1, a2, a3, n, h, wd, 0, st));                     // 2F -> 2F
  // upsample(x2) is applied on read by block 3 (conv_1 and projection), :142-143
  SF_TRY(res_block(w->blocks[3], a1, b0, b1, b2, n, h, wd, 1, st));                     // 2F -> F @2h
  // block 4 (F -> F @4h) reads block 3's output upsampled: on read where its second convolution runs on the Winograd kernel (the
  // identity skip then comes from the half-size tensor too), from a materialised copy otherwise
  int rc = res_block(w->blocks[4], b0, c1, c2, c3, n, 2 * h, 2 * wd, 1, st);
  if (rc == SF_ERR_UNSUPPORTED) {
    SF_HIP(launch_upsample2(b0, c0, n, 2 * h, 2 * wd, F, st));
    rc = res_block(w->blocks[4], c0, c1, c2, c3, n, 4 * h, 4 * wd, 0, st);
  }
  SF_TRY(rc);
  SF_TRY(run1(problem(w->last0, c1, nullptr, c2, n, 4 * h, 4 * wd), EPI_AFFINE, st));
  return run1(problem(w->last1, c2, nullptr, out, n, 4 * h, 4 * wd), EPI_AFFINE, st);
}

// ---- Bottleneck / distribution heads (secondary, BEVerse-compatible classes) ------------------------
size_t sf_bottleneck_ws_bytes(int Cin, int Cout, int n, int H, int W) {
  size_t P = (size_t)n * H * W;
  return (2 * al(P * (Cin / 2)) + al(P * Cin) + al(P * Cout)) * sizeof(float);
}
// Bottleneck.forward (streamingflow/layers/convolutions.py:65-172, beverse basic_modules.py:68-178):
// 1x1 -> BN,ReLU -> 3x3 (stride 2 if downsample) -> BN,ReLU -> 1x1 -> BN,ReLU, + skip (identity, or
// [zero-pad + 2x2 max-pool] + 1x1 + BN).  out: [n][Ho][Wo][Cout], Ho = ceil(H/2) when downsampling.
int sf_bottleneck_fwd(const sf_bottleneck_w* w, const float* x, float* out, int n, int H, int W, float* ws,
                      size_t ws_bytes, void* stream) {
  if (!w || !x || !out || !valid_w(w->down) || !valid_w(w->conv) || !valid_w(w->up)) return SF_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  Arena A(ws, ws_bytes);
  const int Cin = w->down.c0, Cm = w->down.cout, Cout = w->up.cout;
  const int Ho = w->downsample ? (H + 1) / 2 : H, Wo = w->downsample ? (W + 1) / 2 : W;
  const size_t P = (size_t)n * H * W, Po = (size_t)n * Ho * Wo;
  float* t1 = A.take(P * Cm);
  float* t2 = A.take(Po * Cm);
  float* xp = A.take(Po * Cin);
  float* pr = A.take(Po * Cout);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  SF_TRY(run1(problem(w->down, x, nullptr, t1, n, H, W), EPI_AFFINE, st));
  ConvProblem c = problem(w->conv, t1, nullptr, t2, n, H, W);
  if (c.Hout != Ho || c.Wout != Wo) return SF_ERR_INVALID;
  SF_TRY(run1(c, EPI_AFFINE, st));
  ConvProblem u = problem(w->up, t2, nullptr, out, n, Ho, Wo);
  if (w->proj.w) {
    const float* src = x;
    if (w->downsample) {
      SF_HIP(launch_maxpool2(x, xp, n, H, W, Cin, 1, st));
      src = xp;
    }
    SF_TRY(run1(problem(w->proj, src, nullptr, pr, n, Ho, Wo), EPI_AFFINE, st));
    u.add = pr;
  } else {
    if (w->downsample || Cin != Cout) return SF_ERR_INVALID;
    u.add = x;
  }
  return run1(u, EPI_AFFINE, st);
}

// 1x1 head of the (Spatial)DistributionModule (beverse motion_modules.py:34-46, 74-88): optional
// global average pool, 1x1 conv with bias, log-sigma half clamped to [lo, hi] when clamp != 0.
size_t sf_dist_head_ws_bytes(int C, int n) { return (al((size_t)n * 64 * C) + al((size_t)n * C)) * sizeof(float); }
int sf_dist_head_fwd(const sf_conv_w* w, const float* enc, float* out, int n, int H, int W, int global_pool, int clamp,
                     float lo, float hi, float* ws, size_t ws_bytes, void* stream) {
  if (!w || !enc || !out || !valid_w(*w)) return SF_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int C = w->c0;
  const float* src = enc;
  int Hh = H, Wh = W;
  if (global_pool) {
    Arena A(ws, ws_bytes);
    float* part = A.take((size_t)n * 64 * C);
    float* mean = A.take((size_t)n * C);
    if (!A.ok()) return SF_ERR_WORKSPACE;
    SF_HIP(launch_chan_partial(enc, part, n, H * W, C, 64, st));
    SF_HIP(launch_mean_from_partials(part, mean, n, 64, C, H * W, st));
    src = mean; Hh = 1; Wh = 1;
  }
  ConvProblem p = problem(*w, src, nullptr, out, n, Hh, Wh);
  if (clamp) { p.clamp_from = w->cout / 2; p.clamp_lo = lo; p.clamp_hi = hi; }
  return run1(p, EPI_AFFINE, st);
}

// ---- head blocks ----------------------------------------------------------------------------------
size_t sf_convnext_block_ws_bytes(int C, int n, int H, int W) {
  size_t P = (size_t)n * H * W;
  return (al(P * C) + al(P * 4 * C)) * sizeof(float);
}
int sf_convnext_block_fwd(const sf_convnext_w* w, const float* x, float* out, int n, int H, int W, float* ws,
                          size_t ws_bytes, void* stream) {
  if (!w || !x || !out) return SF_ERR_INVALID;
  const int C = w->C;
  hipStream_t st = (hipStream_t)stream;
  Arena A(ws, ws_bytes);
  const size_t P = (size_t)n * H * W;
  float* t = A.take(P * C);
  float* u = A.take(P * 4 * C);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  if (launch_dwconv7_ln(x, t, w->dw_w, w->dw_b, w->ln_w, w->ln_b, n, H, W, C, 1e-6f, st) != hipSuccess)
    return SF_ERR_UNSUPPORTED;
  // 64 -> 256 -> 64 (every ConvNeXt block of the reference's configs): the two pointwise layers in one launch, the hidden tensor
  // never written (convnext_mlp.hip).  Exact fp32 only: the bf16x3 mode keeps its own K loop on the two-launch path
  static const bool fuse = [] { const char* v = std::getenv("SF_MLP_FUSED"); return v ? std::atoi(v) != 0 : true; }();
  const sf_conv_w &a = w->pw1, &b = w->pw2;
  const bool one_by_one = a.kh == 1 && a.kw == 1 && b.kh == 1 && b.kw == 1 && a.stride == 1 && b.stride == 1 && a.pad == 0 && b.pad == 0;
  if (fuse && one_by_one && C == 64 && a.c0 == 64 && a.c1 == 0 && a.cin_pad == 64 && a.cout == 256 && a.cout_pad == 256 && a.act == ACT_GELU &&
      b.c0 == 256 && b.c1 == 0 && b.cin_pad == 256 && b.cout == 64 && b.cout_pad == 64 && b.act == ACT_NONE && a.w && b.w &&
      !(tune().b3 && a.w_bf16x3 && b.w_bf16x3)) {
    SF_TRY(seg_flush());
    if (!g_prof.on) {
      SF_HIP(launch_convnext_mlp(t, x, out, a.w, a.scale, a.bias, b.w, b.scale, b.bias, (long)P, st));
      return SF_OK;
    }
    ProfRec r;
    r.key = 20 * 8 + EPI_AFFINE;
    r.flops = 2.0 * (double)P * (64.0 * 256.0 + 256.0 * 64.0);
    r.bytes = 4.0 * ((double)P * 64.0 * 3.0 + 2.0 * 64.0 * 256.0);     // t and x once, out once, both weight matrices once
    r.a = g_prof.get(); r.b = g_prof.get();
    SF_HIP(hipEventRecord(r.a, st));
    SF_HIP(launch_convnext_mlp(t, x, out, a.w, a.scale, a.bias, b.w, b.scale, b.bias, (long)P, st));
    SF_HIP(hipEventRecord(r.b, st));
    g_prof.recs.push_back(r);
    return SF_OK;
  }
  SF_TRY(run1(problem(w->pw1, t, nullptr, u, n, H, W), EPI_AFFINE, st));
  ConvProblem p2 = problem(w->pw2, u, nullptr, out, n, H, W);
  p2.add = x;
  return run1(p2, EPI_AFFINE, st);
}

size_t sf_deeplab_head_ws_bytes(int C, int hid, int n, int H, int W) {
  size_t P = (size_t)n * H * W;
  return (al(P * 4 * hid) + 2 * al(P * hid) + al((size_t)n * ASPP_SLABS * C) + al((size_t)n * hid)) * sizeof(float);
}
static int deeplab_head(const sf_deeplab_w* w, const float* x, float* out, int n, int H, int W, int planar_group, size_t stride_major,
                        size_t stride_minor, float* ws, size_t ws_bytes, void* stream) {
  if (!w || !x || !out) return SF_ERR_INVALID;
  const int C = w->C, hid = w->hid;
  hipStream_t st = (hipStream_t)stream;
  Arena A(ws, ws_bytes);
  const size_t P = (size_t)n * H * W;
  float* cat = A.take(P * 4 * hid);
  float* y = A.take(P * hid);
  float* z = A.take(P * hid);
  float* part = A.take((size_t)n * ASPP_SLABS * C);
  float* bimg = A.take((size_t)n * hid);
  if (!A.ok()) return SF_ERR_WORKSPACE;
  SF_HIP(launch_aspp_pool(x, part, bimg, n, H * W, C, hid, w->pool_w, w->pool_scale, w->pool_bias, w->proj_pool_w,
                          w->project.scale, w->project.bias, ASPP_SLABS, st));
  ConvProblem ps[4];
  for (int i = 0; i < 4; ++i) {   // convolutions.py:265-269
    ps[i] = problem(w->branch[i], x, nullptr, cat, n, H, W);
    ps[i].out_cs = 4 * hid; ps[i].out_co = i * hid;
  }
  SF_TRY(run(ps, 4, EPI_AFFINE, st));
  ConvProblem pj = problem(w->project, cat, nullptr, y, n, H, W);
  pj.bias = bimg; pj.bias_per_img = 1;
  SF_TRY(run1(pj, EPI_AFFINE, st));
  SF_TRY(run1(problem(w->conv3, y, nullptr, z, n, H, W), EPI_AFFINE, st));
  ConvProblem pc = problem(w->cls, z, nullptr, out, n, H, W);
  if (planar_group > 0) { pc.out_planar = 1; pc.pl_div = planar_group; pc.pl_sa = stride_major; pc.pl_sb = stride_minor; }
  return run1(pc, EPI_AFFINE, st);
}
int sf_deeplab_head_fwd(const sf_deeplab_w* w, const float* x, float* out, int n, int H, int W, float* ws,
                        size_t ws_bytes, void* stream) {
  return deeplab_head(w, x, out, n, H, W, 0, 0, 0, ws, ws_bytes, stream);
}
// the same with the classifier writing the boundary layout itself: image i as [cout][H][W] planes at
// out + (i / group) * stride_major + (i % group) * stride_minor floats (frames (t, b) of a [T][B] run into a [B][T][C][H][W] tensor:
// group = B, stride_major = C*H*W, stride_minor = T*C*H*W) — no transpose launches after the head (future_prediction_ode.py:62-64)
int sf_deeplab_head_planar_fwd(const sf_deeplab_w* w, const float* x, float* out, int n, int H, int W, int group, size_t stride_major,
                               size_t stride_minor, float* ws, size_t ws_bytes, void* stream) {
  if (group < 1 || !w || stride_major < (size_t)w->cls.cout * H * W || stride_minor < (size_t)w->cls.cout * H * W) return SF_ERR_INVALID;
  return deeplab_head(w, x, out, n, H, W, group, stride_major, stride_minor, ws, ws_bytes, stream);
}

// ---- graphs / events ------------------------------------------------------------------------------
int sf_graph_begin(void* stream) {
  SF_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return SF_OK;
}
int sf_graph_end(void* stream, void** exec_out) {
  if (!exec_out) return SF_ERR_INVALID;
  hipGraph_t g = nullptr;
  SF_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t e = nullptr;
  hipError_t r = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (r != hipSuccess) return SF_ERR_LAUNCH;
  *exec_out = (void*)e;
  return SF_OK;
}
int sf_graph_launch(void* exec, void* stream) {
  SF_HIP(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream));
  return SF_OK;
}
int sf_graph_destroy(void* exec) {
  SF_HIP(hipGraphExecDestroy((hipGraphExec_t)exec));
  return SF_OK;
}

/* Diagnostic builds (-DSF_STAMP) only: `buf` = 64 slots x 4096 workgroups x 8 uint64 device buffer (NULL switches the
 * stamps off); conv launches then record in-kernel s_memrealtime stamps into consecutive slots.  SF_ERR_UNSUPPORTED
 * in the product build. */
int sf_set_flow_mode(int on) {
  const int was = g_flow_mode < 0 ? tune().persist : g_flow_mode;
  g_flow_mode = on < 0 ? -1 : (on ? 1 : 0);
  return was;
}
// number of timed-out dependency waits of the calling thread's most recent persistent-flow rollout (0 = healthy; > 0: its outputs were
// overwritten with NaN).  Synchronises `stream`.  SF_ERR_INVALID: this thread has not run one.
int sf_flow_errors(void* stream) {
  if (!g_flow_err_last) return SF_ERR_INVALID;
  unsigned v = 0;
  if (hipMemcpyAsync(&v, g_flow_err_last, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return SF_ERR_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SF_ERR_LAUNCH;
  return (int)(v > 0x3fffffffu ? 0x3fffffffu : v);
}

int sf_debug_stamps(void* buf) {
  if (set_stamp_buffer((unsigned long long*)buf) != hipSuccess || set_stamp_buffer_sp((unsigned long long*)buf) != hipSuccess ||
      set_stamp_buffer_wino((unsigned long long*)buf) != hipSuccess)
    return SF_ERR_UNSUPPORTED;
  g_stamp_on = buf != nullptr;
  g_stamp_slot = 0;
  return SF_OK;
}

/* diagnostic: workgroups per CU of the large LDS-DMA tiles (0: 128x128 fp32, 1: 128x128 bf16x3, 2: 64x128 fp32, 3: 64x128 bf16x3) */
int sf_debug_occupancy(int which) { return glds_occupancy(which); }

int sf_prof_enable(int on) {
  g_prof.on = on != 0;
  return SF_OK;
}
// Aggregates (and clears) the recorded launches by kernel key = cfg*8 + epi.  Arrays of length SF_PROF_KEYS (include/sfnative.h; part of the ABI version):
// calls, total ms, total algorithmic flops, total algorithmic bytes.  Synchronises.
int sf_prof_collect(int32_t* calls, double* ms, double* flops, double* bytes) {
  if (!calls || !ms || !flops || !bytes) return SF_ERR_INVALID;
  for (int i = 0; i < SF_PROF_KEYS; ++i) { calls[i] = 0; ms[i] = 0; flops[i] = 0; bytes[i] = 0; }
  for (auto& r : g_prof.recs) {
    float t = 0.f;
    SF_HIP(hipEventSynchronize(r.b));
    SF_HIP(hipEventElapsedTime(&t, r.a, r.b));
    if (r.key >= 0 && r.key < SF_PROF_KEYS) { calls[r.key] += 1; ms[r.key] += t; flops[r.key] += r.flops; bytes[r.key] += r.bytes; }
    static const bool dump = std::getenv("SF_PROF_DUMP") != nullptr;      // debugging aid: one line per profiled launch, in launch order
    if (dump) std::fprintf(stderr, "[sf-prof] key=%d us=%.1f gflop=%.3f mbytes=%.2f\n", r.key, t * 1e3, r.flops * 1e-9, r.bytes * 1e-6);
    g_prof.pool.push_back(r.a);
    g_prof.pool.push_back(r.b);
  }
  g_prof.recs.clear();
  return SF_OK;
}

int sf_event_create(void** ev) {
  hipEvent_t e;
  SF_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return SF_OK;
}
int sf_event_record(void* ev, void* stream) { SF_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return SF_OK; }
int sf_event_elapsed_ms(void* a, void* b, float* ms) {
  SF_HIP(hipEventSynchronize((hipEvent_t)b));
  SF_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return SF_OK;
}
int sf_event_destroy(void* ev) { SF_HIP(hipEventDestroy((hipEvent_t)ev)); return SF_OK; }

}  // extern "C"
