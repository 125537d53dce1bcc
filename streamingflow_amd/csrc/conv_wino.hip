// Winograd F(2x2, 3x3) convolution for the 3x3 / stride-1 layers of the batched forward (gfx950 only).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 4x4 input tile d (stride 2), 3x3 filter g, 2x2 outputs Y
// in exact fp32 arithmetic on v_mfma_f32_16x16x4_f32: 16 multiplies per 2x2 outputs and (cin, cout) pair instead of 36 — 2.25x fewer
// MACs than the direct form the implicit-GEMM kernels compute; different rounding (the transforms add / subtract before the products):
// tools/r04/winograd_study.py measured <= 7e-6 max-abs on the BEV output of every BASELINE config with all eligible layers switched
// (north-star tolerance 1e-3).  FLOP accounting: bench.py prices these launches at their EXECUTED FLOPs for the roofline and keeps the
// algorithmic (direct-form) count for ODE-steps/s (savings are not credited as achieved FLOPs).
//
// Structure — one 512-thread workgroup per CU, tile = COUT_T output channels x (TH x 8) Winograd tiles (= 2TH x 16 output pixels):
//   * weights are transformed once at pack time (pack.hip: U[cin/16][16 positions][cout_pad][16], sf_conv_w::w_wino);
//   * per 16-channel chunk the (2TH+2) x 18 pixel patch of the input is DMA'd to LDS (buffer_load ... lds, zero fill = the range
//     check), every thread transforms its share (B^T d B: 8 float4 LDS reads, 8 float4 add/sub, 4 float4 LDS writes) into
//     V[16 positions][tile][16 channels];
//   * 8 stages per chunk, two positions each: U[2 positions][COUT_T][16] streams through a ring of 3 LDS buffers (2 stages in flight,
//     counted vmcnt), one barrier per stage between a stage's fragment reads and its MFMAs (as conv_sp.hip / conv_glds_kernel);
//   * wave (wm, wn) owns 32 cout x 16 tiles x ALL 16 positions: 32 accumulator tiles (128 VGPRs), so the output transform A^T M A is
//     register-local: lane (tile j, channel quad g) ends with its four channels of the tile's 2x2 output pixels;
//   * fused epilogues as in the other kernels: AFFINE (scale / bias = conv bias or BN fold, activation, residual add, GRU reset-gate
//     second output) and BLEND (conv-GRU state update).
// LDS rows are 16 floats (64 B); the 16-byte slot s of row r lives at slot s ^ ((r >> 2) & 2): conflict-free for the ds_read_b128
// lane groups of a 16-row fragment (MI355X_MICROARCH.md, LDS table).
#include "sf_math.h"

#include <cstdlib>
#include <type_traits>

namespace sf {

#ifdef SF_STAMP
static __device__ unsigned long long* g_sf_stamps = nullptr;      // diagnostic builds: in-kernel time stamps (sf_device.h)
hipError_t set_stamp_buffer_wino(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sf_stamps), &p, sizeof(p)); }
#else
hipError_t set_stamp_buffer_wino(unsigned long long*) { return hipErrorNotSupported; }
#endif

constexpr int WN_THREADS = 512;
// -DSF_WINO_ROLL builds the rolling input transform (conv_wino_kernel, "ROLLING INPUT TRANSFORM"): measured 1-4 % slower than the
// one-shot transform at the chunk boundary (profiles/r04_z5_winobench_rolling_vs_one_shot_transform.txt), kept for experiments
#if defined(SF_WINO_ROLL)
constexpr bool WN_ROLL_BUILD = true;
#else
constexpr bool WN_ROLL_BUILD = false;
#endif
typedef __attribute__((address_space(3))) void wn_lds_void;

// MW: 16-row cout tiles per wave (2: wave = 32 cout x 16 tiles, 128 accumulator registers, one workgroup per CU;
//     1: wave = 16 cout x 16 tiles, 64 accumulator registers, <= 128 registers and <= 80 KB LDS: TWO workgroups per CU, each
//     one's barriers, transform, prologue and epilogue under the other's MFMAs)
// DIL: dilated 3x3 (pad = dilation) as d x d interleaved undilated problems ("phases": output pixel (y, x) belongs to phase
//     (y mod d, x mod d) and sees only input pixels of its own phase).  The tile grid of a workgroup is the product of two lists —
//     along x: for every phase p its ceil(len(p) / 2) tiles in turn, along y the same — so a block of TH x 8 tiles may span up to
//     RY x RX phases, and the patch keeps 2 n + 2 rows / columns per run of n tiles of one phase.
// CAT: narrow images (a 50x50 latent is 25 tiles wide: blocks of 8 tile columns would be 28 % empty) are concatenated along x — the
//     tile columns of all images form one list, a block of 8 columns may span two images, and the patch keeps a halo column on either
//     side of each image's run (2 n + 2 columns per run, as DIL).  One input scale / bias set per workgroup: layers with per-image
//     epilogue operands keep the plain form.
template <int COUT_T, int TH, int MW, bool DIL = false, bool CAT = false>
struct WinoGeo {
  static_assert(!(DIL && CAT), "one run structure at a time");
  static constexpr int TW = 8, WT = TH * TW;                   // Winograd tiles of a workgroup: TH rows x 8 columns
  static constexpr int RY = 2, RX = 4;                         // DIL: runs (phases) a block may span: >= 3 tiles per phase (wino_takes)
  static constexpr int PH = 2 * TH + (DIL ? 2 * RY : 2), PW = 2 * TW + (DIL ? 2 * RX : (CAT ? 4 : 2));       // input patch (pixels)
  static constexpr int NPX = PH * PW;
  static constexpr int WMW = COUT_T / (16 * MW), WNW = WT / 16; // waves along cout / tiles
  static_assert(WMW * WNW == 8, "eight waves");
  static constexpr int NU = COUT_T / 64;                       // U DMAs per wave and stage (a stage = 2 positions x COUT_T rows of 64 B)
  static constexpr int NP = (NPX * 4 + 511) / 512;             // patch DMAs per wave and chunk
  static constexpr int NVB = (WT == 32 && MW == 2) ? 2 : 1;    // V buffers
  static constexpr int NPB = MW == 2 ? 2 : 1;                  // patch buffers (1: the next patch is issued behind the chunk-boundary barrier)
  static constexpr int U_FLOATS = 2 * COUT_T * 16;             // one stage
  static constexpr int V_FLOATS = 16 * WT * 16;                // one chunk
  static constexpr bool COMPACT = DIL || MW == 1;              // patch = exactly ND DMAs (wave w issues pieces w, w + 8, ...) instead of NP per wave
  static constexpr int ND = (NPX * 4 + 63) / 64;
  static constexpr int P_FLOATS = COMPACT ? ND * 256 : 8 * NP * 64 * 4;      // one patch, padded to whole DMAs
  static constexpr int PARK = DIL ? 512 : ((MW == 1 && WN_ROLL_BUILD) ? ((NPX * 4 + 63) / 64) * 64 : 0);      // loop-invariant words kept in LDS instead of registers:
                                                               // DIL: a patch offset per thread; two workgroups per CU: the pixel offset of every patch piece
#if defined(SF_WINO_RING)
  static constexpr int RING = SF_WINO_RING;
#else
  static constexpr int RING = (MW == 1 && !DIL) ? 4 : 3;       // U stages in LDS (RING - 1 in flight); what fits twice into a CU's 160 KB
#endif
  static constexpr int SB = 2 * COUT_T;                        // the workgroup's scale / bias rows, staged in the prologue for the epilogue
  static constexpr int SC = (MW == 1 && !DIL) ? 256 : 0;       // per-channel input scale of the image (SE gate on in0, <= 256 channels), staged in the prologue
  static constexpr int LDS_FLOATS = RING * U_FLOATS + NVB * V_FLOATS + NPB * P_FLOATS + PARK + SB + SC;
  static constexpr int WG_PER_CU = MW == 2 ? 1 : 2;
};

__device__ __forceinline__ f32x4 wn_lds_read128(const float* p) {
  typedef const __attribute__((address_space(3))) f32x4 lds_f4;
  return *(lds_f4*)p;
}
// the lane index without a live register: two vector instructions where it is needed
// (volatile: otherwise everything derived from it is hoisted out of the stage loop and kept in registers the loop does not have)
__device__ __forceinline__ int wn_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ void wn_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction needs an immediate): all but the n youngest vector-memory operations
// of this wave are done.  n <= the true number of younger operations is always safe (it only waits for more).
__device__ __forceinline__ void wn_wait(const int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
  }
}

// Epilogue arithmetic (semantics of conv_igemm.hip run_epilogue).  fp32 MFMAs and vector-ALU instructions share one issue port, and a
// tile's epilogue runs beside the partner workgroup's stage loop: every vector instruction here is matrix time (in-kernel stamps,
// profiles/r04_x_stamps_wino.txt: 27-36 cycles per instruction, 5 us per tile).  So: packed fp32 adds / fmas on register pairs,
// addresses as a uniform 64-bit base per output pixel + ONE 32-bit lane offset per tensor (scalar arithmetic; the saddr form of
// global_load / global_store), lane masks as scalar conditions, operand loads requested before the output transform.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 wn_sub2(const f32x2 a, const f32x2 b) { return __builtin_elementwise_fma(b, (f32x2){-1.f, -1.f}, a); }   // v_pk_fma_f32
__device__ __forceinline__ f32x4 wn_sub4(const f32x4 a, const f32x4 b) { return __builtin_elementwise_fma(b, (f32x4){-1.f, -1.f, -1.f, -1.f}, a); }       // two v_pk_fma_f32 (a - b, exact: one rounding)
__device__ __forceinline__ f32x2 wn_lo(const f32x4 v) { return (f32x2){v[0], v[1]}; }
__device__ __forceinline__ f32x2 wn_hi(const f32x4 v) { return (f32x2){v[2], v[3]}; }
struct WnOps { float4 a, b; };      // AFFINE: residual, reset-gate state;  BLEND: update gate, state
// uniform per-pixel bases of the tensors an epilogue touches, and the lane's element offsets into them
struct WnPix { const float *ta, *tb; float *out, *out2; };
struct WnLane { unsigned ea, eb, eo, eo2; };
template <int EPI, class PT>
__device__ __forceinline__ WnOps wn_epi_load(const PT& P, const WnPix& px, const WnLane& ln) {
  WnOps o;
  o.a = spm_zero4(); o.b = spm_zero4();
  if constexpr (EPI == EPI_AFFINE) {
    if (P.add) o.a = spm_ld4(px.ta + (size_t)ln.ea);
    if (P.out2) o.b = spm_ld4(px.tb + (size_t)ln.eb);
  } else {
    o.a = spm_ld4(px.ta + (size_t)ln.ea);
    o.b = spm_ld4(px.tb + (size_t)ln.eb);
  }
  return o;
}
// v = the lane's four consecutive output channels c .. c+3 of one pixel; sc / bi = their scale and bias; as = residual scale
// returns the output value (and the reset-gate by-product in y2): the caller stores all four pixels of the tile at the very end —
// a store in the middle made the next pixel's arithmetic wait for its completion (hipcc guards the reuse of a store's data
// registers with s_waitcnt vmcnt(0): one memory round trip per pixel, seen in the ISA and in the in-kernel stamps)
struct WnOut { float4 y, y2; };
template <int EPI, class PT>
__device__ __forceinline__ WnOut wn_epi_finish(const PT& P, const f32x2 vlo, const f32x2 vhi, const WnOps& o, const float4 sc, const float4 bi,
                                               const float4 as, const int c) {
  WnOut r;
  r.y2 = spm_zero4();
  const f32x2 l = __builtin_elementwise_fma(vlo, (f32x2){sc.x, sc.y}, (f32x2){bi.x, bi.y});
  const f32x2 h = __builtin_elementwise_fma(vhi, (f32x2){sc.z, sc.w}, (f32x2){bi.z, bi.w});
  float4 v = make_float4(l[0], l[1], h[0], h[1]);
  float4 y;
  if constexpr (EPI == EPI_AFFINE) {
    const bool act_last = (P.mode & 2) != 0;
    y = act_last ? v : spm_act4(v, P.act);
    if (P.clamp_from >= 0) {
      if (c + 0 >= P.clamp_from) y.x = fminf(fmaxf(y.x, P.clamp_lo), P.clamp_hi);
      if (c + 1 >= P.clamp_from) y.y = fminf(fmaxf(y.y, P.clamp_lo), P.clamp_hi);
      if (c + 2 >= P.clamp_from) y.z = fminf(fmaxf(y.z, P.clamp_lo), P.clamp_hi);
      if (c + 3 >= P.clamp_from) y.w = fminf(fmaxf(y.w, P.clamp_lo), P.clamp_hi);
    }
    if (P.add) {
      if (P.add_scale) { y.x += o.a.x * as.x; y.y += o.a.y * as.y; y.z += o.a.z * as.z; y.w += o.a.w * as.w; }
      else { y.x += o.a.x; y.y += o.a.y; y.z += o.a.z; y.w += o.a.w; }
    }
    if (act_last) y = spm_act4(y, P.act);
    if (P.out2 && c >= P.gate_from)     // GRU gates, reset half: also emit (1 - r) * s, the candidate conv's input
      r.y2 = make_float4(o.b.x * (1.f - y.x), o.b.y * (1.f - y.y), o.b.z * (1.f - y.z), o.b.w * (1.f - y.w));
  } else {      // EPI_BLEND (temporal.py:56)
    v = spm_act4(v, P.act);
    const float4 u = o.a, st = o.b;
    if (P.mode & 1) y = make_float4(u.x * (v.x - st.x), u.y * (v.y - st.y), u.z * (v.z - st.z), u.w * (v.w - st.w));
    else y = make_float4((1.f - u.x) * st.x + u.x * v.x, (1.f - u.y) * st.y + u.y * v.y, (1.f - u.z) * st.z + u.z * v.z, (1.f - u.w) * st.w + u.w * v.w);
  }
  r.y = y;
  return r;
}

template <class PT> __device__ inline int nkc_stamp(const PT& P) { return P.cin_pad >> 4; }
// one axis of the dilated tile grid: N pixels, dilation d: phases p < r have q + 1 pixels (tb tiles), the others q (ts tiles)
struct WnAxis {
  int d, r, tb, ts, nt;
  __host__ __device__ WnAxis(int N, int dil) : d(dil) {
    const int q = N / dil;
    r = N - q * dil; tb = (q + 2) >> 1; ts = (q + 1) >> 1; nt = r * tb + (dil - r) * ts;
  }
  __host__ __device__ int tiles(int p) const { return p < r ? tb : ts; }
  __host__ __device__ void decode(int X, int& p, int& t) const {     // list index -> (phase, tile of the phase)
    const int nb = r * tb;
    if (X < nb) { p = X / tb; t = X - p * tb; }
    else { const int x2 = X - nb; const int pp = x2 / ts; p = r + pp; t = x2 - pp * ts; }
  }
  // patch row / column pc of a block that starts at (phase p0, tile t0) and holds n_blk tiles -> pixel coordinate; false: zero fill
  __host__ __device__ bool patch_coord(int pc, int p0, int t0, int n_blk, int N, int& coord) const {
    int c = 0, ph = p0, tstart = t0, left = n_blk;
    for (int run = 0; run < 4 && left > 0; ++run) {
      const int have = tiles(ph) - tstart;
      const int n = have < left ? have : left;
      if (pc < c + 2 * n + 2) {
        const int xs = 2 * tstart - 1 + (pc - c);
        coord = ph + d * xs;
        return ph < d && xs >= 0 && coord < N;
      }
      c += 2 * n + 2; left -= n; ph += 1; tstart = 0;
    }
    return false;
  }
};

template <int COUT_T, int TH, int MW, int EPI, bool DIL = false, bool CAT = false>
__global__ __launch_bounds__(WN_THREADS, (MW == 2 ? 2 : 4)) void conv_wino_kernel(const ConvLaunch L) {
  typedef WinoGeo<COUT_T, TH, MW, DIL, CAT> G;
  constexpr int TW = G::TW, WT = G::WT, PW = G::PW, NU = G::NU, NP = G::NP, NVB = G::NVB, NPB = G::NPB, RING = G::RING;
  static_assert(!DIL || (NVB == 1 && NPB == 1), "the dilated form exists for the two-workgroup configuration");
  constexpr bool ROLL = WN_ROLL_BUILD && NVB == 1 && NPB == 1 && WT == 32 && MW == 1 && !DIL;      // rolling input transform (below; the dilated form has no registers for it)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ubuf = smem;
  float* const Vbuf = Ubuf + RING * G::U_FLOATS;
  float* const Pbuf = Vbuf + NVB * G::V_FLOATS;
  const ConvProblem& P = L.p[0];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = P.Hout, W = P.Wout;                            // stride 1, pad = dilation: input size = output size
  const WnAxis ax(DIL ? W : 2, DIL ? P.dil : 1), ay(DIL ? H : 2, DIL ? P.dil : 1);      // plain form: constants, folded away
  const int tiles_x = DIL ? ax.nt : (W + 1) >> 1, tiles_y = DIL ? ay.nt : (H + 1) >> 1;
  const int tpi = tiles_x;                                     // CAT: tile columns per image; the column list has n_img * tpi entries
  const int nbx = ((CAT ? P.n_img * tiles_x : tiles_x) + TW - 1) / TW, nby = (tiles_y + TH - 1) / TH;
  // 1-D grid, XCD-aware: workgroup lin runs on XCD lin % 8 (round-robin dispatch).  Every XCD owns a contiguous range of tile blocks
  // (neighbouring blocks share their halo rows / columns through that XCD's L2), and the cout blocks of one tile block are
  // consecutive workgroups of the same XCD: the second one finds the input patch in L2.
  // (A persistent form — two workgroups per CU looping over the items of their XCD, argument pointer and thread index laundered per
  // tile against hoisting — measured 5 % SLOWER than one workgroup per item: profiles/r04_z7_winobench_persistent_vs_per_tile.txt.)
  const int ncb = P.cout_pad / COUT_T;
  const int nblk = nbx * nby * (CAT ? 1 : P.n_img), per_xcd = (nblk + 7) >> 3;
  const int lin = (int)blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  const int tb_ = slot / ncb;
  int b = xcd * per_xcd + tb_;
  if (b >= nblk) return;
  const int bx = b % nbx; b /= nbx;
  const int by = b % nby;
  // CAT: the block's first column is tile ct0 of image img; its columns from cn0 on belong to image img + 1 (tiles 0 ..)
  const int img = CAT ? (bx * TW) / tpi : b / nby;
  const int ct0 = CAT ? bx * TW - img * tpi : 0;
  const int cn0 = CAT ? (tpi - ct0 < TW ? tpi - ct0 : TW) : TW;
  const int ty0 = by * TH, tx0 = CAT ? ct0 : bx * TW;          // DIL: indices into the tile lists of the two axes
  int px0 = 0, pt0 = 0, py0 = 0, qt0 = 0;                      // DIL: (phase, tile) of the block's first column / row
  if constexpr (DIL) { ax.decode(tx0, px0, pt0); ay.decode(ty0, py0, qt0); }
  SF_STAMP_AT(L, 0);
#ifdef SF_STAMP
  SF_STAMP_VAL(L, 8, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));        // HW_ID: wave / SIMD / CU / SH / SE
  SF_STAMP_VAL(L, 9, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));       // XCC_ID
  SF_STAMP_VAL(L, 10, (unsigned long long)nkc_stamp(P));
#endif
  const int cout0 = (slot - tb_ * ncb) * COUT_T;
  const int nkc = P.cin_pad >> 4;                              // 16-channel chunks
  const int NS = nkc * 8;                                      // stages
  const int c0 = P.c0;

  const int img_px_i = P.Hin * P.Win;                          // CAT: pixel offset of the next image inside the buffer
  const int up = DIL ? 0 : P.in_up;                            // nearest x2 upsampling on read (plain form): input pixel (y >> 1, x >> 1)
  const int Win = P.Win;
#if defined(__HIP_DEVICE_COMPILE__)
  auto make_rsrc = [](const float* base, size_t bytes) {
    const unsigned nrec = bytes < 0x7fffffffull ? (unsigned)bytes : 0x7fffffffu;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (int)nrec, 0x00020000);
  };
  const size_t img_px = (size_t)P.Hin * P.Win;
  const size_t n_in = (CAT && img + 1 < P.n_img) ? 2 : 1;      // CAT: the buffer covers this image and the next
  const __amdgpu_buffer_rsrc_t rsrc0 = make_rsrc(P.in0 + (size_t)img * img_px * P.in0_cs, n_in * img_px * P.in0_cs * sizeof(float));
  const __amdgpu_buffer_rsrc_t rsrc1 = make_rsrc(P.in1 ? P.in1 + (size_t)img * img_px * P.in1_cs : P.in0, P.in1 ? n_in * img_px * P.in1_cs * sizeof(float) : 0);
  const __amdgpu_buffer_rsrc_t rsrc_u = make_rsrc(P.w_wino, (size_t)nkc * 16 * P.cout_pad * 16 * sizeof(float));
#endif
  float* const SCbuf = Pbuf + NPB * G::P_FLOATS + G::PARK + G::SB;      // [c0] input scales (SCALED), behind Park and SBuf
  constexpr bool SCALED = G::SC > 0 && EPI == EPI_AFFINE;      // the SE-scaled layers of p_model: the patch is multiplied per channel before the transform
  const bool scaled = SCALED && P.in_scale != nullptr;
  // ---- patch DMA: element e = (pixel, channel quad) of the (2TH+2) x 18 patch, 16 bytes each, LDS linear in e -----------------
  SF_STAMP_AT(L, 14);
  // DIL: DMA `idx = d * 8 + wave` of the G::ND the patch needs (wave w issues npw of them); one input tensor (wino_takes)
  // PQ (two workgroups per CU, two inputs possible): only the pixel offset is kept per DMA (one register instead of two); the byte
  // offset into the chunk's input tensor is formed when the DMA is issued (3 vector instructions, once per chunk)
  constexpr bool PQ = !DIL && G::COMPACT && WN_ROLL_BUILD;
  int pv0[PQ ? 1 : NP], pv1[(DIL || PQ) ? 1 : NP];
  float* const PqPark = Pbuf + NPB * G::P_FLOATS;              // = Park below (PQ only)
  const int npw = G::COMPACT ? (G::ND / 8 + (wave < G::ND % 8 ? 1 : 0)) : NP;      // wave-uniform
#pragma unroll
  for (int d = 0; d < NP; ++d) {
    const int e = (G::COMPACT ? d * 8 + wave : wave * NP + d) * 64 + lane;
    const int pix = e >> 2, quad = e & 3;
    const int py = pix / PW, px = pix - py * PW;
    int iy = 2 * ty0 - 1 + py, ix = 2 * tx0 - 1 + px;
    int run = 0;                                                // CAT: columns from 2 cn0 + 2 on are the next image's run, starting at its halo column x = -1
    if constexpr (CAT) {
      run = px >= 2 * cn0 + 2 ? 1 : 0;
      ix = run ? px - (2 * cn0 + 2) - 1 : ix;
    }
    bool ok = pix < G::NPX && iy >= 0 && iy < H && ix >= 0 && ix < W && (!CAT || (img + run < P.n_img && (run == 0 || cn0 < TW)));
    if constexpr (DIL) {
      const bool oky = ay.patch_coord(py, py0, qt0, TH, H, iy), okx = ax.patch_coord(px, px0, pt0, TW, W, ix);
      ok = pix < G::NPX && oky && okx;
    }
    const int pofs = DIL ? iy * W + ix : (iy >> up) * Win + (ix >> up) + (CAT ? run * img_px_i : 0);
    if constexpr (PQ) {
      if (d < npw) *(__attribute__((address_space(3))) int*)(PqPark + e) = ok ? pofs : -1;      // read back by the same lane when it issues the DMA
    } else pv0[d] = ok ? (pofs * P.in0_cs + quad * 4) * 4 : (int)0x80000000;      // + the chunk's channel offset stays out of range: zero fill
    if constexpr (!DIL && !PQ) pv1[d] = ok ? (pofs * P.in1_cs - c0 + quad * 4) * 4 : (int)0x80000000;
  }
  SF_STAMP_AT(L, 15);
  auto issue_patch = [&](const int kc) {
    float* const dst = Pbuf + (NPB == 2 ? (kc & 1) : 0) * G::P_FLOATS;
    const bool from1 = !DIL && kc * 16 >= c0;                   // wave-uniform: the whole chunk reads in1 (c0 % 16 == 0)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int d = 0; d < NP; ++d) {
      if (G::COMPACT && d >= npw) continue;
      float* const dB = dst + (G::COMPACT ? d * 8 + wave : wave * NP + d) * 256;
      if constexpr (DIL) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (wn_lds_void*)dB, 16, pv0[d] + kc * 64, 0, 0, 0);
      } else if constexpr (PQ) {
        const int cs4 = (from1 ? P.in1_cs : P.in0_cs) * 4, base = (from1 ? -c0 * 4 : 0) + kc * 64;      // scalars
        const int ln_ = wn_lane_id();
        const int pq = *(const __attribute__((address_space(3))) int*)(PqPark + (d * 8 + wave) * 64 + ln_);
        const int off = pq >= 0 ? pq * cs4 + base + (ln_ & 3) * 16 : (int)0x80000000;
        if (from1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (wn_lds_void*)dB, 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (wn_lds_void*)dB, 16, off, 0, 0, 0);
      } else {
        if (from1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (wn_lds_void*)dB, 16, pv1[d] + kc * 64, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (wn_lds_void*)dB, 16, pv0[d] + kc * 64, 0, 0, 0);
      }
    }
#else
    (void)dst; (void)from1;
#endif
  };
  // SCALED (SE gate on in0): every lane multiplies the patch pieces IT fetched by the four channel scales of its quad, in place, once
  // its DMAs have landed — no register lives across the stage loop for it, and the transform stays as it is
  auto scale_patch = [&](const int kc) {
    if constexpr (SCALED) {
      if (scaled && kc * 16 < c0) {
        const int ln_ = wn_lane_id();
        const f32x4 s4 = wn_lds_read128(SCbuf + kc * 16 + (ln_ & 3) * 4);
        typedef __attribute__((address_space(3))) f32x4 lds_f4w;
#pragma unroll
        for (int d = 0; d < NP; ++d) {
          if (d >= npw) continue;
          float* const q = Pbuf + ((d * 8 + wave) * 64 + ln_) * 4;
          *(lds_f4w*)q = wn_lds_read128(q) * s4;
        }
      }
    }
  };
  // ---- U DMA: stage S = (chunk kc, position pair st): rows r of [2 positions][COUT_T], 64 B each; piece q = 16 rows ------------
  int uv[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int r = (wave * NU + u) * 16 + (lane >> 2);          // row of the stage
    const int p = r / COUT_T, row = r - p * COUT_T;
    const int slot = (lane & 3) ^ ((row >> 2) & 2);            // LDS slot (lane & 3) holds source slot `slot`
    int grow = cout0 + row;
    grow = grow < P.cout_pad ? grow : P.cout_pad - 1;
    uv[u] = ((p * P.cout_pad + grow) * 16 + slot * 4) * 4;
  }
  const int u_stage_bytes = 2 * P.cout_pad * 16 * 4;            // two positions
  auto issue_u = [&](const int S) {
    float* const dst = Ubuf + (S % RING) * G::U_FLOATS;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int u = 0; u < NU; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_u, (wn_lds_void*)(dst + (wave * NU + u) * 256), 16, uv[u], S * u_stage_bytes, 0, 0);
#else
    (void)dst;
#endif
  };
  // ---- input transform: task (tile wt, channel quad, row i of B^T d B): 8 reads, 8 add/sub, 4 writes (float4) ------------------
  constexpr int NTASK = WT * 16 / WN_THREADS;                  // 1 (32 tiles) or 2 (64 tiles)
  float* const Park = Pbuf + NPB * G::P_FLOATS;
  float* const SBuf = Park + G::PARK;                          // [scale COUT_T][bias COUT_T]: read behind the stage loop

  if constexpr (DIL) {
    static_assert(!DIL || NTASK == 1, "one transform task per thread");
    const int quad = tid & 3, wt = (tid >> 2) % WT;
    const int tyl = wt / TW, txl = wt - tyl * TW;
    int pX, tX, pY, tY;
    ax.decode(tx0 + txl, pX, tX);
    ay.decode(ty0 + tyl, pY, tY);
    int rx = pX - px0, ry = pY - py0;                          // run of the tile inside the block (clamped: tiles beyond the grid read junk, masked later)
    rx = rx < G::RX ? rx : G::RX - 1; ry = ry < G::RY ? ry : G::RY - 1;
    *(__attribute__((address_space(3))) int*)(Park + tid) = ((2 * tyl + 2 * ry) * PW + 2 * txl + 2 * rx) * 16 + quad * 4;
  }
  auto transform = [&](const int kc) {
    const float* const src = Pbuf + (NPB == 2 ? (kc & 1) : 0) * G::P_FLOATS;
    float* const dst = Vbuf + (NVB == 2 ? (kc & 1) : 0) * G::V_FLOATS;
#pragma unroll
    for (int t = 0; t < NTASK; ++t) {
      // task = (row i of B^T d B, tile wt, channel quad): i is the same for a whole wave (its row pair and sign are scalars) and
      // consecutive lanes walk (quad, tile), so that the four V rows a lane writes — and the eight lanes of a ds_write_b128 group —
      // are 128 contiguous bytes (with i in the low lane bits the writes were 4-way bank conflicts: 33-40 % of the LDS cycles)
      const int task = tid + t * WN_THREADS;
      const int i = __builtin_amdgcn_readfirstlane(task / (WT * 4)), quad = task & 3, wt = (task >> 2) % WT;
      const int tyl = wt / TW, txl = wt - tyl * TW;
      // B^T rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
      const int r1 = (i == 0) ? 0 : (i == 2 ? 2 : 1), r2 = (i == 3) ? 3 : (i == 2 ? 1 : 2);
      const float sg = (i == 1) ? 1.f : -1.f;
      // the tile's first patch pixel: rows / columns 2 t for the plain form; DIL: 2 t + 2 (run of the tile), parked in LDS
      const int tp = DIL ? *(const __attribute__((address_space(3))) int*)(Park + tid)
                         : ((2 * tyl) * PW + 2 * txl + ((CAT && txl >= cn0) ? 2 : 0)) * 16 + quad * 4;      // CAT: the next image's run sits two columns further
      const float* const a = src + tp + r1 * PW * 16;
      const float* const bb = src + tp + r2 * PW * 16;
      float* const o = dst + ((i * 4) * WT + wt) * 16 + ((quad ^ ((wt >> 2) & 2)) << 2);
      typedef __attribute__((address_space(3))) f32x4 lds_f4w;
      if constexpr (MW == 2) {
        f32x4 w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = wn_lds_read128(a + c * 16) + sg * wn_lds_read128(bb + c * 16);
        *(lds_f4w*)(o) = w[0] - w[2];
        *(lds_f4w*)(o + WT * 16) = w[1] + w[2];
        *(lds_f4w*)(o + 2 * WT * 16) = w[2] - w[1];
        *(lds_f4w*)(o + 3 * WT * 16) = w[1] - w[3];
      } else {
        // the 128-register configuration: two columns at a time (24 live registers beside the 64 accumulators instead of 48; the
        // other workgroup of the CU covers the second LDS round trip)
        const f32x4 w0 = wn_lds_read128(a) + sg * wn_lds_read128(bb);
        const f32x4 w2 = wn_lds_read128(a + 32) + sg * wn_lds_read128(bb + 32);
        *(lds_f4w*)(o) = w0 - w2;
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 w1 = wn_lds_read128(a + 16) + sg * wn_lds_read128(bb + 16);
        *(lds_f4w*)(o + WT * 16) = w1 + w2;
        *(lds_f4w*)(o + 2 * WT * 16) = w2 - w1;
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 w3 = wn_lds_read128(a + 48) + sg * wn_lds_read128(bb + 48);
        *(lds_f4w*)(o + 3 * WT * 16) = w1 - w3;
      }
    }
  };
  // ---- fragments: wave (wm, wn) = 32 cout x 16 tiles; lane (j, g): row j of a 16-row fragment, K slot g --------------------------
  const int wm = wave % G::WMW, wn = wave / G::WMW;
  const int j = lane & 15, g = lane >> 4;
  int a_off[MW];
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    const int row = wm * 16 * MW + m * 16 + j;
    a_off[m] = row * 16 + ((g ^ ((row >> 2) & 2)) << 2);
  }
  const int wtl = wn * 16 + j;
  const int b_off = wtl * 16 + ((g ^ ((wtl >> 2) & 2)) << 2);
  f32x4 acc[16][MW];
  f32x4 fa[2][2][MW], fb[2][2];                                 // [set][position of the pair][m]
  auto read_frags = [&](const int S, const int set) {
    const float* const ub = Ubuf + (S % RING) * G::U_FLOATS;
    const int kc = S >> 3, st = S & 7;
    const float* const vb = Vbuf + (NVB == 2 ? (kc & 1) : 0) * G::V_FLOATS + (2 * st) * WT * 16;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int m = 0; m < MW; ++m) fa[set][p][m] = wn_lds_read128(ub + p * COUT_T * 16 + a_off[m]);
      fb[set][p] = wn_lds_read128(vb + p * WT * 16 + b_off);
    }
  };

  auto read_frag = [&](const int S, const int set, const int p) {      // one position of a stage (MW == 1)
    const float* const ub = Ubuf + (S % RING) * G::U_FLOATS;
    const int kc = S >> 3, st = S & 7;
    const float* const vb = Vbuf + (NVB == 2 ? (kc & 1) : 0) * G::V_FLOATS + (2 * st) * WT * 16;
    fa[set][p][0] = wn_lds_read128(ub + p * COUT_T * 16 + a_off[0]);
    fb[set][p] = wn_lds_read128(vb + p * WT * 16 + b_off);
  };
  // ---- prologue: patch 0, three U stages (the ring), transform 0, patch 1 ------------------------------------------------------------
  const int n_u0 = NS < RING ? NS : RING;
  float scv = 1.f;                                              // this thread's input-scale channel: requested in front of the DMAs (older: done when they are)
  if constexpr (SCALED)
    if (scaled && tid < c0) scv = P.in_scale[(size_t)img * c0 + tid];
  issue_patch(0);
  for (int S = 0; S < n_u0; ++S) issue_u(S);
  float sbv = tid < COUT_T ? 1.f : 0.f;                         // scale / bias of the cout block: requested behind the DMAs, parked in LDS below
  if (tid < 2 * COUT_T) {
    const int co = cout0 + (tid < COUT_T ? tid : tid - COUT_T);
    if (co < P.cout) {
      if (tid < COUT_T) { if (P.scale) sbv = P.scale[co]; }
      else if (P.bias) sbv = P.bias[(P.bias_per_img ? (size_t)img * P.cout : 0) + co];
    }
  }
  SF_STAMP_AT(L, 11);
  wn_wait(NU * n_u0);                                           // the patch is the oldest: everything but the U stages (a wave that issued the load above also waits for U(0))
  if constexpr (SCALED) {
    if (scaled) {
      if (tid < G::SC) SCbuf[tid] = scv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wn_barrier();                                             // the scales are in LDS
      scale_patch(0);                                           // this lane's own pieces (landed: wn_wait above)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  wn_barrier();
  SF_STAMP_AT(L, 12);
  transform(0);
  SF_STAMP_AT(L, 13);
  if (tid < 2 * COUT_T) SBuf[tid] = sbv;                        // published by the barrier below
  if (NPB == 2 && nkc > 1) issue_patch(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wn_wait(NU * (n_u0 - 1) + ((NPB == 2 && nkc > 1) ? NP : 0));  // U(0) landed (younger: U(1), U(2), patch(1))
  wn_barrier();
  if (!ROLL && NPB == 1 && nkc > 1) issue_patch(1);             // one patch buffer: every wave has finished transform(0)
  if constexpr (NVB == 2 || ROLL) {      // (the one-buffer, one-shot-transform loop starts its first chunk from a zero C operand instead)
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int m = 0; m < MW; ++m) acc[p][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  read_frags(0, 0);
  SF_STAMP_AT(L, 1);

  // (Measured and not kept, profiles/r04_[hij]_winobench_*: a persistent tile loop with the next tile's DMAs issued before the
  // epilogue; the SIMD-partner stagger of MI355X_MICROARCH.md item 9 as a second code path and as a deferred DMA issue.  Each cost
  // more in the compiler's schedule of this loop — 227 -> 239-254 registers, scalar spills in the stage code — than it bought: 14.6 ms
  // for the 128 -> 128 layer on 224 frames with the loop below, 15.3-16.9 ms with them.)
  // U stages younger than U(S+1) when stage S = (chunk, st) waits for it: RING - 2 of them, fewer at the end of the last chunk
  auto young = [](const int st, const bool more) { return more ? RING - 2 : (6 - st < RING - 2 ? (6 - st < 0 ? 0 : 6 - st) : RING - 2); };
  if constexpr (NVB == 2) {
    // Conditions are written in (kc, st) so that they fold for st < 5 / 6 / 7 after unrolling: with a run-time branch between a
    // stage's fragment reads and its MFMAs hipcc's wait-count pass loses track of which LDS reads are pending at the merge and
    // puts s_waitcnt lgkmcnt(0) in front of the MFMAs — they then wait for the NEXT stage's reads (seen in the ISA of the first
    // version: 51 % of the MFMA peak).
    for (int kc = 0; kc < nkc; ++kc) {
      const bool more = kc + 1 < nkc;                               // wave-uniform
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int S = kc * 8 + st;
        const int set = st & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // fragments of stage S are in registers (and this wave's V writes are out)
        if (st < 7 || more) {
          // U(S+1) landed: younger DMAs of this wave are U(S+2) and — for two stages behind its issue point — the next patch
          const bool patch_young = (st == 4 || st == 5) && (kc + 2 < nkc);
          wn_wait(NU * young(st, more) + (patch_young ? NP : 0));
          wn_barrier();                                             // stage S+1 (and, at st == 7, the next chunk's V) published; buffer S % RING free
          if (st < 8 - RING || more) issue_u(S + RING);
          read_frags(S + 1, set ^ 1);
        }
        // the next stage's fragment reads stay IN FRONT of this stage's MFMAs (the machine scheduler otherwise sinks them behind the
        // MFMAs to save registers and every MFMA group then waits for an LDS round trip)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < MW; ++m)
              acc[2 * st + p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][p][m][e], fb[set][p][e], acc[2 * st + p][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (st == 3 && more) {                                      // next chunk's transform beside this chunk's stages (its patch landed a chunk ago)
          transform(kc + 1);
          if (kc + 2 < nkc) issue_patch(kc + 2);                    // into the buffer transform(kc) read: every wave passed >= 4 barriers since
        }
      }
    }
  } else if constexpr (ROLL) {
    // ROLLING INPUT TRANSFORM.  With the whole transform of chunk kc+1 at the chunk boundary the matrix pipe of this workgroup drains
    // there — barrier, 3 dependent LDS round trips, barrier, and the first fragment reads of the new chunk with nothing to cover them:
    // in-kernel stamps showed a workgroup's stage loop at 52 % of the MFMA rate while its partner was in its prologue / epilogue
    // (profiles/r04_x_stamps_wino.txt).  V[4 i .. 4 i + 3] (row i of B^T d B) is read by stages 2 i and 2 i + 1 only, so the rows of the
    // next chunk are written IN PLACE as they fall free, one row per stage, each thread one (tile, channel quad, output column):
    //     row 0 in stage 5, row 1 in stage 6, row 2 in stage 7 of chunk kc, row 3 in stage 0 of chunk kc + 1
    // (4 LDS reads in front of the stage's MFMAs, 6 packed adds and 1 LDS write behind them).  The single patch buffer is read in
    // those four stages; the next patch is requested in stage 1 and is older than every U stage that stage 5 still has in flight,
    // so the stage's own vmcnt wait + barrier publish it.  Same sums in the same order as the one-shot transform: bitwise equal.
    const int tj = wave >> 1;                                         // output column of this wave's transform tasks
    const int cA = tj == 0 ? 0 : (tj == 2 ? 2 : 1), cB = tj == 0 ? 2 : (tj == 1 ? 2 : (tj == 2 ? 1 : 3));     // columns w[cA] +- w[cB]
    const float sB = tj == 1 ? 1.f : -1.f;
    // the lane's two offsets (patch pixel of its tile / its V row) live in two registers through the loop
    int tp_l, vo_l;
    {
      const int quad = tid & 3, wt = (tid >> 2) & (WT - 1);
      tp_l = ((2 * (wt / TW)) * PW + 2 * (wt % TW)) * 16 + quad * 4;
      vo_l = (tj * WT + wt) * 16 + ((quad ^ ((wt >> 2) & 2)) << 2);
    }
    // half h of a task: column cA (h = 0) or cB (h = 1) of patch rows r1, r2 -> w = d[r1] +- d[r2]
    auto trow_read = [&](const int i, const int h, f32x4 (&r)[2]) {
      const int r1 = (i == 0) ? 0 : (i == 2 ? 2 : 1), r2 = (i == 3) ? 3 : (i == 2 ? 1 : 2);
      const float* const pc = Pbuf + tp_l + (h ? cB : cA) * 16;
      r[0] = wn_lds_read128(pc + r1 * PW * 16); r[1] = wn_lds_read128(pc + r2 * PW * 16);
    };
    auto trow_w = [&](const int i, const f32x4 (&r)[2]) { return (i == 1) ? r[0] + r[1] : r[0] - r[1]; };    // B^T rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    auto trow_write = [&](const int i, const f32x4 wA, const f32x4 wB) {
      typedef __attribute__((address_space(3))) f32x4 lds_f4w;
      *(lds_f4w*)(Vbuf + i * 4 * WT * 16 + vo_l) = wA + sB * wB;
    };
    for (int kc = 0; kc < nkc; ++kc) {
      const bool more = kc + 1 < nkc;                               // wave-uniform
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int S = kc * 8 + st;
        const int set = st & 1;
        // a transform row rides in this stage (stage 0 of chunk 0 rewrites row 3 with the values the prologue wrote: no special case)
        const bool t_now = st == 0 || (st >= 5 && more);
        const int ti = st == 0 ? 3 : st - 5;
        f32x4 tr[2], wA;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // fragments of stage S in registers, this wave's V row written
        if (st < 7 || more) {
          const bool patch_young = more && st >= 2 && st <= RING;   // requested in stage 1: behind U(S+1) for these stages, in front of it later
          wn_wait(NU * young(st, more) + (patch_young ? npw : 0));
          wn_barrier();                                             // U(S+1) and the V rows written in stage S-1 published; buffer S % RING, and the
                                                                    // patch after stage 0, free
          if (st < 8 - RING || more) issue_u(S + RING);
          if (st == 1 && more) issue_patch(kc + 1);
          read_frag(S + 1, set ^ 1, 0);
          if (t_now) trow_read(ti, 0, tr);
        }
        // The stage's MFMAs in two halves.  The next stage's fragments are read one position ahead of each half (24 fragment
        // registers live instead of 32), the transform task's second pair of reads and its arithmetic sit between / behind the
        // halves (8 + 4 registers instead of 16 + 4): the kernel has 128 registers for two workgroups per CU.
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[2 * st + p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][p][0][e], fb[set][p][e], acc[2 * st + p][0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (st < 7 || more) {
            if (p == 0) read_frag(S + 1, set ^ 1, 1);
            if (t_now) {
              if (p == 0) { wA = trow_w(ti, tr); trow_read(ti, 1, tr); }
              else trow_write(ti, wA, trow_w(ti, tr));
            }
          }
        }
      }
    }
  } else {
    // one V buffer: the next chunk's transform sits between the last read of this chunk's V (behind the barrier of stage 7) and the
    // first read of the next one; stage 7's MFMAs cover its LDS writes.  The first chunk is its own copy of the body: its MFMAs start
    // from a zero C operand — no accumulator initialisation (64 vector moves per wave and tile; vector instructions cost matrix
    // time: see the epilogue)
    auto chunk = [&](const int kc, auto first_c) {
      constexpr bool first = decltype(first_c)::value;
      const bool more = kc + 1 < nkc;                               // wave-uniform
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int S = kc * 8 + st;
        const int set = st & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (st < 7) {                                               // inside a chunk: as the two-buffer kernel
          const bool patch_young = st <= 1 && more;                 // the patch is issued at the chunk boundary: young for two stages
          wn_wait(NU * young(st, more) + (patch_young ? npw : 0));
          wn_barrier();
          if (st < 8 - RING || more) issue_u(S + RING);
          read_frags(S + 1, set ^ 1);
        } else if (more) {                                          // chunk boundary
          wn_wait(NU * (RING - 2));
          wn_barrier();                                             // every wave holds this chunk's last fragments: V is free
          issue_u(S + RING);
          transform(kc + 1);
          if (NPB == 2 && kc + 2 < nkc) issue_patch(kc + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < MW; ++m) {
              const f32x4 cin = (first && e == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[2 * st + p][m];
              acc[2 * st + p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][p][m][e], fb[set][p][e], cin, 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (st == 3 && more) scale_patch(kc + 1);                   // its DMAs landed at stage 2's wait; published by the barriers of stages 4 .. 7
        if (st == 7 && more) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          wn_barrier();                                             // the next chunk's V is visible (and every wave is done with the patch)
          if (NPB == 1 && kc + 2 < nkc) issue_patch(kc + 2);
          read_frags(S + 1, set ^ 1);
        }
      }
    };
    chunk(0, std::true_type{});
    for (int kc = 1; kc < nkc; ++kc) chunk(kc, std::false_type{});
  }

  SF_STAMP_AT(L, 2);
  // ---- output transform A^T M A (register-local) + epilogue ---------------------------------------------------------------------------
  // the lane's tile and channel quad are re-derived from a laundered lane index: nothing of them is carried through the stage loop
  // (two registers the 128-register configuration does not have)
  int ln_e = lane;
  asm volatile("" : "+v"(ln_e));
  const int wtl_e = wn * 16 + (ln_e & 15), g_e = ln_e >> 4;
  const int txl_e = wtl_e % TW;
  const bool run_e = CAT && txl_e >= cn0;                      // CAT: the lane's tile belongs to the next image
  const int ty = ty0 + wtl_e / TW, tx = run_e ? txl_e - cn0 : tx0 + txl_e;
  const size_t img_base = (size_t)img * H * W;
  // output pixel (2 t + i) of the tile; DIL: pixel 2 t + i of the tile's phase = phase + d (2 t + i)
  int oy0 = 2 * ty, ox0 = 2 * tx, ostep = 1;
  if constexpr (DIL) {
    int pX, tX, pY, tY;
    ax.decode(tx, pX, tX);
    ay.decode(ty, pY, tY);
    ostep = P.dil;
    ox0 = (tx < ax.nt && pX < ostep) ? pX + ostep * 2 * tX : W;     // beyond the tile list: masked by the range checks below
    oy0 = (ty < ay.nt && pY < ostep) ? pY + ostep * 2 * tY : H;
  }
  // uniform bases of the lane-independent part: tensor + (image, pixel (bq, i) of a tile at the image's origin)
  const bool affine = EPI == EPI_AFFINE;
  const float* const t_a = affine ? P.add : P.e0;
  const float* const t_b = P.e1;
  const int cs_a = affine ? P.add_cs : P.e0_cs, cs_b = P.e1_cs;
  WnPix px[2][2];
#pragma unroll
  for (int bq = 0; bq < 2; ++bq)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t p0 = img_base + (size_t)(bq * ostep) + (size_t)(i * ostep) * W;
      px[bq][i].ta = t_a ? t_a + p0 * cs_a : nullptr;
      px[bq][i].tb = t_b ? t_b + p0 * cs_b : nullptr;
      px[bq][i].out = P.out + p0 * P.out_cs + P.out_co;
      px[bq][i].out2 = P.out2 ? P.out2 + p0 * P.out2_cs : nullptr;
    }
  const bool img_ok = !CAT || img + (run_e ? 1 : 0) < P.n_img;
  const bool x0 = img_ok && ox0 < W, x1 = img_ok && ox0 + ostep < W, y0ok = oy0 < H, y1ok = oy0 + ostep < H;
  const unsigned pix = (x0 && y0ok) ? (unsigned)(oy0 * W + ox0 + (run_e ? H * W : 0)) : 0u;      // lanes without a tile compute on pixel 0 and store nothing
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    const int cl = wm * 16 * MW + m * 16 + 4 * g_e;            // channel inside the workgroup's cout block
    const int c = cout0 + cl;
    const bool c_ok = c < P.cout;
    const int c_ld = c_ok ? c : 0;
    const bool ok00 = c_ok && x0 && y0ok, ok01 = c_ok && x0 && y1ok, ok10 = c_ok && x1 && y0ok, ok11 = c_ok && x1 && y1ok;
    WnLane ln;
    ln.ea = pix * (unsigned)cs_a + (unsigned)c_ld;
    ln.eb = pix * (unsigned)cs_b + (unsigned)((affine && c_ld >= P.gate_from) ? c_ld - P.gate_from : (affine ? 0 : c_ld));
    ln.eo = pix * (unsigned)P.out_cs + (unsigned)c_ld;
    ln.eo2 = pix * (unsigned)P.out2_cs + (unsigned)(c_ld >= P.gate_from ? c_ld - P.gate_from : 0);
    const float4 sc = *(const float4*)(SBuf + cl), bi = *(const float4*)(SBuf + COUT_T + cl);
    float4 as = make_float4(1.f, 1.f, 1.f, 1.f);
    if (affine && P.add && P.add_scale) as = spm_ld4(P.add_scale + (size_t)img * P.cout + c_ld);
    // one output column bq at a time:  t[i] = (M A)[i][bq]:  bq = 0: M[i][0] + M[i][1] + M[i][2],   bq = 1: M[i][1] - M[i][2] - M[i][3]
    //                                  Y[0][bq] = t[0] + t[1] + t[2],   Y[1][bq] = t[1] - t[2] - t[3]
    // column 0's operands are requested before its transform, column 1's before column 0's arithmetic and stores
    WnOps o00, o01, o10, o11;
    o00.a = o00.b = o01.a = o01.b = o10.a = o10.b = o11.a = o11.b = spm_zero4();
    if (ok00) o00 = wn_epi_load<EPI>(P, px[0][0], ln);
    if (ok01) o01 = wn_epi_load<EPI>(P, px[0][1], ln);
    __builtin_amdgcn_sched_barrier(0);
    f32x2 ya[2], yb[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      f32x2 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 a0 = hh ? wn_hi(acc[4 * i + 0][m]) : wn_lo(acc[4 * i + 0][m]), a1 = hh ? wn_hi(acc[4 * i + 1][m]) : wn_lo(acc[4 * i + 1][m]);
        const f32x2 a2 = hh ? wn_hi(acc[4 * i + 2][m]) : wn_lo(acc[4 * i + 2][m]);
        t[i] = a0 + a1 + a2;
      }
      ya[hh] = t[0] + t[1] + t[2]; yb[hh] = wn_sub2(t[1], t[2] + t[3]);
    }
    if (ok10) o10 = wn_epi_load<EPI>(P, px[1][0], ln);
    if (ok11) o11 = wn_epi_load<EPI>(P, px[1][1], ln);
    __builtin_amdgcn_sched_barrier(0);
    SF_STAMP_AT(L, 5);
    WnOut r00, r01, r10, r11;
    r00 = wn_epi_finish<EPI>(P, ya[0], ya[1], o00, sc, bi, as, c);
    r01 = wn_epi_finish<EPI>(P, yb[0], yb[1], o01, sc, bi, as, c);
    SF_STAMP_AT(L, 6);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      f32x2 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 a1 = hh ? wn_hi(acc[4 * i + 1][m]) : wn_lo(acc[4 * i + 1][m]), a2 = hh ? wn_hi(acc[4 * i + 2][m]) : wn_lo(acc[4 * i + 2][m]);
        const f32x2 a3 = hh ? wn_hi(acc[4 * i + 3][m]) : wn_lo(acc[4 * i + 3][m]);
        t[i] = wn_sub2(a1, a2 + a3);
      }
      ya[hh] = t[0] + t[1] + t[2]; yb[hh] = wn_sub2(t[1], t[2] + t[3]);
    }
    SF_STAMP_AT(L, 7);
    r10 = wn_epi_finish<EPI>(P, ya[0], ya[1], o10, sc, bi, as, c);
    r11 = wn_epi_finish<EPI>(P, yb[0], yb[1], o11, sc, bi, as, c);
    // all stores of the tile back to back: no register of a store in flight is written again
    __builtin_amdgcn_sched_barrier(0);
    if (ok00) spm_st4(px[0][0].out + (size_t)ln.eo, r00.y);
    if (ok01) spm_st4(px[0][1].out + (size_t)ln.eo, r01.y);
    if (ok10) spm_st4(px[1][0].out + (size_t)ln.eo, r10.y);
    if (ok11) spm_st4(px[1][1].out + (size_t)ln.eo, r11.y);
    if (affine && P.out2 && c >= P.gate_from) {
      if (ok00) spm_st4(px[0][0].out2 + (size_t)ln.eo2, r00.y2);
      if (ok01) spm_st4(px[0][1].out2 + (size_t)ln.eo2, r01.y2);
      if (ok10) spm_st4(px[1][0].out2 + (size_t)ln.eo2, r10.y2);
      if (ok11) spm_st4(px[1][1].out2 + (size_t)ln.eo2, r11.y2);
    }
  }
  SF_STAMP_AT(L, 3);
#ifdef SF_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SF_STAMP_AT(L, 4);
#endif
}


// =====================================================================================================================================
// Round 5: the same arithmetic (bitwise: same products, same summation orders) on a different decomposition.
//
// What bounded conv_wino_kernel (tools/r05/mfma_cost.hip, profiles/r05_a_mfma_cost.txt): the fp32 MFMA and every other vector-side
// instruction share one issue port, and at two waves per SIMD and workgroup a 1-KB LDS-DMA piece costs ~20 cycles of matrix time, a
// ds_read_b128 ~7, a vector instruction ~5.5, a global load into registers ~4.  Per tile of a 64 -> 64 layer the old kernel spends 8192
// cycles per wave in MFMAs, ~2800 in 520 vector instructions of prologue / epilogue, ~1700 in 40 DMA pieces + 128 fragment reads, ~1100 in
// the input transform: 0.59 of the peak by the issue port alone (measured 0.57-0.58).  Here:
//   * wave (i, h) owns positions 4 i .. 4 i + 3 (row i of B^T d B) of 32 cout x 32 tiles: 2 A + 2 B fragments feed 16 MFMAs (1 + 1 feed 4
//     in the old kernel: half the LDS reads), and row i is exactly what the two waves (i, 0), (i, 1) write in the input transform;
//   * U never touches LDS: every lane loads its A fragments straight into registers (buffer_load_dwordx4, 1 KB contiguous per
//     fragment, two steps ahead, the next-but-one fragment into the registers the current half-step has just released) — no U ring, no
//     per-stage barrier (two barriers per 16-channel chunk instead of nine), and the freed LDS double-buffers V;
//   * the output transform is split: T[i][b] = (M A)[i][b] in registers (4 positions -> 2 values), exchanged through LDS (64 KB over the
//     V buffers, a swizzle that is conflict-free for the writers' and the readers' lane groups), Y = A^T T by the thread that stores it:
//     lane = (tile, channel quad) with the 16 channel quads of a pixel in consecutive lanes — 256 contiguous bytes per pixel and store;
//   * prologue index arithmetic cut to what is per-lane by nature.
// Mode A (plain / concatenated images): V double-buffered, one patch buffer — barriers per chunk: B (chunk start: V(kc) published, patch
// free) and M (patch(kc+1) landed, before the transform that runs between steps 1 and 2).  Mode B (dilated: its patch is 18 KB): one V, two
// patch buffers, the transform at the chunk boundary between two barriers.

// packed subtraction a - b (v_pk_add_f32 with the second operand negated: hipcc turns fma(b, -1, a) and a - b on vectors into four
// scalar v_sub_f32; every vector-side instruction costs ~5.5 cycles of matrix time beside the fp32 MFMAs)
__device__ __forceinline__ f32x2 wn5_sub2(const f32x2 a, const f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x4 wn5_sub4(const f32x4 a, const f32x4 b) {
  const f32x2 lo = wn5_sub2(wn_lo(a), wn_lo(b)), hi = wn5_sub2(wn_hi(a), wn_hi(b));
  return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 wn5_undef4() {      // a register nobody has to initialise (lanes that do not load do not store either)
  f32x4 v;
  return __builtin_nondeterministic_value(v);
}
__device__ __forceinline__ f32x4 wn5_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void wn5_st4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// activation of the tile's 16 values under ONE uniform switch (per element the switch costs a chain of scalar branches and copies)
__device__ __forceinline__ void wn5_act16(f32x4 (&y)[4], const int act) {
  switch (act) {
    case ACT_RELU:
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = __builtin_elementwise_max(y[k], (f32x4){0.f, 0.f, 0.f, 0.f});
      break;
    case ACT_LRELU:
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) y[k][q] = y[k][q] > 0.f ? y[k][q] : 0.1f * y[k][q];
      break;
    case ACT_NONE: break;
    default:
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) y[k][q] = spm_act(y[k][q], act);
  }
}
template <bool DIL, bool CAT>
struct Wino5Geo {
  static_assert(!(DIL && CAT), "one run structure at a time");
  static constexpr int COUT_T = 64, TH = 4, TW = 8, WT = TH * TW;
  static constexpr int RY = 2, RX = 4;
  static constexpr int PH = 2 * TH + (DIL ? 2 * RY : 2), PW = 2 * TW + (DIL ? 2 * RX : (CAT ? 4 : 2));
  static constexpr int NPX = PH * PW;
  static constexpr int ND = (NPX * 4 + 63) / 64;                // 1-KB DMA pieces of a patch; wave w issues pieces w, w + 8, ...
  static constexpr int NP = (ND + 7) / 8;
  static constexpr int P_FLOATS = ND * 256;
  static constexpr int V_FLOATS = 16 * WT * 16;
  static constexpr int NVB = DIL ? 1 : 2, NPB = DIL ? 2 : 1;
  static constexpr int PARK = DIL ? 512 : 0;                    // DIL: the patch offset of every transform task
  static constexpr int SB = 2 * COUT_T, SC = DIL ? 0 : 256;
  static constexpr int T_FLOATS = 4 * 2 * WT * COUT_T;          // the exchange of the output transform: [row i][b][tile][cout]
  static_assert(NVB * V_FLOATS + NPB * P_FLOATS >= T_FLOATS, "the exchange lives over V and the patch");
  static constexpr int LDS_FLOATS = NVB * V_FLOATS + NPB * P_FLOATS + PARK + SB + SC;
  static_assert(2 * LDS_FLOATS * 4 <= 160 * 1024, "two workgroups per CU");
};

// timing-only ablations (tools/r05/ablate.sh; wrong results by construction): -DSF_W5_ABL=<bits>  1: no input transform in the loop,
// 2: no patch DMAs in the loop, 4: no A loads in the loop, 8: no barriers in the loop, 16: no MFMAs, 32: minimal epilogue
#if !defined(SF_W5_ABL)
#define SF_W5_ABL 0
#endif
template <int EPI, bool DIL = false, bool CAT = false>
__global__ __launch_bounds__(WN_THREADS, 4) void conv_wino5_kernel(const ConvLaunch L) {
  typedef Wino5Geo<DIL, CAT> G;
  constexpr int COUT_T = G::COUT_T, TH = G::TH, TW = G::TW, WT = G::WT, PW = G::PW, NP = G::NP;
  constexpr bool MODE_A = G::NVB == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Vbuf = smem;
  float* const Pbuf = Vbuf + G::NVB * G::V_FLOATS;
  float* const Park = Pbuf + G::NPB * G::P_FLOATS;
  float* const SBuf = Park + G::PARK;                          // [scale COUT_T][bias COUT_T]
  float* const SCbuf = SBuf + G::SB;                           // [c0] input scales (SCALED)
  const ConvProblem& P = L.p[0];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = P.Hout, W = P.Wout;
  const WnAxis ax(DIL ? W : 2, DIL ? P.dil : 1), ay(DIL ? H : 2, DIL ? P.dil : 1);
  const int tiles_x = DIL ? ax.nt : (W + 1) >> 1, tiles_y = DIL ? ay.nt : (H + 1) >> 1;
  const int tpi = tiles_x;
  const int nbx = ((CAT ? P.n_img * tiles_x : tiles_x) + TW - 1) / TW, nby = (tiles_y + TH - 1) / TH;
  // 1-D grid, XCD-aware (as conv_wino_kernel): every XCD owns a contiguous range of tile blocks, the cout blocks of a tile block are
  // consecutive workgroups of that XCD
  const int ncb = P.cout_pad / COUT_T;
  const int nblk = nbx * nby * (CAT ? 1 : P.n_img), per_xcd = (nblk + 7) >> 3;
  const int lin = (int)blockIdx.x, xcd = lin & 7, slot_ = lin >> 3;
  const int tb_ = slot_ / ncb;
  int b = xcd * per_xcd + tb_;
  if (b >= nblk) return;
  const int bx = b % nbx; b /= nbx;
  const int by = b % nby;
  const int img = CAT ? (bx * TW) / tpi : b / nby;
  const int ct0 = CAT ? bx * TW - img * tpi : 0;
  const int cn0 = CAT ? (tpi - ct0 < TW ? tpi - ct0 : TW) : TW;
  const int ty0 = by * TH, tx0 = CAT ? ct0 : bx * TW;
  int px0 = 0, pt0 = 0, py0 = 0, qt0 = 0;
  if constexpr (DIL) { ax.decode(tx0, px0, pt0); ay.decode(ty0, py0, qt0); }
  SF_STAMP_AT(L, 0);
#ifdef SF_STAMP
  SF_STAMP_VAL(L, 8, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));
  SF_STAMP_VAL(L, 9, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));
  SF_STAMP_VAL(L, 10, (unsigned long long)nkc_stamp(P));
#endif
  const int cout0 = (slot_ - tb_ * ncb) * COUT_T;
  const int nkc = P.cin_pad >> 4;                              // 16-channel chunks (>= 2: cin_pad is a multiple of 32)
  const int NS = nkc * 4;                                      // steps: (chunk, position of the wave's row)
  const int c0 = P.c0;
  const int img_px_i = P.Hin * P.Win;
  const int up = DIL ? 0 : P.in_up;
  const int Win = P.Win;
#if defined(__HIP_DEVICE_COMPILE__)
  auto make_rsrc = [](const float* base, size_t bytes) {
    const unsigned nrec = bytes < 0x7fffffffull ? (unsigned)bytes : 0x7fffffffu;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (int)nrec, 0x00020000);
  };
  const size_t img_px = (size_t)P.Hin * P.Win;
  const size_t n_in = (CAT && img + 1 < P.n_img) ? 2 : 1;
  const __amdgpu_buffer_rsrc_t rsrc0 = make_rsrc(P.in0 + (size_t)img * img_px * P.in0_cs, n_in * img_px * P.in0_cs * sizeof(float));
  const __amdgpu_buffer_rsrc_t rsrc1 = make_rsrc(P.in1 ? P.in1 + (size_t)img * img_px * P.in1_cs : P.in0, P.in1 ? n_in * img_px * P.in1_cs * sizeof(float) : 0);
  const __amdgpu_buffer_rsrc_t rsrc_u = make_rsrc(P.w_wino, (size_t)nkc * 16 * P.cout_pad * 16 * sizeof(float));
#endif
  constexpr bool SCALED = G::SC > 0 && EPI == EPI_AFFINE;
  const bool scaled = SCALED && P.in_scale != nullptr;
  // ---- patch DMA: element e = (pixel, channel quad) of the patch, 16 bytes each, LDS linear in e; piece idx = d * 8 + wave --------------
  SF_STAMP_AT(L, 14);
  int pv0[NP], pv1[DIL ? 1 : NP];
  const int npw = G::ND / 8 + (wave < G::ND % 8 ? 1 : 0);      // wave-uniform
#pragma unroll
  for (int d = 0; d < NP; ++d) {
    const int e = (d * 8 + wave) * 64 + lane;
    const int pix = e >> 2, quad = e & 3;
    const int py = pix / PW, px = pix - py * PW;
    int iy = 2 * ty0 - 1 + py, ix = 2 * tx0 - 1 + px;
    int run = 0;
    if constexpr (CAT) {
      run = px >= 2 * cn0 + 2 ? 1 : 0;
      ix = run ? px - (2 * cn0 + 2) - 1 : ix;
    }
    bool ok = pix < G::NPX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && (!CAT || (img + run < P.n_img && (run == 0 || cn0 < TW)));
    if constexpr (DIL) {
      const bool oky = ay.patch_coord(py, py0, qt0, TH, H, iy), okx = ax.patch_coord(px, px0, pt0, TW, W, ix);
      ok = pix < G::NPX && oky && okx;
    }
    const int pofs = DIL ? iy * W + ix : (iy >> up) * Win + (ix >> up) + (CAT ? run * img_px_i : 0);
    pv0[d] = ok ? (pofs * P.in0_cs + quad * 4) * 4 : (int)0x80000000;
    if constexpr (!DIL) pv1[d] = ok ? (pofs * P.in1_cs + quad * 4) * 4 : (int)0x80000000;
  }
  SF_STAMP_AT(L, 15);
  auto issue_patch = [&](const int kc) {
    float* const dst = Pbuf + (G::NPB == 2 ? (kc & 1) : 0) * G::P_FLOATS;
    const bool from1 = !DIL && kc * 16 >= c0;                   // wave-uniform: the whole chunk reads in1 (c0 % 16 == 0)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int d = 0; d < NP; ++d) {
      if (d >= npw) continue;
      float* const dB = dst + (d * 8 + wave) * 256;
      // the chunk's channel offset rides in the scalar operand (not part of the range check: an invalid lane stays out of range);
      // the two inputs are two branches (the asm comments keep hipcc from merging them into lane selects: vector instructions)
      if constexpr (DIL) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (wn_lds_void*)dB, 16, pv0[d], kc * 64, 0, 0);
      } else {
        if (from1) {
          asm volatile("; patch from in1");
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (wn_lds_void*)dB, 16, pv1[d], (kc * 16 - c0) * 4, 0, 0);
        } else {
          asm volatile("; patch from in0");
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (wn_lds_void*)dB, 16, pv0[d], kc * 64, 0, 0);
        }
      }
    }
#else
    (void)dst; (void)from1;
#endif
  };
  auto scale_patch = [&](const int kc) {
    if constexpr (SCALED) {
      if (scaled && kc * 16 < c0) {
        const f32x4 s4 = wn_lds_read128(SCbuf + kc * 16 + (lane & 3) * 4);
        typedef __attribute__((address_space(3))) f32x4 lds_f4w;
#pragma unroll
        for (int d = 0; d < NP; ++d) {
          if (d >= npw) continue;
          float* const q = Pbuf + ((d * 8 + wave) * 64 + lane) * 4;
          *(lds_f4w*)q = wn_lds_read128(q) * s4;
        }
      }
    }
  };
  // ---- roles: wave (ih, hh) = row ih of B^T d B (positions 4 ih .. 4 ih + 3), cout half hh; lane (j, g) of a 16-row fragment -------------
  const int ih = wave >> 1, hh = wave & 1;
  const int j = lane & 15, g = lane >> 4;
  // A fragments (U[chunk][position][cout_pad][16]) straight from global memory: 16 rows x 64 B = 1 KB contiguous per fragment
  const int u_voff = ((cout0 + hh * 32 + j) * 16 + g * 4) * 4;
  const int u_pos_bytes = P.cout_pad * 64;
  auto load_A = [&](const int s, const int mb) -> f32x4 {
#if defined(__HIP_DEVICE_COMPILE__)
    const int sc = s < NS ? s : NS - 1;                         // the tail re-loads the last step (no branch in the MFMA stream)
    const int so = (((sc >> 2) * 16 + ih * 4 + (sc & 3)) * u_pos_bytes);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, u_voff + mb * 1024, so, 0));
#else
    (void)s; (void)mb; return (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
  };
  // B fragments: V[position][tile][16], slot swizzle as conv_wino_kernel
  const int b_off = (ih * 4 * WT + j) * 16 + ((g ^ ((j >> 2) & 2)) << 2);
  // ---- input transform: task (row ih, tile wt, channel quad): 8 reads, 8 add / sub, 4 writes (float4) -------------------------------------
  if constexpr (DIL) {
    const int quad = tid & 3, wt = (tid >> 2) % WT;
    const int tyl = wt / TW, txl = wt - tyl * TW;
    int pX, tX, pY, tY;
    ax.decode(tx0 + txl, pX, tX);
    ay.decode(ty0 + tyl, pY, tY);
    int rx = pX - px0, ry = pY - py0;
    rx = rx < G::RX ? rx : G::RX - 1; ry = ry < G::RY ? ry : G::RY - 1;
    *(__attribute__((address_space(3))) int*)(Park + tid) = ((2 * tyl + 2 * ry) * PW + 2 * txl + 2 * rx) * 16 + quad * 4;
  }
  auto transform = [&](const int kc) {
    const float* const src = Pbuf + (G::NPB == 2 ? (kc & 1) : 0) * G::P_FLOATS;
    float* const dst = Vbuf + (G::NVB == 2 ? (kc & 1) : 0) * G::V_FLOATS;
    const int quad = lane & 3, wt = (tid >> 2) & (WT - 1);
    const int tyl = wt / TW, txl = wt - tyl * TW;
    // B^T rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int r1 = (ih == 0) ? 0 : (ih == 2 ? 2 : 1), r2 = (ih == 3) ? 3 : (ih == 2 ? 1 : 2);
    const float sg = (ih == 1) ? 1.f : -1.f;
    const int tp = DIL ? *(const __attribute__((address_space(3))) int*)(Park + tid)
                       : ((2 * tyl) * PW + 2 * txl + ((CAT && txl >= cn0) ? 2 : 0)) * 16 + quad * 4;
    const float* const a = src + tp + r1 * PW * 16;
    const float* const bb = src + tp + r2 * PW * 16;
    float* const o = dst + ((ih * 4) * WT + wt) * 16 + ((quad ^ ((wt >> 2) & 2)) << 2);
    typedef __attribute__((address_space(3))) f32x4 lds_f4w;
    const f32x4 w0 = wn_lds_read128(a) + sg * wn_lds_read128(bb);
    const f32x4 w2 = wn_lds_read128(a + 32) + sg * wn_lds_read128(bb + 32);
    *(lds_f4w*)(o) = wn5_sub4(w0, w2);
    const f32x4 w1 = wn_lds_read128(a + 16) + sg * wn_lds_read128(bb + 16);
    *(lds_f4w*)(o + WT * 16) = w1 + w2;
    *(lds_f4w*)(o + 2 * WT * 16) = wn5_sub4(w2, w1);
    const f32x4 w3 = wn_lds_read128(a + 48) + sg * wn_lds_read128(bb + 48);
    *(lds_f4w*)(o + 3 * WT * 16) = wn5_sub4(w1, w3);
  };
  // ---- prologue -----------------------------------------------------------------------------------------------------------------------------
  float scv = 1.f;
  if constexpr (SCALED)
    if (scaled && tid < c0) scv = P.in_scale[(size_t)img * c0 + tid];
  issue_patch(0);
  f32x4 A[2][2];                                                // [ring slot = step & 1][mb]
  A[0][0] = load_A(0, 0); A[0][1] = load_A(0, 1);
  A[1][0] = load_A(1, 0); A[1][1] = load_A(1, 1);
  float sbv = tid < COUT_T ? 1.f : 0.f;
  if (tid < 2 * COUT_T) {
    const int co = cout0 + (tid < COUT_T ? tid : tid - COUT_T);
    if (co < P.cout) {
      if (tid < COUT_T) { if (P.scale) sbv = P.scale[co]; }
      else if (P.bias) sbv = P.bias[(P.bias_per_img ? (size_t)img * P.cout : 0) + co];
    }
  }
  SF_STAMP_AT(L, 11);
  wn_wait(4);                                                   // the patch is older than the four A loads (a wave that loaded scale / bias waits for one of them too)
  if constexpr (SCALED) {
    if (scaled) {
      if (tid < G::SC) SCbuf[tid] = scv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wn_barrier();
      scale_patch(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  wn_barrier();                                                 // patch(0) complete
  SF_STAMP_AT(L, 12);
  transform(0);
  SF_STAMP_AT(L, 13);
  if (tid < 2 * COUT_T) SBuf[tid] = sbv;
  if (!MODE_A) issue_patch(1);                                  // second patch buffer
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wn_barrier();                                                 // V(0) published; Mode A: the patch buffer is free
  SF_STAMP_AT(L, 1);

  f32x4 acc[4][2][2];
  auto step = [&](const int kc, auto p_c, auto first_c) {
    constexpr int p = decltype(p_c)::value, slot = p & 1;
    constexpr bool first = decltype(first_c)::value;
    const float* const vb = Vbuf + (G::NVB == 2 ? (kc & 1) : 0) * G::V_FLOATS + p * WT * 16 + b_off;
    f32x4 Bf[2];
    Bf[0] = wn_lds_read128(vb);
    Bf[1] = wn_lds_read128(vb + 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const f32x4 cin = (first && e == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[p][mb][nb];
          if (SF_W5_ABL & 16) acc[p][mb][nb] = (e == 0 && nb == 0 && mb == 0) ? cin + A[slot][mb] * Bf[nb] : cin;
          else acc[p][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[slot][mb][e], Bf[nb][e], cin, 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (!(SF_W5_ABL & 4)) A[slot][mb] = load_A(kc * 4 + p + 2, mb);                 // into the registers this half-step has released
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto chunk = [&](const int kc, auto first_c, auto more_c) {
    constexpr bool more = decltype(more_c)::value;
    if (MODE_A && more && !(SF_W5_ABL & 2)) issue_patch(kc + 1);                    // behind barrier B: every wave is done with transform(kc)
    step(kc, std::integral_constant<int, 0>{}, first_c);
    step(kc, std::integral_constant<int, 1>{}, first_c);
    if (MODE_A && more) {
      wn_wait(4);                                               // younger than the patch: the A loads of steps 0 and 1
      scale_patch(kc + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(SF_W5_ABL & 8)) wn_barrier();                                             // M: patch(kc + 1) complete
      if (!(SF_W5_ABL & 1)) transform(kc + 1);
    }
    step(kc, std::integral_constant<int, 2>{}, first_c);
    step(kc, std::integral_constant<int, 3>{}, first_c);
    if (more) {
      if constexpr (MODE_A) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(SF_W5_ABL & 8)) wn_barrier();                                           // B: V(kc + 1) published, patch buffer free
      } else {
        wn_wait(4);                                             // patch(kc + 1) is older than this chunk's A loads
        wn_barrier();                                           // E: every wave holds its last fragments of V(kc); patch(kc + 1) complete
        transform(kc + 1);
        if (kc + 2 < nkc) issue_patch(kc + 2);                  // into the buffer transform(kc) read
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wn_barrier();                                           // F: V(kc + 1) published
      }
    }
  };
  chunk(0, std::true_type{}, std::true_type{});
  for (int kc = 1; kc + 1 < nkc; ++kc) chunk(kc, std::false_type{}, std::true_type{});
  chunk(nkc - 1, std::false_type{}, std::false_type{});
  SF_STAMP_AT(L, 2);

  // ---- output transform, first half in registers: T[ih][b] = (M A)[ih][b] -------------------------------------------------------------------
  //   b = 0: (M0 + M1) + M2      b = 1: M1 - (M2 + M3)        (the orders of conv_wino_kernel)
  if (SF_W5_ABL & 32) {
    f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) sacc += acc[p][mb][nb];
    if (sacc[0] + sacc[1] + sacc[2] + sacc[3] == 1.2345f) P.out[tid] = sacc[0];
    return;
  }
  float* const Tb = Vbuf;                                      // [ih][b][tile][64 cout], 16-byte slot cq of a tile's row at cq ^ (tile & 15)
  wn_barrier();                                                 // every wave is done with V (and nothing is in flight into the patch)
  {
    typedef __attribute__((address_space(3))) f32x4 lds_f4w;
    const int tw_base = ((ih * 2) * WT + j) * 64;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int tw = tw_base + (((hh * 8 + mb * 4 + g) ^ j) << 2);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const f32x4 m0 = acc[0][mb][nb], m1 = acc[1][mb][nb], m2 = acc[2][mb][nb], m3 = acc[3][mb][nb];
        *(lds_f4w*)(Tb + tw + nb * 16 * 64) = (m0 + m1) + m2;
        *(lds_f4w*)(Tb + tw + nb * 16 * 64 + WT * 64) = wn5_sub4(m1, m2 + m3);
      }
    }
  }
  // ---- second half + epilogue: thread = (tile wt, channel quad cq), the 16 quads of a pixel in consecutive lanes --------------------------------
  const int cq = tid & 15, wt_e = tid >> 4;
  const int tyl_e = wt_e >> 3, txl_e = wt_e & 7;
  const bool run_e = CAT && txl_e >= cn0;
  const int ty = ty0 + tyl_e, tx = run_e ? txl_e - cn0 : tx0 + txl_e;
  int oy0 = 2 * ty, ox0 = 2 * tx, ostep = 1;
  if constexpr (DIL) {
    int pX, tX, pY, tY;
    ax.decode(tx, pX, tX);
    ay.decode(ty, pY, tY);
    ostep = P.dil;
    ox0 = (tx < ax.nt && pX < ostep) ? pX + ostep * 2 * tX : W;
    oy0 = (ty < ay.nt && pY < ostep) ? pY + ostep * 2 * tY : H;
  }
  constexpr bool affine = EPI == EPI_AFFINE;
  const float* const t_a = affine ? P.add : P.e0;
  const float* const t_b = P.e1;
  const int cs_a = affine ? P.add_cs : P.e0_cs, cs_b = P.e1_cs;
  const bool img_ok = !CAT || img + (run_e ? 1 : 0) < P.n_img;
  const bool x0 = img_ok && ox0 < W, x1 = img_ok && ox0 + ostep < W, y0ok = oy0 < H, y1ok = oy0 + ostep < H;
  const unsigned pix = (x0 && y0ok) ? (unsigned)(oy0 * W + ox0 + (run_e ? H * W : 0)) : 0u;      // lanes without a tile compute on pixel 0 and store nothing
  const int cl = cq * 4;
  const int c = cout0 + cl;
  const bool c_ok = c < P.cout;
  const int c_ld = c_ok ? c : 0;
  // pixel k of the tile: (b, a) = (k >> 1, k & 1) = (x offset, y offset).  Operands and results move through raw buffer accesses on
  // the image (CAT: the image and the next one): one shared lane offset per tensor, the pixel's offset in the scalar operand, and a
  // lane without that pixel gets an offset beyond the range — its load returns 0, its store is dropped: no lane masks, no registers to
  // initialise, no 64-bit lane arithmetic
  const bool okk[4] = {c_ok && x0 && y0ok, c_ok && x0 && y1ok, c_ok && x1 && y0ok, c_ok && x1 && y1ok};
  const size_t img_base = (size_t)img * H * W;
  const size_t img_span = (size_t)((CAT && img + 1 < P.n_img) ? 2 : 1) * H * W;
  const int pk_[4] = {0, ostep * W, ostep, ostep * W + ostep};      // pixel offset of pixel k
  constexpr int OOB = (int)0x80000000;
  const bool has_a = affine ? P.add != nullptr : true, has_b = affine ? P.out2 != nullptr : true;
  const bool gate_lane = affine && P.out2 != nullptr && c >= P.gate_from;      // GRU gates, reset half: also emits (1 - r) * s
  f32x4 oa[4], ob[4];
#if defined(__HIP_DEVICE_COMPILE__)
  if (has_a) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(t_a + img_base * cs_a, img_span * cs_a * sizeof(float));
    const int v = (int)(pix * (unsigned)cs_a + (unsigned)c_ld) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) oa[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, okk[k] ? v : OOB, pk_[k] * cs_a * 4, 0));
  }
  if (has_b) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(t_b + img_base * cs_b, img_span * cs_b * sizeof(float));
    const int v = (int)(pix * (unsigned)cs_b + (unsigned)(affine ? (c_ld >= P.gate_from ? c_ld - P.gate_from : 0) : c_ld)) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) ob[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (okk[k] && (!affine || gate_lane)) ? v : OOB, pk_[k] * cs_b * 4, 0));
  }
#endif
  f32x4 as = (f32x4){1.f, 1.f, 1.f, 1.f};
  if (affine && P.add && P.add_scale) as = wn5_ld4(P.add_scale + (size_t)img * P.cout + c_ld);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wn_barrier();                                                 // the exchange is complete
  const f32x4 sc = wn_lds_read128(SBuf + cl), bi = wn_lds_read128(SBuf + COUT_T + cl);
  const int tr = wt_e * 64 + ((cq ^ (wt_e & 15)) << 2);
  SF_STAMP_AT(L, 5);
  f32x4 y[4];
#pragma unroll
  for (int bq = 0; bq < 2; ++bq) {
    const float* const tb = Tb + bq * WT * 64 + tr;
    const f32x4 t0 = wn_lds_read128(tb), t1 = wn_lds_read128(tb + 2 * WT * 64), t2 = wn_lds_read128(tb + 4 * WT * 64), t3 = wn_lds_read128(tb + 6 * WT * 64);
    y[2 * bq] = (t0 + t1) + t2;
    y[2 * bq + 1] = wn5_sub4(t1, t2 + t3);
  }
  SF_STAMP_AT(L, 6);
#pragma unroll
  for (int k = 0; k < 4; ++k) y[k] = __builtin_elementwise_fma(y[k], sc, bi);
  f32x4 y2[4];
  if constexpr (affine) {
    const bool act_last = (P.mode & 2) != 0;
    if (!act_last) wn5_act16(y, P.act);
    if (P.clamp_from >= 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c + q >= P.clamp_from) y[k][q] = fminf(fmaxf(y[k][q], P.clamp_lo), P.clamp_hi);
    }
    if (P.add) {
      if (P.add_scale) {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = __builtin_elementwise_fma(oa[k], as, y[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = y[k] + oa[k];
      }
    }
    if (act_last) wn5_act16(y, P.act);
    if (P.out2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y2[k] = ob[k] * ((f32x4){1.f, 1.f, 1.f, 1.f} - y[k]);
    }
  } else {      // EPI_BLEND (temporal.py:56)
    wn5_act16(y, P.act);
    if (P.mode & 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = oa[k] * (y[k] - ob[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = ((f32x4){1.f, 1.f, 1.f, 1.f} - oa[k]) * ob[k] + oa[k] * y[k];
    }
  }
  SF_STAMP_AT(L, 7);
  __builtin_amdgcn_sched_barrier(0);
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(P.out + img_base * P.out_cs + P.out_co, img_span * P.out_cs * sizeof(float) - (size_t)P.out_co * sizeof(float));
    const int v = (int)(pix * (unsigned)P.out_cs + (unsigned)c_ld) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, y[k]), rs, okk[k] ? v : OOB, pk_[k] * P.out_cs * 4, 0);
  }
  if (affine && P.out2) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(P.out2 + img_base * P.out2_cs, img_span * P.out2_cs * sizeof(float));
    const int v = (int)(pix * (unsigned)P.out2_cs + (unsigned)(c_ld >= P.gate_from ? c_ld - P.gate_from : 0)) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, y2[k]), rs, (okk[k] && gate_lane) ? v : OOB, pk_[k] * P.out2_cs * 4, 0);
  }
#endif
  SF_STAMP_AT(L, 3);
#ifdef SF_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SF_STAMP_AT(L, 4);
#endif
}

// weights: packed direct form w[cout_pad][9 * cin_pad] (tap-major, channel-minor) -> U[cin_pad/16][16][cout_pad][16] = G g G^T
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int cout_pad, int cin_pad) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (co, ci)
  if (idx >= (long)cout_pad * cin_pad) return;
  const int co = (int)(idx / cin_pad), ci = (int)(idx - (long)co * cin_pad);
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) g[a][b] = w[(size_t)co * 9 * cin_pad + (a * 3 + b) * cin_pad + ci];
  // G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    t[3][b] = g[2][b];
  }
  const int kc = ci >> 4, cl = ci & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]), u3 = t[i][2];
    const float uu[4] = {u0, u1, u2, u3};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) U[(((size_t)kc * 16 + i * 4 + jj) * cout_pad + co) * 16 + cl] = uu[jj];
  }
}
hipError_t launch_wino_weights(const float* w, float* U, int cout_pad, int cin_pad, hipStream_t stream) {
  const long n = (long)cout_pad * cin_pad;
  hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, U, cout_pad, cin_pad);
  return hipGetLastError();
}

// what the kernel takes: 3x3, stride 1, pad 1, no dilation / upsampling / gather / gate / SE scale / split-K / channel sums; inputs
// in whole 16-channel chunks; transformed weights present; images of at least one workgroup tile
bool wino_takes(const ConvProblem& q, int epi) {
  if (epi != EPI_AFFINE && epi != EPI_BLEND) return false;
  if (!q.w_wino || q.KH != 3 || q.KW != 3 || q.stride != 1 || q.dil < 1 || q.pad != q.dil || (q.in_up && q.dil != 1) || q.gather || q.gate || q.se_sum ||
      (q.in_scale && (epi != EPI_AFFINE || q.dil != 1 || q.c0 > 256)) ||      // SE-scaled input: plain AFFINE form, scales staged in LDS
      q.nsplit > 1 || q.chansum || q.acc_in || q.fuse_w || (epi == EPI_AFFINE && (q.mode & 4)))
    return false;
  if ((q.c0 % 16) || (q.c1 % 16) || q.c0 + q.c1 != q.cin_pad || (q.cout_pad % 64) || (q.cout % 4)) return false;
  if (q.Hout != (q.Hin << q.in_up) || q.Wout != (q.Win << q.in_up) || q.Hout < 16 || q.Wout < 32) return false;      // in_up: nearest x2 upsampling on read
  // dilated (conv_wino_kernel<.., DIL>): one input tensor, AFFINE epilogue, every phase of both axes at least 5 pixels = 3 tiles long
  if (q.dil > 1 && (q.c1 != 0 || epi != EPI_AFFINE || q.Hout / q.dil < 5 || q.Wout / q.dil < 5)) return false;
  const double img_bytes = 4.0 * q.Hin * q.Win;
  // the epilogue addresses its tensors as a per-image base + a 32-bit element offset
  if (img_bytes * q.out_cs >= 2147483648.0 || img_bytes * q.add_cs >= 2147483648.0 || img_bytes * q.e0_cs >= 2147483648.0 ||
      img_bytes * q.e1_cs >= 2147483648.0 || img_bytes * q.out2_cs >= 2147483648.0)
    return false;
  if (img_bytes * q.in0_cs >= 2147483648.0 || img_bytes * q.in1_cs >= 2147483648.0 || 4.0 * 16 * q.cout_pad * q.cin_pad >= 2147483648.0) return false;
  return true;
}

template <int COUT_T, int TH, int MW, int EPI, bool DIL = false, bool CAT = false>
static hipError_t launch_wino_t(const ConvLaunch& L, hipStream_t stream) {
  typedef WinoGeo<COUT_T, TH, MW, DIL, CAT> G;
  auto kern = conv_wino_kernel<COUT_T, TH, MW, EPI, DIL, CAT>;
  constexpr int lds = G::LDS_FLOATS * 4;
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  const ConvProblem& P = L.p[0];
  const int tiles_x = DIL ? WnAxis(P.Wout, P.dil).nt : (P.Wout + 1) / 2, tiles_y = DIL ? WnAxis(P.Hout, P.dil).nt : (P.Hout + 1) / 2;
  const long blocks = CAT ? (long)((tiles_y + TH - 1) / TH) * (((long)P.n_img * tiles_x + G::TW - 1) / G::TW)
                          : (long)P.n_img * ((tiles_y + TH - 1) / TH) * ((tiles_x + G::TW - 1) / G::TW);
  const long grid = ((blocks + 7) / 8) * 8 * (P.cout_pad / COUT_T);      // tile blocks in groups of 8 (one per XCD) x cout blocks
  if (grid > 0x7fffffffL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, 1, 1), dim3(WN_THREADS), lds, stream, L);
  return hipGetLastError();
}
// CAT pays where blocks of 8 tile columns fit the image badly and no epilogue operand is per image (SF_WINO_CAT=0: never)
static bool wino_cat(const ConvProblem& q) {
  static const int on = [] { const char* v = std::getenv("SF_WINO_CAT"); return v ? std::atoi(v) : 1; }();
  const int tpi = (q.Wout + 1) / 2;
  if (!on || q.dil != 1 || q.in_up || q.n_img < 2 || tpi < 8 || q.in_scale || q.bias_per_img || q.add_scale) return false;
  const int plain = (tpi + 7) / 8 * 8;
  if (plain * 100 < tpi * 110) return false;      // less than 10 % empty columns: keep the plain form
  const double img_bytes = 4.0 * q.Hin * q.Win;
  const int cs = q.in0_cs > q.in1_cs ? q.in0_cs : q.in1_cs;
  const int co = q.out_cs > q.add_cs ? q.out_cs : q.add_cs;
  const int ce = q.e0_cs > q.e1_cs ? q.e0_cs : q.e1_cs;
  const int cm = co > ce ? (co > q.out2_cs ? co : q.out2_cs) : (ce > q.out2_cs ? ce : q.out2_cs);
  return 2.0 * img_bytes * cs < 2147483648.0 && 2.0 * img_bytes * cm < 2147483648.0;      // two images behind one base
}
// which tile configuration a problem runs on: 0 = 128 cout x 32 tiles, 1 = 64 cout x 64 tiles, 2 = 64 cout x 32 tiles, two
// workgroups per CU (SF_WINO_TILE = 128 | 64 | 2 forces one: experiments); -1: none
int wino_variant(const ConvProblem& q) {
  static const int force = [] { const char* v = std::getenv("SF_WINO_TILE"); return v ? std::atoi(v) : 0; }();
  if (q.cout_pad % 64) return -1;
  if (q.dil > 1) return 3;      // the dilated form of configuration 2
  if (wino_cat(q)) return 4;    // configuration 2 with the images concatenated along x
#if defined(SF_WINO_ALL_TILES)      // the one-workgroup-per-CU configurations are built for experiments only (tools/r02/build_variant.sh)
  if (force == 64) return 1;
  if (force == 128) return q.cout_pad % 128 == 0 ? 0 : 2;
#else
  (void)force;
#endif
  // measured (profiles/r04_o_winobench_two_wg_per_cu_vs_128x32.txt): two 64 x 32 workgroups per CU win on every layer — the
  // 128 -> 128 layer on 224 frames 12.4 ms against 14.3 ms on 128 x 32 tiles (19.4 ms direct), 64-cout layers 1.59-1.62x the direct
  // form against 1.18-1.28x on 64 x 64 tiles
  return 2;
}
// Winograd tiles a launch executes (the profiler prices 16 products per tile and (cin, cout) pair)
double wino_tiles(const ConvProblem& q) {
  if (q.dil > 1) return (double)q.n_img * WnAxis(q.Hout, q.dil).nt * WnAxis(q.Wout, q.dil).nt;
  return (double)q.n_img * ((q.Hout + 1) / 2) * ((q.Wout + 1) / 2);
}
template <int EPI, bool DIL = false, bool CAT = false>
static hipError_t launch_wino5_t(const ConvLaunch& L, hipStream_t stream) {
  typedef Wino5Geo<DIL, CAT> G;
  auto kern = conv_wino5_kernel<EPI, DIL, CAT>;
  constexpr int lds = G::LDS_FLOATS * 4;
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  const ConvProblem& P = L.p[0];
  const int tiles_x = DIL ? WnAxis(P.Wout, P.dil).nt : (P.Wout + 1) / 2, tiles_y = DIL ? WnAxis(P.Hout, P.dil).nt : (P.Hout + 1) / 2;
  const long blocks = CAT ? (long)((tiles_y + G::TH - 1) / G::TH) * (((long)P.n_img * tiles_x + G::TW - 1) / G::TW)
                          : (long)P.n_img * ((tiles_y + G::TH - 1) / G::TH) * ((tiles_x + G::TW - 1) / G::TW);
  const long grid = ((blocks + 7) / 8) * 8 * (P.cout_pad / G::COUT_T);
  if (grid > 0x7fffffffL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, 1, 1), dim3(WN_THREADS), lds, stream, L);
  return hipGetLastError();
}
// one problem per launch
hipError_t launch_conv_wino(const ConvLaunch& L, int epi, hipStream_t stream) {
  if (L.nprob != 1 || !wino_takes(L.p[0], epi)) return hipErrorInvalidValue;
  const bool affine = epi == EPI_AFFINE;
  static const int ver = [] { const char* v = std::getenv("SF_WINO_V"); return v ? std::atoi(v) : 5; }();      // 4: the round-4 kernel (A/B runs)
  if (ver >= 5) {
    switch (wino_variant(L.p[0])) {
      case 2: return affine ? launch_wino5_t<EPI_AFFINE>(L, stream) : launch_wino5_t<EPI_BLEND>(L, stream);
      case 3: return affine ? launch_wino5_t<EPI_AFFINE, true>(L, stream) : hipErrorInvalidValue;
      case 4: return affine ? launch_wino5_t<EPI_AFFINE, false, true>(L, stream) : launch_wino5_t<EPI_BLEND, false, true>(L, stream);
    }
  }
  switch (wino_variant(L.p[0])) {
#if defined(SF_WINO_ALL_TILES)
    case 0: return affine ? launch_wino_t<128, 4, 2, EPI_AFFINE>(L, stream) : launch_wino_t<128, 4, 2, EPI_BLEND>(L, stream);
    case 1: return affine ? launch_wino_t<64, 8, 2, EPI_AFFINE>(L, stream) : launch_wino_t<64, 8, 2, EPI_BLEND>(L, stream);
#endif
    case 2: return affine ? launch_wino_t<64, 4, 1, EPI_AFFINE>(L, stream) : launch_wino_t<64, 4, 1, EPI_BLEND>(L, stream);
    case 3: return affine ? launch_wino_t<64, 4, 1, EPI_AFFINE, true>(L, stream) : hipErrorInvalidValue;
    case 4: return affine ? launch_wino_t<64, 4, 1, EPI_AFFINE, false, true>(L, stream) : launch_wino_t<64, 4, 1, EPI_BLEND, false, true>(L, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace sf
