// Winograd F(2x2, 3x3) convolution for the 3x3 / stride-1 layers of the batched forward (gfx950 only).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 4x4 input tile d (stride 2), 3x3 filter g, 2x2 outputs Y
// in exact fp32 arithmetic on v_mfma_f32_16x16x4_f32: 16 multiplies per 2x2 outputs and (cin, cout) pair instead of 36 — 2.25x fewer
// MACs than the direct form the implicit-GEMM kernels compute; different rounding (the transforms add / subtract before the products):
// tools/r04/winograd_study.py measured <= 8.4e-6 max-abs on the BEV output of every BASELINE config with all eligible layers switched
// (north-star tolerance 1e-3).  FLOP accounting: bench.py prices these launches at their EXECUTED FLOPs for the roofline and keeps the
// algorithmic (direct-form) count for ODE-steps/s (savings are not credited as achieved FLOPs).
//
// Round 6: a 16-tile / three-workgroups-per-CU form for under-filled launches (Wino5Geo TH_ = 2) and the 7x7 + LayerNorm layer of the batched
// cells as nine 3x3 tap groups (GRP = 9, EPI_LNG); both in front of conv_wino5_kernel.
// Weights are transformed once at pack time (pack.hip: U[cin/16][16 positions][cout_pad][16], sf_conv_w::w_wino).  The kernel
// (conv_wino5_kernel, round 5) is described in front of it; the round-4 kernel it replaced (U through an LDS ring, one wave = 16 cout x
// 16 tiles x all 16 positions: 0.57-0.66 of the fp32 MFMA peak against 0.64-0.74) lives in the history of this file, the F(4x4, 3x3)
// kernel that was built and measured slower in tools/experiments/r05_winograd44_kernel.diff (profiles/r05_winograd44_no_go.md).
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
typedef __attribute__((address_space(3))) void wn_lds_void;

__device__ __forceinline__ f32x4 wn_lds_read128(const float* p) {
  typedef const __attribute__((address_space(3))) f32x4 lds_f4;
  return *(lds_f4*)p;
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

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 wn_lo(const f32x4 v) { return (f32x2){v[0], v[1]}; }
__device__ __forceinline__ f32x2 wn_hi(const f32x4 v) { return (f32x2){v[2], v[3]}; }

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
    case ACT_SIGMOID:      // 1 / (1 + 2^(-x log2 e)) on the hardware exp2 / reciprocal (1 ulp each: ~2e-7 from the expf / division form of the
                           // direct-form kernels, 4 instructions instead of 25 per element: the GRU gates' epilogue is 16 of them per lane)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) y[k][q] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y[k][q]));
      break;
    case ACT_NONE: break;
    default:
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) y[k][q] = spm_act(y[k][q], act);
  }
}
// sum over the 16 lanes of a DPP row (the 16 channel quads of a pixel in the epilogue's lane order), in every lane of the row
__device__ __forceinline__ float wn_row16_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));       // quad_perm [1, 0, 3, 2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));       // quad_perm [2, 3, 0, 1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));      // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));      // row_mirror
#endif
  return v;
}
__device__ __forceinline__ float wn_gelu(float v) { return 0.5f * v * (1.f + spm_erf(v * 0.70710678118654752440f)); }      // (= gelu_f of conv_igemm.hip)

// TH_: tile rows of the workgroup's block — 4 (8 x 16 output pixels, 32 tiles: the kernel as measured in DESIGN 4.3) or 2 (4 x 16 pixels, 16
// tiles, three workgroups per CU: round 6, for launches of fewer than two rounds of the chip's 512 workgroup slots — a single 200x200 frame
// with 64 / 128 output channels is 313 / 626 workgroups of 32 tiles; launch_conv_wino holds the measured rule)
template <bool DIL, bool CAT, int TH_ = 4>
struct Wino5Geo {
  static_assert(!(DIL && CAT), "one run structure at a time");
  static_assert(TH_ == 4 || (TH_ == 2 && !DIL), "tile rows per block");
  static constexpr int COUT_T = 64, TH = TH_, TW = 8, WT = TH * TW, NB = WT / 16;
  static constexpr int RY = 2, RX = 4;
  static constexpr int PH = 2 * TH + (DIL ? 2 * RY : 2), PW = 2 * TW + (DIL ? 2 * RX : (CAT ? 4 : 2));
  static constexpr int NPX = PH * PW;
  static constexpr int ND = (NPX * 4 + 63) / 64;                // 1-KB DMA pieces of a patch; wave w issues pieces w, w + 8, ...
  static constexpr int NP = (ND + 7) / 8;
  static constexpr int P_FLOATS = ND * 256;
  static constexpr int V_FLOATS = 16 * WT * 16;
  static constexpr int NVB = DIL ? 1 : 2, NPB = DIL ? 2 : 1;
  static constexpr int PARK = DIL ? 512 + 64 : 0;               // DIL: the patch offset of every transform task + the per-axis tables of the block
  static constexpr int SB = 2 * COUT_T, SC = DIL ? 0 : (CAT ? 512 : 256);      // SE input scales: [c0 <= 256] of the image (CAT: and of the next one)
  static constexpr int T_FLOATS = 4 * 2 * WT * COUT_T;          // the exchange of the output transform: [row i][b][tile][cout]
  static_assert(NVB * V_FLOATS + NPB * P_FLOATS >= T_FLOATS, "the exchange lives over V and the patch");
  static constexpr int PV = TH_ == 2 ? 2 * 512 : 0;             // 16-tile form (80 registers): the lanes' two patch offsets live in LDS between the chunks
  static constexpr int LDS_FLOATS = NVB * V_FLOATS + NPB * P_FLOATS + PARK + SB + SC + PV;
  static_assert((TH_ == 2 ? 3 : 2) * LDS_FLOATS * 4 <= 160 * 1024, "two workgroups per CU (three of the 16-tile form)");
};

// timing-only ablations (tools/r05/ablate.sh; wrong results by construction): -DSF_W5_ABL=<bits>  1: no input transform in the loop,
// 2: no patch DMAs in the loop, 4: no A loads in the loop, 8: no barriers in the loop, 16: no MFMAs, 32: minimal epilogue
#if !defined(SF_W5_ABL)
#define SF_W5_ABL 0
#endif
#if !defined(SF_W5_MAGIC)
#define SF_W5_MAGIC 1
#endif
// GRP = 9 (round 6, EPI_LNG only): a 7x7 / pad-3 layer as nine 3x3 sub-kernels of its 9x9 zero frame (wino_weights_kernel) — tap group (a, b)
// convolves the input shifted by (3a - 3, 3b - 3), all nine accumulate into the same Winograd-domain sums: the K loop runs over
// 9 x cin/16 chunks, the patch offsets move when a chunk starts a new group, everything else is the 3x3 kernel
template <int EPI, bool DIL = false, bool CAT = false, int TH_ = 4, int GRP = 1>
__global__ __launch_bounds__(WN_THREADS, TH_ == 2 ? 6 : 4) void conv_wino5_kernel(const ConvLaunch L) {
  static_assert(GRP == 1 || (GRP == 9 && !DIL && TH_ == 4), "tap groups: the 7x7 form");
  static_assert((EPI == EPI_LNG) == (GRP == 9), "the LayerNorm epilogue belongs to the 7x7 form");
  static_assert(EPI != EPI_SAMPLE || (!DIL && TH_ == 4), "the sampling layer: plain / concatenated 32-tile forms");
  typedef Wino5Geo<DIL, CAT, TH_> G;
  constexpr int NB = G::NB;                                    // 16-tile fragments of the block
  constexpr int COUT_T = G::COUT_T, TH = G::TH, TW = G::TW, WT = G::WT, PW = G::PW, NP = G::NP;
  constexpr bool MODE_A = G::NVB == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Vbuf = smem;
  float* const Pbuf = Vbuf + G::NVB * G::V_FLOATS;
  float* const Park = Pbuf + G::NPB * G::P_FLOATS;
  float* const SBuf = Park + G::PARK;                          // [scale COUT_T][bias COUT_T]
  float* const SCbuf = SBuf + G::SB;                           // [c0] input scales (SCALED)
  float* const PVbuf = SCbuf + G::SC;                          // [2][512] patch offsets of the lanes (16-tile form)
  (void)PVbuf;
  // a group of layers of identical geometry (the two branches of a dual cell) shares one launch: problem i owns the workgroups
  // [i * wg_base[1], (i + 1) * wg_base[1]) (a multiple of 8 each, so that workgroup -> XCD stays blockIdx & 7); the tails of the
  // single launches (625 - 1250 workgroups over 512 slots) merge into one
  int lin_ = (int)blockIdx.x;
  const int pi_ = (int)__umulhi((unsigned)lin_, L.wn_m[0]);      // wn_m[0] = 1 for a single problem: always 0, no branch
  lin_ -= pi_ * L.wg_base[1];
  const ConvProblem& P = L.p[0];       // geometry, strides, flags: the same for every problem of the group
  const ConvProblem& PX = L.p[pi_];     // tensors: this workgroup's problem
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
  const int lin = lin_, xcd = lin & 7, slot_ = lin >> 3;
  // the block decode divides by four launch constants: the host passes ceil(2^32 / d) (0 for d = 1) and the quotients are one
  // s_mul_hi each instead of a reciprocal sequence through the vector unit (exact: dividend x divisor < 2^32, checked by the host)
#if SF_W5_MAGIC
  auto mdiv = [](const int nn, const unsigned m) { return m ? (int)__umulhi((unsigned)nn, m) : nn; };
  const unsigned m_ncb = L.wn_m[1], m_nbx = L.wn_m[2], m_nby = L.wn_m[3], m_tpi = L.wn_m[4];
  (void)m_tpi;
  const int tb_ = mdiv(slot_, m_ncb);
  int b = xcd * per_xcd + tb_;
  if (b >= nblk) return;
  const int bq_ = mdiv(b, m_nbx);
  const int bx = b - bq_ * nbx; b = bq_;
  const int bi_ = mdiv(b, m_nby);
  const int by = b - bi_ * nby;
  const int img = CAT ? mdiv(bx * TW, m_tpi) : bi_;
#else
  const int tb_ = slot_ / ncb;
  int b = xcd * per_xcd + tb_;
  if (b >= nblk) return;
  const int bx = b % nbx; b /= nbx;
  const int by = b % nby;
  const int img = CAT ? (bx * TW) / tpi : b / nby;
#endif
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
  const int nkc_g = P.cin_pad >> 4;                            // 16-channel chunks (>= 2: cin_pad is a multiple of 32) of a tap group
  const int nkc = GRP * nkc_g;
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
  const __amdgpu_buffer_rsrc_t rsrc0 = make_rsrc(PX.in0 + (size_t)img * img_px * P.in0_cs, n_in * img_px * P.in0_cs * sizeof(float));
  const __amdgpu_buffer_rsrc_t rsrc1 = make_rsrc(PX.in1 ? PX.in1 + (size_t)img * img_px * P.in1_cs : PX.in0, PX.in1 ? n_in * img_px * P.in1_cs * sizeof(float) : 0);
  const __amdgpu_buffer_rsrc_t rsrc_u = make_rsrc(PX.w_wino, (size_t)nkc * 16 * P.cout_pad * 16 * sizeof(float));
#endif
  constexpr bool SCALED = G::SC > 0 && (EPI == EPI_AFFINE || EPI == EPI_SAMPLE);
  const bool scaled = SCALED && PX.in_scale != nullptr;
  // DIL: what depends on one axis only is computed once per block by 36 + 12 lanes and shared through LDS (every lane walking the run
  // lists itself — twice per patch element, twice for its transform task, twice for its output tile — was 300 of the kernel's ~650
  // vector instructions per wave and tile): Tab[0..PW) input x of patch column c (-1: zero fill), [32..32+PH) input y of patch row r,
  // [48..56) first output x of tile column t (W: none), [56..60) first output y of tile row t, [60..68) / [68..72) run of tile column / row
  typedef __attribute__((address_space(3))) int lds_int;
  float* const TabF = Park + 512;
  auto tab = [&](const int i) -> int { return *(const lds_int*)(TabF + i); };
  if constexpr (DIL) {
    if (tid < 72) {
      int v = 0;
      if (tid < PW) { int ix; v = ax.patch_coord(tid, px0, pt0, TW, W, ix) ? ix : -1; }
      else if (tid >= 32 && tid < 32 + G::PH) { int iy; v = ay.patch_coord(tid - 32, py0, qt0, TH, H, iy) ? iy : -1; }
      else if (tid >= 48 && tid < 60) {
        const bool isx = tid < 56;
        const int t = isx ? tx0 + (tid - 48) : ty0 + (tid - 56);
        const WnAxis& a = isx ? ax : ay;
        int pp, tt;
        a.decode(t, pp, tt);
        v = (t < a.nt && pp < P.dil) ? pp + P.dil * 2 * tt : (isx ? W : H);
      } else if (tid >= 60) {
        const bool isx = tid < 68;
        const int t = isx ? tx0 + (tid - 60) : ty0 + (tid - 68);
        int pp, tt;
        (isx ? ax : ay).decode(t, pp, tt);
        const int r = pp - (isx ? px0 : py0), lim = isx ? G::RX : G::RY;
        v = r < lim ? r : lim - 1;
      }
      *(lds_int*)(TabF + tid) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wn_barrier();
  }
  // ---- patch DMA: element e = (pixel, channel quad) of the patch, 16 bytes each, LDS linear in e; piece idx = d * 8 + wave --------------
  SF_STAMP_AT(L, 14);
  int pv0[NP], pv1[DIL ? 1 : NP];
  int piy[GRP > 1 ? NP : 1], pix_[GRP > 1 ? NP : 1];           // tap groups: the element's input pixel without the group's shift (piy < -16: never valid)
  int pvs0[GRP > 1 ? NP : 1], pvs1[GRP > 1 ? NP : 1];           // ... and its two offsets under the current group's shift
  (void)piy; (void)pix_; (void)pvs0; (void)pvs1;
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
      const int pyc = py < G::PH ? py : 0;                      // (pieces are padded to whole DMAs: elements beyond the patch)
      ix = tab(px); iy = tab(32 + pyc);
      ok = pix < G::NPX && ix >= 0 && iy >= 0;
    }
    const int pofs = DIL ? iy * W + ix : (iy >> up) * Win + (ix >> up) + (CAT ? run * img_px_i : 0);
    if constexpr (GRP > 1) {      // the range check waits for the group's shift (issue_patch)
      const bool in_blk = pix < G::NPX && (!CAT || (img + run < P.n_img && (run == 0 || cn0 < TW)));
      piy[d] = in_blk ? iy : -(1 << 20);
      pix_[d] = ix;
      pv0[d] = (pofs * P.in0_cs + quad * 4) * 4;
      pv1[d] = (pofs * P.in1_cs + quad * 4) * 4;
    } else {
      pv0[d] = ok ? (pofs * P.in0_cs + quad * 4) * 4 : (int)0x80000000;
      if constexpr (!DIL) pv1[d] = ok ? (pofs * P.in1_cs + quad * 4) * 4 : (int)0x80000000;
    }
  }
  int g_kc = 0, g_grp = 0;                                      // tap groups: issue_patch is called for chunks 0, 1, 2, ... in order
  // CAT + SE-scaled input (round 6; 32-tile form only — the 16-tile form has no register for it and the host never asks): bit d = patch
  // element d of this lane lies in the NEXT image, whose scales sit behind this image's in SCbuf
  constexpr bool CAT_SC = CAT && G::PV == 0 && G::SC > 0;
  int run_bits = 0;
  if constexpr (CAT_SC) {
#pragma unroll
    for (int d = 0; d < NP; ++d) {
      const int pix = ((d * 8 + wave) * 64 + lane) >> 2;
      run_bits |= ((pix - (pix / PW) * PW) >= 2 * cn0 + 2 ? 1 : 0) << d;
    }
  }
  (void)run_bits;
  (void)g_kc; (void)g_grp;
  SF_STAMP_AT(L, 15);
  if constexpr (G::PV > 0) {
    static_assert(G::PV == 0 || NP == 1, "one piece per wave");
    *(lds_int*)(PVbuf + tid) = pv0[0];
    *(lds_int*)(PVbuf + 512 + tid) = pv1[0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  auto issue_patch = [&](const int kc) {
    float* const dst = Pbuf + (G::NPB == 2 ? (kc & 1) : 0) * G::P_FLOATS;
    const bool from1 = !DIL && kc * 16 >= c0;                   // wave-uniform: the whole chunk reads in1 (c0 % 16 == 0)
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (GRP > 1) {
      if (g_kc == 0) {      // a new tap group: the lanes' offsets under its shift (range check included), once for the group's chunks
        const int ga = (g_grp * 11) >> 5, gb = g_grp - 3 * ga;    // group (a, b): rows 3a .. 3a+2, columns 3b .. 3b+2 of the 9x9 frame
        const int sy = 3 * ga - 3, sx = 3 * gb - 3;
        const int dlt = (sy * Win + sx) * 4;
#pragma unroll
        for (int d = 0; d < NP; ++d) {
          const bool okg = (unsigned)(piy[d] + sy) < (unsigned)H && (unsigned)(pix_[d] + sx) < (unsigned)W;
          pvs0[d] = okg ? pv0[d] + dlt * P.in0_cs : (int)0x80000000;
          pvs1[d] = okg ? pv1[d] + dlt * P.in1_cs : (int)0x80000000;
        }
      }
      const bool f1 = g_kc * 16 >= c0;
      const int so = f1 ? (g_kc * 16 - c0) * 4 : g_kc * 64;
#pragma unroll
      for (int d = 0; d < NP; ++d) {
        if (d >= npw) continue;
        float* const dB = dst + (d * 8 + wave) * 256;
        if (f1) {
          asm volatile("; patch from in1");
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (wn_lds_void*)dB, 16, pvs1[d], so, 0, 0);
        } else {
          asm volatile("; patch from in0");
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (wn_lds_void*)dB, 16, pvs0[d], so, 0, 0);
        }
      }
      if (++g_kc == nkc_g) { g_kc = 0; ++g_grp; }
      (void)kc; (void)from1;
      return;
    }
    if constexpr (G::PV > 0) {
      if (npw > 0) {
        int pv;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pv) : "v"((int)(size_t)(lds_int*)(PVbuf + (from1 ? 512 : 0) + tid)) : "memory");
        float* const dB = dst + wave * 256;
        if (from1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (wn_lds_void*)dB, 16, pv, (kc * 16 - c0) * 4, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (wn_lds_void*)dB, 16, pv, kc * 64, 0, 0);
      }
      return;
    }
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
        f32x4 s4n = s4;
        if constexpr (CAT_SC) s4n = wn_lds_read128(SCbuf + c0 + kc * 16 + (lane & 3) * 4);      // the next image's scales (c0 <= 256)
        typedef __attribute__((address_space(3))) f32x4 lds_f4w;
#pragma unroll
        for (int d = 0; d < NP; ++d) {
          if (d >= npw) continue;
          float* const q = Pbuf + ((d * 8 + wave) * 64 + lane) * 4;
          *(lds_f4w*)q = wn_lds_read128(q) * ((CAT_SC && ((run_bits >> d) & 1)) ? s4n : s4);
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
  (void)NS; (void)u_voff; (void)u_pos_bytes;                    // (read by device code only: the host pass of hipcc sees them unused)
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
    const int rx = tab(60 + txl), ry = tab(68 + tyl);
    *(__attribute__((address_space(3))) int*)(Park + tid) = ((2 * tyl + 2 * ry) * PW + 2 * txl + 2 * rx) * 16 + quad * 4;
  }
  auto transform = [&](const int kc) {
    if constexpr (WT == 16) {
      if (hh) return;                                           // 64 tasks per row of positions: the first wave of the pair has them all
    }
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
    if (scaled && tid < c0) scv = PX.in_scale[(size_t)img * c0 + tid];
    if (CAT_SC && scaled && tid >= c0 && tid < 2 * c0 && img + 1 < P.n_img) scv = PX.in_scale[(size_t)(img + 1) * c0 + (tid - c0)];
  issue_patch(0);
  f32x4 A[2][2];                                                // [ring slot = step & 1][mb]
  A[0][0] = load_A(0, 0); A[0][1] = load_A(0, 1);
  A[1][0] = load_A(1, 0); A[1][1] = load_A(1, 1);
  float sbv = tid < COUT_T ? 1.f : 0.f;
  if (tid < 2 * COUT_T) {
    const int co = cout0 + (tid < COUT_T ? tid : tid - COUT_T);
    if (co < P.cout) {
      if (tid < COUT_T) { if (PX.scale) sbv = PX.scale[co]; }
      else if (PX.bias) sbv = PX.bias[(P.bias_per_img ? (size_t)img * P.cout : 0) + co];
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

  f32x4 acc[4][2][NB];
  int b_cur = b_off;
  auto step = [&](const int kc, auto p_c, auto first_c) {
    constexpr int p = decltype(p_c)::value, slot = p & 1;
    constexpr bool first = decltype(first_c)::value;
    const float* const vb = Vbuf + (G::NVB == 2 ? (kc & 1) : 0) * G::V_FLOATS + p * WT * 16 + b_cur;
    f32x4 Bf[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) Bf[nb] = wn_lds_read128(vb + nb * 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
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
  if constexpr (WT == 16) {      // (80 registers: the last chunk is its own copy of the code, and hipcc carries its fragment offset past the loop in scratch)
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const int j2 = t2 & 15, g2 = (t2 & 63) >> 4;
    b_cur = (ih * 4 * WT + j2) * 16 + ((g2 ^ ((j2 >> 2) & 2)) << 2);
  }
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
        for (int nb = 0; nb < NB; ++nb) sacc += acc[p][mb][nb];
    if (sacc[0] + sacc[1] + sacc[2] + sacc[3] == 1.2345f) PX.out[tid] = sacc[0];
    return;
  }
  float* const Tb = Vbuf;                                      // [ih][b][tile][64 cout], 16-byte slot cq of a tile's row at cq ^ (tile & 15)
  wn_barrier();                                                 // every wave is done with V (and nothing is in flight into the patch)
  // 16-tile form (80 registers): what the exchange and the epilogue derive from the thread index is derived HERE — hipcc would compute it in
  // front of the loop and carry it through in scratch
  int tid_e = tid;
  if constexpr (WT == 16) asm volatile("" : "+v"(tid_e));
  {
    typedef __attribute__((address_space(3))) f32x4 lds_f4w;
    const int j = tid_e & 15, g = (tid_e & 63) >> 4;
    const int tw_base = ((ih * 2) * WT + j) * 64;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int tw = tw_base + (((hh * 8 + mb * 4 + g) ^ j) << 2);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f32x4 m0 = acc[0][mb][nb], m1 = acc[1][mb][nb], m2 = acc[2][mb][nb], m3 = acc[3][mb][nb];
        *(lds_f4w*)(Tb + tw + nb * 16 * 64) = (m0 + m1) + m2;
        *(lds_f4w*)(Tb + tw + nb * 16 * 64 + WT * 64) = wn5_sub4(m1, m2 + m3);
      }
    }
  }
  // ---- second half + epilogue: thread = (tile wt, channel quad cq), the 16 quads of a pixel in consecutive lanes --------------------------------
  const int cq = tid_e & 15, wt_e = WT == 32 ? tid_e >> 4 : (tid_e >> 4) & (WT - 1);
  const bool has_tile = WT == 32 || tid_e < 16 * WT;           // 16-tile blocks: half of the threads have no tile (they load and store nothing)
  const int tyl_e = wt_e >> 3, txl_e = wt_e & 7;
  const bool run_e = CAT && txl_e >= cn0;
  const int ty = ty0 + tyl_e, tx = run_e ? txl_e - cn0 : tx0 + txl_e;
  int oy0 = 2 * ty, ox0 = 2 * tx, ostep = 1;
  if constexpr (DIL) {
    ostep = P.dil;
    ox0 = tab(48 + txl_e);
    oy0 = tab(56 + tyl_e);
  }
  constexpr bool affine = EPI == EPI_AFFINE, lng = EPI == EPI_LNG, smp = EPI == EPI_SAMPLE;
  const float* const t_a = (affine || lng) ? PX.add : PX.e0;
  const float* const t_b = PX.e1;
  const int cs_a = (affine || lng) ? P.add_cs : P.e0_cs, cs_b = P.e1_cs;
  const bool img_ok = has_tile && (!CAT || img + (run_e ? 1 : 0) < P.n_img);
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
  (void)t_a; (void)t_b; (void)cs_a; (void)cs_b; (void)pix; (void)okk; (void)img_base;
  const size_t n_span = (CAT && img + 1 < P.n_img) ? 2 : 1;      // images the workgroup's tiles may lie in
  const size_t img_span = n_span * H * W;
  const int pk_[4] = {0, ostep * W, ostep, ostep * W + ostep};      // pixel offset of pixel k
  constexpr int OOB = (int)0x80000000;
  const bool has_a = smp ? false : ((affine || lng) ? PX.add != nullptr : true), has_b = (lng || smp) ? false : (affine ? PX.out2 != nullptr : true);
  // SAMPLE (round 6): the packed cout rows are interleaved — rows 4q .. 4q+3 = (loc ch, loc ch+1, raw ch, raw ch+1), ch = 8 (row >> 4) + 2 ((row >> 2) & 3)
  // (pack.hip) — so the lane's four channels are two (loc, raw) pairs; eps is an input of the call or drawn here (Philox, keyed by pixel and channel)
  const int s_half = P.cout >> 1, s_ch = ((c >> 4) << 3) + 2 * ((c >> 2) & 3);
  const bool s_ok = c_ok && s_ch < s_half;
  f32x2 ev[4];
  (void)s_half; (void)s_ch; (void)s_ok; (void)ev;
  const bool gate_lane = affine && PX.out2 != nullptr && c >= P.gate_from;      // GRU gates, reset half: also emits (1 - r) * s
  (void)img_span; (void)pk_; (void)OOB; (void)has_a; (void)has_b; (void)gate_lane;
  f32x4 oa[4], ob[4];
#if defined(__HIP_DEVICE_COMPILE__)
  if (has_a && affine && !DIL && P.add_up) {      // block-uniform: the tile's four pixels read source pixel (ty, tx) of the half-size tensor
    const size_t pimg = (size_t)(H >> 1) * (W >> 1);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(t_a + (size_t)img * pimg * cs_a, n_span * pimg * cs_a * sizeof(float));
    const unsigned ppix = (unsigned)(ty * (W >> 1) + tx) + (run_e ? (unsigned)pimg : 0u);
    oa[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, okk[0] ? (int)(ppix * (unsigned)cs_a + (unsigned)c_ld) * 4 : OOB, 0, 0));
    oa[1] = oa[0]; oa[2] = oa[0]; oa[3] = oa[0];
  } else if (has_a) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(t_a + img_base * cs_a, img_span * cs_a * sizeof(float));
    const int v = (int)(pix * (unsigned)cs_a + (unsigned)c_ld) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) oa[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, okk[k] ? v : OOB, pk_[k] * cs_a * 4, 0));
  }
  if constexpr (smp) {
    if (PX.e0) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(PX.e0 + img_base * s_half, img_span * s_half * sizeof(float));
      const int v = (int)(pix * (unsigned)s_half + (unsigned)(s_ok ? s_ch : 0)) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        ev[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (okk[k] && s_ok) ? v : OOB, pk_[k] * s_half * 4, 0));
    }
  }
  if (has_b) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(t_b + img_base * cs_b, img_span * cs_b * sizeof(float));
    const int v = (int)(pix * (unsigned)cs_b + (unsigned)(affine ? (c_ld >= P.gate_from ? c_ld - P.gate_from : 0) : c_ld)) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) ob[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (okk[k] && (!affine || gate_lane)) ? v : OOB, pk_[k] * cs_b * 4, 0));
  }
#endif
  f32x4 as = (f32x4){1.f, 1.f, 1.f, 1.f};
  if (affine && PX.add && PX.add_scale) as = wn5_ld4(PX.add_scale + (size_t)(img + ((CAT && run_e && img_ok) ? 1 : 0)) * P.cout + c_ld);
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
  if constexpr (!lng) {
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = __builtin_elementwise_fma(y[k], sc, bi);
  }
  f32x4 y2[4];
  if constexpr (smp) {      // q = act(conv + bias); p = loc + eps (softplus(raw) + 1e-8)   (model_utils.py:84, 107-108; as conv_igemm.hip / conv_sp.hip)
    wn5_act16(y, P.act);
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2;
    const __amdgpu_buffer_rsrc_t rs_p = make_rsrc(PX.out + img_base * s_half, img_span * s_half * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(PX.out2 ? PX.out2 + img_base * P.cout : PX.out, PX.out2 ? img_span * P.cout * sizeof(float) : 0);
    const unsigned gp0 = (unsigned)img_base + pix;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool okp = okk[k] && s_ok;
      f32x2 e = ev[k];
      if (!PX.e0) {      // block-uniform
        const float2 d = spm_philox_normal2(P.philox, P.draw, okp ? gp0 + (unsigned)pk_[k] : 0u, (unsigned)(s_ok ? s_ch : 0));
        e = (f32x2){d.x, d.y};
      }
      const f32x2 r = {y[k][0] + e[0] * (spm_softplus(y[k][2]) + 1e-8f), y[k][1] + e[1] * (spm_softplus(y[k][3]) + 1e-8f)};
      const int vp = (int)(pix * (unsigned)s_half + (unsigned)s_ch) * 4, vq = (int)(pix * (unsigned)P.cout + (unsigned)s_ch) * 4;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, r), rs_p, okp ? vp : OOB, pk_[k] * s_half * 4, 0);
      if (PX.out2) {      // raw q parameters, reference channel order [loc | raw]
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){y[k][0], y[k][1]}), rs_q, okp ? vq : OOB, pk_[k] * P.cout * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){y[k][2], y[k][3]}), rs_q, okp ? vq + s_half * 4 : OOB, pk_[k] * P.cout * 4, 0);
      }
    }
#endif
    SF_STAMP_AT(L, 3);
    return;
  } else if constexpr (lng) {      // [LayerNorm over the 64 channels of the pixel (convolutions.py:303-308): 16 lanes x 4] -> GELU [-> + residual]
    if (P.mode & 1) {
      const float inv_c = 1.f / (float)P.cout;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float mean = wn_row16_sum((y[k][0] + y[k][1]) + (y[k][2] + y[k][3])) * inv_c;
        const f32x4 dv = y[k] - (f32x4){mean, mean, mean, mean};
        const float var = wn_row16_sum((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3])) * inv_c;
        const float rstd = 1.f / sqrtf(var + P.eps);
        y[k] = __builtin_elementwise_fma(dv * (f32x4){rstd, rstd, rstd, rstd}, sc, bi);      // SBuf carries the LayerNorm's weight / bias here
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) y[k][q] = wn_gelu(y[k][q]);
    if (PX.add) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = y[k] + oa[k];
    }
  } else if constexpr (affine) {
    const bool act_last = (P.mode & 2) != 0;
    if (!act_last) wn5_act16(y, P.act);
    if (P.clamp_from >= 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c + q >= P.clamp_from) y[k][q] = fminf(fmaxf(y[k][q], P.clamp_lo), P.clamp_hi);
    }
    if (PX.add) {
      if (PX.add_scale) {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = __builtin_elementwise_fma(oa[k], as, y[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = y[k] + oa[k];
      }
    }
    if (act_last) wn5_act16(y, P.act);
    if (PX.out2) {
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
  if (affine && !DIL && P.pool2) {      // block-uniform: the tile IS the 2x2 pooling window (H, W even: a tile is whole or absent)
    const size_t pimg = (size_t)(H >> 1) * (W >> 1);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(PX.out + (size_t)img * pimg * P.out_cs + P.out_co,
                                                n_span * pimg * P.out_cs * sizeof(float) - (size_t)P.out_co * sizeof(float));
    const unsigned ppix = (unsigned)(ty * (W >> 1) + tx) + (run_e ? (unsigned)pimg : 0u);
    const f32x4 m = __builtin_elementwise_max(__builtin_elementwise_max(y[0], y[1]), __builtin_elementwise_max(y[2], y[3]));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, m), rs,
                                           okk[0] ? (int)(ppix * (unsigned)P.out_cs + (unsigned)c_ld) * 4 : OOB, 0, 0);
  } else {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(PX.out + img_base * P.out_cs + P.out_co, img_span * P.out_cs * sizeof(float) - (size_t)P.out_co * sizeof(float));
    const int v = (int)(pix * (unsigned)P.out_cs + (unsigned)c_ld) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, y[k]), rs, okk[k] ? v : OOB, pk_[k] * P.out_cs * 4, 0);
  }
  if (affine && PX.out2) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(PX.out2 + img_base * P.out2_cs, img_span * P.out2_cs * sizeof(float));
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

// weights: packed direct form w[cout_pad][kh * kw * cin_pad] (tap-major, channel-minor) -> U[group][cin_pad/16][16][cout_pad][16] = G g G^T.
// 3x3: one group.  7x7 (round 6, the small-P kernel's Winograd block only): the kernel sits in a 9x9 frame of zeros and is cut into 3 x 3
// sub-kernels of 3x3 — group (a, b) = rows 3a .. 3a+2, columns 3b .. 3b+2 of the frame, applied to the input shifted by (3a - 3, 3b - 3) —
// so that the layer is nine F(2x2, 3x3) convolutions accumulating into the same Winograd-domain sums: 9 x 16 = 144 products per 2x2 outputs
// and (cin, cout) pair instead of 4 x 49 = 196.
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int cout_pad, int cin_pad, int ksz) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (group, co, ci)
  const int ngrp = ksz == 7 ? 9 : 1;
  if (idx >= (long)ngrp * cout_pad * cin_pad) return;
  const int grp = (int)(idx / ((long)cout_pad * cin_pad));
  const long rem = idx - (long)grp * cout_pad * cin_pad;
  const int co = (int)(rem / cin_pad), ci = (int)(rem - (long)co * cin_pad);
  const int ga = grp / 3, gb = grp - 3 * ga;
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int ky = ksz == 7 ? 3 * ga + a - 1 : a, kx = ksz == 7 ? 3 * gb + b - 1 : b;
      g[a][b] = (ky >= 0 && ky < ksz && kx >= 0 && kx < ksz) ? w[(size_t)co * ksz * ksz * cin_pad + (size_t)(ky * ksz + kx) * cin_pad + ci] : 0.f;
    }
  // G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    t[3][b] = g[2][b];
  }
  const int kc = ci >> 4, cl = ci & 15, nkc = cin_pad >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]), u3 = t[i][2];
    const float uu[4] = {u0, u1, u2, u3};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) U[((((size_t)grp * nkc + kc) * 16 + i * 4 + jj) * cout_pad + co) * 16 + cl] = uu[jj];
  }
}
hipError_t launch_wino_weights(const float* w, float* U, int cout_pad, int cin_pad, int ksz, hipStream_t stream) {
  const long n = (long)(ksz == 7 ? 9 : 1) * cout_pad * cin_pad;
  hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, U, cout_pad, cin_pad, ksz);
  return hipGetLastError();
}

// what the kernel takes: 3x3, stride 1, pad 1, no dilation / upsampling / gather / gate / SE scale / split-K / channel sums; inputs
// in whole 16-channel chunks; transformed weights present; images of at least one workgroup tile
// EPI_LNG (round 6): the 7x7 / pad-3 + LayerNorm + GELU layer of the batched cells as nine 3x3 tap groups (GRP = 9): 64 output channels = one
// cout block (the LayerNorm needs a pixel's channels in one workgroup), LayerNorm on, plain inputs, [+ residual]
static bool wino_takes_ln7(const ConvProblem& q) {
  static const int on = [] { const char* v = std::getenv("SF_WINO_LN7"); return v ? std::atoi(v) : 1; }();
  if (!on || !q.w_wino || q.KH != 7 || q.KW != 7 || q.stride != 1 || q.dil != 1 || q.pad != 3 || q.in_up || q.gather || q.gate || q.se_sum || q.in_scale ||
      q.nsplit > 1 || q.chansum || q.acc_in || q.fuse_w || q.out_planar || q.pool2 || q.add_up || q.add_scale || q.bias_per_img || q.out2)
    return false;
  if (q.cout != 64 || q.cout_pad != 64 || (q.c0 % 16) || (q.c1 % 16) || q.c0 + q.c1 != q.cin_pad || (q.cin_pad % 32)) return false;
  if (q.Hout != q.Hin || q.Wout != q.Win || q.Hout < 16 || q.Wout < 32) return false;
  // one cout block and 9 x cin/16 chunks per workgroup: the launch needs a full round of workgroups (2 x 256 blocks of 32 tiles = 65536 pixels) to
  // pay — 8 latents of 50x50 are 175 workgroups of ~200 us each and LOSE to the direct form (batch-8 step 549 -> 563 us, profiles/r06_ln7_ab.txt)
  static const double min_p = [] { const char* v = std::getenv("SF_WINO_LN7_MIN_P"); return v ? std::atof(v) : 65536.0; }();
  if ((double)q.n_img * q.Hout * q.Wout < min_p) return false;
  const double img_bytes = 4.0 * q.Hin * q.Win;
  if (img_bytes * q.out_cs >= 2147483648.0 || img_bytes * q.add_cs >= 2147483648.0 || img_bytes * q.in0_cs >= 2147483648.0 ||
      img_bytes * q.in1_cs >= 2147483648.0 || 4.0 * 9 * 16 * q.cout_pad * q.cin_pad >= 2147483648.0)
    return false;
  return true;
}
bool wino_takes(const ConvProblem& q, int epi) {
  if (epi == EPI_LNG) return wino_takes_ln7(q);
  if (epi == EPI_SAMPLE) {      // the sampling layer of the batched infer_state (round 6): the plain AFFINE rules + what its epilogue addresses
    static const int on = [] { const char* v = std::getenv("SF_WINO_SAMPLE"); return v ? std::atoi(v) : 1; }();
    const double img_px = (double)q.Hin * q.Win;
    if (!on || q.dil != 1 || q.in_up || q.add || q.add_scale || q.pool2 || q.add_up || q.bias_per_img || q.scale || (q.cout % 8) || (q.c1 != 0) ||
        2.0 * 4.0 * img_px * q.cout >= 2147483648.0 || (!q.e0 && !q.philox))
      return false;
    return wino_takes(q, EPI_AFFINE);      // (out2 here is the q tensor: no gate_from semantics — the kernel's SAMPLE branch never reads it as a gate)
  }
  if (epi != EPI_AFFINE && epi != EPI_BLEND) return false;
  if (!q.w_wino || q.KH != 3 || q.KW != 3 || q.stride != 1 || q.dil < 1 || q.pad != q.dil || (q.in_up && q.dil != 1) || q.gather || q.gate || q.se_sum ||
      (q.in_scale && (epi != EPI_AFFINE || q.dil != 1 || q.c0 > 256)) ||      // SE-scaled input: plain AFFINE form, scales staged in LDS
      q.nsplit > 1 || q.chansum || q.acc_in || q.fuse_w || q.out_planar || (epi == EPI_AFFINE && (q.mode & 4)))
    return false;
  if (q.pool2 && (epi != EPI_AFFINE || q.dil != 1 || q.out2 || (q.Hout & 1) || (q.Wout & 1))) return false;
  if (q.add_up && (epi != EPI_AFFINE || q.dil != 1 || !q.add || (q.Hout & 1) || (q.Wout & 1))) return false;
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

// CAT pays where blocks of 8 tile columns fit the image badly and no epilogue operand is per image (SF_WINO_CAT=0: never)
static bool wino_cat(const ConvProblem& q) {
  static const int on = [] { const char* v = std::getenv("SF_WINO_CAT"); return v ? std::atoi(v) : 1; }();
  const int tpi = (q.Wout + 1) / 2;
  static const int cat_scaled = [] { const char* v = std::getenv("SF_WINO_CAT_SCALED"); return v ? std::atoi(v) : 1; }();      // round 6: SE scales of both images of a block
  if (!on || q.dil != 1 || q.in_up || q.n_img < 2 || tpi < 8 || q.bias_per_img || ((q.in_scale || q.add_scale) && !cat_scaled)) return false;
  const int plain = (tpi + 7) / 8 * 8;
  if (plain * 100 < tpi * 110) return false;      // less than 10 % empty columns: keep the plain form
  const double img_bytes = 4.0 * q.Hin * q.Win;
  const int cs = q.in0_cs > q.in1_cs ? q.in0_cs : q.in1_cs;
  const int co = q.out_cs > q.add_cs ? q.out_cs : q.add_cs;
  const int ce = q.e0_cs > q.e1_cs ? q.e0_cs : q.e1_cs;
  const int cm = co > ce ? (co > q.out2_cs ? co : q.out2_cs) : (ce > q.out2_cs ? ce : q.out2_cs);
  return 2.0 * img_bytes * cs < 2147483648.0 && 2.0 * img_bytes * cm < 2147483648.0;      // two images behind one base
}
// which form of the kernel a problem runs on: 2 = plain, 3 = dilated, 4 = images concatenated along x; -1: none
int wino_variant(const ConvProblem& q) {
  if (q.cout_pad % 64) return -1;
  if (q.dil > 1) return 3;
  if (wino_cat(q)) return 4;
  return 2;
}
// Winograd tiles a launch executes (the profiler prices 16 products per tile and (cin, cout) pair)
double wino_tiles(const ConvProblem& q) {
  if (q.dil > 1) return (double)q.n_img * WnAxis(q.Hout, q.dil).nt * WnAxis(q.Wout, q.dil).nt;
  return (double)q.n_img * ((q.Hout + 1) / 2) * ((q.Wout + 1) / 2);
}
// workgroups a launch of the group would have with 32-tile blocks (the choice between the two block sizes)
static long wino5_wgs32(const ConvLaunch& L, bool cat) {
  const ConvProblem& P = L.p[0];
  const int tiles_x = (P.Wout + 1) / 2, tiles_y = (P.Hout + 1) / 2;
  const long blocks = cat ? (long)((tiles_y + 3) / 4) * (((long)P.n_img * tiles_x + 7) / 8) : (long)P.n_img * ((tiles_y + 3) / 4) * ((tiles_x + 7) / 8);
  return ((blocks + 7) / 8) * 8 * (P.cout_pad / 64) * L.nprob;
}
template <int EPI, bool DIL = false, bool CAT = false, int TH_ = 4, int GRP = 1>
static hipError_t launch_wino5_t(const ConvLaunch& L, hipStream_t stream) {
  typedef Wino5Geo<DIL, CAT, TH_> G;
  auto kern = conv_wino5_kernel<EPI, DIL, CAT, TH_, GRP>;
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
  const long grid1 = ((blocks + 7) / 8) * 8 * (P.cout_pad / G::COUT_T), grid = grid1 * L.nprob;
  if (grid > 0x7fffffffL) return hipErrorInvalidValue;
  ConvLaunch L2 = L;
  L2.wg_base[0] = 0;
  L2.wg_base[1] = (int)grid1;
  // reciprocals of the block decode's divisors (conv_wino5_kernel): ceil(2^32 / d), 0 for d = 1; exact while dividend x d < 2^32
  const long nbx = ((CAT ? (long)P.n_img * tiles_x : tiles_x) + G::TW - 1) / G::TW, nby = (tiles_y + G::TH - 1) / G::TH, ncb = P.cout_pad / G::COUT_T;
  auto magic = [](long d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); };
  if ((grid1 / 8 + 1) * ncb >= 0x100000000L || (blocks + 8) * nbx >= 0x100000000L || (blocks + 8) * nby >= 0x100000000L ||
      (nbx * G::TW + G::TW) * tiles_x >= 0x100000000L)
    return hipErrorInvalidValue;
  if (L.nprob > 1 && grid * grid1 >= 0x100000000L) return hipErrorInvalidValue;      // (callers launch such groups one by one)
  L2.wn_m[0] = L.nprob > 1 ? magic(grid1) : 1u;
  L2.wn_m[1] = magic(ncb); L2.wn_m[2] = magic(nbx); L2.wn_m[3] = magic(nby); L2.wn_m[4] = magic(tiles_x);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, 1, 1), dim3(WN_THREADS), lds, stream, L2);
  return hipGetLastError();
}
// same tiling, same kernel instantiation, same workgroup count: the layers may share a launch
bool wino_same_geometry(const ConvProblem& a, const ConvProblem& b) {
  return a.n_img == b.n_img && a.Hin == b.Hin && a.Win == b.Win && a.Hout == b.Hout && a.Wout == b.Wout && a.in_up == b.in_up && a.dil == b.dil &&
         a.c0 == b.c0 && a.c1 == b.c1 && a.cin_pad == b.cin_pad && a.cout == b.cout && a.cout_pad == b.cout_pad &&
         (a.in_scale != nullptr) == (b.in_scale != nullptr) && (a.in1 != nullptr) == (b.in1 != nullptr) && (a.add != nullptr) == (b.add != nullptr) &&
         (a.add_scale != nullptr) == (b.add_scale != nullptr) && (a.out2 != nullptr) == (b.out2 != nullptr) && (a.scale != nullptr) == (b.scale != nullptr) &&
         (a.bias != nullptr) == (b.bias != nullptr) && a.in0_cs == b.in0_cs && a.in1_cs == b.in1_cs && a.add_cs == b.add_cs && a.out_cs == b.out_cs &&
         a.out_co == b.out_co && a.out2_cs == b.out2_cs && a.e0_cs == b.e0_cs && a.e1_cs == b.e1_cs && a.act == b.act && a.mode == b.mode &&
         a.gate_from == b.gate_from && a.clamp_from == b.clamp_from && a.clamp_lo == b.clamp_lo && a.clamp_hi == b.clamp_hi &&
         a.bias_per_img == b.bias_per_img && a.pool2 == b.pool2 && a.add_up == b.add_up && wino_variant(a) == wino_variant(b);
}
// one problem per launch, or up to SF_MAX_GROUP of identical geometry
hipError_t launch_conv_wino(const ConvLaunch& L, int epi, hipStream_t stream) {
  if (L.nprob < 1 || L.nprob > SF_MAX_GROUP) return hipErrorInvalidValue;
  for (int i = 0; i < L.nprob; ++i)
    if (!wino_takes(L.p[i], epi) || !wino_same_geometry(L.p[0], L.p[i])) return hipErrorInvalidValue;
  const bool affine = epi == EPI_AFFINE;
  const int var = wino_variant(L.p[0]);
  if (epi == EPI_SAMPLE) {
    if (var == 2) return launch_wino5_t<EPI_SAMPLE>(L, stream);
    if (var == 4) return launch_wino5_t<EPI_SAMPLE, false, true>(L, stream);
    return hipErrorInvalidValue;
  }
  if (epi == EPI_LNG) {
    if (var == 2) return launch_wino5_t<EPI_LNG, false, false, 4, 9>(L, stream);
    if (var == 4) return launch_wino5_t<EPI_LNG, false, true, 4, 9>(L, stream);
    return hipErrorInvalidValue;
  }
  // 16-tile blocks where 32-tile blocks would leave the launch below SF_WINO_SMALL_WGS workgroups (two rounds of the chip's 512 slots; 0: never).
  // The 16-tile form runs THREE workgroups per CU (80 registers, 46 KB of LDS), so a launch of 313 / 626 32-tile workgroups becomes 626 / 1252
  // of 768 slots.  Measured per threshold (profiles/r06_wino16_ab.txt, single-sample forward): BLEND launches 0.608 -> 0.514 ms, AFFINE 2.880 ->
  // 2.841, forward 7.44 -> 7.27 ms; at two samples per forward 13.18 -> 13.14; above ~1 400 workgroups the 32-tile form wins (every workgroup
  // loads the whole U of its 64 output channels whatever its tile count: 16 tiles double the load instructions per MFMA)
  static const long small_wgs = [] { const char* v = std::getenv("SF_WINO_SMALL_WGS"); return v ? std::atol(v) : 1000L; }();
  const bool small = (var == 2 || var == 4) && wino5_wgs32(L, var == 4) < small_wgs && !(var == 4 && L.p[0].in_scale);      // (SE-scaled CAT: 32-tile form only)
  switch (var) {
    case 2:
      if (small) return affine ? launch_wino5_t<EPI_AFFINE, false, false, 2>(L, stream) : launch_wino5_t<EPI_BLEND, false, false, 2>(L, stream);
      return affine ? launch_wino5_t<EPI_AFFINE>(L, stream) : launch_wino5_t<EPI_BLEND>(L, stream);
    case 3: return affine ? launch_wino5_t<EPI_AFFINE, true>(L, stream) : hipErrorInvalidValue;
    case 4:
      if (small) return affine ? launch_wino5_t<EPI_AFFINE, false, true, 2>(L, stream) : launch_wino5_t<EPI_BLEND, false, true, 2>(L, stream);
      return affine ? launch_wino5_t<EPI_AFFINE, false, true>(L, stream) : launch_wino5_t<EPI_BLEND, false, true>(L, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace sf
