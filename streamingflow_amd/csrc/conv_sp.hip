// Small-pixel-count implicit-GEMM convolution for the GRU-ODE latent: the kernel behind every layer of ode_step / the
// Bayesian jump of ONE 50x50 sample (P = n_img*Hout*Wout < 4096; from two samples on the large-tile kernels of
// conv_igemm.hip are as fast or faster).  gfx950 only.
//
// Why a second kernel.  A 2500-pixel layer has 40-80 tiles, fewer than the chip has CUs, and a step is a chain of ten
// such layers: what counts is the latency of one workgroup on one CU.  In-kernel time stamps of the round-1 small-P kernels
// (tools/r02/stamps.py) showed where that went: every wave issued its own LDS-DMAs (~550 cycles of address arithmetic +
// issue per 32-deep chunk), owned ONE 16x16 accumulator, one wave ran the whole epilogue behind a chain of dependent
// global loads (4-7 us for the LayerNorm / trusting-gate tails), and a cross-workgroup split-K hand-off cost 5-14 us in
// the agent-scope release fence.  Structure here, one 768-thread workgroup per CU:
//   * roles: waves 8-11 are LOADERS (all im2col address arithmetic + buffer_load ... lds issue), waves 0-7 CONSUMERS
//     (ds_read_b128 + MFMA only).  Two consumers and one loader share each SIMD.
//   * tile 64 cout x 32 px (NT = 2) or 64 x 64 px (NT = 4); a K chunk is 64 deep (two 32-deep sub-chunks in the packed K
//     order, each a [64 + BN rows][32] block with the 16-B slots XOR-swizzled by (row>>1)&7 through the DMA's per-lane
//     SOURCE address); ring of 3 chunk buffers, 2 chunks in flight, one s_barrier per chunk placed between a chunk's
//     fragment reads and its MFMAs.
//   * consumer (mh, kq): cout half mh, K quarter kq of every chunk (16 K values = one ds_read_b128 per 16-row
//     fragment): 2 + NT fragment reads feed 8 NT MFMAs.  In-workgroup split-K over the 4 quarters, reduced through LDS in
//     a fixed order.
//   * distributed epilogue: after the reduction lane (pixel, channel quad) of the 512 consumer lanes owns ONE float4 of
//     the tile; LayerNorm / softmax statistics are DPP sums over 16-lane rows; every epilogue operand is requested before
//     the K loop.
//   * cross-workgroup split-K (NT = 4): partial tiles leave with sc1 (write-through) stores, one agent-scope ticket per
//     workgroup, the last arriver re-reads all slices with sc1 loads in slice order (MI355X_MICROARCH.md, hand-off table
//     row 1): no release / acquire fence.  Bitwise reproducible.  Compact 1-D grid of at most 256 workgroups.
//   * the SE gate of the input (one image) is computed in the prologue from the producer's per-tile channel sums.
// What bounds it: the fp32 MFMA shares the vector ALU — every DMA instruction costs ~60 cycles of matrix
// time, every VALU instruction 4-8; the shipped loop runs at ~70 % of the MFMA rate on 64x64 tiles.
#include "sf_math.h"

#include <cstring>
#include <type_traits>

namespace sf {

#ifdef SF_STAMP
__device__ unsigned long long* g_sf_stamps = nullptr;     // diagnostic builds: one copy per translation unit (no relocatable device code)
hipError_t set_stamp_buffer_sp(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sf_stamps), &p, sizeof(p)); }
#else
hipError_t set_stamp_buffer_sp(unsigned long long*) { return hipErrorNotSupported; }
#endif

constexpr int SP_BM = 64;                     // output channels per tile
// n / d with the host's ceil(2^32 / d) (exact for n x d < 2^32, which the host checks); m == 0: d is 1, or no reciprocal was made
__device__ __forceinline__ int sp_mdiv(const int n, const unsigned m, const int d) { return m ? (int)__umulhi((unsigned)n, m) : (d == 1 ? n : n / d); }

#if !defined(SF_SP_PIN)
#define SF_SP_PIN 1
#endif
constexpr int SP_NB = 3, SP_LA = SP_NB - 1;   // chunk buffers of the LDS ring / chunks in flight
constexpr int SP_RED_PITCH = 68;              // reduction buffer [4 quarters][BN px][68] (inside the ring)
constexpr int SP_SC_IMGS = 4;                 // SE scale rows kept in LDS: images a pixel tile can touch
constexpr int SP_SC_FLOATS = SP_SC_IMGS * 256 + 1280;   // + scratch of the in-kernel SE gate (6 partial rows + mean + hidden)
constexpr int SP_MISC = 64;                   // hand-off flag
constexpr int SP_THREADS = 768;               // 8 consumer + 4 loader waves
// NT = 16-pixel n-tiles per consumer wave: the tile is 64 cout x (16 NT) px
template <int NT>
struct SpGeo {
  static constexpr int BN = 16 * NT;
  static constexpr int ROWS = SP_BM + BN;     // rows of one sub-chunk block (weights first)
  static constexpr int SUBF = ROWS * 32;      // floats per sub-chunk block
  static constexpr int BUFF = 2 * SUBF;       // floats per chunk buffer
  static constexpr int RING = SP_NB * BUFF;   // NT 2: 72 KB, NT 4: 96 KB
  static constexpr int NBI = BN / 32;         // pixel row blocks (8 rows) per loader and sub-chunk
#if defined(SF_ABL_NO_PIXEL_DMA)      // timing-only ablation (results are garbage): the loaders issue the weight DMAs only — what a pixel operand
  static constexpr int DPC = 2 * 2;           // that is LDS-resident could save at most (tools/r04/abl_pixel_dma.sh)
#elif defined(SF_ABL_NO_WEIGHT_DMA)   // timing-only (garbage results): the pixel DMAs only — what weights that never had to be fetched per workgroup (one
  static constexpr int DPC = 2 * NBI;         // copy per XCD L2, a weight-stationary slice) could save at most (tools/r05/abl_weight_dma.sh)
#else
  static constexpr int DPC = 2 * (2 + NBI);   // DMA instructions per loader and chunk
#endif
  static constexpr int NPX = BN / 32;         // epilogue items (pixel, channel quad) per consumer lane
  static_assert(4 * BN * SP_RED_PITCH + 512 <= RING, "reduction buffer + channel-sum scratch live in the ring");
};
template <int NT>
constexpr int sp_lds_bytes(bool scale) { return (SpGeo<NT>::RING + SP_MISC + (scale ? SP_SC_FLOATS : 0)) * 4; }
// Winograd F(2x2, 3x3) form of a 3x3 layer — and of the trusting gate's 7x7 as nine 3x3 sub-kernels — on one latent (ConvProblem::sp_wino; see the block in sp_body): a 64-pixel tile is 16
// Winograd tiles; per 32-channel sub-chunk the gathered 4x4 patches [16 tiles][16 px][32] and their transform V[16 positions][16 tiles][32],
// two buffers each; the products M[16 positions][16 tiles][SP_RED_PITCH] land over both after the loop
constexpr int SPW_SUB = 16 * 16 * 32;             // floats of one raw / one V buffer (32 KB)
constexpr int SPW_REGION = 4 * SPW_SUB;           // 128 KB
static_assert(16 * 16 * SP_RED_PITCH + 512 <= SPW_REGION, "M + channel-sum scratch live in the region");
// which instantiations carry the Winograd block: 64-pixel tiles, exact fp32, launch path (the flow kernel keeps the direct form), no fused 1x1
template <int EPI, int NT, bool B3, bool PST>
constexpr bool sp_has_wino() { return NT == 4 && !B3 && !PST; }
// floats in front of misc / SE rows / fused-layer buffer: a problem in the Winograd form has the two raw + two V buffers there, a problem in
// the direct form (also in a kernel that carries the block) the ring
template <int EPI, int NT, bool B3, bool PST>
constexpr int sp_region() { return sp_has_wino<EPI, NT, B3, PST>() && SPW_REGION > SpGeo<NT>::RING ? SPW_REGION : SpGeo<NT>::RING; }

#if defined(SF_ABL_NO_WEIGHT_DMA)
#define SP_ABL_NO_W 1
#else
#define SP_ABL_NO_W 0
#endif
typedef __attribute__((address_space(3))) void sp_lds_void;


__device__ __forceinline__ f32x4 sp_lds_read128(const float* p) {
  typedef const __attribute__((address_space(3))) f32x4 lds_f4;
  return *(lds_f4*)p;   // explicit LDS address space: ds_read_b128
}
// raw workgroup barrier fenced for the compiler (no LDS access may move across it); the callers drain their own counters
__device__ __forceinline__ void sp_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// Sum over the 64 lanes of a wave, valid in LANE 63 only: DPP within the 16-lane rows (quad swaps, half mirror, row mirror), then
// row_bcast15 / row_bcast31 across the rows — (r0 + r1) + (r2 + r3), the order of the v_readlane form this replaces (bitwise the
// same sum), all VALU, no LDS round trips (six ds_bpermute steps cost ~1500 cycles in the SE prologue) and no scalar registers
// (four v_readlane results per sum were what spilled SGPRs in the flow kernel's SE prologue).
__device__ __forceinline__ float sp_wave_sum_lane63(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v = spm_row16_sum(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));      // row_bcast15 into rows 1, 3
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));      // row_bcast31 into rows 2, 3
#endif
  return v;
}
__device__ __forceinline__ float sp_reduce16(float v) { return spm_row16_sum(v); }   // sum over the 16 lanes (channel quads) of a pixel

// Global stores of results.  PST (the persistent segment kernel, sp_segment_kernel): write-through (sc1) stores, so that a
// workgroup of a LATER phase of the same launch — on any XCD — finds the bytes behind the phase counter + its acquire
// (MI355X_MICROARCH.md, hand-off forms: every handed-off byte stored sc1, every storing wave drained before the signal).
template <bool PST>
__device__ __forceinline__ void sp_gst4(float* base, const size_t off, const float4 v) {      // base: block-uniform tensor pointer, off: this lane's element offset
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SF_PST_PLAIN)
  if constexpr (PST) {
    // a store the compiler knows (its hazard recognizer and wait counters see it; an inline-asm global_store_dwordx4 sc1 here
    // gave wrong tiles — tests/test_gpu_persistent.py): a volatile store to the GLOBAL address space is emitted as
    // global_store_dwordx4 ... sc0 sc1 (write-through at system scope: the line leaves this XCD's L2, as with sc1 alone).  Round 3
    // used raw buffer stores with aux = sc1: four more scalar registers per store for the resource, which the flow kernel's epilogues
    // do not have (81 SGPR spills with them, none without).
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) volatile f32x4v* gptr;
    *(gptr)(base + off) = (f32x4v){v.x, v.y, v.z, v.w};
  } else {
    spm_st4(base + off, v);
  }
#else
  spm_st4(base + off, v);
#endif
}
template <bool PST>
__device__ __forceinline__ void sp_gst2(float* p, const float2 v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SF_PST_PLAIN) && !defined(SF_PST_PLAIN2)
  if constexpr (PST) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const f32x2v t = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
  } else {
    *reinterpret_cast<float2*>(p) = v;
  }
#else
  *reinterpret_cast<float2*>(p) = v;
#endif
}
template <bool PST>
__device__ __forceinline__ void sp_gst1(float* p, const float v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SF_PST_PLAIN) && !defined(SF_PST_PLAIN1)
  if constexpr (PST) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else *p = v;
#else
  *p = v;
#endif
}

// Flow mode, build flag SF_FLOW_SC1 (default 1): every load of bytes another workgroup of the same launch may have written — pixel
// DMAs, epilogue operands, SE sums — bypasses this CU's L1 (sc0 sc1 / sc1: served by L2, which the hardware keeps coherent with the
// write-through stores of the producers), so an item needs NO agent-scope acquire behind its dependency wait (the buffer_inv sc1
// took 0.8-1.3 us of every item: profiles/r04_c_flow_stamps.txt).  MI355X_MICROARCH.md lists this form for global_/buffer_ loads to
// registers (hand-off table, row 3); for LDS-DMA it is OBSERVED here — tests/test_gpu_persistent.py compares every rollout bit for
// bit with the launch-per-layer path, with cache sweeps in between.  SF_FLOW_SC1=0 builds the acquire form (plain loads).
#ifndef SF_FLOW_SC1
#define SF_FLOW_SC1 1
#endif
template <bool VOL>
__device__ __forceinline__ float4 sp_gld4(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (VOL) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) volatile f32x4v* gptr;      // global_load_dwordx4 ... sc0 sc1
    const f32x4v t = *(gptr)p;
    return make_float4(t.x, t.y, t.z, t.w);
  }
#endif
  return spm_ld4(p);
}
template <bool VOL>
__device__ __forceinline__ float sp_gld1(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (VOL) return *(const __attribute__((address_space(1))) volatile float*)p;
#endif
  return *p;
}

// Epilogue operands of one (pixel, channel quad) item.  They are loaded by the consumer waves BEFORE the K loop (every one of
// them was written by an earlier launch), so that the epilogue is arithmetic + stores only: measured 1.1-2.8 us per launch
// when the loads sat behind the reduction (tools/r02/stamps.py).
struct SpOps {
  float4 a[10];
  float4 pre;      // K-partial sum an earlier launch left (ConvProblem::acc_in), zero without one
  float2 e;
  float c0f, c1f;
};
// `on`: the lane owns a real element (consumer wave, pixel < P, channel < cout); lanes that are not `on` load from safe
// addresses and store nothing.
// PART: 0 = everything; 1 = what depends on the channel only (the same for every pixel of the lane: the Winograd block fetches it once,
// before its loop); 2 = what depends on the pixel
template <int EPI, bool VOL = false, int PART = 0, class PT = ConvProblem>
__device__ __forceinline__ void sp_epi_load(const PT& P, const int gp, const int c, const bool on, const int HWout, SpOps& o, const bool one_img = false) {
  constexpr bool CH = PART != 2, PX = PART != 1;
  const int img = (on && !one_img) ? sp_mdiv(gp, P.sp_m_hw, HWout) : 0;
  // element offsets in 32 bits (the small-P kernel: < 4096 pixels, strides of a few hundred channels): a uniform base + a zero-extended
  // 32-bit lane offset is one multiply-add and the scalar-base form of the load; size_t arithmetic was ten vector instructions per load
  const unsigned gpz = on ? (unsigned)gp : 0u;
  const unsigned cz = on ? (unsigned)c : 0u;
  const unsigned ucout = (unsigned)P.cout;
  if (PX) o.pre = P.acc_in ? sp_gld4<VOL>(P.acc_in + (gpz * (unsigned)P.acc_cs + cz)) : spm_zero4();
  if constexpr (EPI == EPI_AFFINE || EPI == EPI_BLEND) {
    if (CH) {
      o.a[0] = P.scale ? sp_gld4<VOL>(P.scale + cz) : make_float4(1.f, 1.f, 1.f, 1.f);
      o.a[1] = P.bias ? sp_gld4<VOL>(P.bias + ((P.bias_per_img ? (unsigned)img * ucout : 0u) + cz)) : spm_zero4();
    }
    if constexpr (EPI == EPI_AFFINE) {
      // (values, not fields, are chosen under the branches: with `o.a[k] = ...` on both sides hipcc merged the stores into ONE store through a
      // selected address and the whole operand record moved to scratch memory — 416 bytes per lane, the step 151 -> 217 us)
      const bool blend = (P.mode & 4) != 0;      // block-uniform: conv-GRU blend inside an AFFINE launch (a candidate grouped with plain layers)
      float4 a3 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (PART == 2) a3 = o.a[3];
      else if (P.add && P.add_scale) a3 = sp_gld4<VOL>(P.add_scale + ((unsigned)img * ucout + cz));
      if (PX) {
        float4 a2 = spm_zero4();
        if (blend) {
          a2 = sp_gld4<VOL>(P.e0 + (gpz * (unsigned)P.e0_cs + cz));
          a3 = sp_gld4<VOL>(P.e1 + (gpz * (unsigned)P.e1_cs + cz));
        } else if (P.add) {
          a2 = sp_gld4<VOL>(P.add + (gpz * (unsigned)P.add_cs + cz));
        }
        o.a[2] = a2;
        const unsigned cg = (P.out2 && (int)cz >= P.gate_from) ? cz - (unsigned)P.gate_from : 0u;
        o.a[4] = P.out2 ? sp_gld4<VOL>(P.e1 + (gpz * (unsigned)P.e1_cs + cg)) : spm_zero4();
      }
      o.a[3] = a3;
    } else if (PX) {
      o.a[2] = sp_gld4<VOL>(P.e0 + (gpz * (unsigned)P.e0_cs + cz));
      o.a[3] = sp_gld4<VOL>(P.e1 + (gpz * (unsigned)P.e1_cs + cz));
    }
  }
  if constexpr (EPI == EPI_LNG || EPI == EPI_TRUST) {
    const unsigned po = gpz * ucout + cz;
    const bool do_ln = (EPI == EPI_TRUST) || (P.mode & 1);      // a plain GELU layer (the gate's 1x1 projection) has no LayerNorm parameters
    if (CH) {
      o.a[0] = do_ln ? sp_gld4<VOL>(P.scale + cz) : spm_zero4();
      o.a[1] = do_ln ? sp_gld4<VOL>(P.bias + cz) : spm_zero4();
      o.c0f = 0.f; o.c1f = 0.f;
    }
    if constexpr (EPI == EPI_LNG) {
      if (PX) o.a[2] = P.add ? sp_gld4<VOL>(P.add + (gpz * (unsigned)P.add_cs + cz)) : spm_zero4();
    }
    if constexpr (EPI == EPI_TRUST) {
      if (CH) {
        o.a[3] = sp_gld4<VOL>(P.e1 + cz); o.a[4] = sp_gld4<VOL>(P.e1 + (ucout + cz));
        const float* cf = P.coef ? P.coef + (size_t)img * P.coef_stride : nullptr;
        o.c0f = cf ? cf[0] : 0.f;
        o.c1f = (cf && P.out2) ? cf[1] : 0.f;
      }
      if (PX) {
        o.a[2] = sp_gld4<VOL>(P.e0 + po);
        o.a[5] = sp_gld4<VOL>(P.e2 + po); o.a[6] = sp_gld4<VOL>(P.e3 + po);
        const bool deriv = (P.mode & 1) != 0;
        o.a[7] = deriv ? sp_gld4<VOL>(P.e4 + po) : spm_zero4();
        o.a[8] = deriv ? sp_gld4<VOL>(P.e5 + po) : spm_zero4();
        o.a[9] = (P.out2 && (P.mode & 2)) ? sp_gld4<VOL>(P.out2 + po) : spm_zero4();
      }
    }
  }
  if constexpr (EPI == EPI_SAMPLE) {
    const unsigned Chalf = ucout >> 1;
    const unsigned ch = (((unsigned)c >> 4) << 3) + 2 * (((unsigned)c >> 2) & 3);
    const bool ok = on && ch < Chalf;
    if (CH) o.a[0] = P.bias ? sp_gld4<VOL>(P.bias + cz) : spm_zero4();
    if (PX) o.e = P.e0 ? *reinterpret_cast<const float2*>(P.e0 + ((ok ? gpz : 0u) * Chalf + (ok ? ch : 0u))) : make_float2(0.f, 0.f);      // eps: an input of the call
  }
}
// the channel-only operands of one item handed to another item of the same lane (same channel quad)
template <int EPI>
__device__ __forceinline__ void sp_epi_copy_chan(const SpOps& s, SpOps& d) {
  d.a[0] = s.a[0];
  if constexpr (EPI != EPI_SAMPLE) d.a[1] = s.a[1];
  if constexpr (EPI == EPI_AFFINE) d.a[3] = s.a[3];      // (blend mode: the pixel part overwrites it afterwards)
  if constexpr (EPI == EPI_TRUST) { d.a[3] = s.a[3]; d.a[4] = s.a[4]; d.c0f = s.c0f; d.c1f = s.c1f; }
  if constexpr (EPI == EPI_LNG) { d.c0f = s.c0f; d.c1f = s.c1f; }
}

// ---- epilogues in the (pixel, channel-quad) layout -------------------------------------------------------------------
// v: the lane's four consecutive output channels c..c+3 of pixel gp (pre-activation accumulator sums).
// y_out (AFFINE): the stored value, zero where not `on` (for the SE channel sums).
template <int EPI, bool PST = false, class PT = ConvProblem>
__device__ __forceinline__ void sp_epilogue(const PT& P, float4 v, const int gp, const int c, const bool on,
                                            const SpOps& o, float4& y_out) {
  const size_t gpz = on ? (size_t)gp : 0;
  const int cz = on ? c : 0;

  if constexpr (EPI == EPI_AFFINE || EPI == EPI_BLEND) {
    const float4 sc = o.a[0], bi = o.a[1];
    v.x = v.x * sc.x + bi.x; v.y = v.y * sc.y + bi.y; v.z = v.z * sc.z + bi.z; v.w = v.w * sc.w + bi.w;
    float4 y;
    if constexpr (EPI == EPI_AFFINE) {
      const float4 ad = o.a[2], as = o.a[3], sv = o.a[4];
      const bool gate_out = P.out2 && cz >= P.gate_from;
      const int cg = gate_out ? cz - P.gate_from : 0;
      const bool act_last = (P.mode & 2) != 0;
      y = act_last ? v : spm_act4(v, P.act);
      if (P.clamp_from >= 0) {
        if (c + 0 >= P.clamp_from) y.x = fminf(fmaxf(y.x, P.clamp_lo), P.clamp_hi);
        if (c + 1 >= P.clamp_from) y.y = fminf(fmaxf(y.y, P.clamp_lo), P.clamp_hi);
        if (c + 2 >= P.clamp_from) y.z = fminf(fmaxf(y.z, P.clamp_lo), P.clamp_hi);
        if (c + 3 >= P.clamp_from) y.w = fminf(fmaxf(y.w, P.clamp_lo), P.clamp_hi);
      }
      if (P.mode & 4) {      // (1 - u) * s + u * h~ with u = e0 (in a[2]), s = e1 (in a[3]): temporal.py:55-57
        y = make_float4((1.f - ad.x) * as.x + ad.x * y.x, (1.f - ad.y) * as.y + ad.y * y.y, (1.f - ad.z) * as.z + ad.z * y.z, (1.f - ad.w) * as.w + ad.w * y.w);
      } else if (P.add) { y.x += ad.x * as.x; y.y += ad.y * as.y; y.z += ad.z * as.z; y.w += ad.w * as.w; }
      if (act_last) y = spm_act4(y, P.act);
      if (on && gate_out)   // GRU gates, reset half: also emit (1 - r) * s, the candidate conv's input
        sp_gst4<PST>(P.out2, gpz * P.out2_cs + cg, make_float4(sv.x * (1.f - y.x), sv.y * (1.f - y.y), sv.z * (1.f - y.z), sv.w * (1.f - y.w)));
    } else {
      const float4 u = o.a[2], s = o.a[3];
      v = spm_act4(v, P.act);
      if (P.mode & 1) y = make_float4(u.x * (v.x - s.x), u.y * (v.y - s.y), u.z * (v.z - s.z), u.w * (v.w - s.w));
      else y = make_float4((1.f - u.x) * s.x + u.x * v.x, (1.f - u.y) * s.y + u.y * v.y, (1.f - u.z) * s.z + u.z * v.z, (1.f - u.w) * s.w + u.w * v.w);
    }
    if (on) sp_gst4<PST>(P.out, gpz * P.out_cs + P.out_co + c, y);
    y_out = on ? y : spm_zero4();
  }

  if constexpr (EPI == EPI_LNG || EPI == EPI_TRUST) {
    // one 64-row tile holds every channel of its pixels (cout_pad <= 64, checked on the host)
    const bool cv = c < P.cout;
    const float inv_c = 1.f / (float)P.cout;
    const size_t po = gpz * P.cout + cz;
    const float4 lw = o.a[0], lb = o.a[1];
    const bool do_ln = (EPI == EPI_TRUST) || (P.mode & 1);
    if (do_ln) {   // convolutions.py:303-308 (channels_first LayerNorm over the pixel's channels)
      const float mean = sp_reduce16(cv ? (v.x + v.y) + (v.z + v.w) : 0.f) * inv_c;
      const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
      const float var = sp_reduce16(cv ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f) * inv_c;
      const float rstd = 1.f / sqrtf(var + P.eps);
      v.x = lw.x * (dx * rstd) + lb.x; v.y = lw.y * (dy * rstd) + lb.y;
      v.z = lw.z * (dz * rstd) + lb.z; v.w = lw.w * (dw * rstd) + lb.w;
    }
    v.x = spm_gelu(v.x); v.y = spm_gelu(v.y); v.z = spm_gelu(v.z); v.w = spm_gelu(v.w);
    if constexpr (EPI == EPI_LNG) {
      const float4 ad = o.a[2];      // Bottleblock residual (zero without one)
      const float4 y = make_float4(v.x + ad.x, v.y + ad.y, v.z + ad.z, v.w + ad.w);
      if (on && P.out) sp_gst4<PST>(P.out, gpz * P.out_cs + P.out_co + c, y);      // no `out`: the tile feeds the fused 1x1 layer only
      y_out = on ? y : spm_zero4();
    } else {
      // trusting gate tail (temporal_ode_bayes.py:124-131 / :268-275, convolutions.py:375-380)
      const float4 sk = o.a[2], w0 = o.a[3], w1 = o.a[4], r2 = o.a[5], r1 = o.a[6], st = o.a[7], base = o.a[8], b2in = o.a[9];
      const float c0f = o.c0f, c1f = o.c1f;
      const float b0 = v.x + sk.x, b1 = v.y + sk.y, b2 = v.z + sk.z, b3 = v.w + sk.w;
      const float z0 = sp_reduce16(cv ? (w0.x * b0 + w0.y * b1) + (w0.z * b2 + w0.w * b3) : 0.f);
      const float z1 = sp_reduce16(cv ? (w1.x * b0 + w1.y * b1) + (w1.z * b2 + w1.w * b3) : 0.f);
      const float zm = fmaxf(z0, z1);
      const float ez0 = expf(z0 - zm), ez1 = expf(z1 - zm);
      const float g0 = ez0 / (ez0 + ez1), g1 = ez1 / (ez0 + ez1);
      float4 cur;
      cur.x = r2.x * g0 + r1.x * g1; cur.y = r2.y * g0 + r1.y * g1;
      cur.z = r2.z * g0 + r1.z * g1; cur.w = r2.w * g0 + r1.w * g1;
      if (on) {
        if (P.mode & 1) {   // derivative: d = cur - s ; out = base + coef0*d
          const float4 d = make_float4(cur.x - st.x, cur.y - st.y, cur.z - st.z, cur.w - st.w);
          if (P.out2) {
            float4 a2 = (P.mode & 2) ? b2in : base;
            a2.x += c1f * d.x; a2.y += c1f * d.y; a2.z += c1f * d.z; a2.w += c1f * d.w;
            sp_gst4<PST>(P.out2, po, a2);
          }
          sp_gst4<PST>(P.out, po, make_float4(base.x + c0f * d.x, base.y + c0f * d.y, base.z + c0f * d.z, base.w + c0f * d.w));
        } else {
          sp_gst4<PST>(P.out, po, cur);
        }
      }
    }
  }

  if constexpr (EPI == EPI_SAMPLE) {
    // packed cout rows are interleaved: rows 4q..4q+3 = (loc ch, loc ch+1, raw ch, raw ch+1), ch = 8*(row>>4) + 2*((row>>2)&3)
    const int Chalf = P.cout >> 1;
    const int ch = ((c >> 4) << 3) + 2 * ((c >> 2) & 3);
    const bool ok = on && ch < Chalf;
    const int chz = ok ? ch : 0;
    const float4 bi = o.a[0];
    const float2 e = P.e0 ? o.e : spm_philox_normal2(P.philox, P.draw, (unsigned)gpz, (unsigned)chz);      // block-uniform choice
    const float q0 = spm_act(v.x + bi.x, P.act), q1 = spm_act(v.y + bi.y, P.act);
    const float q2 = spm_act(v.z + bi.z, P.act), q3 = spm_act(v.w + bi.w, P.act);
    if (ok) {
      float2 r;
      r.x = q0 + e.x * (spm_softplus(q2) + 1e-8f);     // model_utils.py:84,107-108
      r.y = q1 + e.y * (spm_softplus(q3) + 1e-8f);
      sp_gst2<PST>(P.out + gpz * Chalf + ch, r);
      if (P.out2) {   // raw q parameters, reference channel order [loc | raw]
        sp_gst2<PST>(P.out2 + gpz * P.cout + ch, make_float2(q0, q1));
        sp_gst2<PST>(P.out2 + gpz * P.cout + Chalf + ch, make_float2(q2, q3));
      }
    }
  }
}

// ---- opt-in math mode bf16x3 (see conv_igemm.hip): weights pre-split by sf_pack_conv, pixels split after the fragment read
typedef __attribute__((ext_vector_type(8))) __bf16 sp_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 sp_bf16x2;
typedef __attribute__((ext_vector_type(2))) float sp_f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned sp_u32x4;
__device__ __forceinline__ unsigned sp_pk_bf16(float a, float b) {
  const sp_bf16x2 t = __builtin_convertvector((sp_f32x2){a, b}, sp_bf16x2);
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ void sp_split_bf16x8(const f32x4 x0, const f32x4 x1, sp_bf16x8& hi, sp_bf16x8& lo) {
  sp_u32x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = i < 2 ? x0[2 * i] : x1[2 * i - 4], b = i < 2 ? x0[2 * i + 1] : x1[2 * i - 3];
    const unsigned pk = sp_pk_bf16(a, b);
    h[i] = pk;
    l[i] = sp_pk_bf16(a - __uint_as_float(pk << 16), b - __uint_as_float(pk & 0xffff0000u));
  }
  hi = __builtin_bit_cast(sp_bf16x8, h);
  lo = __builtin_bit_cast(sp_bf16x8, l);
}

struct SpStamp { int stamp_slot; };      // what the SF_STAMP macros read (diagnostic builds)

// Flow mode (sp_flow_kernel): what an item waits for before it loads activations — tile counters of the two phases before it
// (sf_device.h, FlowPhase).  All members are workgroup-uniform.
struct SpDep {
  const unsigned* done;
  unsigned* err;
  int lag_idx, lag_expect;                // done[lag_idx] (this workgroup's replica of phase q-2's total) must reach lag_expect (0: none)
  int full_idx, full_expect;              // done[full_idx] (replica of phase q-1's total) must reach full_expect (0: none)
  int tile_idx, tile_n, tile_expect;      // done[tile_idx + i * SP_FLOW_TILE_STRIDE], i < tile_n (<= 62): tiles of phase q-1 under the halo
  int timeout;                            // polls before the wait gives up (err[0] += 1): a lost signal must not hang the GPU
};
// One wave waits for the counters, lane-parallel: lane 0 the phase-(q-2) total, lane 1 the phase-(q-1) total, lanes 2.. one tile
// counter each — every counter on a line of its own, read with sc1 (L1-bypassing) loads until all have reached their counts.
// The caller runs the workgroup barrier (and the acquire, in builds that need one).
__device__ __forceinline__ void sp_dep_wait(const SpDep& d, const int lane) {
  const bool l0 = lane == 0 && d.lag_expect > 0, l1 = lane == 1 && d.full_expect > 0, lt = lane >= 2 && lane - 2 < d.tile_n;
  const bool mine = l0 || l1 || lt;
  if (!__any(mine)) return;
  const int idx = l0 ? d.lag_idx : (l1 ? d.full_idx : (lt ? d.tile_idx + (lane - 2) * SP_FLOW_TILE_STRIDE : 0));
  const unsigned need = l0 ? (unsigned)d.lag_expect : (l1 ? (unsigned)d.full_expect : (lt ? (unsigned)d.tile_expect : 0u));
  const unsigned* const addr = d.done + (mine ? idx : 0);
  int spins = 0;
  for (;;) {
    const unsigned v = mine ? __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (__all(!mine || v >= need)) break;
    if (++spins >= d.timeout) {                       // wave-uniform (spins is)
      if (lane == 0) __hip_atomic_fetch_add(d.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

template <int EPI, bool SCALE, int NT, bool B3, bool PST, class PT>
__device__ __forceinline__ bool sp_body(const PT& P, const SpStamp L, int bx, int bz, const int m_tile, const int p_tile, float* const smem, const int tid,
                                        const SpDep dep = SpDep()) {
  typedef SpGeo<NT> G;
  constexpr int BN = G::BN;
  constexpr bool VOL = PST && (SF_FLOW_SC1 != 0);      // flow mode: L1-bypassing loads of activations instead of an acquire
  constexpr bool WINO = sp_has_wino<EPI, NT, B3, PST>();
  constexpr int PX_AUX = VOL ? 16 : 0;                 // aux bits of the pixel DMAs (16 = sc1)
  // XOR mask of the 16-byte slot swizzle of the ring: 7 for the fp32 fragment reads (slots c + g), 5 for the bf16x3 loop
  // (slots 2g / 2g + 1: conflict-free for the ds_read_b128 lane groups, see conv_igemm.hip)
  constexpr int SWM = B3 ? 5 : 7;
  constexpr int NKR = B3 ? 2 : 4;      // in-workgroup K split: halves (one 32-deep sub-chunk each) or quarters
#if SF_SP_PIN && defined(__HIP_DEVICE_COMPILE__)
  // The fields of the problem record that the way to the first DMA reads, requested together and waited for once: left to itself
  // hipcc loads each field where a branch first needs it — some fifteen dependent scalar-load round trips between the entry of the
  // kernel and the first weight DMA, on a workgroup that lives for 15-20 us and has no partner on its CU to hide them (the items of
  // the flow kernel read their record from device memory: the same there)
  if constexpr (SCALE) {      // (the SE-scaled instantiations have fewer scalar registers to spare: what the weight DMAs need)
    const int q0 = P.Hout, q1 = P.Wout, q2 = P.n_img, q3 = P.nsplit, q4 = P.cin_pad, q5 = P.KH, q6 = P.KW, q18 = P.cout_pad, q19 = P.ktot, q20 = P.sp_cps;
    const float* r0 = P.w;
    asm volatile("" ::"s"(q0), "s"(q1), "s"(q2), "s"(q3), "s"(q4), "s"(q5), "s"(q6), "s"(q18), "s"(q19), "s"(q20), "s"(r0));
  }
  if constexpr (!SCALE) {
#define SP_PIN(x) asm volatile("" ::"s"(x))
    const int q0 = P.Hout, q1 = P.Wout, q2 = P.n_img, q3 = P.nsplit, q4 = P.cin_pad, q5 = P.KH, q6 = P.KW, q7 = P.c0, q8 = P.c1,
              q9 = P.in0_cs, q10 = P.in1_cs, q11 = P.Hin, q12 = P.Win, q13 = P.in_up, q14 = P.dil, q15 = P.stride, q16 = P.pad,
              q17 = P.cout, q18 = P.cout_pad, q19 = P.ktot, q20 = P.sp_cps;
    const unsigned u0 = P.sp_m_hw, u1 = P.sp_m_w, u2 = P.sp_m_kcpt, u3 = P.sp_m_kw;
    const float *r0 = P.w, *r1 = P.in0, *r2 = P.in1, *r3 = P.fuse_w, *r4 = P.se_sum, *r5 = P.in_scale;
    SP_PIN(q0); SP_PIN(q1); SP_PIN(q2); SP_PIN(q3); SP_PIN(q4); SP_PIN(q5); SP_PIN(q6); SP_PIN(q7); SP_PIN(q8); SP_PIN(q9); SP_PIN(q10);
    SP_PIN(q11); SP_PIN(q12); SP_PIN(q13); SP_PIN(q14); SP_PIN(q15); SP_PIN(q16); SP_PIN(q17); SP_PIN(q18); SP_PIN(q19); SP_PIN(q20);
    SP_PIN(u0); SP_PIN(u1); SP_PIN(u2); SP_PIN(u3); SP_PIN(r0); SP_PIN(r1); SP_PIN(r2); SP_PIN(r3); SP_PIN(r4); SP_PIN(r5);
#undef SP_PIN
  }
#endif
  const int HWout = P.Hout * P.Wout;
  const int Ptot = P.n_img * HWout;
  if (p_tile * BN >= Ptot) return false;               // block-uniform
  const int nsplit = P.nsplit > 1 ? P.nsplit : 1;
  if (bz >= nsplit) return false;                      // block-uniform
  const int kcpt = P.cin_pad >> 5;                     // 32-deep sub-chunks per tap
  const int nsub_all = P.KH * P.KW * kcpt;
  bool wn = false;                                     // block-uniform: this problem runs in the Winograd form (K slices count (tap group, 32-channel sub-chunk) pairs)
  if constexpr (WINO) wn = P.sp_wino != 0;
  // Winograd form: 3x3 sub-kernels of the layer (7x7: the 3 x 3 cuts of its 9x9 frame).  The only 7x7 of the path is the trusting gate's first
  // layer, a LayerNorm launch: the other instantiations compile the group logic away (with it they spilled 11-16 scalar registers and
  // every 3x3 launch of a step lost 0.3 us)
  const int wgroups = (EPI == EPI_LNG && P.KH == 7) ? 9 : 1;
  const int nch_all = wn ? wgroups * kcpt : (nsub_all + 1) >> 1;
  const int cps = P.sp_cps > 0 ? P.sp_cps : (nch_all + nsplit - 1) / nsplit;     // chunks per K slice (host: every slice non-empty)
  const int cb = bz * cps;
#if defined(SF_ABL_K49)      // timing-only ablation (results are garbage): the K loop of every 3x3 layer cut to 4/9 of its chunks — what the products
  const int nchunks_full = (nch_all - cb) < cps ? (nch_all - cb) : cps;      // of a Winograd F(2x2, 3x3) form would cost at most
  const int nchunks = (!wn && P.KH == 3 && P.KW == 3 && P.dil == 1) ? (nchunks_full * 4 + 8) / 9 : nchunks_full;
#else
  const int nchunks = (nch_all - cb) < cps ? (nch_all - cb) : cps;
#endif

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  float* const misc = smem + (wn ? SPW_REGION : G::RING);
  float* const sc_lds = misc + SP_MISC;      // SCALE instantiations only
  const int img0 = sp_mdiv(p_tile * BN, P.sp_m_hw, HWout);              // block-uniform: first image this tile touches
  const int cin_pad = P.cin_pad;

  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // SE-scaled input (res_models.py:161-165 feeding the next conv): the per-(image, channel) scale rows of the images this
  // tile touches go to LDS; the consumers multiply their pixel fragments by them.  Published by the first barrier.
  auto fill_scale_rows = [&]() {
    if (SCALE && P.se_sum) {   // block-uniform: compute the gate here (one image; fixed summation order; host: C <= 128, Cr <= 16)
      // Only the 8 consumer waves work (the loaders' loads would queue behind their first DMAs); every global operand is
      // requested up front and each thread's partial rows are all in flight together (before: four dependent round trips
      // for the rows, three barriers, the loader waves' loads queued behind their DMAs).  What remains, ~3 us over a layer
      // without a gate, is the miss latency of rows another XCD wrote a moment ago plus two barrier / LDS round trips.
      const int C = P.c0, Cr = P.se_cr, nt = P.se_nt;
      float* const part = sc_lds + SP_SC_IMGS * 256;    // [G][C] partial row sums (<= 512 floats)
      float* const hid = part + 1024;                   // [Cr]
      constexpr int SE_Q = 20;                          // rows per thread (host: nt <= SE_Q * G)
      const int G = 512 / C;                            // row groups summed in parallel
      const int ch = tid % C, grp = tid / C;
      float f2[16];
      float f0a = 0.f, f0b = 0.f, f0c = 0.f, f0d = 0.f;
      if (wave < 8) {
        const bool mine = tid < C;                      // the thread that produces the gate of channel tid
        // (every vector load of the prologue queues in the CU's one address unit in front of the first A fragments and patches: the rows
        // took 3.2 us to arrive, profiles/r06_i_*; so only what is needed is requested — fc2 by the two waves that own a channel (C <= 128),
        // the second half of the row slots only when there are that many rows — under wave- / block-uniform branches)
#pragma unroll
        for (int h = 0; h < 16; ++h) f2[h] = 0.f;
        if (wave < 2) {
#pragma unroll
          for (int h = 0; h < 16; ++h) f2[h] = P.se_fc2[min(tid, C - 1) * Cr + min(h, Cr - 1)];      // clamped, never predicated (selected after the pins below)
        }
        // hidden units `wave` and `wave + 8`: their fc0 rows, channels lane and lane + 64
        const int h0 = min(wave, Cr - 1), h1 = min(wave + 8, Cr - 1);
        const int l0 = min(lane, C - 1), l1 = min(lane + 64, C - 1);
        float a0 = P.se_fc0[h0 * C + l0], b0 = P.se_fc0[h0 * C + l1], c0v = P.se_fc0[h1 * C + l0], d0 = P.se_fc0[h1 * C + l1];
        float r[SE_Q];
#pragma unroll
        for (int q = 0; q < SE_Q / 2; ++q) {            // rows grp, grp + G, ...: unconditional loads, selected afterwards
          const int t = grp + q * G;
          r[q] = sp_gld1<VOL>(P.se_sum + (size_t)min(t, nt - 1) * C + ch);
        }
#pragma unroll
        for (int q = SE_Q / 2; q < SE_Q; ++q) r[q] = 0.f;
        if (nt > (SE_Q / 2) * G) {
#pragma unroll
          for (int q = SE_Q / 2; q < SE_Q; ++q) {
            const int t = grp + q * G;
            r[q] = sp_gld1<VOL>(P.se_sum + (size_t)min(t, nt - 1) * C + ch);
          }
        }
        // hipcc sinks a load whose value is only needed under a lane mask into that region and drains vmcnt(0) behind it:
        // every value is "used" here, unconditionally, after ALL the loads were issued -> one wait for the lot
        asm volatile("" : "+v"(a0), "+v"(b0), "+v"(c0v), "+v"(d0));
#pragma unroll
        for (int h = 0; h < 16; h += 4) asm volatile("" : "+v"(f2[h]), "+v"(f2[h + 1]), "+v"(f2[h + 2]), "+v"(f2[h + 3]));
#pragma unroll
        for (int q = 0; q < SE_Q; q += 4) asm volatile("" : "+v"(r[q]), "+v"(r[q + 1]), "+v"(r[q + 2]), "+v"(r[q + 3]));
#pragma unroll
        for (int h = 0; h < 16; ++h) f2[h] = (mine && h < Cr) ? f2[h] : 0.f;
#pragma unroll
        for (int q = 0; q < SE_Q; ++q) r[q] = (grp + q * G < nt && grp < G) ? r[q] : 0.f;
        f0a = (wave < Cr && lane < C) ? a0 : 0.f;
        f0b = (wave < Cr && lane + 64 < C) ? b0 : 0.f;
        f0c = (wave + 8 < Cr && lane < C) ? c0v : 0.f;
        f0d = (wave + 8 < Cr && lane + 64 < C) ? d0 : 0.f;
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < SE_Q; ++q) acc4[q & 3] += r[q];
        if (grp < G) part[grp * C + ch] = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
      }
      SF_STAMP_AT(L, 11);
      __syncthreads();
      SF_STAMP_AT(L, 12);
      if (wave < 8) {
        // channel means of lane and lane + 64: the G <= 8 partial sums are read together (clamped, selected afterwards —
        // a dependent LDS round trip per partial cost ~300 cycles each)
        float pl[8], ph[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          pl[q] = part[min(q, G - 1) * C + min(lane, C - 1)];
          ph[q] = part[min(q, G - 1) * C + min(lane + 64, C - 1)];
        }
        float m_lo = 0.f, m_hi = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          m_lo += (q < G && lane < C) ? pl[q] : 0.f;
          m_hi += (q < G && lane + 64 < C) ? ph[q] : 0.f;
        }
        m_lo *= P.se_inv_hw; m_hi *= P.se_inv_hw;
        const float s = sp_wave_sum_lane63(f0a * m_lo + f0b * m_hi), u = sp_wave_sum_lane63(f0c * m_lo + f0d * m_hi);      // in lane 63
        if (lane == 63 && wave < Cr) hid[wave] = s > 0.f ? s : 0.f;
        if (lane == 63 && wave + 8 < Cr) hid[wave + 8] = u > 0.f ? u : 0.f;
      }
      __syncthreads();
      SF_STAMP_AT(L, 13);
      for (int idx = tid; idx < SP_SC_IMGS * cin_pad; idx += SP_THREADS) {
        float sv = 0.f;
        if (idx < C) {      // idx == tid here (C <= 128: consumer waves 0 and 1, which hold f2)
          float s = 0.f;
#pragma unroll
          for (int h = 0; h < 16; ++h) s += f2[h] * (h < Cr ? hid[h] : 0.f);
          sv = 1.f / (1.f + expf(-s));
          if (P.se_out && bx == 0 && bz == 0) sp_gst1<PST>(P.se_out + idx, sv);
        }
        sc_lds[idx] = sv;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (SCALE) {
      const float* const in_scale = P.in_scale;
      for (int idx = tid; idx < SP_SC_IMGS * cin_pad; idx += SP_THREADS) {
        const int si = idx / cin_pad, ch = idx - si * cin_pad;
        sc_lds[idx] = (in_scale && img0 + si < P.n_img && ch < P.c0) ? sp_gld1<VOL>(in_scale + (size_t)(img0 + si) * P.c0 + ch) : (in_scale ? 0.f : 1.f);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  // epilogue items of this lane — lane (pixel, channel quad): pixels 4*wave + lane/16 (+ 32 i), channels 4*(lane%16) .. +3 —
  // and their operands, fetched now (consumer waves) and used after the K loop
  const int quad = lane & 15;
  const int c_out = m_tile * SP_BM + 4 * quad;
  // Their operands are loaded by the consumers BEFORE the K loop on 32-pixel tiles (one item per lane; those launches have no
  // hand-off that would hide the round trip), and right BEHIND it on 64-pixel tiles (two items per lane: 48-92 registers that the
  // K loop would carry beside its accumulators and fragment sets — the 64-pixel kernels sat at the 168-register cap of a
  // 768-thread workgroup and the TRUST one spilled; the two barriers of the K-quarter reduction and the split-K hand-off cover the loads).
  constexpr bool OPS_EARLY = (NT != 4);
  int px[G::NPX], gpx[G::NPX];      // pixel of the tile / of the tensor
  bool on_item[G::NPX];
  SpOps ops[G::NPX];
#pragma unroll
  for (int i = 0; i < G::NPX; ++i) {
    px[i] = (wave < 8 ? 4 * wave : 0) + (lane >> 4) + 32 * i;
    gpx[i] = p_tile * BN + px[i];
    if constexpr (WINO) {
      if (wn) {      // Winograd form (one image, even H and W): tile pixel 4 t + 2 dy + dx is output (2 ty + dy, 2 tx + dx) of Winograd tile p_tile * 16 + t
        const int TW = P.Wout >> 1;
        const int t = p_tile * (BN / 4) + (px[i] >> 2);
        const int ty = sp_mdiv(t, P.sp_m_tw, TW), tx = t - ty * TW;
        const int gq = (2 * ty + ((px[i] >> 1) & 1)) * P.Wout + 2 * tx + (px[i] & 1);
        gpx[i] = 4 * t < Ptot ? gq : Ptot;
      }
    }
    on_item[i] = (wave < 8) && gpx[i] < Ptot && c_out < P.cout;
  }
  // fused 1x1 layer (LNG launches, block-uniform): its weights and this tile's output meet in a chunk-shaped buffer behind
  // the ring — [2 sub-chunks][64 weight rows | BN pixel rows][32] — and run through the consumers' fragment / MFMA code once more
  const bool fuse = (EPI == EPI_LNG) && P.fuse_w != nullptr;
  // (Winograd form: over the second V buffer, which is dead once the loop is over — the fused weights are fetched behind it)
  float* const fz = wn ? smem + 3 * SPW_SUB : misc + SP_MISC + (SCALE ? SP_SC_FLOATS : 0);
  // ---- loader constants that do not depend on activations (flow mode issues the weight DMAs before the dependency wait) ----
  const int lw = wave - 8;      // loader index (waves 8-11)
  // a sub-chunk is 8 + BN/8 DMA instructions of 8 rows x 128 B: loader lw takes the weight row blocks lw and lw + 4
  // and the pixel row blocks lw + 4 i
  int a_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * ((lw & 3) + 4 * i) + (lane >> 3);
    const int k4 = (lane & 7) ^ ((r >> 1) & SWM);
    int grow = m_tile * SP_BM + r;
    grow = grow < P.cout_pad ? grow : P.cout_pad - 1;   // rows >= cout_pad are never stored
    a_voff[i] = (grow * P.ktot + k4 * 4) * (int)sizeof(float);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  auto make_rsrc = [](const float* base, size_t bytes) {
    const unsigned nrec = bytes < 0x7fffffffull ? (unsigned)bytes : 0x7fffffffu;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (int)nrec, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(B3 ? static_cast<const float*>(P.w3) : P.w, (size_t)P.cout_pad * P.ktot * sizeof(float));
  // the weight half of chunk `cidx` of this K slice (two 32-deep sub-chunks) into ring buffer `buf`: 4 DMAs per loader
  auto issue_weights = [&](const int buf, const int cidx) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int scw = 2 * (cb + cidx) + s2;
      float* const blk = smem + buf * G::BUFF + s2 * G::SUBF;
      const bool live = scw < nsub_all;                    // wave-uniform (odd sub-chunk count: the last half is zero)
      const int minus1 = -1;                               // offset -1 fails the buffer range check: the DMA writes zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (sp_lds_void*)(blk + (8 * lw) * 32), 16, live ? a_voff[0] : minus1, live ? scw * 128 : 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (sp_lds_void*)(blk + (8 * (lw + 4)) * 32), 16, live ? a_voff[1] : minus1, live ? scw * 128 : 0, 0, 0);
    }
  };
  auto issue_fuse_weights = [&]() {      // the fused layer's weights: 2 sub-chunks x 64 rows, this loader's row blocks lw and lw + 4 (zeros past its K)
    const __amdgpu_buffer_rsrc_t rsrc_f = make_rsrc(P.fuse_w, (size_t)P.fuse_cout_pad * P.fuse_kpad * sizeof(float));
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = 8 * (lw + 4 * i) + (lane >> 3);
        const int k4 = (lane & 7) ^ ((r >> 1) & 7);
        const int grow = r < P.fuse_cout_pad ? r : P.fuse_cout_pad - 1;
        const int vo = (s2 * 32 < P.fuse_kpad) ? (grow * P.fuse_kpad + s2 * 32 + k4 * 4) * (int)sizeof(float) : -1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_f, (sp_lds_void*)(fz + s2 * G::SUBF + (8 * (lw + 4 * i)) * 32), 16, vo, 0, 0, 0);
      }
  };
#endif
  // Flow mode: activations (and the epilogue operands, the SE sums) may be outputs of the phase before this one.  Each role first
  // does everything that depends on nothing — the loaders send the weight halves of the first chunks on their way and compute
  // their pixel addresses, the consumers their fragment addresses — then wave 0 waits for the tile counters (+ the agent-scope
  // acquire in SF_FLOW_SC1=0 builds: this CU's L1 may hold lines another CU has rewritten) and ONE workgroup barrier releases
  // every wave's loads of activations.
  constexpr int W_EARLY = PST ? SP_LA : 0;      // chunks whose weight halves are issued before the wait
  if constexpr (PST) SF_STAMP_AT(L, 7);
  float4 fuse_lw = spm_zero4(), fuse_lb = spm_zero4();
  if (fuse && wave < 8) {
    const int cz = c_out < P.fuse_cout ? c_out : 0;
    fuse_lw = spm_ld4(P.fuse_scale + cz);
    fuse_lb = spm_ld4(P.fuse_bias + cz);
  }
  auto consumer_operands = [&]() {              // consumer waves, behind the flow wait
    if (OPS_EARLY) {
#pragma unroll
      for (int i = 0; i < G::NPX; ++i) sp_epi_load<EPI, VOL>(P, gpx[i], c_out, on_item[i], HWout, ops[i]);
    }
  };
  if constexpr (!PST) {
    if (wave < 8) consumer_operands();
    SF_STAMP_AT(L, 0);
  }


  float4 v[G::NPX];
  float* const red = smem;                                // direct form: [4][BN][SP_RED_PITCH]; Winograd form: M[16][16][SP_RED_PITCH]
  bool wn_done = false;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (WINO) {
    if (wn) {      // block-uniform
      // ============================ Winograd F(2x2, 3x3) form of a 3x3 layer on one latent ===========================================
      // Y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 outputs: 16 products instead of 36 (conv_wino.hip has the batched form; the transformed
      // weights U[cin / 16][16 positions][cout_pad][16] are the same copy).  The workgroup owns 16 consecutive Winograd tiles (= the 64 pixels
      // of its tile, in 2x2 blocks) x 64 cout x a slice of the input channels (K is split ACROSS workgroups by channel; the output
      // transform is linear, so every workgroup transforms its own partial products and the slabs / tickets / epilogues below are the
      // direct form's).  Per 32-channel sub-chunk:
      //   loader lw    gathers the 4x4 patches of ITS four tiles (8 LDS-DMAs: 8 pixels x 128 B each, zero fill outside the image),
      //                transforms them — lane = (tile, channel quad, row i of B^T d B): 8 ds_read_b128, 8 packed-width adds, 4 ds_write_b128
      //                — into V[position][tile][32] (slots XOR-swizzled by the tile like the ring's pixel rows) and joins ONE barrier;
      //                it only ever reads what it fetched itself, so nothing but its own vmcnt stands between a DMA and its transform
      //   consumer     (i, mh) owns row i of the 4 x 4 positions and a cout half: per 16 channels 4 ds_read_b128 (V) + 8 buffer_load_dwordx4
      //                straight from U (1 KB contiguous per fragment, a 16-channel group ahead, no LDS) feed 32 MFMAs
      // then the output transform: its row pass in the owner's registers, the column pass through LDS by the lane that owns the pixel.
      float* const raw = smem;                  // [2][16 tiles][16 px][32]
      float* const Vb = smem + 2 * SPW_SUB;     // [2][16 positions][16 tiles][32]
      const int nsc = nchunks;                  // 32-channel sub-chunks of this slice: cb .. cb + nsc - 1
      const int TW = P.Wout >> 1, ntiles = (P.Hout >> 1) * TW;
      f32x4 wacc[4][2];                         // consumers: the four positions of a row x the two 16-row cout fragments of a cout half
      if (wave >= 8) {
        // ---------------------------------------------- loader ------------------------------------------------------------------
        const int slotl = lane & 7, row8 = lane >> 3;
        int vb0[8], vb1[8];
        // patch offsets of this lane's eight DMA rows for tap group `grp` (3x3: the one group, no shift; 7x7: group (a, b) reads the input
        // shifted by (3a - 3, 3b - 3)): recomputed when a sub-chunk starts a new group (a slice of 6 sub-chunks meets at most 3)
        int cur_grp = -1;
        auto set_group = [&](const int grp) {
          const int ga = grp / 3, gb = grp - 3 * ga;
          const int sy = wgroups == 9 ? 3 * ga - 3 : 0, sx = wgroups == 9 ? 3 * gb - 3 : 0;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int t = p_tile * 16 + 4 * lw + (i >> 1);
            const int r = 2 * (i & 1) + (row8 >> 2), c = row8 & 3;
            const int ty = sp_mdiv(t, P.sp_m_tw, TW), tx = t - ty * TW;
            const int iy = 2 * ty - 1 + r + sy, ix = 2 * tx - 1 + c + sx;
            const bool in = (t < ntiles) & (iy >= 0) & (iy < P.Hin) & (ix >= 0) & (ix < P.Win);
            const int pxo = in ? iy * P.Win + ix : 0;
            vb0[i] = in ? (pxo * P.in0_cs + 4 * slotl) * 4 : (int)0x80000000;
            vb1[i] = in ? (pxo * P.in1_cs - P.c0 + 4 * slotl) * 4 : (int)0x80000000;
          }
          cur_grp = grp;
        };
        const __amdgpu_buffer_rsrc_t rsrc0 = make_rsrc(P.in0, (size_t)P.Hin * P.Win * P.in0_cs * sizeof(float));
        const __amdgpu_buffer_rsrc_t rsrc1 = make_rsrc(P.in1 ? P.in1 : P.in0, P.in1 ? (size_t)P.Hin * P.Win * P.in1_cs * sizeof(float) : 0);
        const int c0 = P.c0;
        auto issue_raw = [&](const int sidx) {      // sub-chunk cb + sidx into raw buffer sidx & 1: this loader's 4 tiles
          const int sg = cb + sidx;                 // (tap group, channel sub-chunk), group major
          const int grp = wgroups == 9 ? sp_mdiv(sg, P.sp_m_kcpt, kcpt) : 0;
          const int kc = sg - grp * kcpt;
          if (grp != cur_grp) set_group(grp);       // wave-uniform
          const bool from1 = kc * 32 >= c0;         // wave-uniform
          const int koff = kc * 128;
          float* const dst = raw + (sidx & 1) * SPW_SUB + (4 * lw) * 512;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (from1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (sp_lds_void*)(dst + i * 256), 16, vb1[i] + koff, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (sp_lds_void*)(dst + i * 256), 16, vb0[i] + koff, 0, 0, 0);
          }
        };
        issue_raw(0);
        if (nsc > 1) issue_raw(1);
        // the two items of this lane: (tile tl, row i of the transform, channel quad q)
        int rd_a[2], rd_b[2], wr[2], q4[2];
        float sg[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int id = lane + 64 * n;
          const int q = id & 7, i = (id >> 3) & 3, tl = id >> 5;
          // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: row i of B^T d = d[ra] + sg d[rb]
          const int ra = i == 0 ? 0 : (i == 2 ? 2 : 1), rb = i == 0 ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3));
          sg[n] = i == 1 ? 1.f : -1.f;
          const int tbase = (4 * lw + tl) * 512 + 4 * q;
          rd_a[n] = tbase + ra * 128; rd_b[n] = tbase + rb * 128;
          const int tile_l = 4 * lw + tl;
          wr[n] = ((4 * i) * 16 + tile_l) * 32 + 4 * (q ^ ((tile_l >> 1) & 7));
          q4[n] = 4 * q;
        }
        fill_scale_rows();
        if constexpr (SCALE) sp_barrier();                        // the SE rows (every wave wrote a part) are published
        for (int sidx = 0; sidx < nsc; ++sidx) {
          if (sidx + 1 < nsc) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the older of two sub-chunks in flight has landed
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const float* const rbuf = raw + (sidx & 1) * SPW_SUB;
          float* const vbuf = Vb + (sidx & 1) * SPW_SUB;
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            f32x4 da[4], db[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { da[c] = sp_lds_read128(rbuf + rd_a[n] + c * 32); db[c] = sp_lds_read128(rbuf + rd_b[n] + c * 32); }
            f32x4 t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = da[c] + sg[n] * db[c];
            f32x4 o[4];
            o[0] = t[0] - t[2]; o[1] = t[1] + t[2]; o[2] = t[2] - t[1]; o[3] = t[1] - t[3];
            if constexpr (SCALE) {      // SE gate of the input: per channel, commutes with the transform
              const f32x4 sc4 = sp_lds_read128(sc_lds + ((cb + sidx) % kcpt) * 32 + q4[n]);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) o[jj] = o[jj] * sc4;
            }
            typedef __attribute__((address_space(3))) f32x4 lds_f4w;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) *(lds_f4w*)(vbuf + wr[n] + jj * 512) = o[jj];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // V written, the raw buffer read
          if (sidx + 2 < nsc) issue_raw(sidx + 2);                // into the buffer just transformed (this loader's own region)
          sp_barrier();                                           // V[sidx] published
        }
      } else {
        // ---------------------------------------------- consumer ----------------------------------------------------------------
        // wave (i, mh): row i of the 4 x 4 positions, cout half mh (the ownership of conv_wino.hip's kernel: the row pass of the output
        // transform then happens in this wave's registers and half as much leaves for LDS)
        const int j = lane & 15, g = lane >> 4;
        const int wi = wave >> 1, mh = wave & 1;
        const __amdgpu_buffer_rsrc_t rsrc_u = make_rsrc(P.w_wino, (size_t)wgroups * 16 * P.cout_pad * P.cin_pad * sizeof(float));      // [group][cin / 16][16][cout_pad][16]: consecutive (group, 16-channel) steps are contiguous
        const int a_voff = (j * 16 + 4 * g) * 4;
        const int u_pos = P.cout_pad * 64;                        // bytes between positions
        const int u_grp = 16 * u_pos;                             // bytes between 16-channel groups
        int u_so = (2 * cb * 16 + 4 * wi) * u_pos + (m_tile * SP_BM + 32 * mh) * 64;      // group 2 cb, position (wi, 0), this cout half
        const int b_off = ((4 * wi) * 16 + j) * 32;               // V row of (position (wi, 0), tile j); + 512 per position
        const int sxm = (j >> 1) & 7;
        const int slot_h0 = (g ^ sxm) * 4, slot_h1 = ((4 + g) ^ sxm) * 4;
        // A fragments: two register sets, one 16-channel group ahead of the MFMAs (buffer_load_dwordx4: 1 KB contiguous per fragment).
        // Measured alternatives (profiles/r06_*): three sets / two groups ahead do not fit beside 32 accumulator registers in a 768-thread
        // workgroup (60-146 spills); a ring of four HALF groups in the same 64 registers (8-byte loads, 1.5 groups ahead) made the loop
        // 25 % SLOWER — the address unit spends 16 cycles per load instruction whatever its width, and 16 half-width loads per group and wave
        // are as many cycles as the group's MFMAs; touch loads of the slice by the loaders delayed the patches by 1.8 us for nothing.  With
        // no A loads at all (timing-only) the loop is 0.5 us shorter: two consumers share a SIMD, a sub-chunk is 2 x 64 MFMAs = 4096
        // cycles of its matrix pipe, and the loop runs at 75-80 % of that.
        f32x4 fa[2][4][2], fb[4];
#pragma unroll
        for (int pz = 0; pz < 4; ++pz)
#pragma unroll
          for (int m = 0; m < 2; ++m) wacc[pz][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto load_a = [&](const int set) {      // the 8 A fragments of the next 16-channel group
#pragma unroll
          for (int pz = 0; pz < 4; ++pz)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              fa[set][pz][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, a_voff + m * 1024, u_so + pz * u_pos, 0));
          u_so += u_grp;
        };
        auto mfma_grp = [&](const int set) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int pz = 0; pz < 4; ++pz)
#pragma unroll
              for (int m = 0; m < 2; ++m)
                wacc[pz][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][pz][m][e], fb[pz][e], wacc[pz][m], 0, 0, 0);
        };
        load_a(0);
        if constexpr (SCALE) {      // the SE gate's row loads go out before anything of this wave waits for scalar loads
          fill_scale_rows();
        }
        // operands that depend on the channel only (scale / bias / LayerNorm and logit rows / SE scale of the residual): once per lane, now
        SpOps chan;
        sp_epi_load<EPI, VOL, 1>(P, 0, c_out, c_out < P.cout, HWout, chan, true);
#if SF_SP_PIN
        {      // the problem-record fields that the operand loads, the hand-off and the epilogue read: requested together now, in the shadow of the
               // patch DMAs (left to itself hipcc loads each where the tail first needs it: a chain of dependent scalar-load round trips
               // between the last MFMA and the first operand load)
#define SP_PIN(x) asm volatile("" ::"s"(x))
          const float *t0 = P.acc_in, *t1 = P.scale, *t2 = P.bias, *t3 = P.e0, *t4 = P.e1, *t5 = P.out, *t6 = P.out2, *t7 = P.slab;
          const unsigned* t8 = P.counters;
          const int i0 = P.acc_cs, i1 = P.cout, i2 = P.mode, i3 = P.nsplit, i4 = P.act, i5 = P.fenced;
          SP_PIN(t0); SP_PIN(t1); SP_PIN(t2); SP_PIN(t3); SP_PIN(t4); SP_PIN(t5); SP_PIN(t6); SP_PIN(t7); SP_PIN(t8);
          SP_PIN(i0); SP_PIN(i1); SP_PIN(i2); SP_PIN(i3); SP_PIN(i4); SP_PIN(i5);
          if constexpr (EPI == EPI_AFFINE || EPI == EPI_BLEND) {
            const float *u0 = P.add, *u1 = P.add_scale, *u2 = P.chansum;
            const int j0 = P.bias_per_img, j1 = P.e0_cs, j2 = P.e1_cs, j3 = P.add_cs, j4 = P.out_cs, j5 = P.out_co, j6 = P.out2_cs, j7 = P.gate_from, j8 = P.clamp_from;
            SP_PIN(u0); SP_PIN(u1); SP_PIN(u2); SP_PIN(j0); SP_PIN(j1); SP_PIN(j2); SP_PIN(j3); SP_PIN(j4); SP_PIN(j5); SP_PIN(j6); SP_PIN(j7); SP_PIN(j8);
          }
          if constexpr (EPI == EPI_TRUST) {
            const float *u0 = P.e2, *u1 = P.e3, *u2 = P.e4, *u3 = P.e5, *u4 = P.coef;
            const int j0 = P.coef_stride;
            const float f0 = P.eps;
            SP_PIN(u0); SP_PIN(u1); SP_PIN(u2); SP_PIN(u3); SP_PIN(u4); SP_PIN(j0); SP_PIN(f0);
          }
          if constexpr (EPI == EPI_SAMPLE) {
            const unsigned long long* u0 = P.philox;
            const int j0 = P.draw;
            SP_PIN(u0); SP_PIN(j0);
          }
#undef SP_PIN
        }
#endif
        if constexpr (SCALE) sp_barrier();
        SF_STAMP_AT(L, 1);
        for (int sidx = 0; sidx < nsc; ++sidx) {
          const float* const vbuf = Vb + (sidx & 1) * SPW_SUB + b_off;
          sp_barrier();                                           // V[sidx] published
#ifdef SF_STAMP
          if (sidx == 0) SF_STAMP_AT(L, 2);
#endif
#pragma unroll
          for (int pz = 0; pz < 4; ++pz) fb[pz] = sp_lds_read128(vbuf + pz * 512 + slot_h0);
#if defined(SF_ABL_WSP_NO_A)      // timing-only ablation (garbage results): the loop without its A loads — what their latency costs at most
          if (sidx < 0) load_a(1);
#else
          load_a(1);
#endif
          __builtin_amdgcn_sched_barrier(0);
          mfma_grp(0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int pz = 0; pz < 4; ++pz) fb[pz] = sp_lds_read128(vbuf + pz * 512 + slot_h1);
#if defined(SF_ABL_WSP_NO_A)
          if (sidx < 0) load_a(0);
#else
          if (sidx + 1 < nsc) load_a(0);
#endif
          __builtin_amdgcn_sched_barrier(0);
          mfma_grp(1);
          __builtin_amdgcn_sched_barrier(0);
        }
        SF_STAMP_AT(L, 3);
        // epilogue operands (two items per lane): behind the loop, as in the direct form on 64-pixel tiles; they stay in flight across the
        // two barriers below (raw barriers: __syncthreads would wait for them)
#pragma unroll
        for (int i = 0; i < G::NPX; ++i) {
          sp_epi_copy_chan<EPI>(chan, ops[i]);
          sp_epi_load<EPI, VOL, 2>(P, gpx[i], c_out, on_item[i], HWout, ops[i], true);
        }
        SF_STAMP_AT(L, 10);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      sp_barrier();                                               // every V read and every DMA is done (the loaders drained theirs): T may land over the buffers
      if (fuse && wave >= 8) issue_fuse_weights();                // over the second V buffer; they land under the exchange, the hand-off and the first epilogue
      SF_STAMP_AT(L, 14);
      // output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  Row pass in the registers of the wave that owns row i of M:
      // T[i][0] = M[i][0] + M[i][1] + M[i][2], T[i][1] = M[i][1] - M[i][2] - M[i][3]  ->  T[4 rows][2][16 tiles][SP_RED_PITCH] in LDS ...
      if (wave < 8) {
        const int j = lane & 15, g = lane >> 4, wi = wave >> 1, mh = wave & 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const f32x4 t0 = (wacc[0][m] + wacc[1][m]) + wacc[2][m], t1 = (wacc[1][m] - wacc[2][m]) - wacc[3][m];
          float* const dst = red + ((2 * wi) * 16 + j) * SP_RED_PITCH + 32 * mh + 16 * m + 4 * g;
          spm_st4(dst, make_float4(t0[0], t0[1], t0[2], t0[3]));
          spm_st4(dst + 16 * SP_RED_PITCH, make_float4(t1[0], t1[1], t1[2], t1[3]));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      sp_barrier();
      SF_STAMP_AT(L, 15);
      // ... column pass by the lane that owns the pixel: output (dy, dx) of a tile = T[dy][dx] + T[dy+1][dx] + T[dy+2][dx] with the signs
      // of A^T's row dy (this replaces the direct form's reduction over K quarters).  Fixed order: bitwise reproducible.
#pragma unroll
      for (int i = 0; i < G::NPX; ++i) {
        v[i] = spm_zero4();
        if (wave < 8) {
          const int tl = px[i] >> 2, dy = (px[i] >> 1) & 1, dx = px[i] & 1;
          const float* const mb = red + ((2 * dy + dx) * 16 + tl) * SP_RED_PITCH + 4 * quad;
          const float sgn = dy ? -1.f : 1.f;
          const float4 r0 = spm_ld4(mb), r1 = spm_ld4(mb + 32 * SP_RED_PITCH), r2 = spm_ld4(mb + 64 * SP_RED_PITCH);
          v[i].x = (r0.x + sgn * r1.x) + sgn * r2.x; v[i].y = (r0.y + sgn * r1.y) + sgn * r2.y;
          v[i].z = (r0.z + sgn * r1.z) + sgn * r2.z; v[i].w = (r0.w + sgn * r1.w) + sgn * r2.w;
        }
      }
      SF_STAMP_AT(L, 9);      // (output transform done; the operand loads may still be in flight)
      wn_done = true;
    }
  }
#endif
  if (!wn_done) {
  if (wave >= 8) {
    // ================================= loader =================================================================
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PST) {
      if (fuse) issue_fuse_weights();
#pragma unroll
      for (int c = 0; c < SP_LA; ++c)
        if (c < nchunks) issue_weights(c, c);             // block-uniform
    }
#endif
    int b_c4[G::NBI], iy0[G::NBI], ix0[G::NBI], pbase[G::NBI];
#pragma unroll
    for (int i = 0; i < G::NBI; ++i) {
      const int pr = 8 * (lw + 4 * i) + (lane >> 3);
      b_c4[i] = 4 * ((lane & 7) ^ ((pr >> 1) & SWM));
      const int gp = p_tile * BN + pr;
      const bool pvalid = gp < Ptot;
      const int img = pvalid ? sp_mdiv(gp, P.sp_m_hw, HWout) : 0;
      const int rem = gp - img * HWout;
      const int oy = sp_mdiv(rem, P.sp_m_w, P.Wout), ox = rem - oy * P.Wout;
      iy0[i] = pvalid ? oy * P.stride - P.pad : -(1 << 28);
      ix0[i] = ox * P.stride - P.pad;
      pbase[i] = (pvalid ? img - img0 : 0) * P.Hin * P.Win;
    }
    const size_t img0_px = (size_t)img0 * P.Hin * P.Win;
    const float* const in0 = P.in0 + img0_px * P.in0_cs;
    const float* const in1 = P.in1 ? P.in1 + img0_px * P.in1_cs : nullptr;
    const int c0 = P.c0, c01 = P.c0 + P.c1, in0_cs = P.in0_cs, in1_cs = P.in1_cs;
    const int Win = P.Win, in_up = P.in_up, dil = P.dil, KW = P.KW;
    const int Hlog = P.Hin << P.in_up, Wlog = P.Win << P.in_up;
#if defined(__HIP_DEVICE_COMPILE__)
    const size_t imgs_left = (size_t)(P.n_img - img0);
    const __amdgpu_buffer_rsrc_t rsrc0 = make_rsrc(in0, imgs_left * P.Hin * P.Win * in0_cs * sizeof(float));
    const __amdgpu_buffer_rsrc_t rsrc1 = make_rsrc(in1 ? in1 : in0, in1 ? imgs_left * P.Hin * P.Win * in1_cs * sizeof(float) : 0);
#else
    (void)in0; (void)in1;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    if (!PST && fuse) issue_fuse_weights();
#endif
    if constexpr (PST) sp_barrier();                      // flow: the dependency wait of wave 0 is over (weights were issued above)
    // cursor of the next sub-chunk to fetch
    int sc = 2 * cb;
    const int sc_tap = sp_mdiv(sc, P.sp_m_kcpt, kcpt);
    int cur_kc = sc - sc_tap * kcpt, cur_ty = sp_mdiv(sc_tap, P.sp_m_kw, KW), cur_tx = sc_tap - cur_ty * KW;
    bool tap_fresh = true;
    // Per tap and pixel row block: byte offset of the gathered pixel in either source (channel slot of this lane included),
    // or 0x80000000 outside the image: adding the sub-chunk's channel offset keeps it beyond any buffer (< 2^31 bytes), so the
    // range check zero-fills.  Per sub-chunk the loader then needs ONE vector add per DMA: its instruction stream shares the
    // SIMD's issue with two MFMA streams, and every VALU instruction in it was measured to cost ~8 cycles there.
    int vb0[G::NBI], vb1[G::NBI];
#pragma unroll
    for (int i = 0; i < G::NBI; ++i) { vb0[i] = (int)0x80000000; vb1[i] = (int)0x80000000; }
    const bool whole_chunks = (c0 % 32 == 0) && (c01 % 32 == 0) && c01 == cin_pad;      // block-uniform: no channel padding inside a sub-chunk
    const int minus1 = -1;
    auto issue_chunk = [&](const int buf, const bool with_w = true) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bool live = sc < nsub_all;                       // wave-uniform (odd sub-chunk count: the last half is zero)
        if (tap_fresh) {
#pragma unroll
          for (int i = 0; i < G::NBI; ++i) {
            const int iy = iy0[i] + cur_ty * dil, ix = ix0[i] + cur_tx * dil;
            const bool in = (iy >= 0) & (iy < Hlog) & (ix >= 0) & (ix < Wlog);
            const int px = in ? pbase[i] + (iy >> in_up) * Win + (ix >> in_up) : 0;
            vb0[i] = in ? (px * in0_cs + b_c4[i]) * 4 : (int)0x80000000;
            vb1[i] = in ? (px * in1_cs - c0 + b_c4[i]) * 4 : (int)0x80000000;
          }
        }
        const bool from1 = cur_kc * 32 >= c0;                   // wave-uniform: the whole sub-chunk reads in1
        const int koff = cur_kc * 128;                          // bytes
        float* const blk = smem + buf * G::BUFF + s2 * G::SUBF;
#if defined(__HIP_DEVICE_COMPILE__)
        if (live) {
          if (with_w && !SP_ABL_NO_W) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (sp_lds_void*)(blk + (8 * lw) * 32), 16, a_voff[0], sc * 128, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (sp_lds_void*)(blk + (8 * (lw + 4)) * 32), 16, a_voff[1], sc * 128, 0, 0);
          }
#if !defined(SF_ABL_NO_PIXEL_DMA)
#pragma unroll
          for (int i = 0; i < G::NBI; ++i) {
            float* const dB = blk + (SP_BM + 8 * (lw + 4 * i)) * 32;
            int vob = (from1 ? vb1[i] : vb0[i]) + koff;
            if (!whole_chunks) {                                 // channels past cin inside the sub-chunk: zero
              const int c = cur_kc * 32 + b_c4[i];
              vob = (c < c01) ? vob : -1;
            }
            if (from1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (sp_lds_void*)dB, 16, vob, 0, 0, PX_AUX);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (sp_lds_void*)dB, 16, vob, 0, 0, PX_AUX);
          }
#endif
        } else {      // offset -1 fails the buffer range check: the DMA writes zeros
          if (with_w && !SP_ABL_NO_W) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (sp_lds_void*)(blk + (8 * lw) * 32), 16, minus1, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (sp_lds_void*)(blk + (8 * (lw + 4)) * 32), 16, minus1, 0, 0, 0);
          }
#if !defined(SF_ABL_NO_PIXEL_DMA)
#pragma unroll
          for (int i = 0; i < G::NBI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (sp_lds_void*)(blk + (SP_BM + 8 * (lw + 4 * i)) * 32), 16, minus1, 0, 0, 0);
#endif
        }
#else
        (void)blk; (void)koff; (void)from1; (void)live; (void)whole_chunks; (void)minus1;
#endif
        ++sc;
        ++cur_kc;
        tap_fresh = false;
        if (cur_kc == kcpt) {
          cur_kc = 0;
          tap_fresh = true;
          if (++cur_tx == KW) { cur_tx = 0; ++cur_ty; }
        }
      }
    };
#ifdef SF_STAMP
    unsigned long long l_issue = 0, l_wait = 0, l_bar = 0;
#endif
#pragma unroll
    for (int c = 0; c < SP_LA; ++c)
      if (c < nchunks) issue_chunk(c, c >= W_EARLY);      // block-uniform; flow mode: the weight halves are already on their way
    fill_scale_rows();
    // chunk 0 has landed once everything but the youngest chunk's DMAs is done: a whole chunk in launch mode, its pixel half in
    // flow mode (issue order there: weights 0, weights 1, pixels 0, pixels 1)
    if (nchunks >= SP_LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PST ? 2 * G::NBI : G::DPC) * (SP_LA - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sp_barrier();                                         // chunk 0 published
    int ibuf = SP_LA % SP_NB;
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + SP_LA < nchunks;
#ifdef SF_STAMP
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
      if (more) issue_chunk(ibuf);                        // into the buffer of chunk c-1: every consumer passed the barrier behind its reads
#ifdef SF_STAMP
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      l_issue += t1 - t0;
#endif
      if (c + 1 < nchunks) {
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::DPC * (SP_LA - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SF_STAMP
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        l_wait += t2 - t1;
#endif
        sp_barrier();                                     // chunk c+1 published
#ifdef SF_STAMP
        l_bar += __builtin_amdgcn_s_memtime() - t2;
#endif
      }
      ibuf = ibuf == SP_NB - 1 ? 0 : ibuf + 1;
    }
#ifdef SF_STAMP
    SF_STAMP_VAL_T(L, 13, l_issue, 512);
    SF_STAMP_VAL_T(L, 14, l_wait, 512);
    SF_STAMP_VAL_T(L, 15, l_bar, 512);
#endif
  } else {
    // ================================= consumer ===============================================================
    const int j = lane & 15, g = lane >> 4;
    // fp32: consumer (mh, kq) = cout half, K quarter of the chunk; all NT pixel tiles.
    // bf16x3: consumer (mh, nh, kh) = cout half, pixel half, K half (= one 32-deep sub-chunk: one 16x16x32 MFMA per tile pair
    // and product); lane (j, g) holds the 8 K values 8g .. 8g+7 of its row = slots 2g, 2g+1 (weights: hi / lo pieces)
    constexpr int NTW = B3 ? NT / 2 : NT;      // pixel tiles per consumer wave
    const int mh = wave & 1, kq = wave >> 1;
    const int nh = B3 ? ((wave >> 1) & 1) : 0, kh = wave >> 2;
    const int sxm = (j >> 1) & SWM;
    const int slot4 = B3 ? ((2 * g) ^ sxm) * 4 : ((((kq & 1) << 2) + g) ^ sxm) * 4;
    const int slot4b = B3 ? ((2 * g + 1) ^ sxm) * 4 : 0;
    const int sub_off = (B3 ? kh : (kq >> 1)) * G::SUBF;
    int a_off[2], b_off[NTW];
#pragma unroll
    for (int m = 0; m < 2; ++m) a_off[m] = sub_off + (32 * mh + 16 * m + j) * 32;
#pragma unroll
    for (int n = 0; n < NTW; ++n) b_off[n] = sub_off + (SP_BM + 16 * (nh * NTW + n) + j) * 32;
    // SE-scaled input: the lane's K values of a chunk are channels kc*32 + (kq&1)*16 + 4g .. +3 (fp32) / kc*32 + 8g .. +7 (bf16x3) of its sub-chunk
    const int s_k0 = 2 * cb + (B3 ? kh : (kq >> 1));
    int s_kc = s_k0 - sp_mdiv(s_k0, P.sp_m_kcpt, kcpt) * kcpt;
    const int s_step = kcpt > 2 ? 2 : 0;      // 2 % kcpt
    int simg[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int gp = p_tile * BN + 16 * (nh * NTW + n) + j;
      simg[n] = SCALE ? (gp < Ptot ? sp_mdiv(gp, P.sp_m_hw, HWout) - img0 : 0) * cin_pad + (B3 ? 8 * g : ((kq & 1) << 4) + 4 * g) : 0;
    }
    f32x4 fa[2][2][B3 ? 2 : 1], fb[2][NTW][B3 ? 2 : 1];
    auto read_frags = [&](const int buf, const int set) {
      const float* base = smem + buf * G::BUFF;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        fa[set][m][0] = sp_lds_read128(base + a_off[m] + slot4);
        if constexpr (B3) fa[set][m][1] = sp_lds_read128(base + a_off[m] + slot4b);
      }
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        fb[set][n][0] = sp_lds_read128(base + b_off[n] + slot4);
        if constexpr (B3) fb[set][n][1] = sp_lds_read128(base + b_off[n] + slot4b);
      }
      if (SCALE) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          fb[set][n][0] = fb[set][n][0] * sp_lds_read128(sc_lds + simg[n] + s_kc * 32);
          if constexpr (B3) fb[set][n][1] = fb[set][n][1] * sp_lds_read128(sc_lds + simg[n] + s_kc * 32 + 4);
        }
        s_kc += s_step;
        if (s_kc >= kcpt) s_kc -= kcpt;
      }
    };
    auto mfmas = [&](const int set) {
      if constexpr (B3) {
        sp_bf16x8 bh[NTW], bl[NTW];
#pragma unroll
        for (int n = 0; n < NTW; ++n) sp_split_bf16x8(fb[set][n][0], fb[set][n][1], bh[n], bl[n]);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NTW; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sp_bf16x8, fa[set][m][1]), bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NTW; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sp_bf16x8, fa[set][m][0]), bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NTW; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sp_bf16x8, fa[set][m][0]), bh[n], acc[m][n], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][m][0][e], fb[set][n][0][e], acc[m][n], 0, 0, 0);
      }
    };
    if constexpr (PST) {
      if (wave == 0) {
        sp_dep_wait(dep, lane);
        SF_STAMP_AT(L, 9);
        if constexpr (!VOL) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      sp_barrier();
      SF_STAMP_AT(L, 10);
      consumer_operands();
      SF_STAMP_AT(L, 0);
    }
    fill_scale_rows();
    SF_STAMP_AT(L, 1);
    sp_barrier();                         // chunk 0 published (and the SE scale rows)
    SF_STAMP_AT(L, 2);
    // the barrier that publishes chunk c+1 sits between the fragment reads of chunk c and its MFMAs: the reads of
    // chunk c+1 are in flight under the MFMAs of chunk c
    int buf = 0;
#ifdef SF_STAMP
    unsigned long long c_bar = 0;
    const unsigned long long c_loop0 = __builtin_amdgcn_s_memtime();
#define SP_BAR_T(stmt) { const unsigned long long tb_ = __builtin_amdgcn_s_memtime(); stmt; c_bar += __builtin_amdgcn_s_memtime() - tb_; }
#else
#define SP_BAR_T(stmt) stmt
#endif
    // The steady-state body is ONE basic block: with a run-time branch between a chunk's fragment reads and the previous chunk's
    // MFMAs (the round-2/3 form: `if (c + 1 < nchunks) read ...; mfmas`) hipcc's wait-count pass merges the two paths at the
    // join and puts s_waitcnt lgkmcnt(0) in front of the MFMAs — every second chunk's MFMAs then waited for the NEXT chunk's six
    // fragment reads (found in the ISA, round 4).  sched_barrier keeps the reads in front of the MFMAs.
    read_frags(0, 0);
    int c = 0;
    for (; c + 2 < nchunks; c += 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SP_BAR_T(sp_barrier());
      buf = buf == SP_NB - 1 ? 0 : buf + 1;
      read_frags(buf, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SP_BAR_T(sp_barrier());
      buf = buf == SP_NB - 1 ? 0 : buf + 1;
      read_frags(buf, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c + 1 < nchunks) {            // two chunks left
      SP_BAR_T(sp_barrier());
      buf = buf == SP_NB - 1 ? 0 : buf + 1;
      read_frags(buf, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mfmas(1);
    } else {                          // one
      mfmas(0);
    }
#ifdef SF_STAMP
    SF_STAMP_VAL(L, 8, __builtin_amdgcn_s_memtime() - c_loop0);
    SF_STAMP_VAL(L, 11, c_bar);
    SF_STAMP_VAL(L, 12, (unsigned long long)nchunks);
#endif
  }
  // ================================= reduction over the K quarters + epilogue ===================================
  SF_STAMP_AT(L, 3);
  if (!OPS_EARLY && wave < 8) {
#pragma unroll
    for (int i = 0; i < G::NPX; ++i) sp_epi_load<EPI, VOL>(P, gpx[i], c_out, on_item[i], HWout, ops[i]);
  }
  __syncthreads();                                        // every fragment read and every DMA of the ring is done
  if (wave < 8) {
    const int mh = wave & 1, j = lane & 15, g = lane >> 4;
    constexpr int NTW = B3 ? NT / 2 : NT;
    const int kr = B3 ? (wave >> 2) : (wave >> 1), n0 = B3 ? ((wave >> 1) & 1) * NTW : 0;      // K part / first pixel tile of this wave
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
        spm_st4(red + ((kr * BN) + 16 * (n0 + n) + j) * SP_RED_PITCH + 32 * mh + 16 * m + 4 * g,
                make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]));
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < G::NPX; ++i) {
    v[i] = spm_zero4();
    if (wave < 8) {
#pragma unroll
      for (int kq = 0; kq < NKR; ++kq) {
        const float4 t = spm_ld4(red + (kq * BN + px[i]) * SP_RED_PITCH + 4 * quad);
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      }
    }
  }
  }      // direct form
  if (nsplit > 1) {        // block-uniform: cross-workgroup split-K hand-off, sc1 stores / ticket / sc1 loads
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
    float* const tile_slab = P.slab + (size_t)bx * nsplit * (SP_BM * BN);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_slab, (short)0, nsplit * SP_BM * BN * 4, 0x00020000);
    if (wave < 8) {
#pragma unroll
      for (int i = 0; i < G::NPX; ++i)
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v[i].x), __float_as_uint(v[i].y), __float_as_uint(v[i].z), __float_as_uint(v[i].w)},
                                               rs, (px[i] * SP_BM + 4 * quad) * 4, bz * (SP_BM * BN * 4), 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* const flag = reinterpret_cast<int*>(misc);
    if (tid == 0) {
      unsigned* cnt = P.counters + bx;
      if (P.fenced) {      // reference form (SF_HANDOFF_FENCED=1): agent-scope release before the ticket, acquire behind it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (P.fenced) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const int last = (t == (unsigned)(nsplit - 1));
      if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      *flag = last;
    }
    __syncthreads();
    if (*flag == 0) return false;
    if (wave < 8) {
#pragma unroll
      for (int i = 0; i < G::NPX; ++i) v[i] = spm_zero4();
      for (int z = 0; z < nsplit; ++z) {     // slice order, own slice included: the sum does not depend on who arrives last
#pragma unroll
        for (int i = 0; i < G::NPX; ++i) {
          const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (px[i] * SP_BM + 4 * quad) * 4, z * (SP_BM * BN * 4), 16);
          v[i].x += __uint_as_float(t[0]); v[i].y += __uint_as_float(t[1]); v[i].z += __uint_as_float(t[2]); v[i].w += __uint_as_float(t[3]);
        }
      }
    }
#endif
  }
  SF_STAMP_AT(L, 4);
#pragma unroll
  for (int i = 0; i < G::NPX; ++i) { v[i].x += ops[i].pre.x; v[i].y += ops[i].pre.y; v[i].z += ops[i].pre.z; v[i].w += ops[i].pre.w; }
  const int c = c_out;
  float4 ysum = spm_zero4();
  if (!fuse) {
#pragma unroll
    for (int i = 0; i < G::NPX; ++i) {
      float4 y = spm_zero4();
      sp_epilogue<EPI, PST>(P, v[i], gpx[i], c, on_item[i], ops[i], y);
      ysum.x += y.x; ysum.y += y.y; ysum.z += y.z; ysum.w += y.w;
    }
  }
  if constexpr (EPI == EPI_LNG) {
    if (fuse) {      // block-uniform; only the workgroup that owns the finished tile gets here
      // this lane's four channels of its pixels are one 16-B slot of the pixel rows (K = channel: sub-chunk c / 32)
      // (the loaders' DMAs of the fused weights landed long ago: their K loop ended with vmcnt(0))
#pragma unroll
      for (int i = 0; i < G::NPX; ++i) {
        float4 y = spm_zero4();
        sp_epilogue<EPI, PST>(P, v[i], gpx[i], c, on_item[i], ops[i], y);
        if (wave < 8) {
          const int row = SP_BM + px[i];
          spm_st4(fz + (quad >> 3) * G::SUBF + row * 32 + (((quad & 7) ^ ((row >> 1) & 7)) << 2), y);
        }
      }
      __syncthreads();
      if (wave < 8) {
        const int mh = wave & 1, kq = wave >> 1, j = lane & 15, g = lane >> 4;
        const int slot4 = ((((kq & 1) << 2) + g) ^ ((j >> 1) & 7)) * 4;
        const float* const base = fz + (kq >> 1) * G::SUBF;
        f32x4 fa2[2], fb2[NT];
#pragma unroll
        for (int m = 0; m < 2; ++m) fa2[m] = sp_lds_read128(base + (32 * mh + 16 * m + j) * 32 + slot4);
#pragma unroll
        for (int n = 0; n < NT; ++n) fb2[n] = sp_lds_read128(base + (SP_BM + 16 * n + j) * 32 + slot4);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa2[m][e], fb2[n][e], acc[m][n], 0, 0, 0);
        // K quarters through the reduction buffer once more (every read of the first pass is behind two barriers)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            spm_st4(red + ((kq * BN) + 16 * n + j) * SP_RED_PITCH + 32 * mh + 16 * m + 4 * g,
                    make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]));
      }
      __syncthreads();
      if (wave < 8) {
        const bool cv = c < P.fuse_cout;
        const float inv_c = 1.f / (float)P.fuse_cout;
#pragma unroll
        for (int i = 0; i < G::NPX; ++i) {
          float4 t = spm_zero4();
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) {
            const float4 u = spm_ld4(red + (kq * BN + px[i]) * SP_RED_PITCH + 4 * quad);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
          }
          // channels-first LayerNorm over the pixel's channels + GELU (convolutions.py:303-308, :359-360)
          const float mean = sp_reduce16(cv ? (t.x + t.y) + (t.z + t.w) : 0.f) * inv_c;
          const float dx = t.x - mean, dy = t.y - mean, dz = t.z - mean, dw = t.w - mean;
          const float var = sp_reduce16(cv ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f) * inv_c;
          const float rstd = 1.f / sqrtf(var + P.eps);
          float4 y;
          y.x = spm_gelu(fuse_lw.x * (dx * rstd) + fuse_lb.x); y.y = spm_gelu(fuse_lw.y * (dy * rstd) + fuse_lb.y);
          y.z = spm_gelu(fuse_lw.z * (dz * rstd) + fuse_lb.z); y.w = spm_gelu(fuse_lw.w * (dw * rstd) + fuse_lb.w);
          const int gp = gpx[i];
          if (gp < Ptot && cv) sp_gst4<PST>(P.fuse_out, (size_t)gp * P.fuse_cout + c, y);
        }
      }
#ifdef SF_STAMP
      SF_STAMP_AT(L, 5);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SF_STAMP_AT(L, 6);
#endif
      return true;
    }
  }
  if constexpr (EPI == EPI_AFFINE) {
    if (P.chansum) {   // block-uniform: per-tile channel sums of `out` for the next SE gate, fixed order
      float* const cs_lds = smem + 4 * BN * SP_RED_PITCH;   // behind the reduction buffer
      float4 t = ysum;
      t.x += __shfl_xor(t.x, 16); t.y += __shfl_xor(t.y, 16); t.z += __shfl_xor(t.z, 16); t.w += __shfl_xor(t.w, 16);
      t.x += __shfl_xor(t.x, 32); t.y += __shfl_xor(t.y, 32); t.z += __shfl_xor(t.z, 32); t.w += __shfl_xor(t.w, 32);
      if (wave < 8 && lane < 16) spm_st4(cs_lds + wave * 64 + lane * 4, t);
      __syncthreads();
      if (wave == 0 && lane < 16) {
        float4 a4 = spm_zero4();
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) {
          const float4 b4 = spm_ld4(cs_lds + w8 * 64 + lane * 4);
          a4.x += b4.x; a4.y += b4.y; a4.z += b4.z; a4.w += b4.w;
        }
        if (c < P.cout) sp_gst4<PST>(P.chansum, (size_t)p_tile * P.cout + c, a4);
      }
    }
  }
#ifdef SF_STAMP
  SF_STAMP_AT(L, 5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SF_STAMP_AT(L, 6);
#endif
  return true;
}

// Logical workgroup id -> (problem, K slice, cout tile, pixel tile) on the compact 1-D grid: problem i owns the ids
// [wg_base[i], wg_base[i + 1]), K slice major, then cout tile, then pixel tile.
template <class PT, class IT>
__device__ __forceinline__ void sp_decode(const PT* ps, const IT* wg_base, const int nprob, const int lg, const int BN, int& by, int& bx, int& bz,
                                          int& m_tile, int& p_tile) {
  by = 0;
#pragma unroll
  for (int i = 1; i < SF_MAX_GROUP; ++i)
    if (i < nprob && lg >= wg_base[i]) by = i;
  bx = lg - wg_base[by];
  const PT& P = ps[by];
#if SF_SP_PIN && defined(__HIP_DEVICE_COMPILE__)
  {      // ... and the fields of the problem this decode reads
    const int d0 = P.n_img, d1 = P.Hout, d2 = P.Wout, d3 = P.cout_pad, d4 = P.sp_bn;
    const unsigned d5 = P.sp_m_tiles, d6 = P.sp_m_npt;
    asm volatile("" ::"s"(d0), "s"(d1), "s"(d2), "s"(d3), "s"(d4), "s"(d5), "s"(d6));
  }
#endif
  const int Ptot = P.n_img * P.Hout * P.Wout;
  const int n_mt = (P.cout_pad + SP_BM - 1) / SP_BM;
  const int n_pt = (Ptot + BN - 1) / BN, tiles = n_pt * n_mt;
  const bool mg = P.sp_bn == BN;                      // the reciprocals were made for this tile width
  bz = sp_mdiv(bx, mg ? P.sp_m_tiles : 0u, tiles);
  bx -= bz * tiles;                                  // tile id (slab / ticket index): cout tile major
  m_tile = sp_mdiv(bx, mg ? P.sp_m_npt : 0u, n_pt);
  p_tile = bx - m_tile * n_pt;
}

template <int EPI, bool SCALE, int NT, bool B3 = false>
__global__ __launch_bounds__(SP_THREADS) void conv_sp_kernel(const ConvLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // ring | misc | SE scale rows
  // Compact 1-D grid (L.wg_base): no idle workgroups.  Optionally (L.xcd_shift, off by default) the dispatch id is first mapped
  // so that each XCD gets a contiguous run of logical ids (blocks are dealt round-robin to the 8 XCDs, each with its own L2;
  // cdna_hip_programming.md T1).  Measured on the 50x50 step: 8 % less fabric traffic, 4 % MORE time — thirty workgroups
  // fetching the same lines from one L2 at the same moment is slower than the same fetches spread over eight.
  int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z, m_tile, p_tile;
  // the launch record's scalars in ONE request (they are adjacent in the kernel-argument segment): the count of problems, the XCD
  // flag and the five workgroup bases used to be four dependent round trips
  const int np_ = L.nprob, xs_ = L.xcd_shift;
  int wb_[SF_MAX_GROUP + 1];
#pragma unroll
  for (int i = 0; i <= SF_MAX_GROUP; ++i) wb_[i] = L.wg_base[i];
#if SF_SP_PIN && defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::"s"(np_), "s"(xs_), "s"(wb_[0]), "s"(wb_[1]), "s"(wb_[2]), "s"(wb_[3]), "s"(wb_[4]));
#endif
  int wtot_ = wb_[1];
#pragma unroll
  for (int i = 2; i <= SF_MAX_GROUP; ++i) wtot_ = np_ >= i ? wb_[i] : wtot_;      // wg_base[nprob]
  if (wtot_ > 0) {      // block-uniform
    const int total = (int)gridDim.x, id = bx;
    const int q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
    const int lg = xs_ ? (xcd < r ? xcd * (q + 1) + slot : r * (q + 1) + (xcd - r) * q + slot) : id;
    sp_decode(L.p, wb_, np_, lg, SpGeo<NT>::BN, by, bx, bz, m_tile, p_tile);
  } else {
    const int n_mt = (L.p[by].cout_pad + SP_BM - 1) / SP_BM;
    m_tile = bx % n_mt;
    p_tile = bx / n_mt;
  }
  (void)sp_body<EPI, SCALE, NT, B3, false>(L.p[by], SpStamp{L.stamp_slot}, bx, bz, m_tile, p_tile, smem, (int)threadIdx.x);
}

// ---- persistent flow kernel: every launch group of a rollout as a phase of ONE launch, ordered by tile-level dataflow --------------
// (north star: "the ODE derivative cell ... and the stepping loop fused into one LDS-tiled kernel per step"; sf_device.h, FlowPhase.)
// One workgroup per CU stays resident and runs item `wg` of every phase, in phase order:
//   item start   the loaders send the weight halves of the first chunks (they depend on nothing); wave 0 polls the tile counters
//                of phase q-2 (all) and of phase q-1 (the tiles under the item's halo), then agent-scope acquire + workgroup barrier;
//   item end     results leave with write-through (sc1) stores, every wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE
//                agent-scope atomic add on the (phase, pixel tile) counter by the workgroup that ran the tile's epilogue
// (MI355X_MICROARCH.md "Valid forms"; hand-off table, row 3).  Arithmetic, tiles, split-K slices and summation orders are those of
// the launch-per-layer path: results are bitwise identical to it (tests/test_gpu_persistent.py).  Split-K slabs / tickets alternate
// between two halves of the scratch by phase parity (a slice of phase q+1 may start while a last arriver of phase q still reads).
// Every workgroup of the grid must be resident at once (launch_sp_flow checks the occupancy; the device must be otherwise idle);
// every spin is bounded (SpFlow::timeout_polls) so that a lost signal cannot hang the GPU.
typedef const __attribute__((address_space(4))) FlowPhase FlowPhaseK;      // tables are read with scalar loads
typedef const __attribute__((address_space(4))) ConvProblem ConvProblemK;

template <int EPI, bool SCALE, int NT, bool B3, class FT>
__device__ __forceinline__ bool sp_flow_item(const FT& F, FlowPhaseK& ph, const int wg, float* smem, const int tid, const int stamp_slot,
                                             int& p_tile_out) {
  ConvProblemK* const ps = (ConvProblemK*)F.p + ph.prob0;
  int by, bx, bz, m_tile, p_tile;
  sp_decode(ps, ph.wg_base, ph.nprob, wg, SpGeo<NT>::BN, by, bx, bz, m_tile, p_tile);
  p_tile_out = p_tile;      // the pixel tile whose counter the workgroup signals when it finished one (one scalar carried to the end of
                            // the phase: decoding it again there was a chain of dependent table loads in front of the signal)
  // tiles of phase q-1 under this item's pixels + halo, in phase q-1's own tiling
  SpDep d;
  d.done = F.done; d.err = F.err; d.timeout = F.timeout_polls;
  const int rep = (wg & 7) * SP_FLOW_TOT_STRIDE;      // blocks are dealt round-robin over the XCDs: wg & 7 spreads the pollers (speed only)
  d.lag_idx = ph.lag_tot_base + rep; d.lag_expect = ph.lag_tot_expect;
  d.full_idx = ph.prev_tot_base + rep; d.full_expect = 0;
  d.tile_idx = ph.prev_tile_base; d.tile_n = 0; d.tile_expect = ph.prev_tile_expect;
  if (ph.prev_ntiles > 0) {
    const int p0 = p_tile * SpGeo<NT>::BN - ph.halo_px, p1 = p_tile * SpGeo<NT>::BN + SpGeo<NT>::BN - 1 + ph.halo_px;
    const int sh = ph.prev_bn == 64 ? 6 : 5;            // tiles are 32 or 64 pixels wide: a shift, not a division
    const int lo = (p0 < 0 ? 0 : p0) >> sh;
    int hi = p1 >> sh;
    hi = hi < ph.prev_ntiles - 1 ? hi : ph.prev_ntiles - 1;
    if (ph.dep_full || hi - lo + 1 > 62) {
      d.full_expect = ph.prev_tot_expect;
    } else {
      d.tile_idx += lo * SP_FLOW_TILE_STRIDE;
      d.tile_n = hi - lo + 1;
    }
  }
  return sp_body<EPI, SCALE, NT, B3, true>(ps[by], SpStamp{stamp_slot}, bx, bz, m_tile, p_tile, smem, tid, d);
}

template <bool B3>
__global__ __launch_bounds__(SP_THREADS) void sp_flow_kernel(const SpFlow F_by_value) {
  extern __shared__ __attribute__((aligned(16))) float smem_base[];
  const int wg = (int)blockIdx.x, n_grid = (int)gridDim.x;
  // Nothing may be carried across the phase loop: hipcc otherwise hoists everything that is invariant in it — the argument's
  // fields, thread-index arithmetic, LDS addresses and float constants of all seven inlined bodies — and keeps it in registers
  // for the whole kernel (168 VGPRs, 41-94 SGPR spills, 244-383 VGPR spills in round 3's kernel).  So once per phase the
  // argument pointer (read in place in the kernarg segment), the thread index and the LDS base offset pass through an empty asm:
  // every body computes what it needs from them and it dies with the body.
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) SpFlow* flow_cptr;
  flow_cptr Fk = (flow_cptr)__builtin_amdgcn_kernarg_segment_ptr();      // F is the only argument (offset 0)
#else
  const SpFlow* Fk = &F_by_value;
#endif
  (void)F_by_value;
  const int nphase = Fk->nphase;
  for (int k = 0; k < nphase; ++k) {
    int tid = (int)threadIdx.x;
    unsigned lds_off = 0;
    asm volatile("" : "+s"(Fk), "+v"(tid), "+s"(lds_off));
    float* const smem = smem_base + (lds_off >> 2);
    bool fin = false;
    int p_tile_done = 0;
    {
      const auto& F = *Fk;
      FlowPhaseK& ph = ((FlowPhaseK*)F.ph)[k];
      if (wg < ph.n_wg) {          // block-uniform
        const int key = ph.epi * 4 + (ph.scaled ? 2 : 0) + (ph.nt == 4 ? 1 : 0);
#if defined(SF_STAMP) && !defined(SF_STAMP_FLOW)      // diagnostic builds stamp the launch path; the flow kernel's stamps are their own build (-DSF_STAMP -DSF_STAMP_FLOW:
        (void)key; (void)smem; (void)p_tile_done;       // since the round-5 scalar pins hipcc fails on it with "illegal VGPR to SGPR copy")
#else
        switch (key) {
          case EPI_AFFINE * 4 + 0: fin = sp_flow_item<EPI_AFFINE, false, 2, B3>(F, ph, wg, smem, tid, k & 63, p_tile_done); break;
          case EPI_AFFINE * 4 + 1: fin = sp_flow_item<EPI_AFFINE, false, 4, B3>(F, ph, wg, smem, tid, k & 63, p_tile_done); break;
          case EPI_AFFINE * 4 + 3: fin = sp_flow_item<EPI_AFFINE, true, 4, B3>(F, ph, wg, smem, tid, k & 63, p_tile_done); break;
          case EPI_BLEND * 4 + 1:  fin = sp_flow_item<EPI_BLEND, false, 4, B3>(F, ph, wg, smem, tid, k & 63, p_tile_done); break;
          case EPI_LNG * 4 + 1:    fin = sp_flow_item<EPI_LNG, false, 4, B3>(F, ph, wg, smem, tid, k & 63, p_tile_done); break;
          case EPI_TRUST * 4 + 0:  fin = sp_flow_item<EPI_TRUST, false, 2, B3>(F, ph, wg, smem, tid, k & 63, p_tile_done); break;
          case EPI_SAMPLE * 4 + 3: fin = sp_flow_item<EPI_SAMPLE, true, 4, B3>(F, ph, wg, smem, tid, k & 63, p_tile_done); break;
          default: break;
        }
#endif
      }
    }
    // second cut: what follows re-derives the phase record from the laundered pointer, so nothing of it is carried through the body
    asm volatile("" : "+s"(Fk), "+v"(tid));
    const auto& F = *Fk;
    FlowPhaseK& ph = ((FlowPhaseK*)F.ph)[k];
    asm volatile("" : "+s"(p_tile_done));      // (block-uniform: the pixel tile this workgroup finished, from sp_flow_item)
    // state copy-out riding in this phase: src is an output of phase q-1 (all of it), nobody inside the flow reads dst.  The
    // workgroups without an item copy (all of them when every workgroup has one), after their own wait for phase q-1
    const int n_idle = n_grid - ph.n_wg;
    const bool copier = ph.copy_n4 > 0 && (n_idle > 0 ? wg >= ph.n_wg : true);      // block-uniform
    if (copier) {
      const int part = n_idle > 0 ? wg - ph.n_wg : wg, parts = n_idle > 0 ? n_idle : n_grid;
      if (tid < 64) {
        SpDep d;
        d.done = F.done; d.err = F.err; d.timeout = F.timeout_polls;
        d.lag_idx = 0; d.lag_expect = 0; d.tile_idx = 0; d.tile_n = 0; d.tile_expect = 0;
        d.full_idx = ph.prev_tot_base + (wg & 7) * SP_FLOW_TOT_STRIDE; d.full_expect = ph.prev_ntiles > 0 ? ph.prev_tot_expect : 0;
        sp_dep_wait(d, tid);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the copy reads with plain loads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      const float4* src = reinterpret_cast<const float4*>(ph.copy_src);
      float4* dst = reinterpret_cast<float4*>(ph.copy_dst);
      for (size_t i = (size_t)part * SP_THREADS + tid; i < (size_t)ph.copy_n4; i += (size_t)parts * SP_THREADS) dst[i] = src[i];
    }
    // signal: every wave's stores are out, then one add per workgroup that finished a tile (the others only fed a slab);
    // the barrier also keeps the next item's first LDS-DMAs behind this item's last LDS reads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // lanes 0-7: the eight replicas of the phase total (a finished item and a copy each count once); lane 8: the tile counter
    if (tid < 9 && (fin || copier)) {
      const int n_add = (fin ? 1 : 0) + (copier ? 1 : 0);
      if (tid < 8) __hip_atomic_fetch_add(F.done + ph.tot_base + tid * SP_FLOW_TOT_STRIDE, (unsigned)n_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (fin) __hip_atomic_fetch_add(F.done + ph.tile_base + p_tile_done * SP_FLOW_TILE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// one table piece per launch, passed by value: FlowBlob -> dst (n bytes, a multiple of 16)
__global__ void flow_write_kernel(const FlowBlob blob, unsigned char* dst, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) uint4* src_cptr;
  src_cptr src = (src_cptr)__builtin_amdgcn_kernarg_segment_ptr();      // blob is the first argument (offset 0)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 16 < n) reinterpret_cast<uint4*>(dst)[i] = src[i];
#endif
  (void)blob;
}
hipError_t launch_flow_write(const void* host_src, void* dev_dst, size_t bytes, hipStream_t stream) {
  const unsigned char* s = static_cast<const unsigned char*>(host_src);
  unsigned char* d = static_cast<unsigned char*>(dev_dst);
  if (bytes & 15) return hipErrorInvalidValue;
  for (size_t off = 0; off < bytes; off += SP_WRITER_BYTES) {
    FlowBlob b;
    const size_t n = bytes - off < SP_WRITER_BYTES ? bytes - off : SP_WRITER_BYTES;
    std::memcpy(b.b, s + off, n);
    hipLaunchKernelGGL(flow_write_kernel, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, stream, b, d + off, (int)n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// dynamic LDS of a conv_sp_kernel launch: region (ring, or the Winograd buffers where the instantiation has that block) | misc | SE rows | fused-layer buffer
template <int EPI, bool SCALE, int NT, bool B3>
constexpr int sp_launch_lds(bool fused) {
  // direct form: ring | misc | SE rows | fused-layer buffer; Winograd form (where the instantiation has it): buffers | misc | SE rows, the
  // fused-layer buffer over the second V buffer
  const int direct = SpGeo<NT>::RING + SP_MISC + (SCALE ? SP_SC_FLOATS : 0) + (fused ? SpGeo<NT>::BUFF : 0);
  const int wino = sp_has_wino<EPI, NT, B3, false>() ? SPW_REGION + SP_MISC + (SCALE ? SP_SC_FLOATS : 0) : 0;
  return (direct > wino ? direct : wino) * 4;
}
template <int EPI, bool SCALE, int NT, bool B3>
static hipError_t launch_sp_tb(const ConvLaunch& L, hipStream_t stream);
template <int EPI, bool SCALE, int NT>
static hipError_t launch_sp_t(const ConvLaunch& L, hipStream_t stream) {      // bf16x3: every problem carries split weights (api.hip decides)
  bool b3 = L.nprob > 0;
  for (int i = 0; i < L.nprob; ++i) b3 = b3 && L.p[i].w3 != nullptr && L.p[i].use_w3;
  return b3 ? launch_sp_tb<EPI, SCALE, NT, true>(L, stream) : launch_sp_tb<EPI, SCALE, NT, false>(L, stream);
}
template <int EPI, bool SCALE, int NT, bool B3>
static hipError_t launch_sp_tb(const ConvLaunch& L, hipStream_t stream) {
  auto kern = conv_sp_kernel<EPI, SCALE, NT, B3>;
  static bool attr_done[64] = {};      // the attribute is per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, sp_launch_lds<EPI, SCALE, NT, B3>(EPI == EPI_LNG));      // (only a LayerNorm launch can carry a fused 1x1 layer)
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  int maxblocks = 0, zs = 1;
  bool fused = false;
  for (int i = 0; i < L.nprob; ++i) {
    const ConvProblem& P = L.p[i];
    fused = fused || P.fuse_w != nullptr;
    const int Ptot = P.n_img * P.Hout * P.Wout;
    const int nb = ((Ptot + 16 * NT - 1) / (16 * NT)) * ((P.cout_pad + SP_BM - 1) / SP_BM);
    maxblocks = nb > maxblocks ? nb : maxblocks;
    zs = P.nsplit > zs ? P.nsplit : zs;
  }
  if (maxblocks == 0) return hipSuccess;
  if (fused && EPI != EPI_LNG) return hipErrorInvalidValue;
  const int lds = sp_launch_lds<EPI, SCALE, NT, B3>(fused);      // + the fused 1x1 layer's chunk buffer
  if (L.wg_base[L.nprob] > 0) {      // compact 1-D grid (the host filled wg_base for this tile size)
    hipLaunchKernelGGL(kern, dim3(L.wg_base[L.nprob], 1, 1), dim3(SP_THREADS), lds, stream, L);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(kern, dim3(maxblocks, L.nprob, zs), dim3(SP_THREADS), lds, stream, L);
  return hipGetLastError();
}

template <int NT>
static hipError_t launch_sp_n(const ConvLaunch& L, int epi, bool scaled, hipStream_t stream) {
  if (scaled) {
    if (epi == EPI_AFFINE) return launch_sp_t<EPI_AFFINE, true, NT>(L, stream);
    if (epi == EPI_SAMPLE) return launch_sp_t<EPI_SAMPLE, true, NT>(L, stream);
    return hipErrorInvalidValue;
  }
  switch (epi) {
    case EPI_AFFINE: return launch_sp_t<EPI_AFFINE, false, NT>(L, stream);
    case EPI_BLEND:  return launch_sp_t<EPI_BLEND, false, NT>(L, stream);
    case EPI_LNG:    return launch_sp_t<EPI_LNG, false, NT>(L, stream);
    case EPI_TRUST:  return launch_sp_t<EPI_TRUST, false, NT>(L, stream);
    case EPI_SAMPLE: return launch_sp_t<EPI_SAMPLE, false, NT>(L, stream);
  }
  return hipErrorInvalidValue;
}

// the (epilogue, SE-scaled, tile) variants sp_flow_kernel carries: the ones a 50x50x64 rollout uses; a launch group that needs
// another one ends the flow and runs as an ordinary launch
bool sp_flow_has(int epi, bool scaled, int bn) {
  const int key = epi * 4 + (scaled ? 2 : 0) + (bn == 64 ? 1 : 0);
  return key == EPI_AFFINE * 4 + 0 || key == EPI_AFFINE * 4 + 1 || key == EPI_AFFINE * 4 + 3 || key == EPI_BLEND * 4 + 1 || key == EPI_LNG * 4 + 1 ||
         key == EPI_TRUST * 4 + 0 || key == EPI_SAMPLE * 4 + 3;
}
// Workgroups of the flow kernel that can be resident at once on this device (0: none / error).  The phases wait for each other
// inside the launch, so the grid must not exceed this — and the device must be otherwise idle (another stream's kernel, a CU
// mask or reserved CUs are invisible to the occupancy query; the bounded spins then end the launch with err[0] != 0 instead of
// hanging it).
int sp_flow_capacity(bool b3) {
  const void* kern = b3 ? reinterpret_cast<const void*>(sp_flow_kernel<true>) : reinterpret_cast<const void*>(sp_flow_kernel<false>);
  constexpr int lds = sp_lds_bytes<4>(true) + SpGeo<4>::BUFF * 4;      // the largest variant: 64-px tiles, SE rows, fused 1x1 buffer
  static int cap[64][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  int& c = cap[dev][b3 ? 1 : 0];
  if (c == 0) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 0;
    int n_cu = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    hipError_t e = b3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sp_flow_kernel<true>, SP_THREADS, lds)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sp_flow_kernel<false>, SP_THREADS, lds);
    if (e != hipSuccess || per_cu < 1) { c = -1; return 0; }
    c = n_cu;      // ONE workgroup per CU (the LDS footprint admits no second one; one per CU is also what the items are sized for)
  }
  return c > 0 ? c : 0;
}
// one persistent launch for a flow of dependent phases (sp_flow_kernel); b3: every phase runs the split-bf16 loop
hipError_t launch_sp_flow(const SpFlow& F, int grid, bool b3, hipStream_t stream) {
  if (F.nphase < 1 || grid < 1) return hipSuccess;
  constexpr int lds = sp_lds_bytes<4>(true) + SpGeo<4>::BUFF * 4;
  if (grid > sp_flow_capacity(b3)) return hipErrorInvalidConfiguration;      // every workgroup must be resident: phases wait for each other
  if (b3) hipLaunchKernelGGL(sp_flow_kernel<true>, dim3(grid), dim3(SP_THREADS), lds, stream, F);
  else hipLaunchKernelGGL(sp_flow_kernel<false>, dim3(grid), dim3(SP_THREADS), lds, stream, F);
  return hipGetLastError();
}

// bn: pixels per tile (32 or 64); scaled: every problem carries an SE input scale (single input, cin_pad <= 256)
hipError_t launch_conv_sp(const ConvLaunch& L, int epi, bool scaled, int bn, hipStream_t stream) {
  if (bn == 64) return launch_sp_n<4>(L, epi, scaled, stream);
  if (bn == 32) return launch_sp_n<2>(L, epi, scaled, stream);
  return hipErrorInvalidValue;
}

}  // namespace sf
